// Byte-flip fuzz of the native TrueType reader and text layout (csrc/text.cpp) under AddressSanitizer + UBSan: corrupted and truncated
// fonts must be rejected or parsed without a single out-of-bounds access. Built and run by tests/test_text_sanitizers.py.
#include "contrast_hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
namespace crh { void set_last_error(const std::string&) {} }
#include <string>
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); std::vector<unsigned char> data; int c; while ((c = fgetc(f)) != EOF) data.push_back((unsigned char)c); fclose(f);
    std::mt19937 rng(5); int ok = 0, bad = 0;
    const uint32_t text[] = {'H','a','g',0xC4,'\n','1','2','&','%','g'};
    const int iterations = argc > 2 ? atoi(argv[2]) : 600;
    for (int it = 0; it < iterations; ++it) {
        std::vector<unsigned char> b = data;
        int mode = it % 4; int n = 1 + rng() % 16;
        if (mode == 2) b.resize(100 + rng() % (b.size() - 100));
        for (int k = 0; k < n; ++k) { size_t pos = mode == 0 ? rng() % (12 + 16 * 20) : (mode == 3 ? b.size() / 2 + rng() % (b.size() / 2) : rng() % b.size()); b[pos % b.size()] = (unsigned char)(rng() & 255); }
        crh_font* font = nullptr;
        if (crh_font_create(b.data(), b.size(), &font) != CRH_OK) { ++bad; continue; }
        crh_text_layout lay = {1.0f, 1, 2, 1};
        crh_path_list* list = nullptr;
        if (crh_paths_of_text(font, &lay, text, 10, nullptr, 0, &list) == CRH_OK) crh_path_list_destroy(list);
        crh_font_metrics m; crh_font_get_metrics(font, &m);
        for (int g = 0; g < 30; ++g) { crh_path_list* l2 = nullptr; if (crh_paths_of_glyph(font, (uint16_t)(rng() % (m.number_of_glyphs ? m.number_of_glyphs : 1)), &l2) == CRH_OK) crh_path_list_destroy(l2); }
        uint64_t nl = 0; crh_text_aligned_positions(font, &lay, text, 10, nullptr, nullptr, nullptr, nullptr, nullptr, &nl);
        crh_font_destroy(font); ++ok;
    }
    printf("parsed %d rejected %d\n", ok, bad);
}
