// tests/cpp/path_harness.cpp — prints what the C++ mirror's Path constructors and conversions produce (host code only, no GPU);
// tests/test_path_constructors.py compares the floats bit for bit with the Python mirror.
#include <cstdio>
#include <cstring>

#include "contrast_renderer.hpp"

using namespace contrast_renderer;

static void dump(const char* name, const Path& p) {
    std::printf("%s %zu", name, p.segment_types.size());
    uint32_t bits;
    for (float v : {p.start.first, p.start.second}) {
        std::memcpy(&bits, &v, 4);
        std::printf(" %08x", bits);
    }
    for (SegmentType t : p.segment_types) std::printf(" t%d", (int)t);
    for (float v : p.control) {
        std::memcpy(&bits, &v, 4);
        std::printf(" %08x", bits);
    }
    std::printf("\n");
}

int main() {
    dump("rounded_rect", Path::from_rounded_rect({0.25f, -0.5f}, {1.0f, 0.5f}, 0.2f));
    dump("ellipse", Path::from_ellipse({1.0f, 2.0f}, {0.75f, 0.5f}));
    dump("circle", Path::from_circle({-1.0f, 0.5f}, 0.3f));
    Path p;
    p.start = {0.1f, 0.2f};
    p.push_line({1.0f, 0.3f});
    p.push_integral_quadratic_curve({1.5f, 1.0f}, {0.7f, 1.4f});
    p.push_rational_quadratic_curve(0.6f, {0.2f, 1.9f}, {-0.4f, 1.1f});
    p.push_integral_cubic_curve({-0.9f, 0.8f}, {-0.8f, 0.1f}, {-0.3f, -0.2f});
    const float weights[4] = {1.0f, 1.3f, 0.7f, 1.0f};
    p.push_rational_cubic_curve(weights, {0.0f, -0.6f}, {0.3f, -0.5f}, {0.4f, -0.1f});
    dump("mixed", p);
    Path r = p;
    r.reverse();
    dump("reversed", r);
    r.reverse();
    dump("reversed_twice", r);
    Path c = p;
    c.convert_integral_curves_to_rational_curves();
    dump("rational", c);
    Path d = p;
    d.convert_quadratic_curves_to_cubic_curves();
    dump("cubic", d);
    Path e = p;
    e.close();
    dump("closed", e);
    e.close();
    dump("closed_again", e);
    for (int large = 0; large < 2; ++large)
        for (int sweep = 0; sweep < 2; ++sweep) {
            Path a;
            a.start = {1.0f, 0.25f};
            a.push_elliptical_arc({1.5f, 0.75f}, 0.4f, large != 0, sweep != 0, {-0.5f, 1.0f});
            char name[32];
            std::snprintf(name, sizeof(name), "arc_%d%d", large, sweep);
            dump(name, a);
        }
    Path z;
    z.start = {0.0f, 0.0f};
    z.push_elliptical_arc({0.0f, 1.0f}, 0.0f, false, true, {1.0f, 1.0f});
    dump("arc_zero_radius", z);
    const Path::Plane st = p.get_start_tangent(), en = p.get_end_tangent();
    uint32_t bits[6];
    const float t[6] = {st.c, st.nx, st.ny, en.c, en.nx, en.ny};
    std::memcpy(bits, t, sizeof(bits));
    std::printf("tangents 0 %08x %08x %08x %08x %08x %08x\n", bits[0], bits[1], bits[2], bits[3], bits[4], bits[5]);
    return 0;
}
