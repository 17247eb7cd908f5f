// csrc/text.cpp — the glyph producer of config 3: text.rs (Font, paths_of_glyph, calculate_aligned_positions!, paths_of_text)
// plus Path::transform (path.rs:387-439), host side, behind the C ABI of include/contrast_hip.h.
//
// The reference reads fonts with the crate ttf-parser 0.14.0 (Cargo.toml:20, Cargo.lock:1475-1478), which is not vendored. What
// text.rs observes of it is small: a handful of metrics, cmap lookup, horizontal / vertical advances, the glyph
// bounding box, format-0 kerning and `outline_glyph`, the walk of a TrueType `glyf` outline into move_to / line_to / quad_to / close
// calls. This file restates those from the published TrueType / OpenType table formats:
//   * outline walk: a contour starts at its first on-curve point (or at the midpoint of its first two off-curve points; a single
//     leading off-curve point is kept for the closing curve); consecutive off-curve points imply their midpoint; the contour is
//     closed with a curve through the pending off-curve point(s) or with a line back to the start, then `close()`;
//   * composite glyphs: components are outlined recursively (depth <= 32) under the concatenated 2x2 + offset transform
//     (ARGS_ARE_XY_VALUES offsets; point matching is not supported, as in the crate);
//   * glyphs whose points exceed the i16 range of the returned bounding box, or that emit no point at all, have no outline.
// Everything is integer / f32 arithmetic on big-endian table data; no GPU is involved.
#include <stdint.h>
#include <string.h>

#include <cmath>
#include <limits>
#include <string>
#include <vector>

#include "../../include/contrast_hip.h"

namespace crh {
void set_last_error(const std::string& text); // api.hip
}

namespace {

// ---------------------------------------------------------------------------------------------- big-endian reader
struct Table {
    const uint8_t* p = nullptr;
    size_t n = 0;
    bool has(size_t off, size_t len) const { return p && off <= n && len <= n - off; }
    uint8_t u8(size_t off) const { return p[off]; }
    uint16_t u16(size_t off) const { return (uint16_t)((p[off] << 8) | p[off + 1]); }
    int16_t i16(size_t off) const { return (int16_t)u16(off); }
    uint32_t u32(size_t off) const { return ((uint32_t)u16(off) << 16) | u16(off + 2); }
    Table sub(size_t off, size_t len) const { return has(off, len) ? Table{p + off, len} : Table{}; }
    Table from(size_t off) const { return off <= n && p ? Table{p + off, n - off} : Table{}; }
};

struct Transform { // x' = a x + c y + e, y' = b x + d y + f
    float a = 1.0f, b = 0.0f, c = 0.0f, d = 1.0f, e = 0.0f, f = 0.0f;
    bool is_default() const { return a == 1.0f && b == 0.0f && c == 0.0f && d == 1.0f && e == 0.0f && f == 0.0f; }
    void apply(float& x, float& y) const {
        const float tx = x, ty = y;
        x = a * tx + c * ty + e;
        y = b * tx + d * ty + f;
    }
    static Transform combine(const Transform& t1, const Transform& t2) { // t2 first, then t1
        Transform r;
        r.a = t1.a * t2.a + t1.c * t2.b;
        r.b = t1.b * t2.a + t1.d * t2.b;
        r.c = t1.a * t2.c + t1.c * t2.d;
        r.d = t1.b * t2.c + t1.d * t2.d;
        r.e = t1.a * t2.e + t1.c * t2.f + t1.e;
        r.f = t1.b * t2.e + t1.d * t2.f + t1.f;
        return r;
    }
};

// ---------------------------------------------------------------------------------------------- Vec<Path>
struct PathRec {
    float start[2] = {0.0f, 0.0f};
    std::vector<uint8_t> types;
    std::vector<float> control;
};

} // namespace

struct crh_path_list {
    std::vector<PathRec> paths;
    // flattened view (crh_path_list_view)
    std::vector<uint32_t> shape_path_begin, path_segment_begin, shape_dynamic_begin;
    std::vector<float> path_start, control;
    std::vector<int32_t> path_stroke;
    std::vector<uint8_t> types;
};

struct crh_font {
    std::vector<uint8_t> data;
    Table head, maxp, hhea, hmtx, loca, glyf, cmap, os2, kern, vhea, vmtx;
    uint16_t units_per_em = 0, number_of_glyphs = 0, number_of_h_metrics = 0, number_of_v_metrics = 0;
    int index_to_loc_format = 0;
    Table kern_subtable; // first subtable when it is a horizontal format-0 table
};

namespace {

// OutlineBuilder, text.rs:60-94
struct OutlineBuilder {
    PathRec path;
    std::vector<PathRec>* paths;
    void move_to(float x, float y) {
        path.start[0] = x;
        path.start[1] = y;
    }
    void line_to(float x, float y) {
        path.types.push_back(CRH_SEGMENT_LINE);
        path.control.push_back(x);
        path.control.push_back(y);
    }
    void quad_to(float x1, float y1, float x, float y) {
        path.types.push_back(CRH_SEGMENT_INTEGRAL_QUADRATIC);
        path.control.push_back(x1);
        path.control.push_back(y1);
        path.control.push_back(x);
        path.control.push_back(y);
    }
    void close() { // text.rs:89-93: the contour becomes one Path; no closing segment is added here
        paths->push_back(std::move(path));
        path = PathRec();
    }
};

struct Pt {
    float x, y;
    Pt lerp(Pt other, float t) const { return {x + (other.x - x) * t, y + (other.y - y) * t}; }
};

// the contour state machine
struct ContourWalker {
    OutlineBuilder* builder;
    Transform transform;
    bool default_transform;
    float x_min = std::numeric_limits<float>::max(), y_min = std::numeric_limits<float>::max();
    float x_max = std::numeric_limits<float>::lowest(), y_max = std::numeric_limits<float>::lowest();
    bool any_point = false;
    bool has_first_on = false, has_first_off = false, has_last_off = false;
    Pt first_on{}, first_off{}, last_off{};

    void extend(float x, float y) {
        any_point = true;
        x_min = std::fmin(x_min, x);
        y_min = std::fmin(y_min, y);
        x_max = std::fmax(x_max, x);
        y_max = std::fmax(y_max, y);
    }
    void move_to(float x, float y) {
        if (!default_transform) transform.apply(x, y);
        extend(x, y);
        builder->move_to(x, y);
    }
    void line_to(float x, float y) {
        if (!default_transform) transform.apply(x, y);
        extend(x, y);
        builder->line_to(x, y);
    }
    void quad_to(float x1, float y1, float x, float y) {
        if (!default_transform) {
            transform.apply(x1, y1);
            transform.apply(x, y);
        }
        extend(x1, y1);
        extend(x, y);
        builder->quad_to(x1, y1, x, y);
    }
    void push_point(float x, float y, bool on_curve, bool last_point) {
        const Pt p{x, y};
        if (!has_first_on) {
            if (on_curve) {
                has_first_on = true;
                first_on = p;
                move_to(p.x, p.y);
            } else if (has_first_off) {
                const Pt mid = first_off.lerp(p, 0.5f);
                has_first_on = true;
                first_on = mid;
                has_last_off = true;
                last_off = p;
                move_to(mid.x, mid.y);
            } else {
                has_first_off = true;
                first_off = p;
            }
        } else if (has_last_off) {
            if (on_curve) {
                has_last_off = false;
                quad_to(last_off.x, last_off.y, p.x, p.y);
            } else {
                const Pt mid = last_off.lerp(p, 0.5f);
                const Pt control = last_off;
                last_off = p;
                quad_to(control.x, control.y, mid.x, mid.y);
            }
        } else if (on_curve) {
            line_to(p.x, p.y);
        } else {
            has_last_off = true;
            last_off = p;
        }
        if (last_point) finish_contour();
    }
    void finish_contour() {
        if (has_first_off && has_last_off) {
            has_last_off = false;
            const Pt mid = last_off.lerp(first_off, 0.5f);
            quad_to(last_off.x, last_off.y, mid.x, mid.y);
        }
        if (has_first_on && has_first_off)
            quad_to(first_off.x, first_off.y, first_on.x, first_on.y);
        else if (has_first_on && has_last_off)
            quad_to(last_off.x, last_off.y, first_on.x, first_on.y);
        else if (has_first_on)
            line_to(first_on.x, first_on.y);
        has_first_on = has_first_off = has_last_off = false;
        builder->close();
    }
};

bool glyph_range(const crh_font& f, uint16_t glyph_id, size_t& begin, size_t& end) {
    if (glyph_id >= f.number_of_glyphs) return false;
    if (f.index_to_loc_format == 0) {
        if (!f.loca.has((size_t)glyph_id * 2, 4)) return false;
        begin = (size_t)f.loca.u16((size_t)glyph_id * 2) * 2;
        end = (size_t)f.loca.u16((size_t)glyph_id * 2 + 2) * 2;
    } else {
        if (!f.loca.has((size_t)glyph_id * 4, 8)) return false;
        begin = f.loca.u32((size_t)glyph_id * 4);
        end = f.loca.u32((size_t)glyph_id * 4 + 4);
    }
    return begin < end && f.glyf.has(begin, end - begin);
}

void outline_glyph_data(const crh_font& f, Table g, ContourWalker& w, int depth);

void outline_simple(Table g, int n_contours, ContourWalker& w) {
    size_t at = 10;
    if (!g.has(at, (size_t)n_contours * 2)) return;
    const size_t endpoints = at;
    const uint32_t n_points = (uint32_t)g.u16(endpoints + (size_t)(n_contours - 1) * 2) + 1u;
    at += (size_t)n_contours * 2;
    if (n_points == 1) return; // a glyph of one single point has no outline
    if (!g.has(at, 2)) return;
    const size_t n_instructions = g.u16(at);
    at += 2 + n_instructions;
    // flags (with repeats), then the x deltas, then the y deltas
    std::vector<uint8_t> flags(n_points);
    for (uint32_t i = 0; i < n_points;) {
        if (!g.has(at, 1)) return;
        const uint8_t flag = g.u8(at++);
        uint32_t repeat = 1;
        if (flag & 8u) {
            if (!g.has(at, 1)) return;
            repeat += g.u8(at++);
        }
        for (uint32_t k = 0; k < repeat && i < n_points; ++k) flags[i++] = flag;
    }
    std::vector<int16_t> xs(n_points), ys(n_points);
    int16_t x = 0, y = 0;
    for (uint32_t i = 0; i < n_points; ++i) {
        const uint8_t flag = flags[i];
        if (flag & 2u) { // x is one byte; bit 4 = sign (set = positive)
            if (!g.has(at, 1)) return;
            const int16_t d = g.u8(at++);
            x = (int16_t)(x + ((flag & 16u) ? d : -d));
        } else if (!(flag & 16u)) {
            if (!g.has(at, 2)) return;
            x = (int16_t)(x + g.i16(at));
            at += 2;
        }
        xs[i] = x;
    }
    for (uint32_t i = 0; i < n_points; ++i) {
        const uint8_t flag = flags[i];
        if (flag & 4u) {
            if (!g.has(at, 1)) return;
            const int16_t d = g.u8(at++);
            y = (int16_t)(y + ((flag & 32u) ? d : -d));
        } else if (!(flag & 32u)) {
            if (!g.has(at, 2)) return;
            y = (int16_t)(y + g.i16(at));
            at += 2;
        }
        ys[i] = y;
    }
    uint32_t point = 0;
    for (int c = 0; c < n_contours; ++c) {
        const uint32_t last = g.u16(endpoints + (size_t)c * 2);
        if (last >= n_points || last < point) return; // malformed end points: stop like an exhausted iterator
        for (; point <= last; ++point) w.push_point((float)xs[point], (float)ys[point], (flags[point] & 1u) != 0, point == last);
    }
}

float f2dot14(int16_t v) { return (float)v / 16384.0f; }

void outline_composite(const crh_font& f, Table g, ContourWalker& w, int depth) {
    if (depth >= 32) return;
    size_t at = 10;
    for (;;) {
        if (!g.has(at, 4)) return;
        const uint16_t flags = g.u16(at);
        const uint16_t component = g.u16(at + 2);
        at += 4;
        Transform ts;
        if (flags & 0x0002u) { // ARGS_ARE_XY_VALUES
            if (flags & 0x0001u) { // ARG_1_AND_2_ARE_WORDS
                if (!g.has(at, 4)) return;
                ts.e = (float)g.i16(at);
                ts.f = (float)g.i16(at + 2);
                at += 4;
            } else {
                if (!g.has(at, 2)) return;
                ts.e = (float)(int8_t)g.u8(at);
                ts.f = (float)(int8_t)g.u8(at + 1);
                at += 2;
            }
        } else { // point matching: not supported, the arguments are skipped
            at += (flags & 0x0001u) ? 4 : 2;
        }
        if (flags & 0x0080u) { // WE_HAVE_A_TWO_BY_TWO
            if (!g.has(at, 8)) return;
            ts.a = f2dot14(g.i16(at));
            ts.b = f2dot14(g.i16(at + 2));
            ts.c = f2dot14(g.i16(at + 4));
            ts.d = f2dot14(g.i16(at + 6));
            at += 8;
        } else if (flags & 0x0040u) { // WE_HAVE_AN_X_AND_Y_SCALE
            if (!g.has(at, 4)) return;
            ts.a = f2dot14(g.i16(at));
            ts.d = f2dot14(g.i16(at + 2));
            at += 4;
        } else if (flags & 0x0008u) { // WE_HAVE_A_SCALE
            if (!g.has(at, 2)) return;
            ts.a = f2dot14(g.i16(at));
            ts.d = ts.a;
            at += 2;
        }
        size_t begin, end;
        if (glyph_range(f, component, begin, end)) {
            ContourWalker child;
            child.builder = w.builder;
            child.transform = Transform::combine(w.transform, ts);
            child.default_transform = child.transform.is_default();
            child.x_min = w.x_min, child.y_min = w.y_min, child.x_max = w.x_max, child.y_max = w.y_max, child.any_point = w.any_point;
            outline_glyph_data(f, f.glyf.sub(begin, end - begin), child, depth + 1);
            w.x_min = child.x_min, w.y_min = child.y_min, w.x_max = child.x_max, w.y_max = child.y_max, w.any_point = child.any_point;
        }
        if (!(flags & 0x0020u)) return; // MORE_COMPONENTS
    }
}

void outline_glyph_data(const crh_font& f, Table g, ContourWalker& w, int depth) {
    if (!g.has(0, 10)) return;
    const int16_t n_contours = g.i16(0);
    if (n_contours > 0)
        outline_simple(g, n_contours, w);
    else if (n_contours < 0)
        outline_composite(f, g, w, depth);
}

// Face::outline_glyph: false = None
bool outline_glyph(const crh_font& f, uint16_t glyph_id, std::vector<PathRec>& out) {
    size_t begin, end;
    if (!glyph_range(f, glyph_id, begin, end)) return false;
    OutlineBuilder builder;
    builder.paths = &out;
    ContourWalker w;
    w.builder = &builder;
    w.default_transform = true;
    outline_glyph_data(f, f.glyf.sub(begin, end - begin), w, 0);
    if (!w.any_point) return false;
    // the bounding box is returned as i16: points outside that range make the call fail
    const float lo = -32768.0f, hi = 32767.0f;
    if (!(w.x_min >= lo && w.x_max <= hi && w.y_min >= lo && w.y_max <= hi)) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------- tables
bool typographic_metrics(const crh_font& f) { return f.os2.has(0, 78) && (f.os2.u16(62) & 0x0080u) != 0; } // fsSelection bit 7
int32_t ascender(const crh_font& f) { return typographic_metrics(f) ? f.os2.i16(68) : f.hhea.i16(4); }
int32_t descender(const crh_font& f) { return typographic_metrics(f) ? f.os2.i16(70) : f.hhea.i16(6); }
int32_t line_gap(const crh_font& f) { return typographic_metrics(f) ? f.os2.i16(72) : f.hhea.i16(8); }

bool cmap_subtable_lookup(Table t, uint32_t c, uint16_t& glyph) {
    if (!t.has(0, 2)) return false;
    const uint16_t format = t.u16(0);
    if (format == 0) {
        if (c > 255 || !t.has(6 + c, 1)) return false;
        glyph = t.u8(6 + c);
        return glyph != 0;
    }
    if (format == 4) {
        if (c > 0xFFFFu || !t.has(0, 14)) return false;
        const uint16_t code = (uint16_t)c;
        const size_t seg_count = t.u16(6) / 2u;
        const size_t end_codes = 14, start_codes = 16 + seg_count * 2, deltas = start_codes + seg_count * 2, offsets = deltas + seg_count * 2;
        if (!t.has(offsets, seg_count * 2)) return false;
        size_t lo = 0, hi = seg_count;
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            const uint16_t end_code = t.u16(end_codes + mid * 2);
            if (end_code < code) {
                lo = mid + 1;
                continue;
            }
            const uint16_t start_code = t.u16(start_codes + mid * 2);
            if (start_code > code) {
                hi = mid;
                continue;
            }
            const int16_t delta = t.i16(deltas + mid * 2);
            const uint16_t range_offset = t.u16(offsets + mid * 2);
            if (range_offset == 0) {
                glyph = (uint16_t)(code + (uint16_t)delta);
                return true;
            }
            if (range_offset == 0xFFFFu) return false;
            const size_t pos = offsets + mid * 2 + range_offset + (size_t)(code - start_code) * 2;
            if (!t.has(pos, 2)) return false;
            const uint16_t value = t.u16(pos);
            if (value == 0) return false;
            const int32_t id = (int32_t)(int16_t)(uint16_t)(value + (uint16_t)delta);
            if (id < 0) return false;
            glyph = (uint16_t)id;
            return true;
        }
        return false;
    }
    if (format == 6) {
        if (!t.has(0, 10)) return false;
        const uint32_t first = t.u16(6), count = t.u16(8);
        if (c < first || c - first >= count || !t.has(10 + (size_t)(c - first) * 2, 2)) return false;
        glyph = t.u16(10 + (size_t)(c - first) * 2);
        return true;
    }
    if (format == 12 || format == 13) {
        if (!t.has(0, 16)) return false;
        const uint32_t groups = t.u32(12);
        for (uint32_t g = 0; g < groups; ++g) {
            const size_t at = 16 + (size_t)g * 12;
            if (!t.has(at, 12)) return false;
            const uint32_t start = t.u32(at), end = t.u32(at + 4), id = t.u32(at + 8);
            if (c < start || c > end) continue;
            const uint32_t value = format == 12 ? id + (c - start) : id;
            if (value > 0xFFFFu) return false;
            glyph = (uint16_t)value;
            return true;
        }
        return false;
    }
    return false;
}

bool glyph_index(const crh_font& f, uint32_t c, uint16_t& glyph) {
    if (!f.cmap.has(0, 4)) return false;
    const size_t n = f.cmap.u16(2);
    for (size_t i = 0; i < n; ++i) {
        const size_t rec = 4 + i * 8;
        if (!f.cmap.has(rec, 8)) return false;
        const uint16_t platform = f.cmap.u16(rec), encoding = f.cmap.u16(rec + 2);
        const Table sub = f.cmap.from(f.cmap.u32(rec + 4));
        if (!sub.has(0, 2)) continue;
        const uint16_t format = sub.u16(0);
        // Unicode subtables only: platform 0, or Windows with the BMP / full-repertoire encodings
        const bool unicode = platform == 0 || (platform == 3 && (encoding == 1 || (encoding == 10 && (format == 12 || format == 13))));
        if (!unicode) continue;
        if (cmap_subtable_lookup(sub, c, glyph)) return true;
    }
    return false;
}

bool glyph_advance(const crh_font& f, uint16_t glyph, bool vertical, uint16_t& advance) {
    const Table& t = vertical ? f.vmtx : f.hmtx;
    const uint16_t n_metrics = vertical ? f.number_of_v_metrics : f.number_of_h_metrics;
    if (!t.p || n_metrics == 0 || glyph >= f.number_of_glyphs) return false;
    const size_t index = glyph < n_metrics ? glyph : (size_t)n_metrics - 1; // the last advance repeats for the remaining glyphs
    if (!t.has(index * 4, 2)) return false;
    advance = t.u16(index * 4);
    return true;
}

bool glyph_bounding_box(const crh_font& f, uint16_t glyph, int16_t box[4]) {
    size_t begin, end;
    if (!glyph_range(f, glyph, begin, end)) return false;
    const Table g = f.glyf.sub(begin, end - begin);
    if (!g.has(0, 10)) return false;
    box[0] = g.i16(2);
    box[1] = g.i16(4);
    box[2] = g.i16(6);
    box[3] = g.i16(8);
    return true;
}

bool glyphs_kerning(const crh_font& f, uint16_t left, uint16_t right, int16_t& value) {
    const Table& t = f.kern_subtable; // format 0: nPairs searchRange entrySelector rangeShift, then (left, right, value) sorted
    if (!t.has(0, 8)) return false;
    const size_t n_pairs = t.u16(0);
    const uint32_t key = ((uint32_t)left << 16) | right;
    size_t lo = 0, hi = n_pairs;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2, at = 8 + mid * 6;
        if (!t.has(at, 6)) return false;
        const uint32_t k = t.u32(at);
        if (k == key) {
            value = t.i16(at + 4);
            return true;
        }
        if (k < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------- calculate_aligned_positions! (text.rs:145-230)
struct Positioned {
    int64_t position[2];
    uint16_t glyph;
};
struct Line {
    uint64_t range_end;
    std::vector<Positioned> glyphs; // the last entry is the line terminator (glyph id 0)
};
struct Aligned {
    int64_t extent[2], offset[2];
    std::vector<Line> lines;
};

Aligned aligned_positions(const crh_font& f, const crh_text_layout& layout, const uint32_t* text, size_t n_chars) {
    uint16_t replacement = 0;
    const bool has_replacement = glyph_index(f, 0xFFFDu, replacement);
    const bool kerning = f.kern_subtable.p != nullptr;
    int major_axis;
    int64_t sign_x, sign_y;
    switch (layout.orientation) {
        case CRH_ORIENTATION_RIGHT_TO_LEFT: major_axis = 0, sign_x = -1, sign_y = -1; break;
        case CRH_ORIENTATION_LEFT_TO_RIGHT: major_axis = 0, sign_x = 1, sign_y = -1; break;
        case CRH_ORIENTATION_TOP_TO_BOTTOM: major_axis = 1, sign_x = 1, sign_y = -1; break;
        default: major_axis = 1, sign_x = 1, sign_y = 1; break;
    }
    int64_t line_minor_extent, gap;
    if (major_axis == 0) {
        line_minor_extent = (int16_t)(ascender(f) - descender(f)); // Face::height() is an i16
        gap = line_gap(f);
    } else {
        line_minor_extent = f.vhea.has(0, 10) ? (int16_t)(f.vhea.i16(4) - f.vhea.i16(6)) : 0;
        gap = f.vhea.has(0, 10) ? f.vhea.i16(8) : 0;
    }
    Aligned out;
    out.extent[0] = out.extent[1] = 0;
    int64_t line_major_extent = 0;
    std::vector<Positioned> glyph_positions;
    bool has_prev = false;
    uint16_t prev = 0;
    uint64_t index = 0;
    for (size_t i = 0; i < n_chars; ++i) {
        const uint32_t c = text[i];
        index += 1;
        Positioned gp;
        gp.position[0] = out.extent[0];
        gp.position[1] = out.extent[1];
        gp.position[major_axis] = line_major_extent;
        if (c == (uint32_t)'\n') {
            gp.glyph = 0;
            glyph_positions.push_back(gp);
            out.lines.push_back(Line{index, std::move(glyph_positions)});
            glyph_positions.clear();
            out.extent[major_axis] = std::max(out.extent[major_axis], line_major_extent);
            out.extent[1 - major_axis] += line_minor_extent + gap;
            line_major_extent = 0;
            has_prev = false;
        } else {
            uint16_t glyph = 0;
            if (!glyph_index(f, c, glyph)) glyph = has_replacement ? replacement : 0; // the reference unwrap()s; glyph 0 keeps us total
            int16_t k;
            if (kerning && has_prev && glyphs_kerning(f, prev, glyph, k)) line_major_extent += k;
            has_prev = true;
            prev = glyph;
            uint16_t advance;
            if (glyph_advance(f, glyph, major_axis == 1, advance)) line_major_extent += advance;
            gp.glyph = glyph;
            glyph_positions.push_back(gp);
        }
    }
    {
        Positioned gp;
        gp.position[0] = out.extent[0];
        gp.position[1] = out.extent[1];
        gp.position[major_axis] = line_major_extent;
        gp.glyph = 0;
        glyph_positions.push_back(gp);
        out.lines.push_back(Line{index + 1, std::move(glyph_positions)});
        out.extent[major_axis] = std::max(out.extent[major_axis], line_major_extent);
        out.extent[1 - major_axis] += line_minor_extent;
    }
    int64_t offset[2] = {0, 0};
    switch (layout.minor_alignment) {
        case CRH_ALIGNMENT_BEGIN: offset[1 - major_axis] = -(int64_t)(int16_t)descender(f); break;
        case CRH_ALIGNMENT_BASELINE: offset[1 - major_axis] = 0; break;
        case CRH_ALIGNMENT_CENTER: offset[1 - major_axis] = (f.os2.has(0, 88) && f.os2.u16(0) >= 2 ? (int64_t)f.os2.i16(86) : 0) / 2; break;
        default: offset[1 - major_axis] = -line_minor_extent; break;
    }
    int64_t line_offset[2] = {offset[0], offset[1]};
    for (Line& line : out.lines) {
        const int64_t major_extent = line.glyphs.back().position[major_axis];
        // `let mut offset = offset;` shadows per line, but the minor component below is applied to the copy each time (text.rs:213-221)
        line_offset[0] = offset[0];
        line_offset[1] = offset[1];
        switch (layout.major_alignment) {
            case CRH_ALIGNMENT_BEGIN: line_offset[major_axis] = -out.extent[major_axis] / 2; break;
            case CRH_ALIGNMENT_BASELINE:
            case CRH_ALIGNMENT_CENTER: line_offset[major_axis] = -major_extent / 2; break;
            default: line_offset[major_axis] = out.extent[major_axis] / 2 - major_extent; break;
        }
        line_offset[1 - major_axis] -= (out.extent[1 - major_axis] - line_minor_extent) / 2;
        for (Positioned& gp : line.glyphs) {
            gp.position[0] = sign_x * (gp.position[0] + line_offset[0]);
            gp.position[1] = sign_y * (gp.position[1] + line_offset[1]);
        }
    }
    // the macro returns the OUTER `offset` (text.rs:229): the minor alignment only
    out.offset[0] = sign_x * offset[0];
    out.offset[1] = sign_y * offset[1];
    return out;
}

// ---------------------------------------------------------------------------------------------- Path::transform (path.rs:387-439)
// motor2d_to_mat3 (utils.rs:154-165) of Motor [s, e12, e01, e02]: rotation by the rotor (s, e12) followed by the translation
// translation2d() extracts (utils.rs:137-140). DERIVED (the crate's sandwich product is not vendored); exact for translate2d motors,
// which is all text.rs uses.
void motor_to_mat3(const float m[4], float t[3][2]) {
    const float s = m[0], r = m[1];
    const float norm = s * s + r * r;
    const float cs = (s * s - r * r) / norm, sn = (2.0f * s * r) / norm;
    // translation = motor / rotor: [2 * q3, -2 * q2] with q = m * conj(rotor) / |rotor|^2
    const float q2 = (m[2] * s + m[3] * r) / norm, q3 = (m[3] * s - m[2] * r) / norm;
    t[0][0] = cs;
    t[0][1] = sn;
    t[1][0] = -sn;
    t[1][1] = cs;
    t[2][0] = 2.0f * q3;
    t[2][1] = -2.0f * q2;
}
float canonical(float v) { return v == 0.0f ? 0.0f : v; } // SafeFloat::from: -0 -> +0 (safe_float.rs:44-52)
void transform_point(const float t[3][2], float* p) {
    const float x = p[0], y = p[1];
    p[0] = canonical(t[2][0] + x * t[0][0] + y * t[1][0]);
    p[1] = canonical(t[2][1] + x * t[0][1] + y * t[1][1]);
}
void transform_path(PathRec& path, float scale, const float motor[4]) {
    float t[3][2];
    motor_to_mat3(motor, t);
    t[0][0] *= scale;
    t[1][1] *= scale;
    transform_point(t, path.start);
    size_t at = 0;
    for (uint8_t type : path.types) {
        static const int floats[5] = {2, 4, 6, 5, 10}, first_point[5] = {0, 0, 0, 1, 4};
        for (int k = first_point[type]; k < floats[type]; k += 2) transform_point(t, &path.control[at + k]);
        at += floats[type];
    }
}

// utils.rs:83-98, points (1, x, y), planes by the regressive product
bool convex_polygons_overlap(const std::vector<Pt>& a, const std::vector<Pt>& b) {
    const std::vector<Pt>* pair[2][2] = {{&a, &b}, {&b, &a}};
    for (auto& ab : pair) {
        const std::vector<Pt>& p = *ab[0];
        const std::vector<Pt>& q = *ab[1];
        for (size_t i = 0; i < p.size(); ++i) {
            const Pt u = p[(i + 1) % p.size()], v = p[i]; // plane = u v v
            const float l0 = u.y * v.x - u.x * v.y, l1 = 1.0f * v.y - u.y * 1.0f, l2 = u.x * 1.0f - 1.0f * v.x;
            bool separating = true;
            for (const Pt& point : q)
                if ((1.0f * l0 + point.x * l1) + point.y * l2 <= 0.0f) {
                    separating = false;
                    break;
                }
            if (separating) return false;
        }
    }
    return true;
}

crh_status bad(const char* what) {
    crh::set_last_error(what);
    return CRH_ERR_INVALID_ARGUMENT;
}

} // namespace

// ================================================================================================== C ABI
extern "C" {

crh_status crh_font_create(const void* ttf_bytes, size_t n_bytes, crh_font** out) {
    if (!ttf_bytes || !out || n_bytes < 12) return bad("crh_font_create: no font data");
    crh_font* f = new crh_font();
    f->data.assign((const uint8_t*)ttf_bytes, (const uint8_t*)ttf_bytes + n_bytes);
    const Table file{f->data.data(), f->data.size()};
    size_t base = 0;
    if (file.u32(0) == 0x74746366u) { // 'ttcf': face 0 of a collection
        if (!file.has(12, 4) || file.u32(8) == 0) {
            delete f;
            return bad("crh_font_create: empty font collection");
        }
        base = file.u32(12);
    }
    if (!file.has(base, 12)) {
        delete f;
        return bad("crh_font_create: truncated offset table");
    }
    const uint32_t version = file.u32(base);
    if (version != 0x00010000u && version != 0x74727565u && version != 0x4F54544Fu) { // 1.0, 'true', 'OTTO'
        delete f;
        return bad("crh_font_create: not an sfnt font");
    }
    const size_t n_tables = file.u16(base + 4);
    for (size_t i = 0; i < n_tables; ++i) {
        const size_t rec = base + 12 + i * 16;
        if (!file.has(rec, 16)) break;
        const uint32_t tag = file.u32(rec);
        const Table t = file.sub(file.u32(rec + 8), file.u32(rec + 12));
        switch (tag) {
            case 0x68656164u: f->head = t; break; // head
            case 0x6D617870u: f->maxp = t; break; // maxp
            case 0x68686561u: f->hhea = t; break; // hhea
            case 0x686D7478u: f->hmtx = t; break; // hmtx
            case 0x6C6F6361u: f->loca = t; break; // loca
            case 0x676C7966u: f->glyf = t; break; // glyf
            case 0x636D6170u: f->cmap = t; break; // cmap
            case 0x4F532F32u: f->os2 = t; break;  // OS/2
            case 0x6B65726Eu: f->kern = t; break; // kern
            case 0x76686561u: f->vhea = t; break; // vhea
            case 0x766D7478u: f->vmtx = t; break; // vmtx
            default: break;
        }
    }
    if (!f->head.has(0, 54) || !f->hhea.has(0, 36) || !f->maxp.has(0, 6)) {
        delete f;
        return bad("crh_font_create: head / hhea / maxp missing");
    }
    f->units_per_em = f->head.u16(18);
    f->index_to_loc_format = f->head.i16(50);
    f->number_of_glyphs = f->maxp.u16(4);
    f->number_of_h_metrics = f->hhea.u16(34);
    f->number_of_v_metrics = f->vhea.has(0, 36) ? f->vhea.u16(34) : 0;
    if (f->number_of_glyphs == 0) {
        delete f;
        return bad("crh_font_create: no glyphs");
    }
    if (f->kern.has(0, 4) && f->kern.u16(0) == 0 && f->kern.u16(2) > 0 && f->kern.has(4, 6)) { // version 0: first subtable
        const uint16_t length = f->kern.u16(6), coverage = f->kern.u16(8);
        const bool horizontal = coverage & 1u, format0 = (coverage >> 8) == 0;
        if (horizontal && format0 && f->kern.has(4, length) && length >= 14) f->kern_subtable = f->kern.sub(10, (size_t)length - 6);
    }
    *out = f;
    return CRH_OK;
}
void crh_font_destroy(crh_font* font) { delete font; }

crh_status crh_font_get_metrics(const crh_font* f, crh_font_metrics* out) {
    if (!f || !out) return bad("crh_font_get_metrics: null argument");
    memset(out, 0, sizeof(*out));
    out->units_per_em = f->units_per_em;
    out->number_of_glyphs = f->number_of_glyphs;
    out->ascender = ascender(*f);
    out->descender = descender(*f);
    out->line_gap = line_gap(*f);
    out->height = (int16_t)(out->ascender - out->descender);
    out->has_x_height = f->os2.has(0, 88) && f->os2.u16(0) >= 2;
    out->x_height = out->has_x_height ? f->os2.i16(86) : 0;
    out->has_vertical_metrics = f->vhea.has(0, 36);
    if (out->has_vertical_metrics) {
        out->vertical_height = (int16_t)(f->vhea.i16(4) - f->vhea.i16(6));
        out->vertical_line_gap = f->vhea.i16(8);
    }
    out->has_kerning = f->kern_subtable.p != nullptr;
    return CRH_OK;
}
crh_status crh_font_glyph_index(const crh_font* f, uint32_t code_point, uint16_t* glyph_id, uint32_t* found) {
    if (!f || !glyph_id || !found) return bad("crh_font_glyph_index: null argument");
    *glyph_id = 0;
    *found = glyph_index(*f, code_point, *glyph_id) ? 1u : 0u;
    return CRH_OK;
}
crh_status crh_font_glyph_advance(const crh_font* f, uint16_t glyph_id, uint32_t vertical, uint16_t* advance, uint32_t* found) {
    if (!f || !advance || !found) return bad("crh_font_glyph_advance: null argument");
    *advance = 0;
    *found = glyph_advance(*f, glyph_id, vertical != 0, *advance) ? 1u : 0u;
    return CRH_OK;
}
crh_status crh_font_glyph_bounding_box(const crh_font* f, uint16_t glyph_id, int16_t box[4], uint32_t* found) {
    if (!f || !box || !found) return bad("crh_font_glyph_bounding_box: null argument");
    box[0] = box[1] = box[2] = box[3] = 0;
    *found = glyph_bounding_box(*f, glyph_id, box) ? 1u : 0u;
    return CRH_OK;
}
crh_status crh_font_glyphs_kerning(const crh_font* f, uint16_t left, uint16_t right, int16_t* kerning, uint32_t* found) {
    if (!f || !kerning || !found) return bad("crh_font_glyphs_kerning: null argument");
    *kerning = 0;
    *found = (f->kern_subtable.p && glyphs_kerning(*f, left, right, *kerning)) ? 1u : 0u;
    return CRH_OK;
}

crh_status crh_paths_of_glyph(const crh_font* f, uint16_t glyph_id, crh_path_list** out) { // text.rs:97-104
    if (!f || !out) return bad("crh_paths_of_glyph: null argument");
    crh_path_list* list = new crh_path_list();
    if (!outline_glyph(*f, glyph_id, list->paths)) list->paths.clear();
    *out = list;
    return CRH_OK;
}

crh_status crh_text_aligned_positions(const crh_font* f, const crh_text_layout* layout, const uint32_t* text, size_t n_chars, int64_t extent[2], int64_t offset[2],
                                      int64_t* positions, uint64_t* line_ends, uint64_t* line_lengths, uint64_t* n_lines) {
    if (!f || !layout || (!text && n_chars) || !n_lines) return bad("crh_text_aligned_positions: null argument");
    if (layout->orientation > 3 || layout->major_alignment > 3 || layout->minor_alignment > 3) return bad("crh_text_aligned_positions: bad layout enum");
    const Aligned a = aligned_positions(*f, *layout, text, n_chars);
    *n_lines = a.lines.size();
    if (extent) extent[0] = a.extent[0], extent[1] = a.extent[1];
    if (offset) offset[0] = a.offset[0], offset[1] = a.offset[1];
    size_t at = 0;
    for (size_t l = 0; l < a.lines.size(); ++l) {
        if (line_ends) line_ends[l] = a.lines[l].range_end;
        if (line_lengths) line_lengths[l] = a.lines[l].glyphs.size();
        if (positions)
            for (const Positioned& gp : a.lines[l].glyphs) {
                positions[at * 3 + 0] = gp.position[0];
                positions[at * 3 + 1] = gp.position[1];
                positions[at * 3 + 2] = gp.glyph;
                ++at;
            }
    }
    return CRH_OK;
}

crh_status crh_paths_of_text(const crh_font* f, const crh_text_layout* layout, const uint32_t* text, size_t n_chars, const float* clipping_area, size_t n_clip,
                             crh_path_list** out) { // text.rs:236-263
    if (!f || !layout || (!text && n_chars) || !out) return bad("crh_paths_of_text: null argument");
    if (layout->orientation > 3 || layout->major_alignment > 3 || layout->minor_alignment > 3) return bad("crh_paths_of_text: bad layout enum");
    if (!std::isfinite(layout->size)) return CRH_ERR_NON_FINITE;
    const Aligned a = aligned_positions(*f, *layout, text, n_chars);
    const float scale = layout->size / (float)(int16_t)(ascender(*f) - descender(*f));
    std::vector<Pt> clip;
    for (size_t i = 0; clipping_area && i < n_clip; ++i) clip.push_back(Pt{clipping_area[2 * i], clipping_area[2 * i + 1]});
    crh_path_list* list = new crh_path_list();
    for (const Line& line : a.lines)
        for (size_t g = 0; g + 1 < line.glyphs.size(); ++g) {
            const int64_t x = line.glyphs[g].position[0], y = line.glyphs[g].position[1];
            const uint16_t glyph = line.glyphs[g].glyph;
            int16_t box[4];
            if (clipping_area && glyph_bounding_box(*f, glyph, box)) {
                const float aabb[4] = {(float)((int64_t)box[0] + x) * scale, (float)((int64_t)box[1] + y) * scale, (float)((int64_t)box[2] + x) * scale,
                                       (float)((int64_t)box[3] + y) * scale};
                const std::vector<Pt> polygon = {{aabb[0], aabb[1]}, {aabb[0], aabb[3]}, {aabb[2], aabb[3]}, {aabb[2], aabb[1]}}; // utils.rs:73-80
                if (!convex_polygons_overlap(polygon, clip)) continue;
            }
            const float v[2] = {(float)x * scale, (float)y * scale};
            const float motor[4] = {1.0f, 0.0f, -0.5f * v[1], 0.5f * v[0]}; // translate2d, utils.rs:127-129
            std::vector<PathRec> paths;
            if (!outline_glyph(*f, glyph, paths)) paths.clear();
            for (PathRec& path : paths) {
                transform_path(path, scale, motor);
                list->paths.push_back(std::move(path));
            }
        }
    *out = list;
    return CRH_OK;
}

crh_status crh_path_list_transform(crh_path_list* list, float scale, const float motor[4]) {
    if (!list || !motor) return bad("crh_path_list_transform: null argument");
    for (PathRec& path : list->paths) transform_path(path, scale, motor);
    for (const PathRec& path : list->paths) {
        if (!std::isfinite(path.start[0]) || !std::isfinite(path.start[1])) return CRH_ERR_NON_FINITE; // SafeFloat::from panics
        for (float v : path.control)
            if (!std::isfinite(v)) return CRH_ERR_NON_FINITE;
    }
    return CRH_OK;
}

crh_status crh_path_list_view(const crh_path_list* const_list, crh_path_batch* out) {
    if (!const_list || !out) return bad("crh_path_list_view: null argument");
    crh_path_list* list = const_cast<crh_path_list*>(const_list); // the flattened arrays are a cache owned by the list
    list->shape_path_begin = {0u, (uint32_t)list->paths.size()};
    list->shape_dynamic_begin = {0u, 0u};
    list->path_segment_begin.assign(1, 0u);
    list->path_start.clear();
    list->control.clear();
    list->types.clear();
    list->path_stroke.assign(list->paths.size(), -1);
    for (const PathRec& path : list->paths) {
        list->path_start.push_back(path.start[0]);
        list->path_start.push_back(path.start[1]);
        list->types.insert(list->types.end(), path.types.begin(), path.types.end());
        list->control.insert(list->control.end(), path.control.begin(), path.control.end());
        list->path_segment_begin.push_back((uint32_t)list->types.size());
    }
    memset(out, 0, sizeof(*out));
    out->n_shapes = 1;
    out->shape_path_begin = list->shape_path_begin.data();
    out->n_paths = (uint32_t)list->paths.size();
    out->path_segment_begin = list->path_segment_begin.data();
    out->path_start = list->path_start.data();
    out->path_stroke_options = list->path_stroke.data();
    out->n_segments = (uint32_t)list->types.size();
    out->segment_types = list->types.data();
    out->control_data = list->control.data();
    out->n_control_floats = (uint32_t)list->control.size();
    out->shape_dynamic_begin = list->shape_dynamic_begin.data();
    return CRH_OK;
}
void crh_path_list_destroy(crh_path_list* list) { delete list; }

} // extern "C"
