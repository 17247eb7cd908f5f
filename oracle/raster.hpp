// oracle/raster.hpp — TEST INFRASTRUCTURE (CPU oracle). Software restatement of the reference's
// stencil-then-cover GPU passes: Shape::render(Stencil) / render(Color) (renderer.rs:267-355), their
// fixed-function state (renderer.rs:565-582, :736-754) and the fragment entry points of
// src/shaders.wgsl:155-309. The reference runs these on a hardware rasterizer through wgpu; there are
// no reference pixels anywhere upstream, so this file IS the pixel spec the HIP tile rasterizer is
// checked against. Arithmetic is spelled out operation by operation (explicit
// fmaf, no contraction) so a GPU kernel that follows the same recipe is bit-identical.
//
// Spec summary
//  * framebuffer: y down, row 0 = top; fx = (ndc.x*0.5+0.5)*W, fy = (0.5-ndc.y*0.5)*H; affine instance
//    transforms only (clip.w == 1), z ignored (no depth test on this path).
//  * samples: 1x = pixel centre; 4x = (6,2),(14,6),(2,10),(10,14)/16 (D3D/Vulkan standard pattern).
//  * coverage: top-left rule on float edge functions. Every edge is evaluated from its endpoints in
//    canonical (lexicographic) order, tile-relative (16x16 tiles), so the two triangles sharing an edge
//    see exactly negated values: strips are watertight by construction.
//  * attributes: affine planes through the three vertex values (== perspective interpolation at w == 1),
//    evaluated at the sample (shaders.wgsl:35,40,47,52,58).
//  * stencil: stroke passes = Equal(0) -> IncrementWrap (set-once); fill passes = front(ccw on screen)
//    IncrementWrap / back DecrementWrap, modulo 2^winding_counter_bits; cover = where winding != 0 blend
//    premultiplied "over", and zero the winding of every sample inside the hull strip.
//  * colour: f32 per sample, one quantisation to RGBA8 at resolve (box average).
#pragma once
#include "tessellate.hpp"

namespace oracle {

constexpr int TILE = 16;

struct Frame {
    uint32_t width = 0, height = 0, samples = 1, winding_mask = 15;
    uint32_t clip_mask = 0;  // clip nesting counter bits, above the winding bits (renderer.rs:565)
    uint32_t reference = 0;  // stencil reference = clip_depth << winding_counter_bits (Renderer::set_clip_depth, renderer.rs:932-938)
    std::vector<uint8_t> winding; // [y][x][s] the whole stencil byte: clip nesting counter | winding counter
    std::vector<float> color;     // [y][x][s][4] premultiplied
    std::vector<std::vector<float>> alpha_layers; // [layer][y][x][s] saved alpha (R8 targets of renderer.rs:892-927, kept in f32 like the colour)
    void create(uint32_t w, uint32_t h, uint32_t s, uint32_t winding_bits, uint32_t clip_bits = 0, uint32_t n_alpha_layers = 0) {
        width = w;
        height = h;
        samples = s;
        winding_mask = (1u << winding_bits) - 1u;
        clip_mask = ((1u << clip_bits) - 1u) << winding_bits;
        reference = 0;
        winding.assign((size_t)w * h * s, 0);
        color.assign((size_t)w * h * s * 4, 0.0f);
        alpha_layers.assign(n_alpha_layers, std::vector<float>((size_t)w * h * s, 0.0f));
    }
    void clear() {
        std::fill(winding.begin(), winding.end(), 0);
        std::fill(color.begin(), color.end(), 0.0f);
    }
};

inline void sample_offset(uint32_t samples, uint32_t s, float& ox, float& oy) {
    if (samples == 1) {
        ox = 0.5f;
        oy = 0.5f;
        return;
    }
    static const float X[4] = {0.375f, 0.875f, 0.125f, 0.625f};
    static const float Y[4] = {0.125f, 0.375f, 0.625f, 0.875f};
    ox = X[s & 3];
    oy = Y[s & 3];
}

// vertex stage: clip = M * (x, y, 0, 1) (shaders.wgsl:66-74), then the viewport transform.
inline void to_framebuffer(const float m[16], float w, float h, const float p[2], float out[2]) {
    const float cx = (m[0] * p[0] + m[4] * p[1]) + m[12];
    const float cy = (m[1] * p[0] + m[5] * p[1]) + m[13];
    out[0] = (cx * 0.5f + 0.5f) * w;
    out[1] = (0.5f - cy * 0.5f) * h;
}

struct EdgeSetup {
    float lo[2], bx, nay; // canonical endpoints: E = bx*(y - lo.y) + nay*(x - lo.x)
    bool flip, topleft;
};
struct TriangleSetup {
    bool valid;
    bool front; // counter-clockwise on screen
    EdgeSetup e[3];
    int x0, x1, y0, y1; // inclusive pixel range
    float v0[2], d1[2], d2[2], inv_det;
};

inline bool lex_less(const float a[2], const float b[2]) { return a[0] < b[0] || (a[0] == b[0] && a[1] < b[1]); }

inline TriangleSetup setup_triangle(const float v[3][2], int width, int height) {
    TriangleSetup t;
    t.valid = false;
    for (int k = 0; k < 2; ++k) {
        t.v0[k] = v[0][k];
        t.d1[k] = v[1][k] - v[0][k];
        t.d2[k] = v[2][k] - v[0][k];
    }
    const float det = t.d1[0] * t.d2[1] - t.d2[0] * t.d1[1];
    if (!(det != 0.0f) || !(det == det) || std::isinf(det)) return t;
    t.inv_det = 1.0f / det;
    t.front = det < 0.0f; // y-down cross < 0 == counter-clockwise on screen (FrontFace::Ccw, renderer.rs:477)
    // normalise to clockwise-in-y-down for the edge walk
    const float* n[3] = {v[0], det < 0.0f ? v[2] : v[1], det < 0.0f ? v[1] : v[2]};
    for (int i = 0; i < 3; ++i) {
        const float* a = n[i];
        const float* b = n[(i + 1) % 3];
        EdgeSetup& e = t.e[i];
        const float dx = b[0] - a[0], dy = b[1] - a[1];
        e.topleft = dy < 0.0f || (dy == 0.0f && dx > 0.0f);
        e.flip = !lex_less(a, b);
        const float* lo = e.flip ? b : a;
        const float* hi = e.flip ? a : b;
        e.lo[0] = lo[0];
        e.lo[1] = lo[1];
        e.bx = hi[0] - lo[0];
        e.nay = -(hi[1] - lo[1]);
    }
    float minx = std::fmin(v[0][0], std::fmin(v[1][0], v[2][0])), maxx = std::fmax(v[0][0], std::fmax(v[1][0], v[2][0]));
    float miny = std::fmin(v[0][1], std::fmin(v[1][1], v[2][1])), maxy = std::fmax(v[0][1], std::fmax(v[1][1], v[2][1]));
    if (!(minx == minx && maxx == maxx && miny == miny && maxy == maxy)) return t;
    // inclusive pixel range: clamp in float first (coordinates may exceed the int range), floor, THEN compare — a sliver that begins at
    // x = W - 0.8 still owns the last pixel column although its clamped float range is empty
    t.x0 = (int)std::floor(std::fmin(std::fmax(minx, 0.0f), (float)width));
    t.x1 = (int)std::floor(std::fmax(std::fmin(maxx, (float)(width - 1)), -1.0f));
    t.y0 = (int)std::floor(std::fmin(std::fmax(miny, 0.0f), (float)height));
    t.y1 = (int)std::floor(std::fmax(std::fmin(maxy, (float)(height - 1)), -1.0f));
    if (t.x0 > t.x1 || t.y0 > t.y1) return t;
    t.valid = true;
    return t;
}

// An attribute plane through (v0,a0), (v1,a1), (v2,a2), made tile-relative.
struct AttrPlane {
    float a0, gx, gy;
};
inline AttrPlane setup_attribute(const TriangleSetup& t, float a0, float a1, float a2) {
    const float da1 = a1 - a0, da2 = a2 - a0;
    AttrPlane p;
    p.a0 = a0;
    p.gx = (da1 * t.d2[1] - da2 * t.d1[1]) * t.inv_det;
    p.gy = (da2 * t.d1[0] - da1 * t.d2[0]) * t.inv_det;
    return p;
}
inline float attribute_tile_constant(const TriangleSetup& t, const AttrPlane& p, float tx0, float ty0) {
    return (p.a0 + (tx0 - t.v0[0]) * p.gx) + (ty0 - t.v0[1]) * p.gy;
}

// Rasterise one triangle; `frag(attr_values) -> bool keep`, `stencil(sample_index_in_frame, front)` applies the op.
template <int NATTR, typename Frag, typename Stencil>
inline void raster_triangle(Frame& f, const float v[3][2], const float attr[3][4], Frag frag, Stencil stencil) {
    const TriangleSetup t = setup_triangle(v, (int)f.width, (int)f.height);
    if (!t.valid) return;
    AttrPlane planes[NATTR > 0 ? NATTR : 1];
    for (int a = 0; a < NATTR; ++a) planes[a] = setup_attribute(t, attr[0][a], attr[1][a], attr[2][a]);
    for (int ty = t.y0 / TILE; ty <= t.y1 / TILE; ++ty) {
        for (int tx = t.x0 / TILE; tx <= t.x1 / TILE; ++tx) {
            const float tx0 = (float)(tx * TILE), ty0 = (float)(ty * TILE);
            float c[3];
            for (int i = 0; i < 3; ++i) c[i] = t.e[i].bx * (ty0 - t.e[i].lo[1]) + t.e[i].nay * (tx0 - t.e[i].lo[0]);
            float ac[NATTR > 0 ? NATTR : 1];
            for (int a = 0; a < NATTR; ++a) ac[a] = attribute_tile_constant(t, planes[a], tx0, ty0);
            const int py0 = std::max(t.y0, ty * TILE), py1 = std::min(t.y1, ty * TILE + TILE - 1);
            const int px0 = std::max(t.x0, tx * TILE), px1 = std::min(t.x1, tx * TILE + TILE - 1);
            for (int py = py0; py <= py1; ++py) {
                for (int px = px0; px <= px1; ++px) {
                    for (uint32_t s = 0; s < f.samples; ++s) {
                        float ox, oy;
                        sample_offset(f.samples, s, ox, oy);
                        const float rx = (float)(px - tx * TILE) + ox, ry = (float)(py - ty * TILE) + oy;
                        bool inside = true;
                        for (int i = 0; i < 3 && inside; ++i) {
                            float e = fmaf(rx, t.e[i].nay, fmaf(ry, t.e[i].bx, c[i]));
                            if (t.e[i].flip) e = -e;
                            inside = e > 0.0f || (e == 0.0f && t.e[i].topleft);
                        }
                        if (!inside) continue;
                        float values[NATTR > 0 ? NATTR : 1];
                        for (int a = 0; a < NATTR; ++a) values[a] = fmaf(ry, planes[a].gy, fmaf(rx, planes[a].gx, ac[a]));
                        if (!frag(values)) continue;
                        stencil(((size_t)py * f.width + px) * f.samples + s, t.front);
                    }
                }
            }
        }
    }
}

// ---- shaders.wgsl:165-231 ------------------------------------------------------------------------------------
inline bool cap(float x, float y, uint32_t cap_type) { // shaders.wgsl:165-189
    switch (cap_type & 15u) {
        case 0: return y > 0.5f;                  // Square
        case 1: return x * x + y * y < 0.25f;     // Round: dot(texcoord, texcoord) < 0.25
        case 2: return 0.5f - y > std::fabs(x);   // Out
        case 3: return y < std::fabs(x);          // In
        case 4: return 0.5f - y > x;              // Right
        case 5: return y - 0.5f < x;              // Left
        default: return y < 0.0f;                 // Butt
    }
}
inline bool joint(float radius, bool bevel, uint32_t join) { // shaders.wgsl:191-203
    switch (join) {
        case 1: return bevel;
        case 2: return radius <= 0.5f;
        default: return true;
    }
}
inline bool stroke_dashed(const crh_dynamic_stroke_descriptor& d, float tx, float ty) { // shaders.wgsl:205-231
    const uint32_t last_interval_index = d.count_dashed_join >> 3;
    const float pattern_length = d.gap_end[last_interval_index & 3];
    uint32_t interval_index = 0;
    float position_in_pattern = crh_wgsl_mod(ty - d.phase, pattern_length);
    if (position_in_pattern < 0.0f) position_in_pattern = position_in_pattern + pattern_length;
    float gap_end;
    for (;;) {
        gap_end = d.gap_end[interval_index & 3] - position_in_pattern;
        if (gap_end >= 0.0f || interval_index >= last_interval_index) break;
        interval_index = interval_index + 1;
    }
    const float gap_start = position_in_pattern - d.gap_start[interval_index & 3];
    if (gap_start > 0.0f) {
        const uint32_t caps = d.caps >> (interval_index * 8u);
        const bool start_cap = cap(tx, gap_start, caps >> 4);
        const bool end_cap = cap(tx, gap_end, caps);
        return start_cap || end_cap;
    }
    return true;
}

struct StripWalker { // triangle i of a strip: even (i, i+1, i+2), odd (i, i+2, i+1); provoking vertex = i
    static void triangle(size_t i, size_t idx[3]) {
        idx[0] = i;
        idx[1] = (i & 1) ? i + 2 : i + 1;
        idx[2] = (i & 1) ? i + 1 : i + 2;
    }
};

inline uint8_t wrap_add(uint8_t old, int delta, uint32_t mask) { return (uint8_t)((old & ~mask) | ((uint32_t)(old + delta) & mask)); }

// Shape::render(Stencil) (renderer.rs:275-336) for one instance
inline void render_stencil(Frame& f, const Shape& shape, const float m[16]) {
    const float W = (float)f.width, H = (float)f.height;
    static const crh_dynamic_stroke_descriptor zero_descriptor = {};
    auto descriptor = [&](uint32_t path_index) -> const crh_dynamic_stroke_descriptor& {
        // out-of-range reads of a storage buffer are clamped/zero in WebGPU; never happens for validated input
        return path_index < shape.stroke_buffer.size() ? shape.stroke_buffer[path_index] : zero_descriptor;
    };
    const uint32_t read_mask = f.clip_mask | f.winding_mask;
    auto stroke_stencil = [&](size_t si, bool) { // Equal(ref) -> IncrementWrap, both faces, write mask = winding (renderer.rs:571-576)
        if ((f.winding[si] & read_mask) == (f.reference & read_mask)) f.winding[si] = wrap_add(f.winding[si], 1, f.winding_mask);
    };
    auto fill_stencil = [&](size_t si, bool front) { // LessEqual(ref <= stencil) -> front Increment / back Decrement (renderer.rs:577-582)
        if ((f.reference & read_mask) <= (f.winding[si] & read_mask)) f.winding[si] = wrap_add(f.winding[si], front ? 1 : -1, f.winding_mask);
    };
    // 1. stroke line strips (renderer.rs:278-287, shaders.wgsl:268-285)
    {
        const auto& verts = shape.stroke.line_vertices;
        const auto& idx = shape.stroke.line_indices;
        size_t run_start = 0;
        for (size_t k = 0; k <= idx.size(); ++k) {
            if (k == idx.size() || idx[k] == 0xFFFF) {
                for (size_t i = 0; run_start + i + 2 < k; ++i) {
                    size_t tri[3];
                    StripWalker::triangle(i, tri);
                    float v[3][2], attr[3][4];
                    for (int c = 0; c < 3; ++c) {
                        const Vertex2f1i& vx = verts[idx[run_start + tri[c]]];
                        to_framebuffer(m, W, H, vx.p, v[c]);
                        attr[c][0] = vx.t[0];
                        attr[c][1] = vx.t[1];
                    }
                    const Vertex2f1i& provoking = verts[idx[run_start + i]];
                    const uint32_t flat_u = provoking.u;
                    const float end_texcoord_y = provoking.t[1];
                    const crh_dynamic_stroke_descriptor& d = descriptor(flat_u & 65535u);
                    raster_triangle<2>(
                        f, v, attr,
                        [&](const float* t) {
                            if ((d.count_dashed_join & 4u) != 0u) return stroke_dashed(d, t[0], t[1]);
                            if ((flat_u & 65536u) != 0u) return cap(t[0], t[1] - end_texcoord_y, d.caps >> 4);
                            if (t[1] < 0.0f) return cap(t[0], -t[1], d.caps);
                            return true;
                        },
                        stroke_stencil);
                }
                run_start = k + 1;
            }
        }
    }
    // 2. stroke joint strips (renderer.rs:288-302, shaders.wgsl:287-300)
    {
        const auto& verts = shape.stroke.joint_vertices;
        const auto& idx = shape.stroke.joint_indices;
        const float TAU = crh_acosf(-1.0f) * 2.0f;
        size_t run_start = 0;
        for (size_t k = 0; k <= idx.size(); ++k) {
            if (k == idx.size() || idx[k] == 0xFFFF) {
                for (size_t i = 0; run_start + i + 2 < k; ++i) {
                    size_t tri[3];
                    StripWalker::triangle(i, tri);
                    float v[3][2], attr[3][4];
                    for (int c = 0; c < 3; ++c) {
                        const Vertex3f1i& vx = verts[idx[run_start + tri[c]]];
                        to_framebuffer(m, W, H, vx.p, v[c]);
                        attr[c][0] = vx.t[0];
                        attr[c][1] = vx.t[1];
                        attr[c][2] = vx.t[2];
                    }
                    const uint32_t flat_u = verts[idx[run_start + i]].u;
                    const crh_dynamic_stroke_descriptor& d = descriptor(flat_u & 65535u);
                    raster_triangle<3>(
                        f, v, attr,
                        [&](const float* t) {
                            const float radius = std::sqrt(t[0] * t[0] + t[1] * t[1]);
                            bool fill = joint(radius, (flat_u & 65536u) != 0u, d.count_dashed_join & 3u);
                            if (fill && (d.count_dashed_join & 4u) != 0u) fill = stroke_dashed(d, radius, t[2] + crh_atan2f(t[1], t[0]) / TAU);
                            return fill;
                        },
                        stroke_stencil);
                }
                run_start = k + 1;
            }
        }
    }
    // 3. solid strips (renderer.rs:304-318, shaders.wgsl:233-234)
    {
        const auto& verts = shape.fill.solid_vertices;
        const auto& idx = shape.fill.solid_indices;
        size_t run_start = 0;
        for (size_t k = 0; k <= idx.size(); ++k) {
            if (k == idx.size() || idx[k] == 0xFFFF) {
                for (size_t i = 0; run_start + i + 2 < k; ++i) {
                    size_t tri[3];
                    StripWalker::triangle(i, tri);
                    float v[3][2], attr[3][4] = {};
                    for (int c = 0; c < 3; ++c) to_framebuffer(m, W, H, verts[idx[run_start + tri[c]]].p, v[c]);
                    raster_triangle<0>(
                        f, v, attr, [](const float*) { return true; }, fill_stencil);
                }
                run_start = k + 1;
            }
        }
    }
    // 4. curve triangle lists (renderer.rs:319-335, shaders.wgsl:236-266)
    for (size_t i = 0; i + 2 < shape.fill.integral_quadratic_vertices.size(); i += 3) {
        float v[3][2], attr[3][4] = {};
        for (int c = 0; c < 3; ++c) {
            const Vertex2f& vx = shape.fill.integral_quadratic_vertices[i + c];
            to_framebuffer(m, W, H, vx.p, v[c]);
            attr[c][0] = vx.w[0];
            attr[c][1] = vx.w[1];
        }
        raster_triangle<2>(
            f, v, attr, [](const float* w) { return w[0] * w[0] - w[1] <= 0.0f; }, fill_stencil);
    }
    for (size_t i = 0; i + 2 < shape.fill.integral_cubic_vertices.size(); i += 3) {
        float v[3][2], attr[3][4] = {};
        for (int c = 0; c < 3; ++c) {
            const Vertex3f& vx = shape.fill.integral_cubic_vertices[i + c];
            to_framebuffer(m, W, H, vx.p, v[c]);
            for (int a = 0; a < 3; ++a) attr[c][a] = vx.w[a];
        }
        raster_triangle<3>(
            f, v, attr, [](const float* w) { return w[0] * w[0] * w[0] - w[1] * w[2] <= 0.0f; }, fill_stencil);
    }
    for (size_t i = 0; i + 2 < shape.fill.rational_quadratic_vertices.size(); i += 3) {
        float v[3][2], attr[3][4] = {};
        for (int c = 0; c < 3; ++c) {
            const Vertex3f& vx = shape.fill.rational_quadratic_vertices[i + c];
            to_framebuffer(m, W, H, vx.p, v[c]);
            for (int a = 0; a < 3; ++a) attr[c][a] = vx.w[a];
        }
        raster_triangle<3>(
            f, v, attr, [](const float* w) { return w[0] * w[0] - w[1] * w[2] <= 0.0f; }, fill_stencil);
    }
    for (size_t i = 0; i + 2 < shape.fill.rational_cubic_vertices.size(); i += 3) {
        float v[3][2], attr[3][4] = {};
        for (int c = 0; c < 3; ++c) {
            const Vertex4f& vx = shape.fill.rational_cubic_vertices[i + c];
            to_framebuffer(m, W, H, vx.p, v[c]);
            for (int a = 0; a < 4; ++a) attr[c][a] = vx.w[a];
        }
        raster_triangle<4>(
            f, v, attr, [](const float* w) { return w[0] * w[0] * w[0] - w[1] * w[2] * w[3] <= 0.0f; }, fill_stencil);
    }
}

// The cover operations: Shape::render(Clip | UnClip | Color | SaveAlphaContext | ScaleAlphaContext | RestoreAlphaContext)
// (renderer.rs:338-354) draw the hull strip with the fixed-function state of renderer.rs:692-754 / :761-861 and the fragment stages
// shaders.wgsl:304-355, for one instance. `op` = crh_render_op.
inline void render_cover(Frame& f, const Shape& shape, const float m[16], const float rgba[4], uint32_t op, uint32_t alpha_layer) {
    const float W = (float)f.width, H = (float)f.height;
    const float src[4] = {rgba[0] * rgba[3], rgba[1] * rgba[3], rgba[2] * rgba[3], rgba[3]};
    const float one_minus_a = 1.0f - src[3];
    const uint32_t read_mask = f.clip_mask | f.winding_mask;
    const uint32_t ref = f.reference;
    std::vector<float>* layer = alpha_layer < f.alpha_layers.size() ? &f.alpha_layers[alpha_layer] : nullptr;
    const auto& hull = shape.convex_hull; // already in strip order
    for (size_t i = 0; i + 2 < hull.size(); ++i) {
        size_t tri[3];
        StripWalker::triangle(i, tri);
        float v[3][2], attr[3][4] = {};
        for (int c = 0; c < 3; ++c) to_framebuffer(m, W, H, hull[tri[c]].p, v[c]);
        raster_triangle<0>(
            f, v, attr, [](const float*) { return true; },
            [&](size_t si, bool) {
                const uint32_t st = f.winding[si];
                float* dst = &f.color[si * 4];
                switch (op) {
                    case CRH_OP_COLOR: // Less(ref < stencil): blend premultiplied "over"; pass -> Zero, fail -> Zero on the winding bits (renderer.rs:747-752)
                        if ((ref & read_mask) < (st & read_mask))
                            for (int c = 0; c < 4; ++c) dst[c] = src[c] + dst[c] * one_minus_a;
                        f.winding[si] = (uint8_t)(st & ~f.winding_mask);
                        break;
                    case CRH_OP_CLIP: // NotEqual on the winding bits -> Replace(ref) on clip | winding (renderer.rs:703-708)
                        if ((ref & f.winding_mask) != (st & f.winding_mask)) f.winding[si] = (uint8_t)((st & ~read_mask) | (ref & read_mask));
                        break;
                    case CRH_OP_UNCLIP: // Less on the clip bits (ref < stencil) -> Replace(ref) on clip | winding (renderer.rs:722-727)
                        if ((ref & f.clip_mask) < (st & f.clip_mask)) f.winding[si] = (uint8_t)((st & ~read_mask) | (ref & read_mask));
                        break;
                    default: // the alpha-context covers: LessEqual(ref <= stencil), stencil write mask 0 (renderer.rs:761-766)
                        if (layer && (ref & read_mask) <= (st & read_mask)) {
                            if (op == CRH_OP_SAVE_ALPHA_CONTEXT) { // shaders.wgsl:326-331: the layer receives the frame's alpha
                                (*layer)[si] = dst[3];
                            } else if (op == CRH_OP_SCALE_ALPHA_CONTEXT) { // src = (0,0,0,1-a): alpha' = src.a * One + dst.a * (1 - src.a), renderer.rs:803-828
                                const float sa = 1.0f - rgba[3];
                                dst[3] = sa + dst[3] * (1.0f - sa);
                            } else { // RestoreAlphaContext: src.a = (1 - saved)(1 - a); alpha' = dst.a * One - src.a * One, renderer.rs:829-861
                                const float sa = (1.0f - (*layer)[si]) * (1.0f - rgba[3]);
                                dst[3] = dst[3] - sa;
                            }
                        }
                        break;
                }
            });
    }
}
inline void render_color(Frame& f, const Shape& shape, const float m[16], const float rgba[4]) { render_cover(f, shape, m, rgba, CRH_OP_COLOR, 0); }

// MSAA resolve (box average) + RGBA8 unorm
inline void resolve_rgba8(const Frame& f, uint8_t* out) {
    const float inv = 1.0f / (float)f.samples;
    for (size_t p = 0; p < (size_t)f.width * f.height; ++p) {
        for (int c = 0; c < 4; ++c) {
            float sum = 0.0f;
            for (uint32_t s = 0; s < f.samples; ++s) sum = sum + f.color[(p * f.samples + s) * 4 + c];
            float x = sum * inv;
            x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
            if (!(x == x)) x = 0.0f;
            out[p * 4 + c] = (uint8_t)(int)(x * 255.0f + 0.5f);
        }
    }
}

} // namespace oracle
