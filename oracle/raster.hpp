// oracle/raster.hpp — TEST INFRASTRUCTURE (CPU oracle). Software restatement of the reference's
// stencil-then-cover GPU passes: Shape::render(Stencil) / render(Color) (renderer.rs:267-355), their
// fixed-function state (renderer.rs:565-582, :736-754) and the fragment entry points of
// src/shaders.wgsl:155-309. The reference runs these on a hardware rasterizer through wgpu; there are
// no reference pixels anywhere upstream, so this file IS the pixel spec the HIP tile rasterizer is
// checked against. Arithmetic is spelled out operation by operation (explicit
// fmaf, no contraction) so a GPU kernel that follows the same recipe is bit-identical.
//
// Spec summary
//  * framebuffer: y down, row 0 = top; fx = (ndc.x*0.5+0.5)*W, fy = (0.5-ndc.y*0.5)*H.
//  * instances: a "plain" instance transform (clip.w == 1 and clip.z a constant in [0, 1]: m3 = m7 = m2 = m6 = 0, m15 = 1) takes the
//    affine path below; every other 4x4 matrix (perspective_projection of utils.rs:181-192 times a placement, main.rs:162-202) takes
//    the projective path: homogeneous edge functions (no clipping: the part of a triangle behind the eye fails its own edge tests),
//    per-sample near / far test 0 <= z/w <= 1 (unclipped_depth: false, renderer.rs:478), attributes interpolated perspective-correct
//    at the sample (shaders.wgsl:35-58) as (a/w plane) * (1 / (1/w plane)).
//  * depth: only the colour cover tests / writes depth (renderer.rs:743-745; every other pipeline is Always / no write); the depth
//    attachment is f32 per sample here (the reference's Depth24Plus has implementation-defined precision); stencil depth_fail_op is
//    Keep (renderer.rs:442), so a sample that passes the stencil test but fails the depth test keeps its winding.
//  * samples: 1x = pixel centre; 4x = (6,2),(14,6),(2,10),(10,14)/16 (D3D/Vulkan standard pattern).
//  * coverage: top-left rule on float edge functions. Every edge is evaluated from its endpoints in
//    canonical (lexicographic) order, tile-relative (16x16 tiles), so the two triangles sharing an edge
//    see exactly negated values: strips are watertight by construction.
//  * attributes: affine planes through the three vertex values (== perspective interpolation at w == 1),
//    evaluated at the sample (shaders.wgsl:35,40,47,52,58).
//  * stencil: stroke passes = Equal(0) -> IncrementWrap (set-once); fill passes = front(ccw on screen)
//    IncrementWrap / back DecrementWrap, modulo 2^winding_counter_bits; cover = where winding != 0 blend
//    premultiplied "over", and zero the winding of every sample inside the hull strip.
//  * strips are walked by POSITION (vertex n of a strip is the n-th vertex the builder appended to it), not through the u16 index
//    values: identical as long as a sub-buffer holds at most 65 535 vertices; beyond that the reference's indices wrap (`as u16`,
//    fill.rs:363, stroke.rs:108,128) and a hardware rasterizer would draw whatever the wrapped values point at (and cut strips at a
//    legitimate 65 535). The emitted index BYTES keep the wrap (they are compared bit for bit); the pixels show the intended geometry.
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
    // Configuration::{cull_mode, depth_compare, depth_write_enabled} (renderer.rs:383-390) of the colour cover; depth attachment [y][x][s]
    uint32_t cull_mode = CRH_CULL_NONE, depth_compare = CRH_COMPARE_ALWAYS, depth_write = 0;
    std::vector<float> depth;
    // The colour attachment is an Rgba8Unorm texture the blender reads and writes (CRH_FORMAT_RGBA8_ATTACHMENT; renderer.rs:736-754 blends into the
    // caller's target, examples/showcase/main.rs:205-215): every component a cover writes is rounded to 8 bits where it is written. false: f32 colours
    // for the whole pass, one rounding at resolve (CRH_FORMAT_RGBA8).
    bool attachment8 = attachment8_default();
    static bool& attachment8_default() {
        static bool value = false; // (set by oracle_set_attachment8 for the frames created afterwards: test infrastructure)
        return value;
    }
    static float unorm8(float v) { // resolve_rgba8's rounding, read back as value / 255
        float x = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        if (!(x == x)) x = 0.0f;
        return (float)(uint32_t)(int)(x * 255.0f + 0.5f) * (1.0f / 255.0f);
    }
    void create_depth(float clear_value) { depth.assign((size_t)width * height * samples, clear_value); }
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
    return fmaf(ty0 - t.v0[1], p.gy, fmaf(tx0 - t.v0[0], p.gx, p.a0));
}

// Rasterise one triangle of a plain instance; `frag(attr_values) -> bool keep`, `stencil(sample_index_in_frame, front, depth)` applies the op.
template <int NATTR, typename Frag, typename Stencil>
inline void raster_plain(Frame& f, const float v[3][2], const float attr[3][4], float depth, Frag frag, Stencil stencil) {
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
                            float e = fmaf(ry, t.e[i].bx, fmaf(rx, t.e[i].nay, c[i])); // the column term first: shared by the rows of a pixel column
                            if (t.e[i].flip) e = -e;
                            inside = e > 0.0f || (e == 0.0f && t.e[i].topleft);
                        }
                        if (!inside) continue;
                        float values[NATTR > 0 ? NATTR : 1];
                        for (int a = 0; a < NATTR; ++a) values[a] = fmaf(ry, planes[a].gy, fmaf(rx, planes[a].gx, ac[a]));
                        if (!frag(values)) continue;
                        stencil(((size_t)py * f.width + px) * f.samples + s, t.front, depth);
                    }
                }
            }
        }
    }
}

// ---- projective instances -------------------------------------------------------------------------------------
// clip = M * (x, y, 0, 1) (shaders.wgsl:66-74) kept homogeneous, with the viewport transform applied to x and y:
//   X = (clip.x*0.5 + clip.w*0.5) * W,  Y = (clip.w*0.5 - clip.y*0.5) * H,  Z = clip.z,  Wc = clip.w   (screen point = (X/Wc, Y/Wc))
struct ClipVertex {
    float X, Y, Z, W;
};
inline bool is_plain_instance(const float m[16]) {
    return m[3] == 0.0f && m[7] == 0.0f && m[15] == 1.0f && m[2] == 0.0f && m[6] == 0.0f && m[14] >= 0.0f && m[14] <= 1.0f;
}
inline ClipVertex to_clip(const float m[16], float w, float h, const float p[2]) {
    const float cx = (m[0] * p[0] + m[4] * p[1]) + m[12];
    const float cy = (m[1] * p[0] + m[5] * p[1]) + m[13];
    const float cz = (m[2] * p[0] + m[6] * p[1]) + m[14];
    const float cw = (m[3] * p[0] + m[7] * p[1]) + m[15];
    return ClipVertex{(cx * 0.5f + cw * 0.5f) * w, (cw * 0.5f - cy * 0.5f) * h, cz, cw};
}
inline bool lex_less3(const ClipVertex& a, const ClipVertex& b) {
    return a.X < b.X || (a.X == b.X && (a.Y < b.Y || (a.Y == b.Y && a.W < b.W)));
}
// A screen point p = (x, y, 1) is inside the triangle iff p = sum lambda_i P_i with all lambda_i >= 0 (P_i = (X, Y, Wc)); where the
// triangle is behind the eye the lambdas are all <= 0, so the three edge tests also reject the antipodal image: nothing is clipped.
// lambda_0 * det = det[p, P1, P2] = A x + B y + C with (A, B, C) = P1 x P2: the same E = bx*(y - lo.y) + nay*(x - lo.x) form as the
// plain path, nay = A, bx = B, anchored at the projection of an endpoint. Coefficients come from the endpoints in canonical order, so
// the two triangles sharing an edge see exactly negated values (watertight).
struct ProjectiveSetup {
    bool valid, front;
    EdgeSetup e[3];
    int x0, x1, y0, y1;
    float anchor[2];  // the projection of vertex k, origin of the attribute planes
    int k;
    float lx[2], ly[2], lw[3]; // the other two vertices relative to the anchor, in cyclic order after k: (X - ax*Wc, Y - ay*Wc), and Wc of k, k+1, k+2
    float inv_det;
};
inline ProjectiveSetup setup_projective(const ClipVertex P[3], int width, int height) {
    ProjectiveSetup t;
    t.valid = false;
    int k = -1;
    for (int i = 2; i >= 0; --i)
        if (P[i].W > 0.0f) k = i; // the first vertex in front of the eye
    if (k < 0) return t;
    for (int i = 0; i < 3; ++i)
        if (!(std::isfinite(P[i].X) && std::isfinite(P[i].Y) && std::isfinite(P[i].Z) && std::isfinite(P[i].W))) return t;
    // orientation: det[P0; P1; P2] = w0 w1 w2 * (screen cross product) when all w > 0
    const float c0 = P[1].X * P[2].Y - P[1].Y * P[2].X, a0 = P[1].Y * P[2].W - P[1].W * P[2].Y, b0 = P[1].W * P[2].X - P[1].X * P[2].W;
    const float det = (P[0].X * a0 + P[0].Y * b0) + P[0].W * c0;
    if (!(det != 0.0f) || !std::isfinite(det)) return t;
    t.front = det < 0.0f;
    const ClipVertex* n[3] = {&P[0], det < 0.0f ? &P[2] : &P[1], det < 0.0f ? &P[1] : &P[2]};
    for (int i = 0; i < 3; ++i) {
        const ClipVertex& a = *n[i];
        const ClipVertex& b = *n[(i + 1) % 3];
        EdgeSetup& e = t.e[i];
        e.flip = !lex_less3(a, b);
        const ClipVertex& lo = e.flip ? b : a;
        const ClipVertex& hi = e.flip ? a : b;
        e.nay = lo.Y * hi.W - lo.W * hi.Y; // A of the canonical orientation
        e.bx = lo.W * hi.X - lo.X * hi.W;  // B
        const float A = e.flip ? -e.nay : e.nay, B = e.flip ? -e.bx : e.bx;
        e.topleft = A > 0.0f || (A == 0.0f && B > 0.0f); // the plain rule (dy < 0 || (dy == 0 && dx > 0)) in terms of the half-plane normal
        const ClipVertex& anchor = lo.W > 0.0f ? lo : (hi.W > 0.0f ? hi : (lo.W != 0.0f ? lo : hi));
        if (anchor.W == 0.0f) return t; // an edge at infinity
        e.lo[0] = anchor.X / anchor.W;
        e.lo[1] = anchor.Y / anchor.W;
        if (!(std::isfinite(e.lo[0]) && std::isfinite(e.lo[1]))) return t;
    }
    if (P[0].W > 0.0f && P[1].W > 0.0f && P[2].W > 0.0f) {
        float px[3], py[3];
        for (int i = 0; i < 3; ++i) {
            px[i] = P[i].X / P[i].W;
            py[i] = P[i].Y / P[i].W;
        }
        const float minx = std::fmin(px[0], std::fmin(px[1], px[2])), maxx = std::fmax(px[0], std::fmax(px[1], px[2]));
        const float miny = std::fmin(py[0], std::fmin(py[1], py[2])), maxy = std::fmax(py[0], std::fmax(py[1], py[2]));
        if (!(minx == minx && maxx == maxx && miny == miny && maxy == maxy)) return t;
        t.x0 = (int)std::floor(std::fmin(std::fmax(minx, 0.0f), (float)width));
        t.x1 = (int)std::floor(std::fmax(std::fmin(maxx, (float)(width - 1)), -1.0f));
        t.y0 = (int)std::floor(std::fmin(std::fmax(miny, 0.0f), (float)height));
        t.y1 = (int)std::floor(std::fmax(std::fmin(maxy, (float)(height - 1)), -1.0f));
        if (t.x0 > t.x1 || t.y0 > t.y1) return t;
    } else { // crosses the eye plane: its screen extent is unbounded, every pixel is a candidate
        t.x0 = 0;
        t.y0 = 0;
        t.x1 = width - 1;
        t.y1 = height - 1;
    }
    // attribute planes are set up relative to the anchor (the projection of vertex k) so that small triangles far from the frame
    // origin do not cancel: with P'_k = (0, 0, w_k) the lambda gradients reduce to the expressions of setup_projective_plane
    t.k = k;
    const ClipVertex& K = P[k];
    const ClipVertex& U = P[(k + 1) % 3];
    const ClipVertex& V = P[(k + 2) % 3];
    t.anchor[0] = K.X / K.W;
    t.anchor[1] = K.Y / K.W;
    t.lx[0] = U.X - t.anchor[0] * U.W;
    t.ly[0] = U.Y - t.anchor[1] * U.W;
    t.lx[1] = V.X - t.anchor[0] * V.W;
    t.ly[1] = V.Y - t.anchor[1] * V.W;
    t.lw[0] = K.W;
    t.lw[1] = U.W;
    t.lw[2] = V.W;
    const float local_det = K.W * (t.lx[0] * t.ly[1] - t.ly[0] * t.lx[1]);
    if (!(local_det != 0.0f) || !std::isfinite(local_det)) return t;
    t.inv_det = 1.0f / local_det;
    t.valid = true;
    return t;
}
// The plane F(x, y) = sum_i f_i lambda_i(x, y) through the per-vertex values f (index 0 = vertex k, then cyclic): F = f/w interpolated
// linearly on screen. f = attribute -> "a/w"; f = 1 -> "1/w"; f = clip.z -> z/w itself (NDC depth is affine on screen).
inline AttrPlane setup_projective_plane(const ProjectiveSetup& t, float fk, float fu, float fv) {
    const float ak = t.ly[0] * t.lw[2] - t.lw[1] * t.ly[1], bk = t.lw[1] * t.lx[1] - t.lx[0] * t.lw[2]; // P'_u x P'_v
    const float au = t.ly[1] * t.lw[0], bu = -(t.lx[1] * t.lw[0]);                                       // P'_v x P'_k
    const float av = -(t.lw[0] * t.ly[0]), bv = t.lw[0] * t.lx[0];                                       // P'_k x P'_u
    AttrPlane p;
    p.a0 = fk / t.lw[0];
    p.gx = ((fk * ak + fu * au) + fv * av) * t.inv_det;
    p.gy = ((fk * bk + fu * bu) + fv * bv) * t.inv_det;
    return p;
}

template <int NATTR, typename Frag, typename Stencil>
inline void raster_projective(Frame& f, const ClipVertex P[3], const float attr[3][4], Frag frag, Stencil stencil) {
    const ProjectiveSetup t = setup_projective(P, (int)f.width, (int)f.height);
    if (!t.valid) return;
    const int k = t.k, u = (k + 1) % 3, v = (k + 2) % 3;
    AttrPlane planes[NATTR > 0 ? NATTR : 1];
    for (int a = 0; a < NATTR; ++a) planes[a] = setup_projective_plane(t, attr[k][a], attr[u][a], attr[v][a]);
    const AttrPlane qp = setup_projective_plane(t, 1.0f, 1.0f, 1.0f);       // 1/w
    const AttrPlane zp = setup_projective_plane(t, P[k].Z, P[u].Z, P[v].Z); // z/w
    auto tile_constant = [&](const AttrPlane& p, float tx0, float ty0) { return fmaf(ty0 - t.anchor[1], p.gy, fmaf(tx0 - t.anchor[0], p.gx, p.a0)); };
    for (int ty = t.y0 / TILE; ty <= t.y1 / TILE; ++ty) {
        for (int tx = t.x0 / TILE; tx <= t.x1 / TILE; ++tx) {
            const float tx0 = (float)(tx * TILE), ty0 = (float)(ty * TILE);
            float c[3];
            for (int i = 0; i < 3; ++i) c[i] = t.e[i].bx * (ty0 - t.e[i].lo[1]) + t.e[i].nay * (tx0 - t.e[i].lo[0]);
            float ac[NATTR > 0 ? NATTR : 1];
            for (int a = 0; a < NATTR; ++a) ac[a] = tile_constant(planes[a], tx0, ty0);
            const float qc = tile_constant(qp, tx0, ty0), zc = tile_constant(zp, tx0, ty0);
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
                            float e = fmaf(ry, t.e[i].bx, fmaf(rx, t.e[i].nay, c[i])); // the column term first: shared by the rows of a pixel column
                            if (t.e[i].flip) e = -e;
                            inside = e > 0.0f || (e == 0.0f && t.e[i].topleft);
                        }
                        if (!inside) continue;
                        const float z = fmaf(ry, zp.gy, fmaf(rx, zp.gx, zc));
                        if (!(z >= 0.0f && z <= 1.0f)) continue; // near / far clip of the fixed-function pipeline, per sample
                        const float q = fmaf(ry, qp.gy, fmaf(rx, qp.gx, qc));
                        const float w = 1.0f / q;
                        float values[NATTR > 0 ? NATTR : 1];
                        for (int a = 0; a < NATTR; ++a) values[a] = fmaf(ry, planes[a].gy, fmaf(rx, planes[a].gx, ac[a])) * w;
                        if (!frag(values)) continue;
                        stencil(((size_t)py * f.width + px) * f.samples + s, t.front, z);
                    }
                }
            }
        }
    }
}

// One triangle of an instance: `pos` are the model-space positions (the vertex attribute at location 4, shaders.wgsl:66-153).
template <int NATTR, typename Frag, typename Stencil>
inline void raster_triangle(Frame& f, const float m[16], const float pos[3][2], const float attr[3][4], Frag frag, Stencil stencil) {
    const float W = (float)f.width, H = (float)f.height;
    if (is_plain_instance(m)) {
        float v[3][2];
        for (int c = 0; c < 3; ++c) to_framebuffer(m, W, H, pos[c], v[c]);
        raster_plain<NATTR>(f, v, attr, m[14], frag, stencil);
    } else {
        ClipVertex P[3];
        for (int c = 0; c < 3; ++c) P[c] = to_clip(m, W, H, pos[c]);
        raster_projective<NATTR>(f, P, attr, frag, stencil);
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
    static const crh_dynamic_stroke_descriptor zero_descriptor = {};
    auto descriptor = [&](uint32_t path_index) -> const crh_dynamic_stroke_descriptor& {
        // out-of-range reads of a storage buffer are clamped/zero in WebGPU; never happens for validated input
        return path_index < shape.stroke_buffer.size() ? shape.stroke_buffer[path_index] : zero_descriptor;
    };
    const uint32_t read_mask = f.clip_mask | f.winding_mask;
    auto stroke_stencil = [&](size_t si, bool, float) { // Equal(ref) -> IncrementWrap, both faces, write mask = winding (renderer.rs:571-576)
        if ((f.winding[si] & read_mask) == (f.reference & read_mask)) f.winding[si] = wrap_add(f.winding[si], 1, f.winding_mask);
    };
    auto fill_stencil = [&](size_t si, bool front, float) { // LessEqual(ref <= stencil) -> front Increment / back Decrement (renderer.rs:577-582)
        if ((f.reference & read_mask) <= (f.winding[si] & read_mask)) f.winding[si] = wrap_add(f.winding[si], front ? 1 : -1, f.winding_mask);
    };
    // 1. stroke line strips (renderer.rs:278-287, shaders.wgsl:268-285)
    {
        const auto& verts = shape.stroke.line_vertices;
        const auto& idx = shape.stroke.line_indices;
        const auto& restarts = shape.stroke.line_restarts;
        size_t run_start = 0, vertex_base = 0, next_restart = 0;
        for (size_t k = 0; k <= idx.size(); ++k) {
            if (k == idx.size() || (next_restart < restarts.size() && restarts[next_restart] == k)) {
                for (size_t i = 0; run_start + i + 2 < k; ++i) {
                    size_t tri[3];
                    StripWalker::triangle(i, tri);
                    float v[3][2], attr[3][4];
                    for (int c = 0; c < 3; ++c) {
                        const Vertex2f1i& vx = verts[vertex_base + tri[c]];
                        v[c][0] = vx.p[0], v[c][1] = vx.p[1];
                        attr[c][0] = vx.t[0];
                        attr[c][1] = vx.t[1];
                    }
                    const Vertex2f1i& provoking = verts[vertex_base + i];
                    const uint32_t flat_u = provoking.u;
                    const float end_texcoord_y = provoking.t[1];
                    const crh_dynamic_stroke_descriptor& d = descriptor(flat_u & 65535u);
                    raster_triangle<2>(
                        f, m, v, attr,
                        [&](const float* t) {
                            if ((d.count_dashed_join & 4u) != 0u) return stroke_dashed(d, t[0], t[1]);
                            if ((flat_u & 65536u) != 0u) return cap(t[0], t[1] - end_texcoord_y, d.caps >> 4);
                            if (t[1] < 0.0f) return cap(t[0], -t[1], d.caps);
                            return true;
                        },
                        stroke_stencil);
                }
                vertex_base += k - run_start;
                run_start = k + 1;
                ++next_restart;
            }
        }
    }
    // 2. stroke joint strips (renderer.rs:288-302, shaders.wgsl:287-300)
    {
        const auto& verts = shape.stroke.joint_vertices;
        const auto& idx = shape.stroke.joint_indices;
        const auto& restarts = shape.stroke.joint_restarts;
        const float TAU = crh_acosf(-1.0f) * 2.0f;
        size_t run_start = 0, vertex_base = 0, next_restart = 0;
        for (size_t k = 0; k <= idx.size(); ++k) {
            if (k == idx.size() || (next_restart < restarts.size() && restarts[next_restart] == k)) {
                for (size_t i = 0; run_start + i + 2 < k; ++i) {
                    size_t tri[3];
                    StripWalker::triangle(i, tri);
                    float v[3][2], attr[3][4];
                    for (int c = 0; c < 3; ++c) {
                        const Vertex3f1i& vx = verts[vertex_base + tri[c]];
                        v[c][0] = vx.p[0], v[c][1] = vx.p[1];
                        attr[c][0] = vx.t[0];
                        attr[c][1] = vx.t[1];
                        attr[c][2] = vx.t[2];
                    }
                    const uint32_t flat_u = verts[vertex_base + i].u;
                    const crh_dynamic_stroke_descriptor& d = descriptor(flat_u & 65535u);
                    raster_triangle<3>(
                        f, m, v, attr,
                        [&](const float* t) {
                            const float radius = std::sqrt(t[0] * t[0] + t[1] * t[1]);
                            bool fill = joint(radius, (flat_u & 65536u) != 0u, d.count_dashed_join & 3u);
                            if (fill && (d.count_dashed_join & 4u) != 0u) fill = stroke_dashed(d, radius, t[2] + crh_atan2f(t[1], t[0]) / TAU);
                            return fill;
                        },
                        stroke_stencil);
                }
                vertex_base += k - run_start;
                run_start = k + 1;
                ++next_restart;
            }
        }
    }
    // 3. solid strips (renderer.rs:304-318, shaders.wgsl:233-234)
    {
        const auto& verts = shape.fill.solid_vertices;
        const auto& idx = shape.fill.solid_indices;
        const auto& restarts = shape.fill.solid_restarts;
        size_t run_start = 0, vertex_base = 0, next_restart = 0;
        for (size_t k = 0; k <= idx.size(); ++k) {
            if (k == idx.size() || (next_restart < restarts.size() && restarts[next_restart] == k)) {
                for (size_t i = 0; run_start + i + 2 < k; ++i) {
                    size_t tri[3];
                    StripWalker::triangle(i, tri);
                    float v[3][2], attr[3][4] = {};
                    for (int c = 0; c < 3; ++c) v[c][0] = verts[vertex_base + tri[c]].p[0], v[c][1] = verts[vertex_base + tri[c]].p[1];
                    raster_triangle<0>(
                        f, m, v, attr, [](const float*) { return true; }, fill_stencil);
                }
                vertex_base += k - run_start;
                run_start = k + 1;
                ++next_restart;
            }
        }
    }
    // 4. curve triangle lists (renderer.rs:319-335, shaders.wgsl:236-266)
    for (size_t i = 0; i + 2 < shape.fill.integral_quadratic_vertices.size(); i += 3) {
        float v[3][2], attr[3][4] = {};
        for (int c = 0; c < 3; ++c) {
            const Vertex2f& vx = shape.fill.integral_quadratic_vertices[i + c];
            v[c][0] = vx.p[0], v[c][1] = vx.p[1];
            attr[c][0] = vx.w[0];
            attr[c][1] = vx.w[1];
        }
        raster_triangle<2>(
            f, m, v, attr, [](const float* w) { return w[0] * w[0] - w[1] <= 0.0f; }, fill_stencil);
    }
    for (size_t i = 0; i + 2 < shape.fill.integral_cubic_vertices.size(); i += 3) {
        float v[3][2], attr[3][4] = {};
        for (int c = 0; c < 3; ++c) {
            const Vertex3f& vx = shape.fill.integral_cubic_vertices[i + c];
            v[c][0] = vx.p[0], v[c][1] = vx.p[1];
            for (int a = 0; a < 3; ++a) attr[c][a] = vx.w[a];
        }
        raster_triangle<3>(
            f, m, v, attr, [](const float* w) { return w[0] * w[0] * w[0] - w[1] * w[2] <= 0.0f; }, fill_stencil);
    }
    for (size_t i = 0; i + 2 < shape.fill.rational_quadratic_vertices.size(); i += 3) {
        float v[3][2], attr[3][4] = {};
        for (int c = 0; c < 3; ++c) {
            const Vertex3f& vx = shape.fill.rational_quadratic_vertices[i + c];
            v[c][0] = vx.p[0], v[c][1] = vx.p[1];
            for (int a = 0; a < 3; ++a) attr[c][a] = vx.w[a];
        }
        raster_triangle<3>(
            f, m, v, attr, [](const float* w) { return w[0] * w[0] - w[1] * w[2] <= 0.0f; }, fill_stencil);
    }
    for (size_t i = 0; i + 2 < shape.fill.rational_cubic_vertices.size(); i += 3) {
        float v[3][2], attr[3][4] = {};
        for (int c = 0; c < 3; ++c) {
            const Vertex4f& vx = shape.fill.rational_cubic_vertices[i + c];
            v[c][0] = vx.p[0], v[c][1] = vx.p[1];
            for (int a = 0; a < 4; ++a) attr[c][a] = vx.w[a];
        }
        raster_triangle<4>(
            f, m, v, attr, [](const float* w) { return w[0] * w[0] * w[0] - w[1] * w[2] * w[3] <= 0.0f; }, fill_stencil);
    }
}

// The cover operations: Shape::render(Clip | UnClip | Color | SaveAlphaContext | ScaleAlphaContext | RestoreAlphaContext)
// (renderer.rs:338-354) draw the hull strip with the fixed-function state of renderer.rs:692-754 / :761-861 and the fragment stages
// shaders.wgsl:304-355, for one instance. `op` = crh_render_op.
inline bool depth_test(uint32_t compare, float fragment, float stored) { // wgpu::CompareFunction
    switch (compare) {
        case CRH_COMPARE_NEVER: return false;
        case CRH_COMPARE_LESS: return fragment < stored;
        case CRH_COMPARE_EQUAL: return fragment == stored;
        case CRH_COMPARE_LESS_EQUAL: return fragment <= stored;
        case CRH_COMPARE_GREATER: return fragment > stored;
        case CRH_COMPARE_NOT_EQUAL: return fragment != stored;
        case CRH_COMPARE_GREATER_EQUAL: return fragment >= stored;
        default: return true;
    }
}
inline void render_cover(Frame& f, const Shape& shape, const float m[16], const float rgba[4], uint32_t op, uint32_t alpha_layer) {
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
        for (int c = 0; c < 3; ++c) v[c][0] = hull[tri[c]].p[0], v[c][1] = hull[tri[c]].p[1];
        raster_triangle<0>(
            f, m, v, attr, [](const float*) { return true; },
            [&](size_t si, bool front, float z) {
                const uint32_t st = f.winding[si];
                float* dst = &f.color[si * 4];
                switch (op) {
                    case CRH_OP_COLOR: // Less(ref < stencil): blend premultiplied "over"; pass -> Zero, fail -> Zero on the winding bits (renderer.rs:747-752)
                        if ((f.cull_mode == CRH_CULL_FRONT && front) || (f.cull_mode == CRH_CULL_BACK && !front)) break; // Configuration::cull_mode, renderer.rs:743
                        if ((ref & read_mask) < (st & read_mask)) {
                            // the depth test follows the stencil test; depth_fail_op = Keep (renderer.rs:442): the winding survives
                            if (!f.depth.empty() && !depth_test(f.depth_compare, z, f.depth[si])) break;
                            for (int c = 0; c < 4; ++c) dst[c] = src[c] + dst[c] * one_minus_a;
                            if (f.attachment8)
                                for (int c = 0; c < 4; ++c) dst[c] = Frame::unorm8(dst[c]);
                            if (!f.depth.empty() && f.depth_write) f.depth[si] = z;
                        }
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
                                if (f.attachment8) dst[3] = Frame::unorm8(dst[3]);
                            } else { // RestoreAlphaContext: src.a = (1 - saved)(1 - a); alpha' = dst.a * One - src.a * One, renderer.rs:829-861
                                const float sa = (1.0f - (*layer)[si]) * (1.0f - rgba[3]);
                                dst[3] = dst[3] - sa;
                                if (f.attachment8) dst[3] = Frame::unorm8(dst[3]);
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
