// csrc/raster_common.hpp — what the two raster paths share: the set-up triangle records, candidate numbering of a Shape, the vertex
// stage and the stroke fragment stages of shaders.wgsl:165-231. raster.hip is the general path (recorded passes with clip nesting / alpha
// contexts, perspective instances, depth, culling: triangle strips exactly as the reference draws them); raster_edges.hip is the plain
// Stencil + Color pass (boundary edges + backdrop for the polygon interiors, one binning traversal).
#pragma once
#include "ga.hpp"
#include "raster_params.hpp"
#include "scene.hpp"

#ifndef CRH_TILE_WAVES
#define CRH_TILE_WAVES 6
#endif
#ifndef CRH_WALK_WAVES
#define CRH_WALK_WAVES 2
#endif


namespace crh {

constexpr int kTile = 16;
constexpr uint32_t kSortBytesMax = 32u * 1024u; // dynamic LDS per workgroup for the tile sort: 8 192 primitives per tile at msaa 1, 2 048 at msaa 4

CRH_D float2 to_framebuffer(const float* m, float w, float h, float x, float y) { // oracle/raster.hpp to_framebuffer
    const float cx = (m[0] * x + m[4] * y) + m[12];
    const float cy = (m[1] * x + m[5] * y) + m[13];
    return make_float2((cx * 0.5f + 0.5f) * w, (0.5f - cy * 0.5f) * h);
}

// RGBA8 unorm of a resolved colour: clamp to [0, 1] with NaN -> 0, x * 255 + 0.5 truncated (oracle/raster.hpp resolve_rgba8) — four instructions per channel
CRH_D uint32_t pack_unorm8(float c0, float c1, float c2, float c3) {
    const float v[4] = {c0, c1, c2, c3};
    uint32_t out = 0;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
#if defined(__HIP_DEVICE_COMPILE__)
        const float x = __builtin_amdgcn_fmed3f(v[ch], 0.0f, 1.0f); // a NaN operand makes v_med3_f32 return the minimum of the others: 0
#else
        const float x = v[ch];
#endif
        out |= (uint32_t)(int)(x * 255.0f + 0.5f) << (8 * ch);
    }
    return out;
}
// What an Rgba8Unorm attachment keeps of a colour component a blend wrote (CRH_FORMAT_RGBA8_ATTACHMENT): pack_unorm8's rounding, read back as
// load_pixel reads it — idempotent, so it may be applied to components the blend left alone
CRH_D float attachment_unorm8(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float x = __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f);
#else
    const float x = v;
#endif
    return (float)(uint32_t)(int)(x * 255.0f + 0.5f) * (1.0f / 255.0f);
}
// The target's pixel (gx, gy): the resolved premultiplied colour, clamped to [0, 1] (NaN -> 0) and stored as RGBA8 unorm or — the layers
// of the multi-GPU exchange — as four binary16 values (round to nearest even; the clamp is the same, so a 16F layer holds what the RGBA8
// target would have quantised).
CRH_D void store_pixel(const RasterParams& r, uint32_t gx, uint32_t gy, float c0, float c1, float c2, float c3) {
    const size_t at = (size_t)gy * r.width + gx;
    if (r.format == CRH_FORMAT_RGBA16F) {
        float v[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            float x = v[ch];
            x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
            if (!(x == x)) x = 0.0f;
            v[ch] = x;
        }
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        reinterpret_cast<h4*>(r.rgba8)[at] = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    } else {
        reinterpret_cast<uint32_t*>(r.rgba8)[at] = pack_unorm8(c0, c1, c2, c3);
    }
}
CRH_D float4 load_pixel(const RasterParams& r, uint32_t gx, uint32_t gy) {
    const size_t at = (size_t)gy * r.width + gx;
    if (r.format == CRH_FORMAT_RGBA16F) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 d = reinterpret_cast<const h4*>(r.rgba8)[at];
        return make_float4((float)d[0], (float)d[1], (float)d[2], (float)d[3]);
    }
    const uchar4 d = reinterpret_cast<const uchar4*>(r.rgba8)[at];
    return make_float4((float)d.x * (1.0f / 255.0f), (float)d.y * (1.0f / 255.0f), (float)d.z * (1.0f / 255.0f), (float)d.w * (1.0f / 255.0f));
}

enum : uint32_t { KIND_SOLID = 0, KIND_IQ = 1, KIND_IC = 2, KIND_RQ = 3, KIND_RC = 4, KIND_LINE = 5, KIND_JOINT = 6, KIND_COVER = 7 };

// One set-up triangle, tile independent: two 64-byte halves, each fetched with one scalar load.
// Edge i evaluates, relative to a tile origin (tx0, ty0):
//   c = bx*(ty0 - lo_y) + nay*(tx0 - lo_x);  E = fma(rx, nay, fma(ry, bx, c))   — the canonical-orientation sign is folded in.
struct PrimCoverage { // everything the coverage test needs
    uint32_t flags; // bits 0-2 top-left per edge, bit 3 front (ccw on screen), bits 4-6 kind, 7-9 cover op, 16-23 clip ref, 24-27 alpha layer, 28 projective
    uint32_t desc;  // stroke: index of the 48-byte descriptor
    ushort4 box;    // inclusive pixel box x0 x1 y0 y1; x0 == 0xFFFF: nothing to draw
    float lo_x[3], lo_y[3], bx[3], nay[3];
};
struct PrimFragment { // what the fragment stage needs
    float a0[4], gx[4], gy[4]; // attribute planes through vertex 0; cover: a0 = premultiplied source colour, gx[0] = depth of a plain instance
    float v0x, v0y;
    uint32_t flat_u; // stroke: provoking vertex' u32 (group | 0x10000)
    float end_y;     // stroke line: provoking vertex' texcoord.y
};
struct PrimRec {
    PrimCoverage cov;
    PrimFragment frag;
};
static_assert(sizeof(PrimCoverage) == 64 && sizeof(PrimFragment) == 64 && sizeof(PrimRec) == 128, "PrimRec");
// Primitives of projective instances (flags bit 28) also carry the planes of 1/w and z/w (oracle/raster.hpp raster_projective), anchored
// at (frag.v0x, frag.v0y) like the attribute planes — which then hold a/w. A side array, so the plain pass never touches it.
struct PrimProj {
    float q0, qgx, qgy, z0, zgx, zgy, ax, ay; // (ax, ay): the anchor, = frag.v0x / v0y (repeated: solid triangles never load their fragment half)
};
static_assert(sizeof(PrimProj) == 32, "PrimProj");
constexpr uint32_t kFlagProjective = 1u << 28;

// Wave-uniform loads through the constant address space become s_load_dwordx4..x16 (the records are written by an earlier kernel,
// so the scalar cache is coherent with them). The host pass of the compiler never runs this code.
#if defined(__HIP_DEVICE_COMPILE__)
#define CRH_CONST __attribute__((address_space(4)))
#else
#define CRH_CONST
#endif
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
CRH_D f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); } // two IEEE fmas (v_pk_fma_f32)
CRH_D f32x2 splat2(float v) { return f32x2{v, v}; }
template <class T>
CRH_D T load_uniform(const T* p) { // p must be wave uniform
    static_assert(sizeof(T) % 16 == 0, "16-byte multiples");
    T out;
    const u32x4 CRH_CONST* src = (const u32x4 CRH_CONST*)p;
    u32x4* dst = reinterpret_cast<u32x4*>(&out);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; ++i) dst[i] = src[i];
    return out;
}

CRH_D uint32_t shape_candidates(const SceneDev& s, uint32_t shape, uint32_t c[8]) {
    const uint32_t* b0 = s.shape_base + shape * kShapeRow; // begin[NCH], end[NCH]
    const uint32_t* b1 = b0 + NCH;
    const uint32_t lvn = b1[CH_LINE_V] - b0[CH_LINE_V], svn = b1[CH_SOLID_V] - b0[CH_SOLID_V], hn = s.hull_count[shape];
    c[0] = lvn >= 3u ? lvn - 2u : 0u;                  // stroke line strip triangles
    c[1] = c[0] + 3u * (b1[CH_JOINT] - b0[CH_JOINT]);  // joint strips: 3 triangles per join
    c[2] = c[1] + (svn >= 3u ? svn - 2u : 0u);         // solid strips
    c[3] = c[2] + (b1[CH_IQ] - b0[CH_IQ]);
    c[4] = c[3] + (b1[CH_IC_V] - b0[CH_IC_V]) / 3u;
    c[5] = c[4] + (b1[CH_RQ] - b0[CH_RQ]);
    c[6] = c[5] + (b1[CH_RC_V] - b0[CH_RC_V]) / 3u;
    c[7] = c[6] + (hn >= 3u ? hn - 2u : 0u);           // cover: hull strip
    return c[7];
}

CRH_D DrawItem item_of(const RasterParams& r, uint32_t i) {
    if (r.items) return r.items[i];
    return DrawItem{i, i, 1u | ((uint32_t)(CRH_OP_COLOR + 1) << 4), 0u}; // the plain pass: Stencil + Color of Shape i at clip depth 0
}
// the candidate triangles of an item: [first, last) in the Shape's candidate numbering (stencil kinds first, cover strip last)
CRH_D void item_candidates(const SceneDev& s, const DrawItem& it, uint32_t cb[8], uint32_t& first, uint32_t& last) {
    shape_candidates(s, it.shape, cb);
    first = (it.ops & 1u) ? 0u : cb[6];
    last = (it.ops >> 4) ? cb[7] : cb[6];
}

CRH_D bool cap_test(float x, float y, uint32_t cap_type) { // shaders.wgsl:165-189
    switch (cap_type & 15u) {
        case 0: return y > 0.5f;
        case 1: return x * x + y * y < 0.25f;
        case 2: return 0.5f - y > fabsf(x);
        case 3: return y < fabsf(x);
        case 4: return 0.5f - y > x;
        case 5: return y - 0.5f < x;
        default: return y < 0.0f;
    }
}
// The dashed pattern walk of shaders.wgsl:205-231 with the 48-byte descriptor in scalar registers (it is wave uniform) and the interval
// search unrolled into selects: interval = number of leading intervals that end before the position (at most `last`). No memory access
// and no loop per sample — the loop over global-memory descriptor fields this replaces was half of the dashed workload's raster time.
CRH_D float wave_uniform(float v) { // v is the same in every lane: keep it in a scalar register
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
#else
    return v;
#endif
}
CRH_D bool stroke_dashed(const crh_dynamic_stroke_descriptor& d, float tx, float ty) {
    // The descriptor's fields as scalar VALUES first. Selected straight out of the struct ("a ? d.gap_end[1] : d.gap_end[0]"), the compiler
    // turns the select of two loads into one load at a selected address — from a copy of the descriptor in scratch memory: 52 bytes per lane
    // and three vector-memory round trips per sample in every kernel that draws strokes.
    const float ge0 = wave_uniform(d.gap_end[0]), ge1 = wave_uniform(d.gap_end[1]), ge2 = wave_uniform(d.gap_end[2]), ge3 = wave_uniform(d.gap_end[3]);
    const float gs0 = wave_uniform(d.gap_start[0]), gs1 = wave_uniform(d.gap_start[1]), gs2 = wave_uniform(d.gap_start[2]), gs3 = wave_uniform(d.gap_start[3]);
    const uint32_t last = d.count_dashed_join >> 3;
    const float ge_last = last == 0u ? ge0 : (last == 1u ? ge1 : (last == 2u ? ge2 : ge3)); // wave uniform
    const float pattern_length = ge_last;
    float position = crh_wgsl_mod(ty - d.phase, pattern_length);
    if (position < 0.0f) position = position + pattern_length;
    // for (;;) { gap_end = gap_end[interval] - position; if (gap_end >= 0 || interval >= last) break; ++interval; }
    const bool a0 = !(ge0 - position >= 0.0f) && 0u < last;
    const bool a1 = a0 && !(ge1 - position >= 0.0f) && 1u < last;
    const bool a2 = a1 && !(ge2 - position >= 0.0f) && 2u < last;
    const uint32_t interval = (uint32_t)a0 + (uint32_t)a1 + (uint32_t)a2;
    const float ge = a2 ? ge3 : (a1 ? ge2 : (a0 ? ge1 : ge0));
    const float gs = a2 ? gs3 : (a1 ? gs2 : (a0 ? gs1 : gs0));
    const float gap_end = ge - position;
    const float gap_start = position - gs;
    if (gap_start > 0.0f) {
        const uint32_t caps = d.caps >> (interval * 8u);
        const bool start_cap = cap_test(tx, gap_start, caps >> 4);
        const bool end_cap = cap_test(tx, gap_end, caps);
        return start_cap || end_cap;
    }
    return true;
}
CRH_D bool stroke_dashed_joint(const crh_dynamic_stroke_descriptor& d, float radius, float a0, float a1, float a2) { // shaders.wgsl:296-299
    const float tau = crh_acosf(-1.0f) * 2.0f;
    return stroke_dashed(d, radius, a2 + crh_atan2f(a1, a0) / tau);
}

} // namespace crh
