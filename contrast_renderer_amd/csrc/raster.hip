// csrc/raster.hip — tile-binned compute rasterizer that replaces the reference's stencil-then-cover passes
// (Shape::render renderer.rs:267-355, stencil states renderer.rs:565-582,692-754,761-861, fragment stages shaders.wgsl:155-355).
//
// The frame is cut into 16x16-pixel tiles; one wavefront owns one tile (msaa 1: lane = column x row group, four pixels per lane; msaa 4:
// four wavefronts per tile, one pixel x four samples per lane), so the stencil byte of the reference (winding counter, clip nesting
// counter), the saved alpha layers and the colour of a sample never leave the owning lane's registers, and no workgroup barrier is
// needed anywhere in the raster kernel (except in the rare tile whose list is too long for LDS and is sorted in global memory).
//
//   k_scan_*           two-kernel exclusive scan (primitive ranges per Shape / draw item, tile list offsets per frame)
//   k_prim_setup<S,P>  one wavefront per draw item (= Shape in the plain pass), one lane per triangle: vertex stage (shaders.wgsl:13-27,
//                      66-151), edge functions in canonical orientation, clamped pixel box, attribute planes -> a 128-byte record, once
//                      per frame. Primitive ids are contiguous per item and ascend in draw order (item, then line / joint / solid / IQ /
//                      IC / RQ / RC / cover = renderer.rs:275-354).
//   k_tile_walk<S,F>   count pass (F = false) and fill pass (F = true) of the per-tile lists: kWalkWaves wavefronts per item, lane =
//                      triangle, exact tile test (an edge function is monotone in x and y under fmaf, so the best tile corner
//                      decides), ballot + popcount -> ONE atomic per (64-triangle chunk, tile); the fill pass writes prim ids.
//   k_raster_tile<..>  sorts the tile's list by prim id (= draw order) in registers / LDS / global memory and walks it; see the kernel's
//                      comment. Workgroup b runs on XCD b % 8: tiles are dealt to the XCDs in 8x8 blocks (L2 locality of the records).
//   Coverage and attribute arithmetic follow oracle/raster.hpp operation by operation (tile-relative constants, explicit fmaf), so
//   pixels are bit-identical to the CPU spec.
//
// Roofline note: algorithmic traffic = emitted vertices read once + W*H*4 bytes written once; the raster kernel is VALU-issue bound
// (edge functions per sample), the others latency bound; there is no GEMM shape for MFMA anywhere.
#include "raster_common.hpp"

namespace crh {

// ---------------------------------------------------------------------------------------------- scans
// Two-kernel exclusive scan over a u32 array (1024 items per block): out[i] = sum of in[0..i), out[n] = total.
struct ScanJob {
    const uint32_t* in;
    uint32_t* out;
    uint32_t* block_sum;
    uint32_t n, blocks;
    uint32_t* max_out; // optional: atomicMax of the items (the longest tile list)
};
// candidate counts are transform independent: one lane per Shape (runs at the end of tessellation, before the scan)
__global__ __launch_bounds__(256) void k_shape_ncand(SceneDev s, uint32_t* shape_ncand) {
    const uint32_t shape = blockIdx.x * 256u + threadIdx.x;
    if (shape >= s.n_shapes) return;
    uint32_t c[8];
    shape_ncand[shape] = shape_candidates(s, shape, c);
}
// the same per draw item of a recorded pass (per frame)
__global__ __launch_bounds__(256) void k_item_ncand(SceneDev s, RasterParams r, uint32_t* item_ncand) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= r.n_items) return;
    uint32_t cb[8], first, last;
    item_candidates(s, r.items[i], cb, first, last);
    item_ncand[i] = last - first;
}
CRH_D void scan_local_body(const ScanJob& j) {
    __shared__ uint32_t wave_sum[4];
    const uint32_t i0 = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i0 + k < j.n ? j.in[i0 + k] : 0u;
    const uint32_t mine = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = mine;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (j.max_out) {
        uint32_t longest = max(max(v[0], v[1]), max(v[2], v[3]));
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, d, 64));
        if (lane == 0 && longest > 0u) atomicMax(j.max_out, longest);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += up;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += wave_sum[w];
    uint32_t run = base + incl - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < j.n) j.out[i0 + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 255) j.block_sum[blockIdx.x] = run;
}
__global__ __launch_bounds__(256) void k_scan_local(ScanJob j) { scan_local_body(j); }
// two scans of equally many items in one launch (blockIdx.y picks the job): the primitive ids and the slot ranges per Shape
__global__ __launch_bounds__(256) void k_scan_local2(ScanJob a, ScanJob b) { scan_local_body(blockIdx.y ? b : a); }
// mode 0: primitive ranges; mode 1: tile list offsets (also publishes the pair count and the overflow flag)
CRH_D void scan_add_body(const ScanJob& j, const RasterParams& r, int mode) {
    __shared__ uint32_t partial[256];
    uint32_t sum = 0;
    for (uint32_t k = threadIdx.x; k < blockIdx.x; k += 256u) sum += j.block_sum[k];
    partial[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < (uint32_t)d) partial[threadIdx.x] += partial[threadIdx.x + d];
        __syncthreads();
    }
    const uint32_t base = partial[0];
    const uint32_t i0 = blockIdx.x * 1024u + threadIdx.x * 4u;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < j.n) j.out[i0 + k] += base;
    if (blockIdx.x + 1u == j.blocks && threadIdx.x == 0) {
        const uint32_t total = base + j.block_sum[blockIdx.x];
        j.out[j.n] = total;
        if (mode == 1) {
            r.overflow[1] = total;
            r.overflow[0] = total > r.pair_capacity ? 1u : 0u;
        }
    }
}
__global__ __launch_bounds__(256) void k_scan_add(ScanJob j, RasterParams r, int mode) { scan_add_body(j, r, mode); }
__global__ __launch_bounds__(256) void k_scan_add2(ScanJob a, ScanJob b) {
    RasterParams unused = {};
    scan_add_body(blockIdx.y ? b : a, unused, 0);
}

// ---------------------------------------------------------------------------------------------- exact tile test
// An edge function E = fma(ry, bx, fma(rx, nay, c)) is monotone in rx and in ry (fmaf rounds monotonically), so its extremes over a
// box of sample positions sit at the corners. Sample positions inside a tile span [s_lo, 15 + s_hi] in x and in y. A tile is a hit of
// a triangle when the triangle's pixel box overlaps it and no edge rejects its best tile corner — a conservative superset of "some
// sample of the tile is covered" (three half planes each touching the tile do not imply a common point); the raster kernel decides per
// sample. The test itself is written out in k_tile_walk, with the tile-invariant parts hoisted.

// tile rectangle that bounds the pixel boxes of the wave's (up to 64) triangles; false when no lane draws anything
CRH_D bool wave_tile_rect(bool valid, const PrimCoverage& cov, uint32_t& tx0, uint32_t& tx1, uint32_t& ty0, uint32_t& ty1) {
    uint32_t x0 = valid ? cov.box.x : 0xFFFFu, x1 = valid ? cov.box.y : 0u, y0 = valid ? cov.box.z : 0xFFFFu, y1 = valid ? cov.box.w : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        x0 = min(x0, (uint32_t)__shfl_xor((int)x0, d, 64));
        y0 = min(y0, (uint32_t)__shfl_xor((int)y0, d, 64));
        x1 = max(x1, (uint32_t)__shfl_xor((int)x1, d, 64));
        y1 = max(y1, (uint32_t)__shfl_xor((int)y1, d, 64));
    }
    tx0 = x0 / kTile;
    tx1 = x1 / kTile;
    ty0 = y0 / kTile;
    ty1 = y1 / kTile;
    return x0 != 0xFFFFu;
}

// ---------------------------------------------------------------------------------------------- k_prim_setup
// PROJ == false: every instance is plain (the host checked), the projective setup is compiled out
template <int S, bool PROJ>
__global__ __launch_bounds__(64) void k_prim_setup(SceneDev s, RasterParams r) {
    const uint32_t item = blockIdx.x, lane = threadIdx.x;
    const DrawItem it = item_of(r, item);
    const uint32_t shape = it.shape;
    const float* m = r.transforms + 16u * it.instance;
    // oracle/raster.hpp is_plain_instance: clip.w == 1 and clip.z a constant in [0, 1]; every other matrix takes the projective setup
    const bool plain = !PROJ || (m[3] == 0.0f && m[7] == 0.0f && m[15] == 1.0f && m[2] == 0.0f && m[6] == 0.0f && m[14] >= 0.0f && m[14] <= 1.0f);
    uint32_t cb[8], first_candidate, last_candidate;
    item_candidates(s, it, cb, first_candidate, last_candidate);
    const uint32_t n_candidates = last_candidate - first_candidate;
    const uint32_t prim0 = r.shape_prim_begin[item];
    const uint32_t cover_op = (it.ops >> 4) ? (it.ops >> 4) - 1u : (uint32_t)CRH_OP_COLOR;
    if (prim0 + n_candidates > r.prim_capacity) return; // cannot happen: the capacity is an upper bound derived from the totals
    const uint32_t* b0 = s.shape_base + shape * kShapeRow;
    const uint32_t lv0 = b0[CH_LINE_V], j0 = b0[CH_JOINT], sv0 = b0[CH_SOLID_V], iq0 = b0[CH_IQ], ic0 = b0[CH_IC_V], rq0 = b0[CH_RQ], rc0 = b0[CH_RC_V],
                   hull0 = b0[CH_HULL];
    const uint32_t dyn0 = s.shape_dyn_begin[shape];
    const float W = (float)r.width, H = (float)r.height;
    for (uint32_t c0 = 0; c0 < n_candidates; c0 += 64u) { // all 64 lanes stay in the loop: the tile walk below is wave-wide
        const uint32_t c = first_candidate + c0 + lane; // in the Shape's candidate numbering
        const bool in_range_c = c0 + lane < n_candidates;
        float2 p[3] = {make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f)};
        float attr[3][4] = {};
        uint32_t kind = KIND_SOLID, flat_u = 0, desc = 0;
        float end_y = 0.0f;
        int n_attr = 0;
        bool valid = true;
        uint32_t i0 = 0, i1 = 0, i2 = 0;
        auto strip = [&](uint32_t k, uint32_t base) { // strip triangle k: even (k, k+1, k+2), odd (k, k+2, k+1); provoking vertex k
            i0 = base + k;
            i1 = base + ((k & 1u) ? k + 2u : k + 1u);
            i2 = base + ((k & 1u) ? k + 1u : k + 2u);
        };
        if (!in_range_c) {
            valid = false;
        } else if (c < cb[0]) { // stroke line strips (vertex2f1u / stencil_stroke_line)
            const uint32_t k = c;
            valid = s.line_pair_cut[(lv0 + k) >> 1] == 0;
            strip(k, lv0);
            const Vertex2f1i a = s.line_v[i0], b = s.line_v[i1], d = s.line_v[i2];
            p[0] = make_float2(a.x, a.y);
            p[1] = make_float2(b.x, b.y);
            p[2] = make_float2(d.x, d.y);
            attr[0][0] = a.u, attr[0][1] = a.v;
            attr[1][0] = b.u, attr[1][1] = b.v;
            attr[2][0] = d.u, attr[2][1] = d.v;
            flat_u = a.i; // i0 is the provoking vertex
            end_y = a.v;
            desc = dyn0 + (a.i & 65535u);
            kind = KIND_LINE;
            n_attr = 2;
        } else if (c < cb[1]) { // stroke joint strips: 5 vertices, 3 triangles per join
            const uint32_t q = c - cb[0], jn = q / 3u, k = q - 3u * jn;
            strip(k, 5u * (j0 + jn));
            const Vertex3f1i a = s.joint_v[i0], b = s.joint_v[i1], d = s.joint_v[i2];
            p[0] = make_float2(a.x, a.y);
            p[1] = make_float2(b.x, b.y);
            p[2] = make_float2(d.x, d.y);
            attr[0][0] = a.u, attr[0][1] = a.v, attr[0][2] = a.w;
            attr[1][0] = b.u, attr[1][1] = b.v, attr[1][2] = b.w;
            attr[2][0] = d.u, attr[2][1] = d.v, attr[2][2] = d.w;
            flat_u = a.i;
            desc = dyn0 + (flat_u & 65535u);
            kind = KIND_JOINT;
            n_attr = 3;
        } else if (c < cb[2]) { // solid strips (vertex0 / stencil_solid)
            const uint32_t k = c - cb[1];
            const uint8_t f0 = s.solid_flag[sv0 + k], f1 = s.solid_flag[sv0 + k + 1u];
            valid = ((f0 | f1) & 2u) == 0;
            const uint32_t parity = f0 & 1u;
            i0 = sv0 + k;
            i1 = sv0 + (parity ? k + 2u : k + 1u);
            i2 = sv0 + (parity ? k + 1u : k + 2u);
            const Vertex0 a = s.solid_v[i0], b = s.solid_v[i1], d = s.solid_v[i2];
            p[0] = make_float2(a.x, a.y);
            p[1] = make_float2(b.x, b.y);
            p[2] = make_float2(d.x, d.y);
        } else if (c < cb[3]) {
            const uint32_t at = 3u * (iq0 + (c - cb[2]));
            for (int v = 0; v < 3; ++v) {
                const Vertex2f a = s.iq_v[at + v];
                p[v] = make_float2(a.x, a.y);
                attr[v][0] = a.u, attr[v][1] = a.v;
            }
            kind = KIND_IQ;
            n_attr = 2;
        } else if (c < cb[4]) {
            const uint32_t at = ic0 + 3u * (c - cb[3]);
            for (int v = 0; v < 3; ++v) {
                const Vertex3f a = s.ic_v[at + v];
                p[v] = make_float2(a.x, a.y);
                attr[v][0] = a.u, attr[v][1] = a.v, attr[v][2] = a.w;
            }
            kind = KIND_IC;
            n_attr = 3;
        } else if (c < cb[5]) {
            const uint32_t at = 3u * (rq0 + (c - cb[4]));
            for (int v = 0; v < 3; ++v) {
                const Vertex3f a = s.rq_v[at + v];
                p[v] = make_float2(a.x, a.y);
                attr[v][0] = a.u, attr[v][1] = a.v, attr[v][2] = a.w;
            }
            kind = KIND_RQ;
            n_attr = 3;
        } else if (c < cb[6]) {
            const uint32_t at = rc0 + 3u * (c - cb[5]);
            for (int v = 0; v < 3; ++v) {
                const Vertex4f a = s.rc_v[at + v];
                p[v] = make_float2(a.x, a.y);
                attr[v][0] = a.k, attr[v][1] = a.l, attr[v][2] = a.m, attr[v][3] = a.n;
            }
            kind = KIND_RC;
            n_attr = 4;
        } else { // cover: hull strip (vertex_color / color_cover)
            strip(c - cb[6], hull0);
            const Vertex0 a = s.hull_v[i0], b = s.hull_v[i1], d = s.hull_v[i2];
            p[0] = make_float2(a.x, a.y);
            p[1] = make_float2(b.x, b.y);
            p[2] = make_float2(d.x, d.y);
            kind = KIND_COVER;
        }
        // ---- oracle/raster.hpp setup_triangle + setup_attribute
        PrimRec rec;
        PrimProj proj = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        rec.cov.box = make_ushort4(0xFFFFu, 0, 0, 0);
        bool drawn = false;
        const uint32_t clip_ref = kind == KIND_COVER ? (it.refs >> 8) & 255u : it.refs & 255u;
        // bits 0-2 top-left, 3 front, 4-6 kind, 7-9 cover operation, 16-23 stencil reference (clip depth), 24-27 alpha layer, 28 projective
        const uint32_t flags_common = (kind << 4) | (cover_op << 7) | (clip_ref << 16) | (((it.refs >> 16) & 15u) << 24);
        const bool culled_kind = kind == KIND_COVER && cover_op == CRH_OP_COLOR && r.cull_mode != 0u; // Configuration::cull_mode: the colour cover only
        if (!PROJ || plain) {
#pragma unroll
        for (int v = 0; v < 3; ++v) p[v] = to_framebuffer(m, W, H, p[v].x, p[v].y);
        const float d1x = p[1].x - p[0].x, d1y = p[1].y - p[0].y;
        const float d2x = p[2].x - p[0].x, d2y = p[2].y - p[0].y;
        const float det = d1x * d2y - d2x * d1y;
        if (in_range_c && valid && det != 0.0f && det == det && is_finite(det)) {
            float minx = fminf(p[0].x, fminf(p[1].x, p[2].x)), maxx = fmaxf(p[0].x, fmaxf(p[1].x, p[2].x));
            float miny = fminf(p[0].y, fminf(p[1].y, p[2].y)), maxy = fmaxf(p[0].y, fmaxf(p[1].y, p[2].y));
            const bool nan_free = minx == minx && maxx == maxx && miny == miny && maxy == maxy;
            // inclusive pixel range: clamp, floor, THEN compare (oracle/raster.hpp setup_triangle)
            const int x0 = (int)floorf(fminf(fmaxf(minx, 0.0f), W)), x1 = (int)floorf(fmaxf(fminf(maxx, W - 1.0f), -1.0f));
            const int y0 = (int)floorf(fminf(fmaxf(miny, 0.0f), H)), y1 = (int)floorf(fmaxf(fminf(maxy, H - 1.0f), -1.0f));
            if (nan_free && x0 <= x1 && y0 <= y1 && !(culled_kind && (r.cull_mode == CRH_CULL_FRONT) == (det < 0.0f))) {
                drawn = true;
                rec.cov.box = make_ushort4((unsigned short)x0, (unsigned short)x1, (unsigned short)y0, (unsigned short)y1);
                const float inv_det = 1.0f / det;
                const bool front = det < 0.0f; // y-down cross < 0 == counter-clockwise on screen (FrontFace::Ccw, renderer.rs:477)
                const float2 nv[3] = {p[0], det < 0.0f ? p[2] : p[1], det < 0.0f ? p[1] : p[2]}; // clockwise-in-y-down edge walk
                uint32_t flags = (front ? 8u : 0u) | flags_common;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float2 a = nv[i], b = nv[(i + 1) % 3];
                    const float dx = b.x - a.x, dy = b.y - a.y;
                    if (dy < 0.0f || (dy == 0.0f && dx > 0.0f)) flags |= 1u << i; // top-left rule
                    const bool flip = !(a.x < b.x || (a.x == b.x && a.y < b.y));   // canonical (lexicographic) endpoint order
                    const float2 el = flip ? b : a, eh = flip ? a : b;
                    const float sg = flip ? -1.0f : 1.0f;
                    rec.cov.lo_x[i] = el.x;
                    rec.cov.lo_y[i] = el.y;
                    rec.cov.bx[i] = (eh.x - el.x) * sg;
                    rec.cov.nay[i] = -(eh.y - el.y) * sg;
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (a < n_attr) {
                        const float da1 = attr[1][a] - attr[0][a], da2 = attr[2][a] - attr[0][a];
                        rec.frag.a0[a] = attr[0][a];
                        rec.frag.gx[a] = (da1 * d2y - da2 * d1y) * inv_det;
                        rec.frag.gy[a] = (da2 * d1x - da1 * d2x) * inv_det;
                    } else {
                        rec.frag.a0[a] = rec.frag.gx[a] = rec.frag.gy[a] = 0.0f;
                    }
                }
                if (kind == KIND_COVER) { // color_cover: (rgb * a, a), shaders.wgsl:304-309
                    const float* color = r.colors + 4u * it.instance;
                    rec.frag.a0[0] = color[0] * color[3];
                    rec.frag.a0[1] = color[1] * color[3];
                    rec.frag.a0[2] = color[2] * color[3];
                    rec.frag.a0[3] = color[3];
                    rec.frag.gx[0] = m[14]; // the fragment depth of a plain instance
                }
                rec.frag.v0x = p[0].x;
                rec.frag.v0y = p[0].y;
                rec.frag.flat_u = flat_u;
                rec.frag.end_y = end_y;
                rec.cov.flags = flags;
                rec.cov.desc = desc;
            }
        }
        } else if (PROJ) {
        // ---- oracle/raster.hpp setup_projective + setup_projective_plane, operation by operation
        float PX[3], PY[3], PZ[3], PW[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const float cx = (m[0] * p[v].x + m[4] * p[v].y) + m[12];
            const float cy = (m[1] * p[v].x + m[5] * p[v].y) + m[13];
            const float cz = (m[2] * p[v].x + m[6] * p[v].y) + m[14];
            const float cw = (m[3] * p[v].x + m[7] * p[v].y) + m[15];
            PX[v] = (cx * 0.5f + cw * 0.5f) * W;
            PY[v] = (cw * 0.5f - cy * 0.5f) * H;
            PZ[v] = cz;
            PW[v] = cw;
        }
        const int k = PW[0] > 0.0f ? 0 : (PW[1] > 0.0f ? 1 : (PW[2] > 0.0f ? 2 : -1)); // the first vertex in front of the eye
        bool ok = in_range_c && valid && k >= 0;
#pragma unroll
        for (int v = 0; v < 3; ++v) ok = ok && is_finite(PX[v]) && is_finite(PY[v]) && is_finite(PZ[v]) && is_finite(PW[v]);
        const float c0 = PX[1] * PY[2] - PY[1] * PX[2], a0 = PY[1] * PW[2] - PW[1] * PY[2], b0 = PW[1] * PX[2] - PX[1] * PW[2];
        const float det = (PX[0] * a0 + PY[0] * b0) + PW[0] * c0;
        ok = ok && det != 0.0f && is_finite(det);
        const bool front = det < 0.0f;
        uint32_t flags = (front ? 8u : 0u) | flags_common | kFlagProjective;
        const int order[3] = {0, det < 0.0f ? 2 : 1, det < 0.0f ? 1 : 2};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int ia = order[i], ib = order[(i + 1) % 3];
            const float aX = PX[ia], aY = PY[ia], aW = PW[ia], bX = PX[ib], bY = PY[ib], bW = PW[ib];
            const bool flip = !(aX < bX || (aX == bX && (aY < bY || (aY == bY && aW < bW)))); // canonical (lexicographic) endpoint order
            const float lX = flip ? bX : aX, lY = flip ? bY : aY, lW = flip ? bW : aW, hX = flip ? aX : bX, hY = flip ? aY : bY, hW = flip ? aW : bW;
            const float nay = lY * hW - lW * hY, bx = lW * hX - lX * hW;
            const float sg = flip ? -1.0f : 1.0f;
            const float A = nay * sg, B = bx * sg; // exact: the oracle negates
            if (A > 0.0f || (A == 0.0f && B > 0.0f)) flags |= 1u << i;
            const bool use_lo = lW > 0.0f || (!(hW > 0.0f) && lW != 0.0f);
            const float nX = use_lo ? lX : hX, nY = use_lo ? lY : hY, nW = use_lo ? lW : hW;
            ok = ok && nW != 0.0f;
            const float ax = nX / nW, ay = nY / nW;
            ok = ok && is_finite(ax) && is_finite(ay);
            rec.cov.lo_x[i] = ax;
            rec.cov.lo_y[i] = ay;
            rec.cov.bx[i] = B;
            rec.cov.nay[i] = A;
        }
        int x0 = 0, y0 = 0, x1 = (int)r.width - 1, y1 = (int)r.height - 1; // crossing the eye plane: every pixel is a candidate
        if (PW[0] > 0.0f && PW[1] > 0.0f && PW[2] > 0.0f) {
            const float q0x = PX[0] / PW[0], q1x = PX[1] / PW[1], q2x = PX[2] / PW[2], q0y = PY[0] / PW[0], q1y = PY[1] / PW[1], q2y = PY[2] / PW[2];
            const float minx = fminf(q0x, fminf(q1x, q2x)), maxx = fmaxf(q0x, fmaxf(q1x, q2x));
            const float miny = fminf(q0y, fminf(q1y, q2y)), maxy = fmaxf(q0y, fmaxf(q1y, q2y));
            ok = ok && minx == minx && maxx == maxx && miny == miny && maxy == maxy;
            x0 = (int)floorf(fminf(fmaxf(minx, 0.0f), W));
            x1 = (int)floorf(fmaxf(fminf(maxx, W - 1.0f), -1.0f));
            y0 = (int)floorf(fminf(fmaxf(miny, 0.0f), H));
            y1 = (int)floorf(fmaxf(fminf(maxy, H - 1.0f), -1.0f));
            ok = ok && x0 <= x1 && y0 <= y1;
        }
        // planes relative to the anchor = the projection of vertex k (index 0 = k, then cyclic)
        const int kk = k < 0 ? 0 : k, ku = (kk + 1) % 3, kv = (kk + 2) % 3;
        float KX = 0.0f, KY = 0.0f, KZ = 0.0f, KW = 1.0f, UX = 0.0f, UY = 0.0f, UZ = 0.0f, UW = 1.0f, VX = 0.0f, VY = 0.0f, VZ = 0.0f, VW = 1.0f;
        float fk[4], fu[4], fv[4];
#pragma unroll
        for (int v = 0; v < 3; ++v) { // register-only selection (no dynamic indexing)
            if (v == kk) KX = PX[v], KY = PY[v], KZ = PZ[v], KW = PW[v];
            if (v == ku) UX = PX[v], UY = PY[v], UZ = PZ[v], UW = PW[v];
            if (v == kv) VX = PX[v], VY = PY[v], VZ = PZ[v], VW = PW[v];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (v == kk) fk[a] = attr[v][a];
                if (v == ku) fu[a] = attr[v][a];
                if (v == kv) fv[a] = attr[v][a];
            }
        }
        const float ancx = KX / KW, ancy = KY / KW;
        const float lx0 = UX - ancx * UW, ly0 = UY - ancy * UW, lx1 = VX - ancx * VW, ly1 = VY - ancy * VW;
        const float local_det = KW * (lx0 * ly1 - ly0 * lx1);
        ok = ok && local_det != 0.0f && is_finite(local_det);
        ok = ok && !(culled_kind && (r.cull_mode == CRH_CULL_FRONT) == front);
        if (ok) {
            drawn = true;
            rec.cov.box = make_ushort4((unsigned short)x0, (unsigned short)x1, (unsigned short)y0, (unsigned short)y1);
            const float inv_det = 1.0f / local_det;
            const float ak = ly0 * VW - UW * ly1, bk = UW * lx1 - lx0 * VW; // P'_u x P'_v
            const float au = ly1 * KW, bu = -(lx1 * KW);                    // P'_v x P'_k
            const float av = -(KW * ly0), bv = KW * lx0;                    // P'_k x P'_u
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (a < n_attr) {
                    rec.frag.a0[a] = fk[a] / KW;
                    rec.frag.gx[a] = ((fk[a] * ak + fu[a] * au) + fv[a] * av) * inv_det;
                    rec.frag.gy[a] = ((fk[a] * bk + fu[a] * bu) + fv[a] * bv) * inv_det;
                } else {
                    rec.frag.a0[a] = rec.frag.gx[a] = rec.frag.gy[a] = 0.0f;
                }
            }
            proj.ax = ancx;
            proj.ay = ancy;
            proj.q0 = 1.0f / KW;
            proj.qgx = ((1.0f * ak + 1.0f * au) + 1.0f * av) * inv_det;
            proj.qgy = ((1.0f * bk + 1.0f * bu) + 1.0f * bv) * inv_det;
            proj.z0 = KZ / KW;
            proj.zgx = ((KZ * ak + UZ * au) + VZ * av) * inv_det;
            proj.zgy = ((KZ * bk + UZ * bu) + VZ * bv) * inv_det;
            if (kind == KIND_COVER) {
                const float* color = r.colors + 4u * it.instance;
                rec.frag.a0[0] = color[0] * color[3];
                rec.frag.a0[1] = color[1] * color[3];
                rec.frag.a0[2] = color[2] * color[3];
                rec.frag.a0[3] = color[3];
            }
            rec.frag.v0x = ancx;
            rec.frag.v0y = ancy;
            rec.frag.flat_u = flat_u;
            rec.frag.end_y = end_y;
            rec.cov.flags = flags;
            rec.cov.desc = desc;
        }
        } // projective
        if (in_range_c) {
            if (drawn) {
                r.prim_rec[prim0 + c0 + lane] = rec;
                if (PROJ && !plain) r.prim_proj[prim0 + c0 + lane] = proj; // the host allocates the side array whenever an instance is not plain
            } else {
                r.prim_rec[prim0 + c0 + lane].cov.box = rec.cov.box;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- k_tile_walk
// Count pass (FILL = false) and fill pass (FILL = true) of the tile lists. One workgroup of kWalkWaves wavefronts per draw item; its
// triangles are taken 64 at a time (lane = triangle) and wavefront w walks every kWalkWaves-th tile ROW of the chunk's tile rectangle,
// so the largest Shapes (hundreds of tiles) do not leave one long serial tail. Stand-alone four wavefronts per item are fastest
// (1: 0.92 ms tail; 16: idle waves dominate); with frames overlapping two are, because the walks then share the CUs
// with the raster kernel and total work counts, not the tail.
//   count: ballot of the lanes whose triangle can touch the tile -> ONE atomic per (chunk, tile)
//   fill : lane 0 reserves popcount(ballot) slots of the tile's list with one returning atomic (consumed one iteration later, after
//          the next tile's test has been computed, so its latency is hidden); every hit lane writes
//          prim id at its rank.
constexpr uint32_t kWalkWaves = CRH_WALK_WAVES;
template <int S, bool FILL>
__global__ __launch_bounds__(64 * kWalkWaves) void k_tile_walk(SceneDev s, RasterParams r) {
    const uint32_t shape = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (FILL && r.overflow[0]) return;
    const uint32_t prim0 = r.shape_prim_begin[shape], n_candidates = r.shape_prim_begin[shape + 1] - prim0;
    if (prim0 + n_candidates > r.prim_capacity) return;
    for (uint32_t c0 = 0; c0 < n_candidates; c0 += 64u) {
        const uint32_t c = c0 + lane;
        ushort4 box = make_ushort4(0xFFFFu, 0, 0, 0);
        if (c < n_candidates) box = r.prim_rec[prim0 + c].cov.box;
        PrimCoverage cov;
        cov.box = box;
        const bool drawn = box.x != 0xFFFFu;
        uint32_t tx_a, tx_b, ty_a, ty_b;
        if (!wave_tile_rect(drawn, cov, tx_a, tx_b, ty_a, ty_b)) continue;
        if (ty_a + wave > ty_b) continue; // this wavefront has no row in the chunk's rectangle
        if (drawn) cov = r.prim_rec[prim0 + c].cov;
        unsigned long long pending_ballot = 0;
        uint32_t pending_entry = 0, pending_base = 0;
        auto flush = [&]() {
            if (!pending_ballot) return;
            const uint32_t base = __shfl(pending_base, 0, 64);
            if ((pending_ballot >> lane) & 1ull) r.tile_list[base + (uint32_t)__popcll(pending_ballot & ((1ull << lane) - 1ull))] = pending_entry;
        };
        // TileTest with everything that does not depend on the tile hoisted out of the loops (the same operations in the same order: the
        // result is bit-identical): the best corner of every edge, the row term of the edge constants, the row half of the box test
        float best_x[3], best_y[3];
        {
            const float lo = S == 1 ? 0.5f : 0.125f, hi = (float)(kTile - 1) + (S == 1 ? 0.5f : 0.875f);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                best_x[i] = cov.nay[i] > 0.0f ? hi : lo;
                best_y[i] = cov.bx[i] > 0.0f ? hi : lo;
            }
        }
        const int tl0 = (int)(cov.flags & 1u), tl1 = (int)((cov.flags >> 1) & 1u), tl2 = (int)((cov.flags >> 2) & 1u);
        for (uint32_t ty = ty_a + wave; ty <= ty_b; ty += kWalkWaves) {
            const int tpy = (int)(ty * kTile);
            const bool row_overlap = drawn && (int)cov.box.z <= tpy + kTile - 1 && (int)cov.box.w >= tpy;
            const float ty0 = (float)tpy;
            float row_term[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) row_term[i] = cov.bx[i] * (ty0 - cov.lo_y[i]);
            for (uint32_t tx = tx_a; tx <= tx_b; ++tx) {
                const int tpx = (int)(tx * kTile);
                const float tx0 = (float)tpx;
                bool hit = row_overlap && (int)cov.box.x <= tpx + kTile - 1 && (int)cov.box.y >= tpx;
                const float e0 = fmaf(best_y[0], cov.bx[0], fmaf(best_x[0], cov.nay[0], row_term[0] + cov.nay[0] * (tx0 - cov.lo_x[0])));
                const float e1 = fmaf(best_y[1], cov.bx[1], fmaf(best_x[1], cov.nay[1], row_term[1] + cov.nay[1] * (tx0 - cov.lo_x[1])));
                const float e2 = fmaf(best_y[2], cov.bx[2], fmaf(best_x[2], cov.nay[2], row_term[2] + cov.nay[2] * (tx0 - cov.lo_x[2])));
                hit = hit && (e0 > 0.0f || (e0 == 0.0f && tl0)) && (e1 > 0.0f || (e1 == 0.0f && tl1)) && (e2 > 0.0f || (e2 == 0.0f && tl2));
                const unsigned long long ballot = __ballot(hit);
                if (!ballot) continue;
                const uint32_t tile = ty * r.tiles_x + tx;
                if (!FILL) {
                    if (lane == 0) atomicAdd(&r.tile_count[tile], (uint32_t)__popcll(ballot));
                } else {
                    uint32_t base = 0;
                    if (lane == 0) base = r.tile_offset[tile] + atomicAdd(&r.tile_cursor[tile], (uint32_t)__popcll(ballot));
                    flush(); // the previous tile's atomic has had a whole iteration to return
                    pending_ballot = ballot;
                    pending_entry = prim0 + c;
                    pending_base = base;
                }
            }
        }
        if (FILL) flush();
    }
}

// ---------------------------------------------------------------------------------------------- k_raster_tile

// One workgroup per 16x16 tile, lane = (column px, row group rq), ROWS pixel rows per lane:
//   ROWS == 4 (msaa 1): ONE wavefront per tile; the lane owns the pixels (px, 4b + rq), b = 0..3, so a primitive's entry setup, tile
//                       constants and kind dispatch are paid once per (tile, primitive) and only the per-sample arithmetic repeats;
//   ROWS == 1 (msaa 4): four wavefronts per tile, wavefront w owns rows 4w..4w+3 — 4 samples x 1 row per lane keeps the per-lane state
//                       (winding + colour of every sample) at the same 20 registers instead of 80.
//   OPS == true: the full RenderOperation set — every sample also carries the clip nesting counter and up to kMaxAlphaLayers saved
//                alphas; OPS == false is the plain Stencil + Color pass at clip depth 0 (what the benchmark runs).
constexpr int kMaxAlphaLayers = 4;
//   STROKES == false: the scene has no stroked path, so the stroke fragment stages (and the registers their out-of-line dashed pattern
//                walk reserves) are compiled out: fewer VGPRs, more waves per SIMD.
template <int S, int ROWS, bool OPS, bool STROKES>
__global__ __launch_bounds__(64 * (4 / ROWS)) __attribute__((amdgpu_waves_per_eu((OPS || STROKES || S == 4) ? 1 : CRH_TILE_WAVES))) void k_raster_tile(SceneDev s, RasterParams r) {
    extern __shared__ uint32_t sort_buffer[]; // [waves][r.sort_capacity], wave-private; only used by tiles with more than 64 primitives
    __shared__ float4 entry_buffer[4 / ROWS][64 * 3];        // wave-private: the set-up values of the current chunk's 64 entries

    // XCD-aware tile order: workgroup b runs on XCD b % 8 (each XCD has its own L2). The frame is cut into 8x8-tile blocks dealt to the
    // XCDs in turn (spatially interleaved, so an unevenly filled frame still loads all eight), and an XCD walks a block's 64 tiles back to
    // back: a primitive record shared by neighbouring tiles is fetched into one L2 instead of up to eight. launch_raster pads the grid.
    constexpr uint32_t kB = CRH_XCD_BLOCK_LOG2, kBlock = 1u << kB;
    const uint32_t turn = blockIdx.x >> 3;
    const uint32_t blocks_x = (r.tiles_x + kBlock - 1u) >> kB, block = (turn >> (2u * kB)) * 8u + (blockIdx.x & 7u);
    uint32_t tx = (block % blocks_x) * kBlock + (turn & (kBlock - 1u)), ty = (block / blocks_x) * kBlock + ((turn >> kB) & (kBlock - 1u));
    if (r.tile_order) { // the host's order for this frame: every XCD's heavy tiles first (api.hip order_tiles_heavy_first)
        const uint32_t mine = r.tile_order[blockIdx.x];
        if (mine == 0xFFFFFFFFu) return;
        ty = mine / r.tiles_x, tx = mine - ty * r.tiles_x;
    }
    if (tx >= r.tiles_x || ty >= r.tiles_y || ty < r.slab_ty0 || ty >= r.slab_ty1) return; // (beyond the frame, or not in this pass' slab of tile rows)
    const uint32_t tile = ty * r.tiles_x + tx;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* __restrict__ keys = sort_buffer + wave * r.sort_capacity;
    const uint32_t px = lane & 15u, rq = lane >> 4;
    const uint32_t first_row = 4u * ROWS * wave; // local row b of this lane is pixel row first_row + 4b + rq (ROWS == 4: one wavefront, wave == 0)
    const uint32_t gx = tx * kTile + px;
    const float tx0 = (float)(tx * kTile), ty0 = (float)(ty * kTile);
    const int tpx = (int)(tx * kTile), tpy = (int)(ty * kTile);

    float sx[S], sy0[S]; // sample positions of the lane's first row; local row b adds 4b (exact in f32)
    if (S == 1) {
        sx[0] = (float)px + 0.5f;
        sy0[0] = (float)(first_row + rq) + 0.5f;
    } else {
        const float ox[4] = {0.375f, 0.875f, 0.125f, 0.625f}, oy[4] = {0.125f, 0.375f, 0.625f, 0.875f};
#pragma unroll
        for (int k = 0; k < S; ++k) {
            sx[k] = (float)px + ox[k & 3];
            sy0[k] = (float)(first_row + rq) + oy[k & 3];
        }
    }
    const uint32_t row_shift = first_row + rq; // the lane's row b is bit row_shift + 4b of a 16-bit row mask
    int winding[ROWS][S];
    int clipc[OPS ? ROWS : 1][OPS ? S : 1];                        // clip nesting counter (the upper stencil bits, renderer.rs:565)
    float saved[OPS ? ROWS : 1][OPS ? S : 1][kMaxAlphaLayers];     // alpha-context layers (renderer.rs:892-927)
#pragma unroll
    for (int b = 0; b < (OPS ? ROWS : 1); ++b)
#pragma unroll
        for (int k = 0; k < (OPS ? S : 1); ++k) {
            clipc[OPS ? b : 0][OPS ? k : 0] = 0;
#pragma unroll
            for (int l = 0; l < kMaxAlphaLayers; ++l) saved[b][k][l] = 0.0f;
        }
    float col[ROWS][S][4];
#pragma unroll
    for (int b = 0; b < ROWS; ++b)
#pragma unroll
        for (int k = 0; k < S; ++k) {
            winding[b][k] = 0;
            col[b][k][0] = col[b][k][1] = col[b][k][2] = col[b][k][3] = 0.0f;
        }
    // the depth attachment (OPS only): tested / written by the colour cover alone (renderer.rs:743-745)
    float depth[OPS ? ROWS : 1][OPS ? S : 1];
    const bool has_depth = OPS && r.depth != nullptr;
    if (OPS) {
#pragma unroll
        for (int b = 0; b < ROWS; ++b) {
            const uint32_t gy = ty * kTile + first_row + 4u * b + rq;
#pragma unroll
            for (int k = 0; k < S; ++k)
                depth[OPS ? b : 0][OPS ? k : 0] = (has_depth && gx < r.width && gy < r.height) ? r.depth[((size_t)gy * r.width + gx) * S + k] : 0.0f;
        }
    }
    if (r.load_existing) {
#pragma unroll
        for (int b = 0; b < ROWS; ++b) {
            const uint32_t gy = ty * kTile + first_row + 4u * b + rq;
            if (gx < r.width && gy < r.height) {
                const float4 d = load_pixel(r, gx, gy);
#pragma unroll
                for (int k = 0; k < S; ++k) col[b][k][0] = d.x, col[b][k][1] = d.y, col[b][k][2] = d.z, col[b][k][3] = d.w;
            }
        }
    }

    const uint32_t list_begin = r.tile_offset[tile];
    uint32_t n = r.overflow[0] ? 0u : r.tile_offset[tile + 1] - list_begin;
    constexpr uint32_t kLdsSortMax = kSortBytesMax / (4u * (4u / ROWS));
    if (n > r.sort_capacity && n <= kLdsSortMax) n = 0; // the host sizes the sort buffer from overflow[3] (the longest list) and runs the frame again
    // Pass state kept with the frame (renderer.rs:148-158, 257-266: the stencil attachment and the alpha layers outlive a Shape::render call):
    // the tile starts from what the earlier passes left — stencil byte = clip nesting counter << winding bits | winding counter
    // (renderer.rs:565-566, 936), the saved alphas, the colour of every SAMPLE — and leaves its own behind (below, in front of the resolve).
    const bool keeps_state = OPS && r.state_stencil != nullptr;
    if (keeps_state && r.state_load) {
        if (n == 0u && r.load_existing) return; // nothing of this pass touches the tile: planes and pixels stay as they are (a cleared frame's tiles are all written)
#pragma unroll
        for (int b = 0; b < ROWS; ++b) {
            const uint32_t gy = ty * kTile + first_row + 4u * b + rq;
            if (gx < r.width && gy < r.height) {
                const size_t at = ((size_t)gy * r.width + gx) * S;
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    const uint32_t st = r.state_stencil[at + k];
                    winding[b][k] = (int)(st & r.winding_mask);
                    clipc[OPS ? b : 0][OPS ? k : 0] = (int)((st >> r.winding_bits) & r.clip_mask_count);
                    const float4 c = reinterpret_cast<const float4*>(r.state_color)[at + k];
                    col[b][k][0] = c.x, col[b][k][1] = c.y, col[b][k][2] = c.z, col[b][k][3] = c.w;
#pragma unroll
                    for (int l = 0; l < kMaxAlphaLayers; ++l)
                        if ((uint32_t)l < r.state_layers) saved[OPS ? b : 0][OPS ? k : 0][l] = r.state_alpha[(size_t)l * r.width * r.height * S + at + k];
                }
            }
        }
    }
    // ---- draw order = ascending prim id: bitonic network in registers (<= 64 entries), in LDS, or — a list longer than LDS holds — in
    //      place in global memory
    uint32_t my_key = 0xFFFFFFFFu;
    const bool sorted_in_place = n > kLdsSortMax;
    uint32_t* const segment = r.tile_list + list_begin;
    if (sorted_in_place) {
        // Thousands of primitives over one tile (one Shape with 10^4 slivers through a point, hundreds of Shapes stacked): rare, so simple.
        // A normalised bitonic network — every compare-exchange leaves the smaller key at the lower index — sorts any length: positions
        // beyond n behave as +inf and are skipped. All wavefronts of the tile's workgroup take part; keys move through L2 (agent-scope
        // atomics) so that every lane sees what the others wrote.
        const uint32_t tid = threadIdx.x, n_threads = 64u * (4u / ROWS);
        uint32_t padded = 1;
        while (padded < n) padded <<= 1;
        auto exchange = [&](uint32_t i, uint32_t partner) {
            if (partner < n) {
                const uint32_t a = __hip_atomic_load(segment + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t b = __hip_atomic_load(segment + partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a > b) {
                    __hip_atomic_store(segment + i, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(segment + partner, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        };
        for (uint32_t k = 2; k <= padded; k <<= 1) {
            const uint32_t half = k >> 1;
            for (uint32_t p = tid; p < (padded >> 1); p += n_threads) { // the mirror step of the block of k
                const uint32_t block = p / half, t = p - block * half;
                exchange(block * k + t, block * k + k - 1u - t);
            }
            __threadfence();
            __syncthreads();
            for (uint32_t j = half >> 1; j > 0; j >>= 1) {
                for (uint32_t p = tid; p < (padded >> 1); p += n_threads) {
                    const uint32_t i = 2u * j * (p / j) + (p % j);
                    exchange(i, i + j);
                }
                __threadfence();
                __syncthreads();
            }
        }
    } else if (n <= 64u) {
        if (lane < n) my_key = r.tile_list[list_begin + lane];
#pragma unroll
        for (uint32_t k = 2; k <= 64u; k <<= 1) {
#pragma unroll
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                const uint32_t other = __shfl_xor(my_key, j, 64);
                const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
                my_key = keep_min ? min(my_key, other) : max(my_key, other);
            }
        }
    } else {
        uint32_t padded = 128;
        while (padded < n) padded <<= 1;
        for (uint32_t i = lane; i < padded; i += 64u) keys[i] = i < n ? r.tile_list[list_begin + i] : 0xFFFFFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = 2; k <= padded; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = lane; i < padded; i += 64u) {
                    const uint32_t partner = i ^ j;
                    if (partner > i) {
                        const uint32_t a = keys[i], b = keys[partner];
                        if (((i & k) == 0) ? (a > b) : (a < b)) {
                            keys[i] = b;
                            keys[partner] = a;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
    }

    const PrimRec* recs = r.prim_rec;
    const int wmask = (int)r.winding_mask;
    for (uint32_t q0 = 0; q0 < n; q0 += 64u) {
        if (sorted_in_place)
            my_key = q0 + lane < n ? __hip_atomic_load(segment + q0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
        else if (n > 64u)
            my_key = q0 + lane < n ? keys[q0 + lane] : 0xFFFFFFFFu;
        const uint32_t count = min(64u, n - q0);
        // ---- entry setup, vectorised across the chunk: lane j prepares entry j (one gathered 64-byte record per lane)
        uint32_t e_bits = 0, e_flags = 0, e_desc = 0;
        float e_c[3] = {0.0f, 0.0f, 0.0f}, e_bx[3] = {0.0f, 0.0f, 0.0f}, e_nay[3] = {0.0f, 0.0f, 0.0f};
        if (lane < count) {
            const PrimCoverage mine = recs[my_key].cov;
            const int bx0 = max((int)mine.box.x, tpx) - tpx, bx1 = min((int)mine.box.y, tpx + kTile - 1) - tpx;
            const int by0 = max((int)mine.box.z, tpy) - tpy, by1 = min((int)mine.box.w, tpy + kTile - 1) - tpy;
            const uint32_t col_bits = bx1 >= bx0 ? (2u << bx1) - (1u << bx0) : 0u, row_bits = by1 >= by0 ? (2u << by1) - (1u << by0) : 0u;
            e_bits = col_bits | (row_bits << 16); // columns / rows of the tile inside the triangle's clamped pixel box
            e_flags = mine.flags;
            e_desc = mine.desc;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                e_c[i] = mine.bx[i] * (ty0 - mine.lo_y[i]) + mine.nay[i] * (tx0 - mine.lo_x[i]);
                e_bx[i] = mine.bx[i];
                e_nay[i] = mine.nay[i];
            }
        }
        // staged in LDS so that the loop reads entry j with three uniform-address (broadcast) ds_read_b128 — the LDS pipe instead of a
        // dozen v_readlane on the VALU pipe, which is what bounds this kernel
        float4* __restrict__ entries = entry_buffer[wave];
        __builtin_amdgcn_wave_barrier(); // the previous chunk's reads are done
        entries[lane * 3u + 0u] = make_float4(e_c[0], e_c[1], e_c[2], __uint_as_float(e_bits));
        entries[lane * 3u + 1u] = make_float4(e_bx[0], e_bx[1], e_bx[2], __uint_as_float(e_flags));
        entries[lane * 3u + 2u] = make_float4(e_nay[0], e_nay[1], e_nay[2], __uint_as_float(e_desc));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t j = 0; j < count; ++j) {
            const uint32_t prim = __builtin_amdgcn_readlane(my_key, j);
            const float4 ea4 = entries[j * 3u + 0u], eb4 = entries[j * 3u + 1u], ec4 = entries[j * 3u + 2u];
            const uint32_t flags = __builtin_amdgcn_readfirstlane(__float_as_uint(eb4.w));
            const uint32_t kind = (flags >> 4) & 7u;
            const int clip_ref = OPS ? (int)((flags >> 16) & 255u) : 0; // the stencil reference of this draw: its clip depth
#ifndef CRH_NO_DEAD_COVER_SKIP
            // A colour cover over a tile in which no sample can pass its stencil test (Less: a non-zero winding, or a deeper clip level)
            // changes nothing: the blend needs a pass and the Zero operation leaves a winding that is already zero modulo the counter.
            // The long thin triangles of a hull strip mostly find the tile already cleared by their predecessors.
            if (kind == KIND_COVER && (!OPS || ((flags >> 7) & 7u) == (uint32_t)CRH_OP_COLOR)) {
                int live = 0;
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int k = 0; k < S; ++k) live |= OPS ? (int)((winding[b][k] & wmask) != 0 || clipc[OPS ? b : 0][OPS ? k : 0] > clip_ref) : (winding[b][k] & wmask);
                if (!__any(live != 0)) continue;
            }
#endif
            // the second half of the record (attribute planes / cover colour) comes through a scalar load issued up front
            PrimFragment frag;
            if (kind != KIND_SOLID) frag = load_uniform(&recs[prim].frag);
            // inside[b][k]  <=>  sample k of pixel (px, 4b + rq) is covered:
            //   edge i accepts e  <=>  e > 0 || (e == 0 && top_left_i)  <=>  as_int(e) >= 1 - top_left_i
            //   (edge values are finite and never -0: an exact zero sum rounds to +0 unless both addends are -0, which would need
            //    bx == nay == 0, i.e. a zero-length edge, and those triangles have det == 0 and are never set up)
            // Two (row, sample) combinations are evaluated together with packed f32 FMAs (v_pk_fma_f32: two IEEE fmas
            // per instruction, bit-identical to the scalar ones); the three compares produce lane masks that are combined on the scalar
            // unit; "the pixel is inside the clamped box" is a bit lookup in (column mask, row mask).
            bool inside[ROWS][S];
            {
                const uint32_t bits = __float_as_uint(ea4.w);
                const float c0 = ea4.x, c1 = ea4.y, c2 = ea4.z, bx_0 = eb4.x, bx_1 = eb4.y, bx_2 = eb4.z, nay_0 = ec4.x, nay_1 = ec4.y, nay_2 = ec4.z;
                const uint32_t lane_rows = ((bits >> px) & 1u) ? (bits >> 16) >> row_shift : 0u; // bit 4b: the lane's row b is inside the box
                const int thr0 = 1 - (int)(flags & 1u), thr1 = 1 - (int)((flags >> 1) & 1u), thr2 = 1 - (int)((flags >> 2) & 1u);
                // E = fma(ry, bx, fma(rx, nay, c)): the column term is shared by the rows of the lane (one fma per edge and sample position)
                float ha[S], hb[S], hc[S];
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    ha[k] = fmaf(sx[k], nay_0, c0);
                    hb[k] = fmaf(sx[k], nay_1, c1);
                    hc[k] = fmaf(sx[k], nay_2, c2);
                }
                // the ROWS * S (row, sample) combinations are taken two at a time
#pragma unroll
                for (int c = 0; c < ROWS * S; c += 2) {
                    const int b0 = c / S, k0 = c % S, b1 = (c + 1) / S, k1 = (c + 1) % S;
                    const f32x2 y = {sy0[k0] + (float)(4 * b0), sy0[k1] + (float)(4 * b1)};
                    const f32x2 ea = fma2(y, splat2(bx_0), f32x2{ha[k0], ha[k1]});
                    const f32x2 eb = fma2(y, splat2(bx_1), f32x2{hb[k0], hb[k1]});
                    const f32x2 ec = fma2(y, splat2(bx_2), f32x2{hc[k0], hc[k1]});
                    inside[b0][k0] = (__float_as_int(ea[0]) >= thr0) & (__float_as_int(eb[0]) >= thr1) & (__float_as_int(ec[0]) >= thr2) & ((lane_rows & (1u << (4 * b0))) != 0u);
                    inside[b1][k1] = (__float_as_int(ea[1]) >= thr0) & (__float_as_int(eb[1]) >= thr1) & (__float_as_int(ec[1]) >= thr2) & ((lane_rows & (1u << (4 * b1))) != 0u);
                }
            }
            // projective instances (oracle/raster.hpp raster_projective): per-sample near / far test on z/w, and 1 / (1/w) for the attributes
            float rw[OPS ? ROWS : 1][OPS ? S : 1], zs[OPS ? ROWS : 1][OPS ? S : 1];
            if (OPS) {
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int k = 0; k < S; ++k) {
                        rw[OPS ? b : 0][OPS ? k : 0] = 1.0f;
                        zs[OPS ? b : 0][OPS ? k : 0] = 0.0f;
                    }
                if (flags & kFlagProjective) { // wave uniform
                    const PrimProj pp = load_uniform(&r.prim_proj[prim]);
                    const float dxa = tx0 - pp.ax, dya = ty0 - pp.ay;
                    const float zc = fmaf(dya, pp.zgy, fmaf(dxa, pp.zgx, pp.z0)), qc = fmaf(dya, pp.qgy, fmaf(dxa, pp.qgx, pp.q0));
#pragma unroll
                    for (int b = 0; b < ROWS; ++b)
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            const float y = sy0[k] + (float)(4 * b);
                            const float z = fmaf(y, pp.zgy, fmaf(sx[k], pp.zgx, zc));
                            const float q = fmaf(y, pp.qgy, fmaf(sx[k], pp.qgx, qc));
                            inside[b][k] = inside[b][k] & (z >= 0.0f) & (z <= 1.0f); // unclipped_depth: false (renderer.rs:478)
                            zs[OPS ? b : 0][OPS ? k : 0] = z;
                            rw[OPS ? b : 0][OPS ? k : 0] = 1.0f / q;
                        }
                }
            }
            const int delta = (flags & 8u) ? 1 : -1; // front (ccw on screen) increments, back decrements (renderer.rs:577-582)
            // Every kind only produces the change of the winding counters (dw) and, for the colour cover, which samples blend; the state
            // itself is updated once after the dispatch. (Updating winding / colour inside the multi-way dispatch made every iteration end
            // with ~18 register-pair copies: the SSA join of 20 state registers over all kinds.)
            int dw[ROWS][S];
            bool blend[ROWS][S];
#pragma unroll
            for (int b = 0; b < ROWS; ++b)
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    dw[b][k] = 0;
                    blend[b][k] = false;
                }
            const uint32_t cover_op = OPS ? (flags >> 7) & 7u : (uint32_t)CRH_OP_COLOR;
            const bool color_cover = kind == KIND_COVER && cover_op == CRH_OP_COLOR;
            if (kind == KIND_SOLID) { // stencil_solid
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int k = 0; k < S; ++k) dw[b][k] = (inside[b][k] && (!OPS || clipc[OPS ? b : 0][OPS ? k : 0] >= clip_ref)) ? delta : 0; // LessEqual(ref <= stencil)
            } else if (kind == KIND_COVER) {
                if (cover_op == CRH_OP_COLOR) { // stencil Less / Zero of color_cover (renderer.rs:747-752); the blend itself follows the dispatch
#pragma unroll
                    for (int b = 0; b < ROWS; ++b)
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            // Less(ref < stencil) on clip | winding: a deeper clip level, or this level with a non-zero winding
                            const bool stencil_pass = OPS ? (clipc[OPS ? b : 0][OPS ? k : 0] > clip_ref || (clipc[OPS ? b : 0][OPS ? k : 0] == clip_ref && (winding[b][k] & wmask) != 0)) : (winding[b][k] & wmask) != 0;
                            if (OPS) {
                                // the depth test follows the stencil test; depth_fail_op = Keep (renderer.rs:442): the winding survives a depth fail
                                const float z = (flags & kFlagProjective) ? zs[OPS ? b : 0][OPS ? k : 0] : frag.gx[0];
                                const float stored = depth[OPS ? b : 0][OPS ? k : 0];
                                const uint32_t relation = (z < stored ? 1u : 0u) | (z == stored ? 2u : 0u) | (z > stored ? 4u : 0u) | 8u;
                                const bool depth_pass = !has_depth || (relation & r.depth_pass_mask) != 0u;
                                const bool depth_fail = inside[b][k] && stencil_pass && !depth_pass;
                                blend[b][k] = inside[b][k] && stencil_pass && depth_pass;
                                dw[b][k] = (inside[b][k] && !depth_fail) ? -winding[b][k] : 0;
                                depth[OPS ? b : 0][OPS ? k : 0] = (blend[b][k] && has_depth && r.depth_write != 0u) ? z : stored;
                            } else {
                                blend[b][k] = inside[b][k] && stencil_pass;
                                dw[b][k] = inside[b][k] ? -winding[b][k] : 0; // pass -> Zero, fail -> Zero
                            }
                        }
                } else if (OPS) {
                    // Clip / UnClip / the alpha-context covers, branch-free per sample (the operation is wave uniform)
                    const uint32_t layer = (flags >> 24) & 15u;
                    const float ca = frag.a0[3]; // the instance colour's alpha
                    const bool is_clip = cover_op == CRH_OP_CLIP, is_unclip = cover_op == CRH_OP_UNCLIP, is_save = cover_op == CRH_OP_SAVE_ALPHA_CONTEXT;
                    const bool is_scale = cover_op == CRH_OP_SCALE_ALPHA_CONTEXT, is_restore = cover_op == CRH_OP_RESTORE_ALPHA_CONTEXT;
                    const float scale_src = 1.0f - ca; // scale_alpha_context_cover: src = (0, 0, 0, 1 - a), shaders.wgsl:311-316
#pragma unroll
                    for (int b = 0; b < ROWS; ++b)
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            const int cb_ = OPS ? b : 0, ck_ = OPS ? k : 0;
                            const bool in = inside[b][k];
                            const int clip_now = clipc[cb_][ck_];
                            // Clip: NotEqual on the winding bits -> Replace(ref) (renderer.rs:703-708); UnClip: Less on the clip bits (ref < stencil)
                            // -> Replace(ref) (renderer.rs:722-727); both rewrite clip | winding
                            const bool replace = in && ((is_clip && (winding[b][k] & wmask) != 0) || (is_unclip && clip_ref < clip_now));
                            clipc[cb_][ck_] = replace ? clip_ref : clip_now;
                            dw[b][k] = replace ? -winding[b][k] : 0;
                            // alpha-context covers: LessEqual(ref <= stencil), stencil untouched (renderer.rs:761-766)
                            const bool pass = in && clip_now >= clip_ref;
                            const float alpha = col[b][k][3];
                            float mine = 0.0f;
#pragma unroll
                            for (int l = 0; l < kMaxAlphaLayers; ++l) mine = (uint32_t)l == layer ? saved[cb_][ck_][l] : mine;
                            const float scaled = scale_src + alpha * (1.0f - scale_src);      // alpha' = src.a * One + dst.a * (1 - src.a), renderer.rs:803-828
                            const float restored = alpha - (1.0f - mine) * (1.0f - ca);        // alpha' = dst.a - (1 - saved)(1 - a), renderer.rs:829-861
                            float next = is_scale ? scaled : (is_restore ? restored : alpha);
                            if (r.format == CRH_FORMAT_RGBA8_ATTACHMENT) next = attachment_unorm8(next); // an Rgba8Unorm attachment keeps 8 bits of what the blender writes
                            col[b][k][3] = pass ? next : alpha;
#pragma unroll
                            for (int l = 0; l < kMaxAlphaLayers; ++l) // save_alpha_context_cover: the layer receives the frame's alpha, shaders.wgsl:326-331
                                saved[cb_][ck_][l] = (pass && is_save && (uint32_t)l == layer) ? alpha : saved[cb_][ck_][l];
                        }
                }
            } else {
            // attribute planes, tile relative: ac = fma(ty0 - v0y, gy, fma(tx0 - v0x, gx, a0)); a = fma(sy, gy, fma(sx, gx, ac))
            const float dx0 = tx0 - frag.v0x, dy0 = ty0 - frag.v0y;
            float hx[4][S]; // the row-independent inner fma
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float ac = fmaf(dy0, frag.gy[t], fmaf(dx0, frag.gx[t], frag.a0[t]));
#pragma unroll
                for (int k = 0; k < S; ++k) hx[t][k] = fmaf(sx[k], frag.gx[t], ac);
            }
            if (kind <= KIND_RC) { // the four implicit-curve tests (shaders.wgsl:236-266)
#pragma unroll
                for (int b = 0; b < ROWS; ++b) {
                    bool row_touched = false;
#pragma unroll
                    for (int k = 0; k < S; ++k) row_touched = row_touched | inside[b][k];
                    if (!__any(row_touched)) continue; // curve triangles are small: most rows of the tile are not touched
#pragma unroll
                    for (int k = 0; k < S; ++k) {
                        const float y = sy0[k] + (float)(4 * b);
                        // (x * 1.0f is exact, so the plain instances of an OPS pass keep their bits)
                        const float w = OPS ? rw[OPS ? b : 0][OPS ? k : 0] : 1.0f;
                        const float a0 = OPS ? fmaf(y, frag.gy[0], hx[0][k]) * w : fmaf(y, frag.gy[0], hx[0][k]);
                        const float a1 = OPS ? fmaf(y, frag.gy[1], hx[1][k]) * w : fmaf(y, frag.gy[1], hx[1][k]);
                        const float a2 = OPS ? fmaf(y, frag.gy[2], hx[2][k]) * w : fmaf(y, frag.gy[2], hx[2][k]);
                        const float a3 = OPS ? fmaf(y, frag.gy[3], hx[3][k]) * w : fmaf(y, frag.gy[3], hx[3][k]);
                        const float lhs = (kind == KIND_IQ || kind == KIND_RQ) ? a0 * a0 : a0 * a0 * a0;
                        const float rhs = kind == KIND_IQ ? a1 : (kind == KIND_RC ? a1 * a2 * a3 : a1 * a2);
                        dw[b][k] = (inside[b][k] && (!OPS || clipc[OPS ? b : 0][OPS ? k : 0] >= clip_ref) && lhs - rhs <= 0.0f) ? delta : 0;
                    }
                }
            } else if (STROKES) { // KIND_LINE / KIND_JOINT: the stroke fragment stages
                int any_inside = 0;
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int k = 0; k < S; ++k) any_inside |= (int)inside[b][k];
                if (__any(any_inside)) {
                    const crh_dynamic_stroke_descriptor d = load_uniform(&s.descriptors[__builtin_amdgcn_readfirstlane(__float_as_uint(ec4.w))]); // 48 B, scalar loads
                    const uint32_t caps = d.caps, count_dashed_join = d.count_dashed_join; // wave uniform
                    const uint32_t flat_u = frag.flat_u;
                    const float end_y = frag.end_y;
                    const bool dashed = (count_dashed_join & 4u) != 0u;
#pragma unroll
                    for (int b = 0; b < ROWS; ++b)
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            // stroke stencil: Equal(0) -> IncrementWrap, both faces (renderer.rs:571-576)
                            if (inside[b][k] && (winding[b][k] & wmask) == 0 && (!OPS || clipc[OPS ? b : 0][OPS ? k : 0] == clip_ref)) { // Equal(ref) on clip | winding
                                const float y = sy0[k] + (float)(4 * b);
                                const float w = OPS ? rw[OPS ? b : 0][OPS ? k : 0] : 1.0f;
                                const float a0 = OPS ? fmaf(y, frag.gy[0], hx[0][k]) * w : fmaf(y, frag.gy[0], hx[0][k]);
                                const float a1 = OPS ? fmaf(y, frag.gy[1], hx[1][k]) * w : fmaf(y, frag.gy[1], hx[1][k]);
                                const float a2 = OPS ? fmaf(y, frag.gy[2], hx[2][k]) * w : fmaf(y, frag.gy[2], hx[2][k]);
                                bool fill;
                                if (kind == KIND_LINE) { // stencil_stroke_line, shaders.wgsl:268-285
                                    if (dashed)
                                        fill = stroke_dashed(d, a0, a1);
                                    else if ((flat_u & 65536u) != 0u)
                                        fill = cap_test(a0, a1 - end_y, caps >> 4);
                                    else if (a1 < 0.0f)
                                        fill = cap_test(a0, -a1, caps);
                                    else
                                        fill = true;
                                } else { // stencil_stroke_joint, shaders.wgsl:287-300
                                    const float radius = sqrtf(a0 * a0 + a1 * a1);
                                    const uint32_t join = count_dashed_join & 3u;
                                    fill = join == 1u ? (flat_u & 65536u) != 0u : (join == 2u ? radius <= 0.5f : true);
                                    if (fill && dashed) fill = stroke_dashed_joint(d, radius, a0, a1, a2);
                                }
                                dw[b][k] = fill ? 1 : 0;
                            }
                        }
                }
            }
            } // kinds with attribute planes
#pragma unroll
            for (int b = 0; b < ROWS; ++b)
#pragma unroll
                for (int k = 0; k < S; ++k) winding[b][k] += dw[b][k];
            if (color_cover) { // color_cover: premultiplied "over" (shaders.wgsl:304-309, blending of examples/showcase/main.rs:32-43)
                const float s0 = frag.a0[0], s1 = frag.a0[1], s2 = frag.a0[2], ca = frag.a0[3];
                const float one_minus_a = 1.0f - ca;
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int k = 0; k < S; ++k) {
                        const float n0 = s0 + col[b][k][0] * one_minus_a, n1 = s1 + col[b][k][1] * one_minus_a;
                        const float n2 = s2 + col[b][k][2] * one_minus_a, n3 = ca + col[b][k][3] * one_minus_a;
                        col[b][k][0] = blend[b][k] ? n0 : col[b][k][0];
                        col[b][k][1] = blend[b][k] ? n1 : col[b][k][1];
                        col[b][k][2] = blend[b][k] ? n2 : col[b][k][2];
                        col[b][k][3] = blend[b][k] ? n3 : col[b][k][3];
                    }
                if (r.format == CRH_FORMAT_RGBA8_ATTACHMENT) { // an Rgba8Unorm attachment keeps 8 bits of what the blender writes (idempotent on the others)
#pragma unroll
                    for (int b = 0; b < ROWS; ++b)
#pragma unroll
                        for (int k = 0; k < S; ++k)
#pragma unroll
                            for (int ch = 0; ch < 4; ++ch) col[b][k][ch] = attachment_unorm8(col[b][k][ch]);
                }
            }
        }
    }
    if (OPS && has_depth && r.depth_write != 0u) {
#pragma unroll
        for (int b = 0; b < ROWS; ++b) {
            const uint32_t gy = ty * kTile + first_row + 4u * b + rq;
            if (gx < r.width && gy < r.height) {
#pragma unroll
                for (int k = 0; k < S; ++k) r.depth[((size_t)gy * r.width + gx) * S + k] = depth[OPS ? b : 0][OPS ? k : 0];
            }
        }
    }
    if (keeps_state) { // what this pass leaves to the next one (the winding counter wraps inside its bits: IncrementWrap / DecrementWrap under the write mask, renderer.rs:577-582)
#pragma unroll
        for (int b = 0; b < ROWS; ++b) {
            const uint32_t gy = ty * kTile + first_row + 4u * b + rq;
            if (gx < r.width && gy < r.height) {
                const size_t at = ((size_t)gy * r.width + gx) * S;
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    r.state_stencil[at + k] = (uint8_t)(((uint32_t)winding[b][k] & r.winding_mask) | (((uint32_t)clipc[OPS ? b : 0][OPS ? k : 0] & r.clip_mask_count) << r.winding_bits));
                    reinterpret_cast<float4*>(r.state_color)[at + k] = make_float4(col[b][k][0], col[b][k][1], col[b][k][2], col[b][k][3]);
#pragma unroll
                    for (int l = 0; l < kMaxAlphaLayers; ++l)
                        if ((uint32_t)l < r.state_layers) r.state_alpha[(size_t)l * r.width * r.height * S + at + k] = saved[OPS ? b : 0][OPS ? k : 0][l];
                }
            }
        }
    }
    // ---- MSAA resolve (box average) + RGBA8 unorm store
#pragma unroll
    for (int b = 0; b < ROWS; ++b) {
        const uint32_t gy = ty * kTile + first_row + 4u * b + rq;
        if (gx < r.width && gy < r.height) {
            const float inv = 1.0f / (float)S;
            float avg[4];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                float sum = 0.0f;
#pragma unroll
                for (int k = 0; k < S; ++k) sum = sum + col[b][k][ch];
                avg[ch] = sum * inv;
            }
            store_pixel(r, gx, gy, avg[0], avg[1], avg[2], avg[3]);
        }
    }
}

// ordered premultiplied "over" of n RGBA8 layers: dst = L0 under L1 under ...
__global__ __launch_bounds__(256) void k_composite(const uint8_t* const* layers, uint32_t n_layers, uint64_t n_pixels, uint8_t* dst) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n_pixels) return;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (uint32_t l = 0; l < n_layers; ++l) {
        const uchar4 p = reinterpret_cast<const uchar4*>(layers[l])[i];
        const float sr[4] = {(float)p.x * (1.0f / 255.0f), (float)p.y * (1.0f / 255.0f), (float)p.z * (1.0f / 255.0f), (float)p.w * (1.0f / 255.0f)};
        const float k = 1.0f - sr[3];
        for (int c = 0; c < 4; ++c) acc[c] = sr[c] + acc[c] * k;
    }
    uint32_t packed = 0;
    for (int c = 0; c < 4; ++c) {
        float x = acc[c] < 0.0f ? 0.0f : (acc[c] > 1.0f ? 1.0f : acc[c]);
        packed |= (uint32_t)(int)(x * 255.0f + 0.5f) << (8 * c);
    }
    reinterpret_cast<uint32_t*>(dst)[i] = packed;
}

__global__ __launch_bounds__(256) void k_state_colors_from_image(RasterParams r, uint32_t samples) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint64_t)r.width * r.height) return;
    const uint32_t gy = (uint32_t)(i / r.width), gx = (uint32_t)(i - (uint64_t)gy * r.width);
    const float4 c = load_pixel(r, gx, gy);
    for (uint32_t k = 0; k < samples; ++k) reinterpret_cast<float4*>(r.state_color)[i * samples + k] = c;
}

// ---------------------------------------------------------------------------------------------- launchers
static ScanJob scan_job(const uint32_t* in, uint32_t* out, uint32_t* block_sum, uint32_t n, uint32_t* max_out = nullptr) {
    return ScanJob{in, out, block_sum, n, (n + 1023u) / 1024u, max_out};
}

// transform independent: contiguous primitive ids per Shape (runs at the end of tessellation)
void launch_prim_ranges(const SceneDev& s, uint32_t* shape_ncand, uint32_t* shape_prim_begin, uint32_t* scratch, hipStream_t stream) {
    if (s.n_shapes == 0) {
        (void)hipMemsetAsync(shape_prim_begin, 0, 4, stream);
        return;
    }
    hipLaunchKernelGGL(k_shape_ncand, dim3((s.n_shapes + 255u) / 256u), dim3(256), 0, stream, s, shape_ncand);
    const ScanJob j = scan_job(shape_ncand, shape_prim_begin, scratch, s.n_shapes);
    RasterParams unused = {};
    hipLaunchKernelGGL(k_scan_local, dim3(j.blocks), dim3(256), 0, stream, j);
    hipLaunchKernelGGL(k_scan_add, dim3(j.blocks), dim3(256), 0, stream, j, unused, 0);
}
// per frame, for a recorded pass: candidate counts and contiguous primitive ids per draw item
void launch_item_ranges(const SceneDev& s, const RasterParams& r, uint32_t* item_ncand, uint32_t* item_prim_begin, uint32_t* scratch, hipStream_t stream) {
    hipLaunchKernelGGL(k_item_ncand, dim3((r.n_items + 255u) / 256u), dim3(256), 0, stream, s, r, item_ncand);
    const ScanJob j = scan_job(item_ncand, item_prim_begin, scratch, r.n_items);
    RasterParams unused = {};
    hipLaunchKernelGGL(k_scan_local, dim3(j.blocks), dim3(256), 0, stream, j);
    hipLaunchKernelGGL(k_scan_add, dim3(j.blocks), dim3(256), 0, stream, j, unused, 0);
}
// `after_setup` (optional) is recorded when k_prim_setup, the last reader of the tessellated vertex streams, has been enqueued
void launch_bin(const SceneDev& s, const RasterParams& r, uint32_t samples, hipStream_t stream, void (*mark)(void*, const char*, uint64_t), void* ctx, hipEvent_t after_setup) {
    // tile_cursor, tile_count and the overflow words are adjacent: one memset ([5] belongs to the edge pass (raster_edges.hip); cleared so that the host never sees a stale flag)
    (void)hipMemsetAsync(r.tile_cursor, 0, sizeof(uint32_t) * 2u * r.n_tiles + 32, stream);
    if (r.n_items) {
        if (samples == 4) {
            if (r.prim_proj)
                hipLaunchKernelGGL((k_prim_setup<4, true>), dim3(r.n_items), dim3(64), 0, stream, s, r);
            else
                hipLaunchKernelGGL((k_prim_setup<4, false>), dim3(r.n_items), dim3(64), 0, stream, s, r);
        } else {
            if (r.prim_proj)
                hipLaunchKernelGGL((k_prim_setup<1, true>), dim3(r.n_items), dim3(64), 0, stream, s, r);
            else
                hipLaunchKernelGGL((k_prim_setup<1, false>), dim3(r.n_items), dim3(64), 0, stream, s, r);
        }
    }
    if (after_setup) (void)hipEventRecord(after_setup, stream);
    if (mark) mark(ctx, "raster_prim_setup", 0);
    if (r.n_items) {
        if (samples == 4)
            hipLaunchKernelGGL((k_tile_walk<4, false>), dim3(r.n_items), dim3(64 * kWalkWaves), 0, stream, s, r);
        else
            hipLaunchKernelGGL((k_tile_walk<1, false>), dim3(r.n_items), dim3(64 * kWalkWaves), 0, stream, s, r);
    }
    if (mark) mark(ctx, "raster_tile_count", 0);
    const ScanJob j = scan_job(r.tile_count, r.tile_offset, r.scan_scratch, r.n_tiles, r.overflow + 3); // overflow[3] = longest tile list
    hipLaunchKernelGGL(k_scan_local, dim3(j.blocks), dim3(256), 0, stream, j);
    hipLaunchKernelGGL(k_scan_add, dim3(j.blocks), dim3(256), 0, stream, j, r, 1);
    if (mark) mark(ctx, "raster_tile_scan", 0);
}
// `after_fill` (optional) is recorded when the fill pass, the last reader of the per-item primitive ranges, has been enqueued
void launch_fill(const SceneDev& s, const RasterParams& r, uint32_t samples, hipStream_t stream, void (*mark)(void*, const char*, uint64_t), void* ctx,
                 hipEvent_t after_fill) {
    if (r.n_items) {
        if (samples == 4)
            hipLaunchKernelGGL((k_tile_walk<4, true>), dim3(r.n_items), dim3(64 * kWalkWaves), 0, stream, s, r);
        else
            hipLaunchKernelGGL((k_tile_walk<1, true>), dim3(r.n_items), dim3(64 * kWalkWaves), 0, stream, s, r);
    }
    if (after_fill) (void)hipEventRecord(after_fill, stream);
    if (mark) mark(ctx, "raster_tile_fill", 0);
}
void launch_raster(const SceneDev& s, const RasterParams& r, uint32_t samples, hipStream_t stream, void (*mark)(void*, const char*, uint64_t), void* ctx,
                   uint64_t raster_bytes, bool has_stroke) {
    // 8x8-tile blocks, an equal number per XCD (k_raster_tile's tile order)
    constexpr uint32_t kBlock = 1u << CRH_XCD_BLOCK_LOG2;
    const uint32_t blocks = ((r.tiles_x + kBlock - 1u) / kBlock) * ((r.tiles_y + kBlock - 1u) / kBlock);
    const dim3 grid((r.tile_order && r.order_places) ? r.order_places : ((blocks + 7u) / 8u) * kBlock * kBlock * 8u);
#define CRH_LAUNCH_TILE(S_, ROWS_, OPS_, STROKES_) \
    hipLaunchKernelGGL((k_raster_tile<S_, ROWS_, OPS_, STROKES_>), grid, dim3(64 * (4 / ROWS_)), (4 / ROWS_) * r.sort_capacity * 4u, stream, s, r)
    if (samples == 4) {
        if (r.general) { // clip nesting / alpha contexts / depth / projective instances: the OPS variant
            if (has_stroke) CRH_LAUNCH_TILE(4, 1, true, true); else CRH_LAUNCH_TILE(4, 1, true, false);
        } else {
            if (has_stroke) CRH_LAUNCH_TILE(4, 1, false, true); else CRH_LAUNCH_TILE(4, 1, false, false);
        }
    } else {
        if (r.general) { // clip nesting / alpha contexts / depth / projective instances: the OPS variant
            if (has_stroke) CRH_LAUNCH_TILE(1, 4, true, true); else CRH_LAUNCH_TILE(1, 4, true, false);
        } else {
            if (has_stroke) CRH_LAUNCH_TILE(1, 4, false, true); else CRH_LAUNCH_TILE(1, 4, false, false);
        }
    }
#undef CRH_LAUNCH_TILE
    if (mark) mark(ctx, "raster_tiles", raster_bytes);
}
// exported to raster_edges.hip
void launch_scan_u32(const uint32_t* in, uint32_t* out, uint32_t* block_sum, uint32_t n, hipStream_t stream) {
    const ScanJob j = scan_job(in, out, block_sum, n);
    RasterParams unused = {};
    hipLaunchKernelGGL(k_scan_local, dim3(j.blocks), dim3(256), 0, stream, j);
    hipLaunchKernelGGL(k_scan_add, dim3(j.blocks), dim3(256), 0, stream, j, unused, 0);
}
void launch_scan_u32_pair(const uint32_t* in0, uint32_t* out0, uint32_t* block_sum0, const uint32_t* in1, uint32_t* out1, uint32_t* block_sum1, uint32_t n, hipStream_t stream) {
    const ScanJob a = scan_job(in0, out0, block_sum0, n), b = scan_job(in1, out1, block_sum1, n);
    hipLaunchKernelGGL(k_scan_local2, dim3(a.blocks, 2), dim3(256), 0, stream, a, b);
    hipLaunchKernelGGL(k_scan_add2, dim3(a.blocks, 2), dim3(256), 0, stream, a, b);
}
void launch_scan_tiles(const RasterParams& r, hipStream_t stream) { // tile_count -> tile_offset; publishes the pair total and the longest list
    const ScanJob j = scan_job(r.tile_count, r.tile_offset, r.scan_scratch, r.n_tiles, r.overflow + 3);
    hipLaunchKernelGGL(k_scan_local, dim3(j.blocks), dim3(256), 0, stream, j);
    hipLaunchKernelGGL(k_scan_add, dim3(j.blocks), dim3(256), 0, stream, j, r, 1);
}
// A frame that starts keeping its pass state while it already shows an image: every sample of a pixel starts from the pixel's resolved colour
// (what a pass over existing content has always read, load_pixel).
void launch_state_colors_from_image(const RasterParams& r, uint32_t samples, hipStream_t stream) {
    const uint64_t n = (uint64_t)r.width * r.height;
    hipLaunchKernelGGL(k_state_colors_from_image, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, r, samples);
}
void launch_composite(const uint8_t* const* layers_dev, uint32_t n_layers, uint64_t n_pixels, uint8_t* dst, hipStream_t stream) {
    hipLaunchKernelGGL(k_composite, dim3((uint32_t)((n_pixels + 255) / 256)), dim3(256), 0, stream, layers_dev, n_layers, n_pixels, dst);
}

} // namespace crh
