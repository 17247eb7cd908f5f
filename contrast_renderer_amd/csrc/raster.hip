// csrc/raster.hip — tile-binned compute rasterizer that replaces the reference's stencil-then-cover passes
// (Shape::render renderer.rs:267-355, stencil states renderer.rs:565-582,736-754, fragment stage shaders.wgsl:155-309).
//
//   k_shape_bounds  one lane per Shape : framebuffer bounding box of its geometry -> 16x16-tile rectangle, per-tile counts
//   k_tile_scan     one workgroup      : exclusive scan of the per-tile counts
//   k_bin           one lane per Shape : append the shape to every tile of its rectangle (atomic slot, sorted later)
//   k_raster<S>     one workgroup per tile, one lane per pixel, S samples per lane:
//        sort the tile's shape list (painter's order == shape index order), then per shape
//        - lanes set up candidate triangles (one per lane): transform, edge functions, attribute planes, tile overlap;
//          survivors are compacted IN ORDER into an LDS primitive list (ballot + popcount),
//        - every lane walks the list and updates the winding counters of its own samples (registers, the reference's
//          8-bit stencil), evaluating the implicit-curve / cap / join / dash fragment tests at the sample,
//        - the cover primitives (hull strip) blend the shape's colour where winding != 0 and zero the counter.
//   The arithmetic of coverage and attributes follows oracle/raster.hpp operation by operation (canonical edge
//   orientation, tile-relative constants, explicit fmaf) so pixels are bit-identical to the CPU spec.
//
// Roofline note: this kernel reads each emitted vertex once per overlapped tile from L2 and writes W*H*4 bytes once; it is
// VALU-bound (polynomial evaluation per sample), not HBM-bound — no MFMA shape exists in it.
#include "ga.hpp"
#include "raster_params.hpp"
#include "scene.hpp"

namespace crh {

constexpr int kTile = 16;
constexpr int kMaxTileShapes = 2048; // shapes overlapping one tile that the in-LDS sort handles


CRH_D float2 to_framebuffer(const float* m, float w, float h, float x, float y) { // oracle/raster.hpp to_framebuffer
    const float cx = (m[0] * x + m[4] * y) + m[12];
    const float cy = (m[1] * x + m[5] * y) + m[13];
    return make_float2((cx * 0.5f + 0.5f) * w, (0.5f - cy * 0.5f) * h);
}

// ---------------------------------------------------------------------------------------------- binning
__global__ __launch_bounds__(256) void k_shape_bounds(SceneDev s, RasterParams r) {
    const uint32_t shape = blockIdx.x * 256u + threadIdx.x;
    if (shape >= s.n_shapes) return;
    const float* m = r.transforms + 16u * shape;
    // affine instances only this round (clip.w == 1): SURVEY.md §8(f) ranks perspective "next"
    if (!(m[3] == 0.0f && m[7] == 0.0f && m[15] == 1.0f)) raise_error(s, s.elem_path[s.shape_elem_begin[shape]], CRH_ERR_UNSUPPORTED);
    const uint32_t base = s.shape_base[shape * NCH + CH_HULL];
    const uint32_t n = s.shape_base[(shape + 1) * NCH + CH_HULL] - base;
    const float W = (float)r.width, H = (float)r.height;
    float minx = __uint_as_float(0x7f800000u), miny = minx, maxx = -minx, maxy = -minx;
    // every vertex any primitive of the shape uses is a hull candidate (fill.rs:245-247,272,282,293; stroke.rs:89-92,125)
    for (uint32_t i = 0; i < n; ++i) {
        const Vertex0 c = s.hull_cand[base + i];
        const float2 f = to_framebuffer(m, W, H, c.x, c.y);
        minx = fminf(minx, f.x);
        maxx = fmaxf(maxx, f.x);
        miny = fminf(miny, f.y);
        maxy = fmaxf(maxy, f.y);
    }
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    minx = fmaxf(minx, 0.0f);
    miny = fmaxf(miny, 0.0f);
    maxx = fminf(maxx, W - 1.0f);
    maxy = fminf(maxy, H - 1.0f);
    if (n > 0 && minx <= maxx && miny <= maxy) {
        const uint32_t tx0 = (uint32_t)floorf(minx) / kTile, tx1 = (uint32_t)floorf(maxx) / kTile;
        const uint32_t ty0 = (uint32_t)floorf(miny) / kTile, ty1 = (uint32_t)floorf(maxy) / kTile;
        lo = tx0 | (ty0 << 16);
        hi = tx1 | (ty1 << 16);
        for (uint32_t ty = ty0; ty <= ty1; ++ty)
            for (uint32_t tx = tx0; tx <= tx1; ++tx) atomicAdd(&r.tile_count[ty * r.tiles_x + tx], 1u);
    }
    r.shape_rect[shape] = lo;
    r.shape_rect_hi[shape] = hi;
}

__global__ __launch_bounds__(1024) void k_tile_scan(RasterParams r) {
    __shared__ uint32_t partial[1024];
    const uint32_t chunk = (r.n_tiles + 1023u) / 1024u;
    const uint32_t begin = threadIdx.x * chunk, end = min(begin + chunk, r.n_tiles);
    uint32_t sum = 0;
    for (uint32_t t = begin; t < end; ++t) sum += r.tile_count[t];
    partial[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t v = threadIdx.x >= (uint32_t)d ? partial[threadIdx.x - d] : 0u;
        __syncthreads();
        partial[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = partial[threadIdx.x] - sum;
    for (uint32_t t = begin; t < end; ++t) {
        r.tile_offset[t] = run;
        r.tile_cursor[t] = 0;
        run += r.tile_count[t];
    }
    if (threadIdx.x == 1023) {
        r.tile_offset[r.n_tiles] = partial[1023];
        r.overflow[1] = partial[1023];
        r.overflow[0] = partial[1023] > r.pair_capacity ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void k_bin(SceneDev s, RasterParams r) {
    const uint32_t shape = blockIdx.x * 256u + threadIdx.x;
    if (shape >= s.n_shapes) return;
    if (r.overflow[0]) return;
    const uint32_t lo = r.shape_rect[shape], hi = r.shape_rect_hi[shape];
    if (lo == 0xFFFFFFFFu) return;
    for (uint32_t ty = lo >> 16; ty <= (hi >> 16); ++ty)
        for (uint32_t tx = lo & 0xFFFFu; tx <= (hi & 0xFFFFu); ++tx) {
            const uint32_t tile = ty * r.tiles_x + tx;
            const uint32_t slot = atomicAdd(&r.tile_cursor[tile], 1u);
            r.tile_list[r.tile_offset[tile] + slot] = shape;
        }
}

// ---------------------------------------------------------------------------------------------- per-tile raster
enum : uint32_t { KIND_SOLID = 0, KIND_IQ = 1, KIND_IC = 2, KIND_RQ = 3, KIND_RC = 4, KIND_LINE = 5, KIND_JOINT = 6, KIND_COVER = 7 };

// One primitive staged in LDS (struct of arrays over the 256 slots of a chunk; every lane reads the same slot => broadcast).
struct PrimList {
    float e_nay[3][256], e_bx[3][256], e_c[3][256]; // edge i: E = fma(rx, nay, fma(ry, bx, c)), sign already folded in
    float a_gx[4][256], a_gy[4][256], a_c[4][256];  // attribute planes
    uint32_t flags[256];  // bits 0-2 top-left per edge, bit 3 front, bits 4-6 kind, bits 8-23 pixel range nibbles x0 x1 y0 y1
    uint32_t flat_u[256]; // stroke: provoking vertex' u32 (group | 0x10000)
    float end_y[256];     // stroke line: provoking vertex' texcoord.y
    uint32_t desc[256];   // stroke: index of the 48-byte descriptor
    float color[4][256];  // cover: premultiplied source colour
};

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
CRH_D bool stroke_dashed(const crh_dynamic_stroke_descriptor& d, float tx, float ty) { // shaders.wgsl:205-231
    const uint32_t last = d.count_dashed_join >> 3;
    const float pattern_length = d.gap_end[last & 3u];
    uint32_t interval = 0;
    float position = crh_wgsl_mod(ty - d.phase, pattern_length);
    if (position < 0.0f) position = position + pattern_length;
    float gap_end;
    for (;;) {
        gap_end = d.gap_end[interval & 3u] - position;
        if (gap_end >= 0.0f || interval >= last) break;
        interval = interval + 1u;
    }
    const float gap_start = position - d.gap_start[interval & 3u];
    if (gap_start > 0.0f) {
        const uint32_t caps = d.caps >> (interval * 8u);
        const bool start_cap = cap_test(tx, gap_start, caps >> 4);
        const bool end_cap = cap_test(tx, gap_end, caps);
        return start_cap || end_cap;
    }
    return true;
}

struct Candidate { // a triangle before setup
    float2 p[3];
    float attr[3][4];
    uint32_t kind, flat_u, desc;
    float end_y;
};

template <int S>
__global__ __launch_bounds__(256) void k_raster(SceneDev s, RasterParams r) {
    __shared__ PrimList prims;
    __shared__ uint32_t order[kMaxTileShapes];
    __shared__ uint32_t wave_count[4];

    const uint32_t tile = blockIdx.x;
    const uint32_t tx = tile % r.tiles_x, ty = tile / r.tiles_x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t px = tid & 15u, py = tid >> 4;
    const uint32_t gx = tx * kTile + px, gy = ty * kTile + py;
    const bool in_frame = gx < r.width && gy < r.height;
    const float tx0 = (float)(tx * kTile), ty0 = (float)(ty * kTile);
    const float W = (float)r.width, H = (float)r.height;

    float sx[S], sy[S];
    if (S == 1) {
        sx[0] = (float)px + 0.5f;
        sy[0] = (float)py + 0.5f;
    } else {
        const float ox[4] = {0.375f, 0.875f, 0.125f, 0.625f}, oy[4] = {0.125f, 0.375f, 0.625f, 0.875f};
#pragma unroll
        for (int k = 0; k < S; ++k) {
            sx[k] = (float)px + ox[k & 3];
            sy[k] = (float)py + oy[k & 3];
        }
    }
    int winding[S];
    float col[S][4];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        winding[k] = 0;
        col[k][0] = col[k][1] = col[k][2] = col[k][3] = 0.0f;
    }
    if (r.load_existing && in_frame) {
        const uchar4 d = reinterpret_cast<const uchar4*>(r.rgba8)[(size_t)gy * r.width + gx];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            col[k][0] = (float)d.x * (1.0f / 255.0f);
            col[k][1] = (float)d.y * (1.0f / 255.0f);
            col[k][2] = (float)d.z * (1.0f / 255.0f);
            col[k][3] = (float)d.w * (1.0f / 255.0f);
        }
    }

    const bool overflowed = r.overflow[0] != 0u;
    const uint32_t list_begin = r.tile_offset[tile];
    uint32_t n_list = overflowed ? 0u : r.tile_offset[tile + 1] - list_begin;
    if (n_list > (uint32_t)kMaxTileShapes) {
        if (tid == 0) raise_error(s, 0, CRH_ERR_UNSUPPORTED);
        n_list = 0;
    }
    // ---- painter's order: sort the shape indices of this tile (bitonic, padded with 0xFFFFFFFF)
    if (n_list > 1) {
        uint32_t padded = 1;
        while (padded < n_list) padded <<= 1;
        for (uint32_t i = tid; i < padded; i += 256) order[i] = i < n_list ? r.tile_list[list_begin + i] : 0xFFFFFFFFu;
        __syncthreads();
        for (uint32_t k = 2; k <= padded; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < padded; i += 256) {
                    const uint32_t partner = i ^ j;
                    if (partner > i) {
                        const uint32_t a = order[i], b = order[partner];
                        if (((i & k) == 0) ? (a > b) : (a < b)) {
                            order[i] = b;
                            order[partner] = a;
                        }
                    }
                }
                __syncthreads();
            }
    } else if (n_list == 1) {
        if (tid == 0) order[0] = r.tile_list[list_begin];
        __syncthreads();
    }

    for (uint32_t li = 0; li < n_list; ++li) {
        const uint32_t shape = order[li];
        const uint32_t* b0 = s.shape_base + shape * NCH;
        const uint32_t* b1 = b0 + NCH;
        const float* m = r.transforms + 16u * shape;
        // candidate ranges, in the order Shape::render issues the draws (renderer.rs:275-336) and then the cover (:345-354)
        const uint32_t lv0 = b0[CH_LINE_V], lvn = b1[CH_LINE_V] - lv0;
        const uint32_t n_line = lvn >= 3u ? lvn - 2u : 0u;
        const uint32_t j0 = b0[CH_JOINT], n_joint = 3u * (b1[CH_JOINT] - j0);
        const uint32_t sv0 = b0[CH_SOLID_V], svn = b1[CH_SOLID_V] - sv0;
        const uint32_t n_solid = svn >= 3u ? svn - 2u : 0u;
        const uint32_t iq0 = b0[CH_IQ], n_iq = b1[CH_IQ] - iq0;
        const uint32_t ic0 = b0[CH_IC_V], n_ic = (b1[CH_IC_V] - ic0) / 3u;
        const uint32_t rq0 = b0[CH_RQ], n_rq = b1[CH_RQ] - rq0;
        const uint32_t rc0 = b0[CH_RC_V], n_rc = (b1[CH_RC_V] - rc0) / 3u;
        const uint32_t hull0 = b0[CH_HULL], hn = s.hull_count[shape];
        const uint32_t n_cover = hn >= 3u ? hn - 2u : 0u;
        const uint32_t c1 = n_line, c2 = c1 + n_joint, c3 = c2 + n_solid, c4 = c3 + n_iq, c5 = c4 + n_ic, c6 = c5 + n_rq, c7 = c6 + n_rc;
        const uint32_t n_candidates = c7 + n_cover;
        const uint32_t dyn0 = s.shape_dyn_begin[shape];
        const float ca = r.colors[4u * shape + 3];
        const float src[4] = {r.colors[4u * shape] * ca, r.colors[4u * shape + 1] * ca, r.colors[4u * shape + 2] * ca, ca};

        for (uint32_t chunk = 0; chunk < n_candidates; chunk += 256u) {
            // ---------------- setup: one candidate triangle per lane
            const uint32_t c = chunk + tid;
            bool keep = false;
            Candidate cd;
            cd.kind = KIND_SOLID;
            cd.flat_u = 0;
            cd.desc = 0;
            cd.end_y = 0.0f;
            int n_attr = 0;
            if (c < n_candidates) {
                bool valid = true;
                uint32_t i0, i1, i2;
                auto strip = [&](uint32_t k, uint32_t base) { // strip triangle k: even (k, k+1, k+2), odd (k, k+2, k+1)
                    i0 = base + k;
                    i1 = base + ((k & 1u) ? k + 2u : k + 1u);
                    i2 = base + ((k & 1u) ? k + 1u : k + 2u);
                };
                if (c < c1) { // stroke line strips (vertex2f1u / stencil_stroke_line)
                    const uint32_t k = c;
                    valid = s.line_pair_cut[(lv0 + k) >> 1] == 0;
                    strip(k, lv0);
                    if (valid) {
                        const Vertex2f1i a = s.line_v[i0], b = s.line_v[i1], d = s.line_v[i2];
                        cd.p[0] = to_framebuffer(m, W, H, a.x, a.y);
                        cd.p[1] = to_framebuffer(m, W, H, b.x, b.y);
                        cd.p[2] = to_framebuffer(m, W, H, d.x, d.y);
                        cd.attr[0][0] = a.u, cd.attr[0][1] = a.v;
                        cd.attr[1][0] = b.u, cd.attr[1][1] = b.v;
                        cd.attr[2][0] = d.u, cd.attr[2][1] = d.v;
                        const Vertex2f1i pv = s.line_v[lv0 + k]; // provoking vertex = first vertex of the primitive
                        cd.flat_u = pv.i;
                        cd.end_y = pv.v;
                        cd.desc = dyn0 + (pv.i & 65535u);
                    }
                    cd.kind = KIND_LINE;
                    n_attr = 2;
                } else if (c < c2) { // stroke joint strips: 5 vertices, 3 triangles per join
                    const uint32_t q = c - c1, jn = q / 3u, k = q - 3u * jn;
                    strip(k, 5u * (j0 + jn));
                    const Vertex3f1i a = s.joint_v[i0], b = s.joint_v[i1], d = s.joint_v[i2];
                    cd.p[0] = to_framebuffer(m, W, H, a.x, a.y);
                    cd.p[1] = to_framebuffer(m, W, H, b.x, b.y);
                    cd.p[2] = to_framebuffer(m, W, H, d.x, d.y);
                    cd.attr[0][0] = a.u, cd.attr[0][1] = a.v, cd.attr[0][2] = a.w;
                    cd.attr[1][0] = b.u, cd.attr[1][1] = b.v, cd.attr[1][2] = b.w;
                    cd.attr[2][0] = d.u, cd.attr[2][1] = d.v, cd.attr[2][2] = d.w;
                    cd.flat_u = s.joint_v[5u * (j0 + jn) + k].i;
                    cd.desc = dyn0 + (cd.flat_u & 65535u);
                    cd.kind = KIND_JOINT;
                    n_attr = 3;
                } else if (c < c3) { // solid strips (vertex0 / stencil_solid)
                    const uint32_t k = c - c2;
                    const uint8_t f0 = s.solid_flag[sv0 + k], f1 = s.solid_flag[sv0 + k + 1u];
                    valid = ((f0 | f1) & 2u) == 0;
                    const uint32_t parity = f0 & 1u;
                    i0 = sv0 + k;
                    i1 = sv0 + (parity ? k + 2u : k + 1u);
                    i2 = sv0 + (parity ? k + 1u : k + 2u);
                    if (valid) {
                        const Vertex0 a = s.solid_v[i0], b = s.solid_v[i1], d = s.solid_v[i2];
                        cd.p[0] = to_framebuffer(m, W, H, a.x, a.y);
                        cd.p[1] = to_framebuffer(m, W, H, b.x, b.y);
                        cd.p[2] = to_framebuffer(m, W, H, d.x, d.y);
                    }
                    cd.kind = KIND_SOLID;
                } else if (c < c4) {
                    const uint32_t at = 3u * (iq0 + (c - c3));
                    for (int v = 0; v < 3; ++v) {
                        const Vertex2f a = s.iq_v[at + v];
                        cd.p[v] = to_framebuffer(m, W, H, a.x, a.y);
                        cd.attr[v][0] = a.u, cd.attr[v][1] = a.v;
                    }
                    cd.kind = KIND_IQ;
                    n_attr = 2;
                } else if (c < c5) {
                    const uint32_t at = ic0 + 3u * (c - c4);
                    for (int v = 0; v < 3; ++v) {
                        const Vertex3f a = s.ic_v[at + v];
                        cd.p[v] = to_framebuffer(m, W, H, a.x, a.y);
                        cd.attr[v][0] = a.u, cd.attr[v][1] = a.v, cd.attr[v][2] = a.w;
                    }
                    cd.kind = KIND_IC;
                    n_attr = 3;
                } else if (c < c6) {
                    const uint32_t at = 3u * (rq0 + (c - c5));
                    for (int v = 0; v < 3; ++v) {
                        const Vertex3f a = s.rq_v[at + v];
                        cd.p[v] = to_framebuffer(m, W, H, a.x, a.y);
                        cd.attr[v][0] = a.u, cd.attr[v][1] = a.v, cd.attr[v][2] = a.w;
                    }
                    cd.kind = KIND_RQ;
                    n_attr = 3;
                } else if (c < c7) {
                    const uint32_t at = rc0 + 3u * (c - c6);
                    for (int v = 0; v < 3; ++v) {
                        const Vertex4f a = s.rc_v[at + v];
                        cd.p[v] = to_framebuffer(m, W, H, a.x, a.y);
                        cd.attr[v][0] = a.k, cd.attr[v][1] = a.l, cd.attr[v][2] = a.m, cd.attr[v][3] = a.n;
                    }
                    cd.kind = KIND_RC;
                    n_attr = 4;
                } else { // cover: hull strip (vertex_color / color_cover)
                    strip(c - c7, hull0);
                    const Vertex0 a = s.hull_v[i0], b = s.hull_v[i1], d = s.hull_v[i2];
                    cd.p[0] = to_framebuffer(m, W, H, a.x, a.y);
                    cd.p[1] = to_framebuffer(m, W, H, b.x, b.y);
                    cd.p[2] = to_framebuffer(m, W, H, d.x, d.y);
                    cd.kind = KIND_COVER;
                }
                // ---- oracle/raster.hpp setup_triangle
                const float d1x = cd.p[1].x - cd.p[0].x, d1y = cd.p[1].y - cd.p[0].y;
                const float d2x = cd.p[2].x - cd.p[0].x, d2y = cd.p[2].y - cd.p[0].y;
                const float det = d1x * d2y - d2x * d1y;
                if (valid && det != 0.0f && det == det && is_finite(det)) {
                    float minx = fminf(cd.p[0].x, fminf(cd.p[1].x, cd.p[2].x)), maxx = fmaxf(cd.p[0].x, fmaxf(cd.p[1].x, cd.p[2].x));
                    float miny = fminf(cd.p[0].y, fminf(cd.p[1].y, cd.p[2].y)), maxy = fmaxf(cd.p[0].y, fmaxf(cd.p[1].y, cd.p[2].y));
                    const bool nan_free = minx == minx && maxx == maxx && miny == miny && maxy == maxy;
                    minx = fmaxf(minx, 0.0f);
                    miny = fmaxf(miny, 0.0f);
                    maxx = fminf(maxx, W - 1.0f);
                    maxy = fminf(maxy, H - 1.0f);
                    if (nan_free && minx <= maxx && miny <= maxy) {
                        const int x0 = (int)floorf(minx), x1 = (int)floorf(maxx), y0 = (int)floorf(miny), y1 = (int)floorf(maxy);
                        const int tpx = (int)(tx * kTile), tpy = (int)(ty * kTile);
                        const int rx0 = max(x0, tpx) - tpx, rx1 = min(x1, tpx + kTile - 1) - tpx;
                        const int ry0 = max(y0, tpy) - tpy, ry1 = min(y1, tpy + kTile - 1) - tpy;
                        if (rx0 <= rx1 && ry0 <= ry1) {
                            keep = true;
                            const float inv_det = 1.0f / det;
                            const bool front = det < 0.0f;
                            // clockwise-in-y-down vertex order for the edge walk
                            const float2 nv[3] = {cd.p[0], det < 0.0f ? cd.p[2] : cd.p[1], det < 0.0f ? cd.p[1] : cd.p[2]};
                            uint32_t flags = (front ? 8u : 0u) | (cd.kind << 4) | ((uint32_t)rx0 << 8) | ((uint32_t)rx1 << 12) | ((uint32_t)ry0 << 16) |
                                             ((uint32_t)ry1 << 20);
                            // slot is assigned after the ballot below; stash the numbers in registers
                            float nay[3], bxs[3], cc[3];
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                const float2 a = nv[i], b = nv[(i + 1) % 3];
                                const float dx = b.x - a.x, dy = b.y - a.y;
                                if (dy < 0.0f || (dy == 0.0f && dx > 0.0f)) flags |= 1u << i;
                                const bool flip = !(a.x < b.x || (a.x == b.x && a.y < b.y));
                                const float2 lo = flip ? b : a, hi = flip ? a : b;
                                const float bx = hi.x - lo.x, na = -(hi.y - lo.y);
                                const float cst = bx * (ty0 - lo.y) + na * (tx0 - lo.x);
                                const float sg = flip ? -1.0f : 1.0f;
                                nay[i] = na * sg;
                                bxs[i] = bx * sg;
                                cc[i] = cst * sg;
                            }
                            float agx[4], agy[4], acs[4];
#pragma unroll
                            for (int a = 0; a < 4; ++a) {
                                if (a < n_attr) {
                                    const float da1 = cd.attr[1][a] - cd.attr[0][a], da2 = cd.attr[2][a] - cd.attr[0][a];
                                    agx[a] = (da1 * d2y - da2 * d1y) * inv_det;
                                    agy[a] = (da2 * d1x - da1 * d2x) * inv_det;
                                    acs[a] = (cd.attr[0][a] + (tx0 - cd.p[0].x) * agx[a]) + (ty0 - cd.p[0].y) * agy[a];
                                } else {
                                    agx[a] = agy[a] = acs[a] = 0.0f;
                                }
                            }
                            // ---- ordered compaction: each wave packs its survivors into its own 64-slot region of the LDS list
                            // (ballot over the lanes that reached this point + popcount of the lower lanes). Waves own consecutive
                            // candidate ranges, so walking region 0, 1, 2, 3 visits the survivors in candidate (= draw) order.
                            const unsigned long long ballot = __ballot(1);
                            const uint32_t rank = __popcll(ballot & ((1ull << lane) - 1ull));
                            if (rank == 0) wave_count[wave] = (uint32_t)__popcll(ballot);
                            const uint32_t slot = wave * 64u + rank;
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                prims.e_nay[i][slot] = nay[i];
                                prims.e_bx[i][slot] = bxs[i];
                                prims.e_c[i][slot] = cc[i];
                            }
#pragma unroll
                            for (int a = 0; a < 4; ++a) {
                                prims.a_gx[a][slot] = agx[a];
                                prims.a_gy[a][slot] = agy[a];
                                prims.a_c[a][slot] = acs[a];
                            }
                            prims.flags[slot] = flags;
                            prims.flat_u[slot] = cd.flat_u;
                            prims.end_y[slot] = cd.end_y;
                            prims.desc[slot] = cd.desc;
#pragma unroll
                            for (int k = 0; k < 4; ++k) prims.color[k][slot] = src[k];
                        }
                    }
                }
            }
            if (!__any(keep)) { // a wave without survivors still has to publish its count
                if (lane == 0) wave_count[wave] = 0;
            }
            __syncthreads();
            // ---------------- coverage: every lane walks the survivors wave by wave (wave-private regions keep candidate order)
            for (uint32_t w = 0; w < 4; ++w) {
                const uint32_t cnt = wave_count[w];
                for (uint32_t q = 0; q < cnt; ++q) {
                    const uint32_t slot = w * 64u + q;
                    const uint32_t flags = prims.flags[slot];
                    const uint32_t kind = (flags >> 4) & 7u;
                    const bool in_range = px >= ((flags >> 8) & 15u) && px <= ((flags >> 12) & 15u) && py >= ((flags >> 16) & 15u) && py <= ((flags >> 20) & 15u);
                    bool inside[S];
                    bool any_inside = false;
#pragma unroll
                    for (int k = 0; k < S; ++k) {
                        bool in = in_range;
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            const float e = fmaf(sx[k], prims.e_nay[i][slot], fmaf(sy[k], prims.e_bx[i][slot], prims.e_c[i][slot]));
                            in = in && (e > 0.0f || (e == 0.0f && ((flags >> i) & 1u)));
                        }
                        inside[k] = in;
                        any_inside = any_inside || in;
                    }
                    if (!__any(any_inside)) continue;
                    const bool front = (flags & 8u) != 0u;
                    if (kind == KIND_COVER) { // color_cover + stencil Less / Zero (renderer.rs:747-752, shaders.wgsl:304-309)
                        const float one_minus_a = 1.0f - prims.color[3][slot];
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            if (inside[k]) {
                                if ((winding[k] & (int)r.winding_mask) != 0) {
#pragma unroll
                                    for (int ch = 0; ch < 4; ++ch) col[k][ch] = prims.color[ch][slot] + col[k][ch] * one_minus_a;
                                }
                                winding[k] = 0;
                            }
                        }
                    } else if (kind == KIND_SOLID) { // stencil_solid: front +1, back -1 (renderer.rs:577-582)
#pragma unroll
                        for (int k = 0; k < S; ++k)
                            if (inside[k]) winding[k] += front ? 1 : -1;
                    } else {
                        float a0[S], a1[S], a2[S], a3[S];
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            a0[k] = fmaf(sy[k], prims.a_gy[0][slot], fmaf(sx[k], prims.a_gx[0][slot], prims.a_c[0][slot]));
                            a1[k] = fmaf(sy[k], prims.a_gy[1][slot], fmaf(sx[k], prims.a_gx[1][slot], prims.a_c[1][slot]));
                            a2[k] = fmaf(sy[k], prims.a_gy[2][slot], fmaf(sx[k], prims.a_gx[2][slot], prims.a_c[2][slot]));
                            a3[k] = fmaf(sy[k], prims.a_gy[3][slot], fmaf(sx[k], prims.a_gx[3][slot], prims.a_c[3][slot]));
                        }
                        if (kind <= KIND_RC) { // the four implicit-curve tests (shaders.wgsl:236-266)
#pragma unroll
                            for (int k = 0; k < S; ++k) {
                                bool fill;
                                if (kind == KIND_IQ)
                                    fill = a0[k] * a0[k] - a1[k] <= 0.0f;
                                else if (kind == KIND_IC)
                                    fill = a0[k] * a0[k] * a0[k] - a1[k] * a2[k] <= 0.0f;
                                else if (kind == KIND_RQ)
                                    fill = a0[k] * a0[k] - a1[k] * a2[k] <= 0.0f;
                                else
                                    fill = a0[k] * a0[k] * a0[k] - a1[k] * a2[k] * a3[k] <= 0.0f;
                                if (inside[k] && fill) winding[k] += front ? 1 : -1;
                            }
                        } else {
                            const crh_dynamic_stroke_descriptor d = s.descriptors[prims.desc[slot]];
                            const uint32_t flat_u = prims.flat_u[slot];
                            const float end_y = prims.end_y[slot];
#pragma unroll
                            for (int k = 0; k < S; ++k) {
                                if (!inside[k]) continue;
                                bool fill;
                                if (kind == KIND_LINE) { // stencil_stroke_line, shaders.wgsl:268-285
                                    if ((d.count_dashed_join & 4u) != 0u)
                                        fill = stroke_dashed(d, a0[k], a1[k]);
                                    else if ((flat_u & 65536u) != 0u)
                                        fill = cap_test(a0[k], a1[k] - end_y, d.caps >> 4);
                                    else if (a1[k] < 0.0f)
                                        fill = cap_test(a0[k], -a1[k], d.caps);
                                    else
                                        fill = true;
                                } else { // stencil_stroke_joint, shaders.wgsl:287-300
                                    const float radius = sqrtf(a0[k] * a0[k] + a1[k] * a1[k]);
                                    const uint32_t join = d.count_dashed_join & 3u;
                                    fill = join == 1u ? (flat_u & 65536u) != 0u : (join == 2u ? radius <= 0.5f : true);
                                    if (fill && (d.count_dashed_join & 4u) != 0u) {
                                        const float tau = crh_acosf(-1.0f) * 2.0f;
                                        fill = stroke_dashed(d, radius, a2[k] + crh_atan2f(a1[k], a0[k]) / tau);
                                    }
                                }
                                // stroke stencil: Equal(0) -> IncrementWrap, both faces (renderer.rs:571-576)
                                if (fill && (winding[k] & (int)r.winding_mask) == 0) winding[k] += 1;
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    // ---- MSAA resolve (box average) + RGBA8 unorm store
    if (in_frame) {
        const float inv = 1.0f / (float)S;
        uint32_t packed = 0;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < S; ++k) sum = sum + col[k][ch];
            float x = sum * inv;
            x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
            if (!(x == x)) x = 0.0f;
            packed |= (uint32_t)(int)(x * 255.0f + 0.5f) << (8 * ch);
        }
        reinterpret_cast<uint32_t*>(r.rgba8)[(size_t)gy * r.width + gx] = packed;
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

// ---------------------------------------------------------------------------------------------- launchers
void launch_bin(const SceneDev& s, const RasterParams& r, hipStream_t stream, void (*mark)(void*, const char*, uint64_t), void* ctx) {
    hipMemsetAsync(r.tile_count, 0, sizeof(uint32_t) * r.n_tiles, stream);
    hipLaunchKernelGGL(k_shape_bounds, dim3((s.n_shapes + 255) / 256), dim3(256), 0, stream, s, r);
    if (mark) mark(ctx, "raster_bounds", 0);
    hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, stream, r);
    if (mark) mark(ctx, "raster_tile_scan", 0);
}
void launch_raster(const SceneDev& s, const RasterParams& r, uint32_t samples, hipStream_t stream, void (*mark)(void*, const char*, uint64_t), void* ctx,
                   uint64_t raster_bytes) {
    hipLaunchKernelGGL(k_bin, dim3((s.n_shapes + 255) / 256), dim3(256), 0, stream, s, r);
    if (mark) mark(ctx, "raster_bin", 0);
    if (samples == 4)
        hipLaunchKernelGGL(k_raster<4>, dim3(r.n_tiles), dim3(256), 0, stream, s, r);
    else
        hipLaunchKernelGGL(k_raster<1>, dim3(r.n_tiles), dim3(256), 0, stream, s, r);
    if (mark) mark(ctx, "raster_tiles", raster_bytes);
}
void launch_composite(const uint8_t* const* layers_dev, uint32_t n_layers, uint64_t n_pixels, uint8_t* dst, hipStream_t stream) {
    hipLaunchKernelGGL(k_composite, dim3((uint32_t)((n_pixels + 255) / 256)), dim3(256), 0, stream, layers_dev, n_layers, n_pixels, dst);
}

} // namespace crh
