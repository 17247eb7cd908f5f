// csrc/raster.hip — tile-binned compute rasterizer that replaces the reference's stencil-then-cover passes
// (Shape::render renderer.rs:267-355, stencil states renderer.rs:565-582,736-754, fragment stage shaders.wgsl:155-309).
//
//   k_shape_setup   one wavefront per Shape: framebuffer bounding box -> 16x16-tile rectangle, per-tile counts, candidate count
//   k_scan_local/add two-kernel exclusive scans: tile offsets and per-Shape primitive ranges
//   k_shape_emit    one wavefront per Shape: append the Shape to its tiles (atomic slot, sorted later) and set up every triangle
//                   ONCE per frame (transform, edge functions, attribute planes -> 128-byte record + 8-byte pixel box)
//   k_raster<S>     one workgroup per tile, one lane per pixel, S samples per lane:
//        sort the tile's shape list (painter's order == shape index order); the candidate triangles of consecutive shapes fill
//        256 lanes at a time: box test against the tile, survivors load their record, make it tile relative and are compacted
//        IN ORDER into an LDS primitive list (ballot + popcount),
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
constexpr int kMaxTileShapes = 1024; // shapes overlapping one tile that the in-LDS sort handles


CRH_D float2 to_framebuffer(const float* m, float w, float h, float x, float y) { // oracle/raster.hpp to_framebuffer
    const float cx = (m[0] * x + m[4] * y) + m[12];
    const float cy = (m[1] * x + m[5] * y) + m[13];
    return make_float2((cx * 0.5f + 0.5f) * w, (0.5f - cy * 0.5f) * h);
}

// ---------------------------------------------------------------------------------------------- per-frame setup
enum : uint32_t { KIND_SOLID = 0, KIND_IQ = 1, KIND_IC = 2, KIND_RQ = 3, KIND_RC = 4, KIND_LINE = 5, KIND_JOINT = 6, KIND_COVER = 7 };

// One set-up triangle, tile independent (128 bytes). Edge i evaluates, relative to a tile origin (tx0, ty0):
//   c = bx*(ty0 - lo_y) + nay*(tx0 - lo_x);  E = fma(rx, nay, fma(ry, bx, c))   — the canonical-orientation sign is folded in.
struct PrimRec {
    float lo_x[3], lo_y[3], bx[3], nay[3];
    float a0[4], gx[4], gy[4]; // attribute planes through vertex 0
    float v0x, v0y;
    uint32_t flags;  // bits 0-2 top-left per edge, bit 3 front (ccw on screen), bits 4-6 kind
    uint32_t flat_u; // stroke: provoking vertex' u32 (group | 0x10000)
    float end_y;     // stroke line: provoking vertex' texcoord.y
    uint32_t shape;
    uint32_t desc;   // stroke: index of the 48-byte descriptor
    uint32_t pad;
};
static_assert(sizeof(PrimRec) == 128, "PrimRec");

CRH_D uint32_t shape_candidates(const SceneDev& s, uint32_t shape, uint32_t c[8]) {
    const uint32_t* b0 = s.shape_base + shape * NCH;
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

// k_shape_setup — one wavefront per Shape: framebuffer bounding box of everything the Shape draws (every vertex of every primitive is
// a hull candidate: fill.rs:245-247,272,282,293; stroke.rs:89-92,125) -> tile rectangle, per-tile counts, candidate count.
__global__ __launch_bounds__(64) void k_shape_setup(SceneDev s, RasterParams r) {
    const uint32_t shape = blockIdx.x, lane = threadIdx.x;
    const float* m = r.transforms + 16u * shape;
    if (lane == 0 && !(m[3] == 0.0f && m[7] == 0.0f && m[15] == 1.0f)) // affine instances only this round (clip.w == 1)
        raise_error(s, s.elem_path[min(s.shape_elem_begin[shape], s.n_elems - 1u)], CRH_ERR_UNSUPPORTED);
    const uint32_t base = s.shape_base[shape * NCH + CH_HULL];
    const uint32_t n = s.shape_base[(shape + 1) * NCH + CH_HULL] - base;
    const float W = (float)r.width, H = (float)r.height;
    const float inf = __uint_as_float(0x7f800000u);
    float minx = inf, miny = inf, maxx = -inf, maxy = -inf;
    for (uint32_t i = lane; i < n; i += 64u) {
        const Vertex0 c = s.hull_cand[base + i];
        const float2 f = to_framebuffer(m, W, H, c.x, c.y);
        minx = fminf(minx, f.x);
        maxx = fmaxf(maxx, f.x);
        miny = fminf(miny, f.y);
        maxy = fmaxf(maxy, f.y);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        minx = fminf(minx, __shfl_xor(minx, d, 64));
        maxx = fmaxf(maxx, __shfl_xor(maxx, d, 64));
        miny = fminf(miny, __shfl_xor(miny, d, 64));
        maxy = fmaxf(maxy, __shfl_xor(maxy, d, 64));
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
        const uint32_t tw = tx1 - tx0 + 1u, count = tw * (ty1 - ty0 + 1u);
        for (uint32_t i = lane; i < count; i += 64u) atomicAdd(&r.tile_count[(ty0 + i / tw) * r.tiles_x + tx0 + i % tw], 1u);
    }
    if (lane == 0) {
        uint32_t c[8];
        r.shape_rect[shape] = lo;
        r.shape_rect_hi[shape] = hi;
        r.shape_ncand[shape] = lo == 0xFFFFFFFFu ? 0u : shape_candidates(s, shape, c); // off-screen Shapes draw nothing
    }
}

// Two-kernel exclusive scan over u32 arrays (1024 items per block), used for the tile offsets and the per-Shape primitive ranges.
struct ScanJob {
    const uint32_t* in;
    uint32_t* out; // [n + 1]
    uint32_t* block_sum;
    uint32_t n, blocks;
};
__global__ __launch_bounds__(256) void k_scan_local(ScanJob a, ScanJob b) {
    __shared__ uint32_t wave_sum[4];
    const bool second = blockIdx.x >= a.blocks;
    const ScanJob j = second ? b : a;
    const uint32_t block = second ? blockIdx.x - a.blocks : blockIdx.x;
    const uint32_t i0 = block * 1024u + threadIdx.x * 4u;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i0 + k < j.n ? j.in[i0 + k] : 0u;
    const uint32_t mine = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = mine;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
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
    if (threadIdx.x == 255) j.block_sum[block] = run;
}
__global__ __launch_bounds__(256) void k_scan_add(ScanJob a, ScanJob b, RasterParams r) {
    __shared__ uint32_t partial[256];
    const bool second = blockIdx.x >= a.blocks;
    const ScanJob j = second ? b : a;
    const uint32_t block = second ? blockIdx.x - a.blocks : blockIdx.x;
    uint32_t sum = 0;
    for (uint32_t k = threadIdx.x; k < block; k += 256u) sum += j.block_sum[k];
    partial[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < (uint32_t)d) partial[threadIdx.x] += partial[threadIdx.x + d];
        __syncthreads();
    }
    const uint32_t base = partial[0];
    const uint32_t i0 = block * 1024u + threadIdx.x * 4u;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < j.n) j.out[i0 + k] += base;
    if (block + 1u == j.blocks && threadIdx.x == 0) {
        const uint32_t total = base + j.block_sum[block];
        j.out[j.n] = total;
        if (!second) { // job a = tiles: publish the pair count and the overflow flag
            r.overflow[1] = total;
            r.overflow[0] = total > r.pair_capacity ? 1u : 0u;
        }
    }
}

// k_shape_emit — one wavefront per Shape: (1) append the Shape to every tile of its rectangle, (2) set up its triangles once
// for the whole frame: transform (vertex stage, shaders.wgsl:66-151), edge functions, attribute planes, pixel box.
__global__ __launch_bounds__(64) void k_shape_emit(SceneDev s, RasterParams r) {
    const uint32_t shape = blockIdx.x, lane = threadIdx.x;
    const uint32_t lo = r.shape_rect[shape], hi = r.shape_rect_hi[shape];
    if (lo == 0xFFFFFFFFu) return;
    if (!r.overflow[0]) {
        const uint32_t tx0 = lo & 0xFFFFu, ty0 = lo >> 16, tw = (hi & 0xFFFFu) - tx0 + 1u, count = tw * ((hi >> 16) - ty0 + 1u);
        for (uint32_t i = lane; i < count; i += 64u) {
            const uint32_t tile = (ty0 + i / tw) * r.tiles_x + tx0 + i % tw;
            const uint32_t slot = atomicAdd(&r.tile_cursor[tile], 1u);
            r.tile_list[r.tile_offset[tile] + slot] = shape;
        }
    }
    uint32_t cb[8];
    const uint32_t n_candidates = shape_candidates(s, shape, cb);
    const uint32_t prim0 = r.shape_prim_begin[shape];
    const uint32_t* b0 = s.shape_base + shape * NCH;
    const uint32_t lv0 = b0[CH_LINE_V], j0 = b0[CH_JOINT], sv0 = b0[CH_SOLID_V], iq0 = b0[CH_IQ], ic0 = b0[CH_IC_V], rq0 = b0[CH_RQ], rc0 = b0[CH_RC_V],
                   hull0 = b0[CH_HULL];
    const uint32_t dyn0 = s.shape_dyn_begin[shape];
    const float* m = r.transforms + 16u * shape;
    const float W = (float)r.width, H = (float)r.height;
    for (uint32_t c = lane; c < n_candidates; c += 64u) {
        float2 p[3];
        float attr[3][4];
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
        if (c < cb[0]) { // stroke line strips (vertex2f1u / stencil_stroke_line)
            const uint32_t k = c;
            valid = s.line_pair_cut[(lv0 + k) >> 1] == 0;
            strip(k, lv0);
            const Vertex2f1i a = s.line_v[i0], b = s.line_v[i1], d = s.line_v[i2];
            p[0] = to_framebuffer(m, W, H, a.x, a.y);
            p[1] = to_framebuffer(m, W, H, b.x, b.y);
            p[2] = to_framebuffer(m, W, H, d.x, d.y);
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
            p[0] = to_framebuffer(m, W, H, a.x, a.y);
            p[1] = to_framebuffer(m, W, H, b.x, b.y);
            p[2] = to_framebuffer(m, W, H, d.x, d.y);
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
            p[0] = to_framebuffer(m, W, H, a.x, a.y);
            p[1] = to_framebuffer(m, W, H, b.x, b.y);
            p[2] = to_framebuffer(m, W, H, d.x, d.y);
        } else if (c < cb[3]) {
            const uint32_t at = 3u * (iq0 + (c - cb[2]));
            for (int v = 0; v < 3; ++v) {
                const Vertex2f a = s.iq_v[at + v];
                p[v] = to_framebuffer(m, W, H, a.x, a.y);
                attr[v][0] = a.u, attr[v][1] = a.v;
            }
            kind = KIND_IQ;
            n_attr = 2;
        } else if (c < cb[4]) {
            const uint32_t at = ic0 + 3u * (c - cb[3]);
            for (int v = 0; v < 3; ++v) {
                const Vertex3f a = s.ic_v[at + v];
                p[v] = to_framebuffer(m, W, H, a.x, a.y);
                attr[v][0] = a.u, attr[v][1] = a.v, attr[v][2] = a.w;
            }
            kind = KIND_IC;
            n_attr = 3;
        } else if (c < cb[5]) {
            const uint32_t at = 3u * (rq0 + (c - cb[4]));
            for (int v = 0; v < 3; ++v) {
                const Vertex3f a = s.rq_v[at + v];
                p[v] = to_framebuffer(m, W, H, a.x, a.y);
                attr[v][0] = a.u, attr[v][1] = a.v, attr[v][2] = a.w;
            }
            kind = KIND_RQ;
            n_attr = 3;
        } else if (c < cb[6]) {
            const uint32_t at = rc0 + 3u * (c - cb[5]);
            for (int v = 0; v < 3; ++v) {
                const Vertex4f a = s.rc_v[at + v];
                p[v] = to_framebuffer(m, W, H, a.x, a.y);
                attr[v][0] = a.k, attr[v][1] = a.l, attr[v][2] = a.m, attr[v][3] = a.n;
            }
            kind = KIND_RC;
            n_attr = 4;
        } else { // cover: hull strip (vertex_color / color_cover)
            strip(c - cb[6], hull0);
            const Vertex0 a = s.hull_v[i0], b = s.hull_v[i1], d = s.hull_v[i2];
            p[0] = to_framebuffer(m, W, H, a.x, a.y);
            p[1] = to_framebuffer(m, W, H, b.x, b.y);
            p[2] = to_framebuffer(m, W, H, d.x, d.y);
            kind = KIND_COVER;
        }
        // ---- oracle/raster.hpp setup_triangle + setup_attribute
        ushort4 box = make_ushort4(0xFFFFu, 0, 0, 0);
        const float d1x = p[1].x - p[0].x, d1y = p[1].y - p[0].y;
        const float d2x = p[2].x - p[0].x, d2y = p[2].y - p[0].y;
        const float det = d1x * d2y - d2x * d1y;
        if (valid && det != 0.0f && det == det && is_finite(det)) {
            float minx = fminf(p[0].x, fminf(p[1].x, p[2].x)), maxx = fmaxf(p[0].x, fmaxf(p[1].x, p[2].x));
            float miny = fminf(p[0].y, fminf(p[1].y, p[2].y)), maxy = fmaxf(p[0].y, fmaxf(p[1].y, p[2].y));
            const bool nan_free = minx == minx && maxx == maxx && miny == miny && maxy == maxy;
            minx = fmaxf(minx, 0.0f);
            miny = fmaxf(miny, 0.0f);
            maxx = fminf(maxx, W - 1.0f);
            maxy = fminf(maxy, H - 1.0f);
            if (nan_free && minx <= maxx && miny <= maxy) {
                box = make_ushort4((unsigned short)floorf(minx), (unsigned short)floorf(maxx), (unsigned short)floorf(miny), (unsigned short)floorf(maxy));
                PrimRec rec;
                const float inv_det = 1.0f / det;
                const bool front = det < 0.0f; // y-down cross < 0 == counter-clockwise on screen (FrontFace::Ccw, renderer.rs:477)
                const float2 nv[3] = {p[0], det < 0.0f ? p[2] : p[1], det < 0.0f ? p[1] : p[2]}; // clockwise-in-y-down edge walk
                uint32_t flags = (front ? 8u : 0u) | (kind << 4);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float2 a = nv[i], b = nv[(i + 1) % 3];
                    const float dx = b.x - a.x, dy = b.y - a.y;
                    if (dy < 0.0f || (dy == 0.0f && dx > 0.0f)) flags |= 1u << i; // top-left rule
                    const bool flip = !(a.x < b.x || (a.x == b.x && a.y < b.y));   // canonical (lexicographic) endpoint order
                    const float2 el = flip ? b : a, eh = flip ? a : b;
                    const float sg = flip ? -1.0f : 1.0f;
                    rec.lo_x[i] = el.x;
                    rec.lo_y[i] = el.y;
                    rec.bx[i] = (eh.x - el.x) * sg;
                    rec.nay[i] = -(eh.y - el.y) * sg;
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (a < n_attr) {
                        const float da1 = attr[1][a] - attr[0][a], da2 = attr[2][a] - attr[0][a];
                        rec.a0[a] = attr[0][a];
                        rec.gx[a] = (da1 * d2y - da2 * d1y) * inv_det;
                        rec.gy[a] = (da2 * d1x - da1 * d2x) * inv_det;
                    } else {
                        rec.a0[a] = rec.gx[a] = rec.gy[a] = 0.0f;
                    }
                }
                rec.v0x = p[0].x;
                rec.v0y = p[0].y;
                rec.flags = flags;
                rec.flat_u = flat_u;
                rec.end_y = end_y;
                rec.shape = shape;
                rec.desc = desc;
                rec.pad = 0;
                r.prim_rec[prim0 + c] = rec;
            }
        }
        r.prim_box[prim0 + c] = box;
    }
}

// ---------------------------------------------------------------------------------------------- per-tile raster
// One primitive staged in LDS. Every lane reads the same slot (broadcast), so the fields a coverage step needs together are
// packed into float4s (one ds_read_b128 each) instead of one array per scalar.
struct PrimList {
    float4 e0[256]; // nay0 bx0 c0 nay1      edge i: E = fma(rx, nay_i, fma(ry, bx_i, c_i))
    float4 e1[256]; // bx1 c1 nay2 bx2
    float4 e2[256]; // c2, flags (bits), flat_u (bits), end_y
    float4 a[4][256]; // attribute plane k: gx gy c, (k == 0: descriptor index bits in .w)
    float4 color[256]; // cover: premultiplied source colour
};
// flags: bits 0-2 top-left per edge, bit 3 front, bits 4-6 kind, bits 8-23 pixel range nibbles x0 x1 y0 y1 inside the tile,
//        bits 24-27 bands (pixel rows 4b..4b+3 = the rows wave b owns) that may contain covered samples,
//        bits 28-31 bands in which EVERY sample of the pixel range is inside the triangle (edge evaluation can be skipped)

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

template <int S>
__global__ __launch_bounds__(256) void k_raster(SceneDev s, RasterParams r) {
    __shared__ PrimList prims;
    __shared__ uint32_t order[kMaxTileShapes];       // the tile's shapes in painter's order
    __shared__ uint32_t cand_end[kMaxTileShapes];    // inclusive prefix of their candidate counts
    __shared__ uint32_t wave_count[4];

    const uint32_t tile = blockIdx.x;
    const uint32_t tx = tile % r.tiles_x, ty = tile / r.tiles_x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t px = tid & 15u, py = tid >> 4;
    const uint32_t gx = tx * kTile + px, gy = ty * kTile + py;
    const bool in_frame = gx < r.width && gy < r.height;
    const float tx0 = (float)(tx * kTile), ty0 = (float)(ty * kTile);
    const int tpx = (int)(tx * kTile), tpy = (int)(ty * kTile);

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
    if (n_list > 0) {
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
        // inclusive prefix of the candidate counts (Hillis-Steele in LDS; lists are short)
        for (uint32_t i = tid; i < n_list; i += 256) cand_end[i] = r.shape_prim_begin[order[i] + 1u] - r.shape_prim_begin[order[i]];
        __syncthreads();
        for (uint32_t d = 1; d < n_list; d <<= 1) {
            uint32_t add[kMaxTileShapes / 256];
#pragma unroll
            for (uint32_t q = 0; q < kMaxTileShapes / 256; ++q) {
                const uint32_t i = tid + q * 256u;
                add[q] = (i < n_list && i >= d) ? cand_end[i - d] : 0u;
            }
            __syncthreads();
#pragma unroll
            for (uint32_t q = 0; q < kMaxTileShapes / 256; ++q) {
                const uint32_t i = tid + q * 256u;
                if (i < n_list) cand_end[i] += add[q];
            }
            __syncthreads();
        }
    }
    uint32_t n_candidates = n_list ? cand_end[n_list - 1u] : 0u;
    if (r.debug & 4u) {
        if (tid == 0) {
            atomicAdd(&r.overflow[2], n_list);
            atomicAdd(&r.overflow[3], n_candidates);
            atomicMax(&r.overflow[5], n_list);
            atomicMax(&r.overflow[6], n_candidates);
        }
    }
    if (r.debug & 1u) n_candidates = 0;

    for (uint32_t chunk = 0; chunk < n_candidates; chunk += 256u) {
        // ---------------- gather: candidates of several consecutive shapes fill the 256 lanes; survivors of the box test are
        // packed per wave into the LDS list in candidate (= draw) order
        const uint32_t c = chunk + tid;
        bool keep = false;
        uint32_t prim = 0, shape = 0;
        int rx0 = 0, rx1 = 0, ry0 = 0, ry1 = 0;
        if (c < n_candidates) {
            uint32_t lo_i = 0, hi_i = n_list - 1u; // first i with cand_end[i] > c
            while (lo_i < hi_i) {
                const uint32_t mid = (lo_i + hi_i) >> 1;
                if (cand_end[mid] > c)
                    hi_i = mid;
                else
                    lo_i = mid + 1u;
            }
            shape = order[lo_i];
            prim = r.shape_prim_begin[shape] + (c - (lo_i ? cand_end[lo_i - 1u] : 0u));
            const ushort4 box = r.prim_box[prim];
            if (box.x != 0xFFFFu) {
                rx0 = max((int)box.x, tpx) - tpx;
                rx1 = min((int)box.y, tpx + kTile - 1) - tpx;
                ry0 = max((int)box.z, tpy) - tpy;
                ry1 = min((int)box.w, tpy + kTile - 1) - tpy;
                keep = rx0 <= rx1 && ry0 <= ry1;
            }
        }
        // exact cull + band classification: an edge function is monotone in rx and in ry (fmaf rounds monotonically), so its
        // extremes over a box of sample positions sit at the corners; a primitive is dropped when some edge rejects its best
        // corner, and a band is "full" when every edge accepts its worst corner.
        PrimRec rec;
        float ec[3];
        uint32_t bands = 0, full = 0;
        if (keep) {
            rec = r.prim_rec[prim];
            const float s_lo = S == 1 ? 0.5f : 0.125f, s_hi = S == 1 ? 0.5f : 0.875f;
            const float x_lo = (float)rx0 + s_lo, x_hi = (float)rx1 + s_hi;
#pragma unroll
            for (int i = 0; i < 3; ++i) ec[i] = rec.bx[i] * (ty0 - rec.lo_y[i]) + rec.nay[i] * (tx0 - rec.lo_x[i]);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int by0 = max(ry0, 4 * b), by1 = min(ry1, 4 * b + 3);
                if (by0 > by1) continue;
                const float y_lo = (float)by0 + s_lo, y_hi = (float)by1 + s_hi;
                bool some = true, all = true;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const bool tl = (rec.flags >> i) & 1u;
                    const float xa = rec.nay[i] > 0.0f ? x_hi : x_lo, xb = rec.nay[i] > 0.0f ? x_lo : x_hi;
                    const float ya = rec.bx[i] > 0.0f ? y_hi : y_lo, yb = rec.bx[i] > 0.0f ? y_lo : y_hi;
                    const float e_max = fmaf(xa, rec.nay[i], fmaf(ya, rec.bx[i], ec[i]));
                    const float e_min = fmaf(xb, rec.nay[i], fmaf(yb, rec.bx[i], ec[i]));
                    some = some && (e_max > 0.0f || (e_max == 0.0f && tl));
                    all = all && (e_min > 0.0f || (e_min == 0.0f && tl));
                }
                if (some) bands |= 1u << b;
                if (all) full |= 1u << b;
            }
            keep = bands != 0u;
        }
        const unsigned long long ballot = __ballot(keep);
        if (lane == 0) wave_count[wave] = (uint32_t)__popcll(ballot);
        if (keep) {
            const uint32_t slot = wave * 64u + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
            const uint32_t flags = (rec.flags & 0xFFu) | ((uint32_t)rx0 << 8) | ((uint32_t)rx1 << 12) | ((uint32_t)ry0 << 16) | ((uint32_t)ry1 << 20) |
                                   (bands << 24) | (full << 28);
            prims.e0[slot] = make_float4(rec.nay[0], rec.bx[0], ec[0], rec.nay[1]);
            prims.e1[slot] = make_float4(rec.bx[1], ec[1], rec.nay[2], rec.bx[2]);
            prims.e2[slot] = make_float4(ec[2], __uint_as_float(flags), __uint_as_float(rec.flat_u), rec.end_y);
            const uint32_t kind = (rec.flags >> 4) & 7u;
            if (kind != KIND_SOLID && kind != KIND_COVER) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    prims.a[a][slot] = make_float4(rec.gx[a], rec.gy[a], (rec.a0[a] + (tx0 - rec.v0x) * rec.gx[a]) + (ty0 - rec.v0y) * rec.gy[a],
                                                   a == 0 ? __uint_as_float(rec.desc) : 0.0f);
            }
            if (kind == KIND_COVER) {
                const float ca = r.colors[4u * shape + 3];
                prims.color[slot] = make_float4(r.colors[4u * shape] * ca, r.colors[4u * shape + 1] * ca, r.colors[4u * shape + 2] * ca, ca);
            }
        }
        __syncthreads();
        if ((r.debug & 4u) && tid == 0) atomicAdd(&r.overflow[4], wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3]);
        // ---------------- coverage: every lane walks the survivors, wave region by wave region (= candidate order)
        for (uint32_t w = 0; w < 4; ++w) {
            const uint32_t cnt = (r.debug & 2u) ? 0u : wave_count[w];
            for (uint32_t q = 0; q < cnt; ++q) {
                const uint32_t slot = w * 64u + q;
                const float4 e2 = prims.e2[slot];
                const uint32_t flags = __float_as_uint(e2.y);
                if (!((flags >> (24u + wave)) & 1u)) continue; // nothing of this primitive in the 4 rows this wave owns (wave-uniform)
                const float4 e0 = prims.e0[slot], e1 = prims.e1[slot];
                const uint32_t kind = (flags >> 4) & 7u;
                // Straight-line predicates (bitwise, no short-circuit): the compiler otherwise emits an exec-mask branch ladder per '&&'.
                const uint32_t bx0 = (flags >> 8) & 15u, bx1 = (flags >> 12) & 15u, by0 = (flags >> 16) & 15u, by1 = (flags >> 20) & 15u;
                const int in_range = (int)((px - bx0) <= (bx1 - bx0)) & (int)((py - by0) <= (by1 - by0));
                const int full = (int)((flags >> (28u + wave)) & 1u); // every sample of the pixel range in this band is covered
                const int tl0 = (int)(flags & 1u), tl1 = (int)((flags >> 1) & 1u), tl2 = (int)((flags >> 2) & 1u);
                int inside[S];
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    const float ea = fmaf(sx[k], e0.x, fmaf(sy[k], e0.y, e0.z));
                    const float eb = fmaf(sx[k], e0.w, fmaf(sy[k], e1.x, e1.y));
                    const float ecv = fmaf(sx[k], e1.z, fmaf(sy[k], e1.w, e2.x));
                    const int ia = (int)(ea > 0.0f) | ((int)(ea == 0.0f) & tl0);
                    const int ib = (int)(eb > 0.0f) | ((int)(eb == 0.0f) & tl1);
                    const int ic = (int)(ecv > 0.0f) | ((int)(ecv == 0.0f) & tl2);
                    inside[k] = in_range & (full | (ia & ib & ic));
                }
                const int delta = (flags & 8u) ? 1 : -1; // front (ccw on screen) increments, back decrements (renderer.rs:577-582)
                switch (kind) {
                    case KIND_SOLID: { // stencil_solid
#pragma unroll
                        for (int k = 0; k < S; ++k) winding[k] += inside[k] ? delta : 0;
                        break;
                    }
                    case KIND_COVER: { // color_cover + stencil Less / Zero (renderer.rs:747-752, shaders.wgsl:304-309)
                        const float4 src = prims.color[slot];
                        const float one_minus_a = 1.0f - src.w;
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            const bool blend = inside[k] && (winding[k] & (int)r.winding_mask) != 0;
                            const float n0 = src.x + col[k][0] * one_minus_a, n1 = src.y + col[k][1] * one_minus_a;
                            const float n2 = src.z + col[k][2] * one_minus_a, n3 = src.w + col[k][3] * one_minus_a;
                            col[k][0] = blend ? n0 : col[k][0];
                            col[k][1] = blend ? n1 : col[k][1];
                            col[k][2] = blend ? n2 : col[k][2];
                            col[k][3] = blend ? n3 : col[k][3];
                            winding[k] = inside[k] ? 0 : winding[k];
                        }
                        break;
                    }
                    case KIND_IQ: { // u^2 - v <= 0 (shaders.wgsl:236-242)
                        const float4 p0 = prims.a[0][slot], p1 = prims.a[1][slot];
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            const float a0 = fmaf(sy[k], p0.y, fmaf(sx[k], p0.x, p0.z)), a1 = fmaf(sy[k], p1.y, fmaf(sx[k], p1.x, p1.z));
                            winding[k] += (inside[k] & (int)(a0 * a0 - a1 <= 0.0f)) ? delta : 0;
                        }
                        break;
                    }
                    case KIND_IC:   // k^3 - l m <= 0 (shaders.wgsl:244-250)
                    case KIND_RQ: { // u^2 - v w <= 0 (shaders.wgsl:252-258)
                        const float4 p0 = prims.a[0][slot], p1 = prims.a[1][slot], p2 = prims.a[2][slot];
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            const float a0 = fmaf(sy[k], p0.y, fmaf(sx[k], p0.x, p0.z)), a1 = fmaf(sy[k], p1.y, fmaf(sx[k], p1.x, p1.z));
                            const float a2 = fmaf(sy[k], p2.y, fmaf(sx[k], p2.x, p2.z));
                            const float lhs = kind == KIND_IC ? a0 * a0 * a0 : a0 * a0;
                            winding[k] += (inside[k] & (int)(lhs - a1 * a2 <= 0.0f)) ? delta : 0;
                        }
                        break;
                    }
                    case KIND_RC: { // k^3 - l m n <= 0 (shaders.wgsl:260-266)
                        const float4 p0 = prims.a[0][slot], p1 = prims.a[1][slot], p2 = prims.a[2][slot], p3 = prims.a[3][slot];
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            const float a0 = fmaf(sy[k], p0.y, fmaf(sx[k], p0.x, p0.z)), a1 = fmaf(sy[k], p1.y, fmaf(sx[k], p1.x, p1.z));
                            const float a2 = fmaf(sy[k], p2.y, fmaf(sx[k], p2.x, p2.z)), a3 = fmaf(sy[k], p3.y, fmaf(sx[k], p3.x, p3.z));
                            winding[k] += (inside[k] & (int)(a0 * a0 * a0 - a1 * a2 * a3 <= 0.0f)) ? delta : 0;
                        }
                        break;
                    }
                    default: { // KIND_LINE / KIND_JOINT: the stroke fragment stages
                        int any_inside = 0;
#pragma unroll
                        for (int k = 0; k < S; ++k) any_inside |= inside[k];
                        if (!__any(any_inside)) break;
                        const float4 p0 = prims.a[0][slot], p1 = prims.a[1][slot], p2 = prims.a[2][slot];
                        const crh_dynamic_stroke_descriptor d = s.descriptors[__float_as_uint(p0.w)];
                        const uint32_t flat_u = __float_as_uint(e2.z);
                        const float end_y = e2.w;
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            if (!inside[k]) continue;
                            const float a0 = fmaf(sy[k], p0.y, fmaf(sx[k], p0.x, p0.z)), a1 = fmaf(sy[k], p1.y, fmaf(sx[k], p1.x, p1.z));
                            const float a2 = fmaf(sy[k], p2.y, fmaf(sx[k], p2.x, p2.z));
                            bool fill;
                            if (kind == KIND_LINE) { // stencil_stroke_line, shaders.wgsl:268-285
                                if ((d.count_dashed_join & 4u) != 0u)
                                    fill = stroke_dashed(d, a0, a1);
                                else if ((flat_u & 65536u) != 0u)
                                    fill = cap_test(a0, a1 - end_y, d.caps >> 4);
                                else if (a1 < 0.0f)
                                    fill = cap_test(a0, -a1, d.caps);
                                else
                                    fill = true;
                            } else { // stencil_stroke_joint, shaders.wgsl:287-300
                                const float radius = sqrtf(a0 * a0 + a1 * a1);
                                const uint32_t join = d.count_dashed_join & 3u;
                                fill = join == 1u ? (flat_u & 65536u) != 0u : (join == 2u ? radius <= 0.5f : true);
                                if (fill && (d.count_dashed_join & 4u) != 0u) {
                                    const float tau = crh_acosf(-1.0f) * 2.0f;
                                    fill = stroke_dashed(d, radius, a2 + crh_atan2f(a1, a0) / tau);
                                }
                            }
                            // stroke stencil: Equal(0) -> IncrementWrap, both faces (renderer.rs:571-576)
                            if (fill && (winding[k] & (int)r.winding_mask) == 0) winding[k] += 1;
                        }
                        break;
                    }
                }
            }
        }
        __syncthreads();
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
static ScanJob scan_job(const uint32_t* in, uint32_t* out, uint32_t* block_sum, uint32_t n) { return ScanJob{in, out, block_sum, n, (n + 1023u) / 1024u}; }

void launch_bin(const SceneDev& s, const RasterParams& r, hipStream_t stream, void (*mark)(void*, const char*, uint64_t), void* ctx) {
    (void)hipMemsetAsync(r.tile_count, 0, sizeof(uint32_t) * (r.n_tiles + r.n_tiles), stream); // tile_count and tile_cursor are adjacent
    hipLaunchKernelGGL(k_shape_setup, dim3(s.n_shapes), dim3(64), 0, stream, s, r);
    if (mark) mark(ctx, "raster_shape_setup", 0);
    const ScanJob tiles = scan_job(r.tile_count, r.tile_offset, r.scan_scratch, r.n_tiles);
    const ScanJob prims = scan_job(r.shape_ncand, r.shape_prim_begin, r.scan_scratch + tiles.blocks, s.n_shapes);
    hipLaunchKernelGGL(k_scan_local, dim3(tiles.blocks + prims.blocks), dim3(256), 0, stream, tiles, prims);
    hipLaunchKernelGGL(k_scan_add, dim3(tiles.blocks + prims.blocks), dim3(256), 0, stream, tiles, prims, r);
    if (mark) mark(ctx, "raster_scans", 0);
}
void launch_raster(const SceneDev& s, const RasterParams& r, uint32_t samples, hipStream_t stream, void (*mark)(void*, const char*, uint64_t), void* ctx,
                   uint64_t raster_bytes) {
    hipLaunchKernelGGL(k_shape_emit, dim3(s.n_shapes), dim3(64), 0, stream, s, r);
    if (mark) mark(ctx, "raster_shape_emit", 0);
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
