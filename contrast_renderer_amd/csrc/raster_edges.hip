// csrc/raster_edges.hip — the plain Stencil + Color pass (Shape::render renderer.rs:267-355 with the stencil states renderer.rs:565-582,
// 736-754 and the fragment stages shaders.wgsl:233-309) as boundary edges + backdrop.
//
// The reference draws the interior of a filled path as a triangle strip (triangle_fan_to_strip, vertex.rs:28-35; renderer.rs:304-318) and
// covers the Shape with the strip of its convex hull (renderer.rs:340-354). Both strips are long thin triangles across the whole Shape —
// five of six (tile, triangle) pairs of the benchmark scene. Their sum is the winding number of the strip's BOUNDARY chain (interior strip
// edges are shared by two triangles that see exactly negated edge functions under the top-left rule, so they cancel sample by sample), and
// that is what this file evaluates, with the same canonical-orientation edge function  E = fma(ry, bx, fma(rx, nay, c))  per boundary edge:
//
//   g_e(p) = E_e(p) > 0 || (E_e(p) == 0 && top-left of the canonical direction)          (what a strip triangle on that side would accept)
//   w(p)   = sum_e sigma_e * Y_e(p.y) * (g_e(p) - down_e)                                 ray to -x; Y = half-open y range, sigma = chain direction
//
// per 16x16 tile T with q_k = (left tile boundary, y of sample row k):
//   w(p)   = BD(T) + sum_{e touching T} sigma_e * [ xr_e * (g_e(q_k) - g_e(q_0)) + Y_e(k) * (g_e(p) - g_e(q_k)) ]
//   BD(T)  = w(q_0), the backdrop, summed over ALL edges of the chain by the binning kernel (one lane per edge, ballots);
//   the bracket is the crossing count of the path q_0 -> q_k -> p with e, non-zero only for edges whose g is not constant over the tile.
// Every term is an evaluation of the same f32 expression the triangle path uses, at sample positions or at q_k, so the result equals the
// strip's sample for sample (tools/proto_edges.cpp checks this formulation against oracle/raster.hpp on the CPU, bit for bit; the GPU
// parity tests check this file). Curve and stroke triangles stay triangles.
//
//   k_bin_edges<S>     ONE traversal per draw item: waves 0-1 set up and walk the triangles, waves 2-3 the boundary edges (fill chain +
//                      hull chain) over the item's tile rectangle; per tile they count entries (one atomic) and append (tile, key) pairs
//                      to a wave-private LDS stage that is flushed to the pair stream in blocks (one atomic per block).
//   k_scatter          pair -> its slot in the tile's list (offsets from the scan of the counts).
//   k_raster_edges<..> one wavefront per tile as in raster.hip; entries are triangles, edges and one COVER entry per (item, tile).
// Keys are slot numbers of a 32-byte primitive heap (a triangle owns four slots = its 128-byte record); they ascend in draw order.
#include <algorithm>
#include <type_traits>
#include <vector>

#include "raster_common.hpp"

namespace crh {

void launch_scan_tiles(const RasterParams& r, hipStream_t stream); // raster.hip: exclusive scan of tile_count -> tile_offset, pair total, longest list

constexpr uint32_t EK_EDGE = 0, EK_SYNTH = 7, EK_COVER_TRI = 8; // kinds 1..6 = KIND_IQ .. KIND_JOINT as in raster_common.hpp (flags bits 4-7)
// Two refinements of a COVER entry's code, decided by the bin kernel per (item, tile):
//   + kCoverHull    the whole tile lies inside the item's hull (no hull edge matters there, hull backdrop non-zero): the cover resets the
//                   winding of EVERY sample of the tile;
//   + kCoverOpaque  also: the item is opaque and the whole tile lies inside its fill (no fill edge matters, backdrop winding non-zero under
//                   the winding rule): unless a sample inherits a winding that cancels the backdrop, the cover REPLACES the tile.
// k_raster_edges uses them to start a tile's list late (see there): painter's-order occlusion, verified per tile, exact.
constexpr uint32_t kCoverHull = 9u, kCoverOpaque = 18u;
constexpr uint32_t kEdgeTl = 1u, kEdgeSigmaPos = 2u, kEdgeHull = 4u;
// Synthetic slots of an item (flags bits 8-11 = code): 0 BD+1, 1 BD-1 (fill winding of the whole tile), 2 HBD+1, 3 HBD-1 (hull winding of the
// whole tile); 4 + (bd + 1) + 3 * (hbd + 1): COVER with one unit of both backdrops folded in (bd, hbd in -1..1).
// Slot layout of an item, in key (= draw) order, every region a multiple of 4 slots:
//   triangles (stroke lines, joints, the four curve lists; 4 slots each) | fill chain edges | BD / HBD slots (4) |
//   hull region: the hull chain's edges (1 slot each) or — a hull strip whose triangles do not all face the same way — its triangles as
//   cover triangles (4 slots each; the region is sized for those) | the 9 COVER slots (12)
struct EdgeRec {
    uint32_t flags, pad0;
    float lo_x, lo_y, hi_x, hi_y, bx, nay;
};
struct SynthRec {
    uint32_t flags, first_slot; // first_slot: the item's first slot (its triangles and fill edges lie in [first_slot, synth_a))
    float r, g, b, a;
    uint32_t synth_a, pad;      // the item's first backdrop slot
};
static_assert(sizeof(EdgeRec) == 32 && sizeof(SynthRec) == 32, "slots");

struct ItemSlots {
    uint32_t n_tri, n_fe, n_hull; // triangles, fill chain edges (= polygon vertices), hull vertices (0: no cover)
    uint32_t fe0, synth_a, hull0, synth_b, total; // region offsets from the item's first slot
    uint32_t cb[8];
};
CRH_D ItemSlots item_slots(const SceneDev& s, const DrawItem& it) {
    ItemSlots k;
    shape_candidates(s, it.shape, k.cb);
    const uint32_t* b0 = s.shape_base + it.shape * kShapeRow;
    const bool stencil = (it.ops & 1u) != 0u, cover = (it.ops >> 4) != 0u;
    const uint32_t hn = s.hull_count[it.shape];
    k.n_tri = stencil ? k.cb[1] + (k.cb[6] - k.cb[2]) : 0u; // stroke line + joint triangles, then the four curve lists
    k.n_fe = stencil ? b0[NCH + CH_SOLID_V] - b0[CH_SOLID_V] : 0u; // one boundary edge per polygon vertex
    k.n_hull = (cover && hn >= 3u) ? hn : 0u;
    k.fe0 = 4u * k.n_tri;
    k.synth_a = k.fe0 + ((k.n_fe + 3u) & ~3u);
    k.hull0 = k.synth_a + 4u;
    k.synth_b = k.hull0 + (k.n_hull ? 4u * (k.n_hull - 2u) : 0u);
    k.total = k.synth_b + 28u; // 9 COVER codes, the same 9 as "the whole tile inside the hull" (kCoverHull) and as "opaque over the whole tile" (kCoverOpaque)
    return k;
}
// What the primitives of one draw item are set up from: gathered once per item (a wavefront's scalar registers in k_bin_edges, an LDS
// record in k_bin_flat) instead of through the Shape's rows of shape_base per primitive.
struct ItemCtx {
    uint32_t shape, instance, dyn0;
    uint32_t lv0, jn0, iq0, ic0, rq0, rc0, hull0, sv0; // the Shape's first record in every stream
    uint32_t cb[8];                                    // shape_candidates()
    float m0, m4, m12, m1, m5, m13;                    // the rows of the instance matrix to_framebuffer() uses
    float col[4];                                      // straight-alpha colour of the instance
};
CRH_D ItemCtx item_ctx(const SceneDev& s, const RasterParams& r, const DrawItem& it, const uint32_t cb[8]) {
    ItemCtx c;
    const uint32_t* b0 = s.shape_base + it.shape * kShapeRow;
    c.shape = it.shape, c.instance = it.instance, c.dyn0 = s.shape_dyn_begin[it.shape];
    c.lv0 = b0[CH_LINE_V], c.jn0 = b0[CH_JOINT], c.iq0 = b0[CH_IQ], c.ic0 = b0[CH_IC_V], c.rq0 = b0[CH_RQ], c.rc0 = b0[CH_RC_V], c.hull0 = b0[CH_HULL], c.sv0 = b0[CH_SOLID_V];
#pragma unroll
    for (int i = 0; i < 8; ++i) c.cb[i] = cb[i];
    const float* m = r.transforms + 16u * it.instance;
    c.m0 = m[0], c.m4 = m[4], c.m12 = m[12], c.m1 = m[1], c.m5 = m[5], c.m13 = m[13];
    const float* color = r.colors + 4u * it.instance;
    c.col[0] = color[0], c.col[1] = color[1], c.col[2] = color[2], c.col[3] = color[3];
    return c;
}
CRH_D float2 to_framebuffer(const ItemCtx& c, float w, float h, float x, float y) { // raster_common.hpp to_framebuffer, operation for operation
    const float cx = (c.m0 * x + c.m4 * y) + c.m12;
    const float cy = (c.m1 * x + c.m5 * y) + c.m13;
    return make_float2((cx * 0.5f + 0.5f) * w, (0.5f - cy * 0.5f) * h);
}
// The tile split of the multi-GPU path: does the item's Shape, as this instance places it, miss the rows of the pass' slab altogether? Then the
// binning kernels need not set its primitives up (one rank of eight used to set all 100 000 items of config 4 up to find that 7 in 8 have no row
// in its slab). The box is the Shape's own (k_shape_bounds), its four corners go through the instance's affine map — the extremes of y are at
// corners —, a tile row of margin absorbs the rounding of that map against the vertices' own. Unbounded boxes (stroked Shapes, degenerate hulls)
// and anything not finite never pass the test.
CRH_D bool item_misses_slab(const RasterParams& r, const DrawItem& it) {
    if (!r.shape_bounds) return false;
    const float* bb = r.shape_bounds + 4u * it.shape;
    const float* m = r.transforms + 16u * it.instance;
    const float m1 = m[1], m5 = m[5], m13 = m[13]; // (the row of the instance matrix to_framebuffer() takes y from)
    const float y00 = (m1 * bb[0] + m5 * bb[1]) + m13, y10 = (m1 * bb[2] + m5 * bb[1]) + m13, y01 = (m1 * bb[0] + m5 * bb[3]) + m13, y11 = (m1 * bb[2] + m5 * bb[3]) + m13;
    const float h = (float)r.height;
    const float cy_hi = fmaxf(fmaxf(y00, y10), fmaxf(y01, y11)), cy_lo = fminf(fminf(y00, y10), fminf(y01, y11));
    const float top = (0.5f - cy_hi * 0.5f) * h, bottom = (0.5f - cy_lo * 0.5f) * h; // (to_framebuffer: y grows downwards)
    const float slab_top = (float)(r.slab_ty0 * kTile), slab_bottom = (float)(min(r.slab_ty1, r.tiles_y) * kTile);
    const bool finite = is_finite(y00) && is_finite(y10) && is_finite(y01) && is_finite(y11);
    return finite && (bottom + (float)kTile < slab_top || top - (float)kTile >= slab_bottom);
}
__global__ __launch_bounds__(256) void k_shape_bounds(SceneDev s, float* bounds) {
    const uint32_t shape = blockIdx.x * 256u + threadIdx.x;
    if (shape >= s.n_shapes) return;
    const uint32_t* b0 = s.shape_base + shape * kShapeRow;
    const uint32_t* b1 = b0 + NCH;
    const uint32_t hn = s.hull_count[shape];
    const float inf = __uint_as_float(0x7f800000u);
    float4 box = make_float4(-inf, -inf, inf, inf); // unbounded: never left out
    // a filled Shape draws polygon vertices and curve control points — all of them hull candidates (fill.rs:263-367), so the hull's box holds
    // them; a stroked one also draws join triangles around the path's own control points (stroke.rs:53-121), which an offset stroke leaves outside
    const bool stroked = (b1[CH_LINE_V] - b0[CH_LINE_V]) + (b1[CH_JOINT] - b0[CH_JOINT]) != 0u;
    if (!stroked && hn >= 3u) {
        box = make_float4(inf, inf, -inf, -inf);
        const Vertex0* v = s.hull_v + b0[CH_HULL];
        for (uint32_t i = 0; i < hn; ++i) box.x = fminf(box.x, v[i].x), box.y = fminf(box.y, v[i].y), box.z = fmaxf(box.z, v[i].x), box.w = fmaxf(box.w, v[i].y);
    }
    reinterpret_cast<float4*>(bounds)[shape] = box;
}
void launch_shape_bounds(const SceneDev& s, float* bounds, hipStream_t stream) {
    if (s.n_shapes) hipLaunchKernelGGL(k_shape_bounds, dim3((s.n_shapes + 255u) / 256u), dim3(256), 0, stream, s, bounds);
}
// one lane per item of a pass with a slab: r.item_elsewhere[item] = the item misses the slab. (A kernel of its own in front of the binning kernels:
// the test inside k_bin_flat's first phase cost that kernel its last free registers — 12 B of scratch, whose accesses wait with the record stores.)
__global__ __launch_bounds__(256) void k_slab_items(RasterParams r, uint8_t* elsewhere) {
    const uint32_t item = blockIdx.x * 256u + threadIdx.x;
    if (item < r.n_items) elsewhere[item] = item_misses_slab(r, item_of(r, item)) ? 1u : 0u;
}
void launch_slab_items(const RasterParams& r, uint8_t* elsewhere, hipStream_t stream) {
    if (r.n_items) hipLaunchKernelGGL(k_slab_items, dim3((r.n_items + 255u) / 256u), dim3(256), 0, stream, r, elsewhere);
}
__global__ __launch_bounds__(256) void k_item_nslots(SceneDev s, RasterParams r, uint32_t n_items, uint32_t* out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_items) return;
    out[i] = item_slots(s, item_of(r, i)).total;
}

// ---------------------------------------------------------------------------------------------- triangle setup (plain instances)
// oracle/raster.hpp setup_triangle + setup_attribute for candidate c of the Shape (lines, joints, curve lists); false: nothing to draw
CRH_D bool setup_plain_triangle(const SceneDev& s, const RasterParams& r, const ItemCtx& ctx, uint32_t c, PrimRec& rec) {
    const uint32_t* cb = ctx.cb;
    const uint32_t dyn0 = ctx.dyn0;
    const float W = (float)r.width, H = (float)r.height;
    float2 p[3];
    float attr[3][4] = {};
    uint32_t kind, flat_u = 0, desc = 0;
    float end_y = 0.0f;
    int n_attr;
    bool valid = true;
    if (c < cb[0]) { // stroke line strips
        const uint32_t lv0 = ctx.lv0, k = c;
        valid = s.line_pair_cut[(lv0 + k) >> 1] == 0;
        const uint32_t i0 = lv0 + k, i1 = lv0 + ((k & 1u) ? k + 2u : k + 1u), i2 = lv0 + ((k & 1u) ? k + 1u : k + 2u);
        const Vertex2f1i a = s.line_v[i0], b = s.line_v[i1], d = s.line_v[i2];
        p[0] = make_float2(a.x, a.y), p[1] = make_float2(b.x, b.y), p[2] = make_float2(d.x, d.y);
        attr[0][0] = a.u, attr[0][1] = a.v, attr[1][0] = b.u, attr[1][1] = b.v, attr[2][0] = d.u, attr[2][1] = d.v;
        flat_u = a.i;
        end_y = a.v;
        desc = dyn0 + (a.i & 65535u);
        kind = KIND_LINE;
        n_attr = 2;
    } else if (c < cb[1]) { // joint strips: 5 vertices, 3 triangles per join
        const uint32_t q = c - cb[0], jn = q / 3u, k = q - 3u * jn, base = 5u * (ctx.jn0 + jn);
        const uint32_t i0 = base + k, i1 = base + ((k & 1u) ? k + 2u : k + 1u), i2 = base + ((k & 1u) ? k + 1u : k + 2u);
        const Vertex3f1i a = s.joint_v[i0], b = s.joint_v[i1], d = s.joint_v[i2];
        p[0] = make_float2(a.x, a.y), p[1] = make_float2(b.x, b.y), p[2] = make_float2(d.x, d.y);
        attr[0][0] = a.u, attr[0][1] = a.v, attr[0][2] = a.w, attr[1][0] = b.u, attr[1][1] = b.v, attr[1][2] = b.w, attr[2][0] = d.u, attr[2][1] = d.v, attr[2][2] = d.w;
        flat_u = a.i;
        desc = dyn0 + (flat_u & 65535u);
        kind = KIND_JOINT;
        n_attr = 3;
    } else if (c < cb[3]) {
        const uint32_t at = 3u * (ctx.iq0 + (c - cb[2]));
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const Vertex2f a = s.iq_v[at + v];
            p[v] = make_float2(a.x, a.y);
            attr[v][0] = a.u, attr[v][1] = a.v;
        }
        kind = KIND_IQ;
        n_attr = 2;
    } else if (c < cb[4]) {
        const uint32_t at = ctx.ic0 + 3u * (c - cb[3]);
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const Vertex3f a = s.ic_v[at + v];
            p[v] = make_float2(a.x, a.y);
            attr[v][0] = a.u, attr[v][1] = a.v, attr[v][2] = a.w;
        }
        kind = KIND_IC;
        n_attr = 3;
    } else if (c < cb[5]) {
        const uint32_t at = 3u * (ctx.rq0 + (c - cb[4]));
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const Vertex3f a = s.rq_v[at + v];
            p[v] = make_float2(a.x, a.y);
            attr[v][0] = a.u, attr[v][1] = a.v, attr[v][2] = a.w;
        }
        kind = KIND_RQ;
        n_attr = 3;
    } else if (c < cb[6]) {
        const uint32_t at = ctx.rc0 + 3u * (c - cb[5]);
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const Vertex4f a = s.rc_v[at + v];
            p[v] = make_float2(a.x, a.y);
            attr[v][0] = a.k, attr[v][1] = a.l, attr[v][2] = a.m, attr[v][3] = a.n;
        }
        kind = KIND_RC;
        n_attr = 4;
    } else { // a triangle of the hull strip as a cover triangle (hull strips whose triangles face both ways)
        const uint32_t k = c - cb[6], hull0 = ctx.hull0;
        const Vertex0 a = s.hull_v[hull0 + k], b = s.hull_v[hull0 + ((k & 1u) ? k + 2u : k + 1u)], d = s.hull_v[hull0 + ((k & 1u) ? k + 1u : k + 2u)];
        p[0] = make_float2(a.x, a.y), p[1] = make_float2(b.x, b.y), p[2] = make_float2(d.x, d.y);
        kind = EK_COVER_TRI;
        n_attr = 0;
    }
#pragma unroll
    for (int v = 0; v < 3; ++v) p[v] = to_framebuffer(ctx, W, H, p[v].x, p[v].y);
    const float d1x = p[1].x - p[0].x, d1y = p[1].y - p[0].y;
    const float d2x = p[2].x - p[0].x, d2y = p[2].y - p[0].y;
    const float det = d1x * d2y - d2x * d1y;
    if (!(valid && det != 0.0f && det == det && is_finite(det))) return false;
    const float minx = fminf(p[0].x, fminf(p[1].x, p[2].x)), maxx = fmaxf(p[0].x, fmaxf(p[1].x, p[2].x));
    const float miny = fminf(p[0].y, fminf(p[1].y, p[2].y)), maxy = fmaxf(p[0].y, fmaxf(p[1].y, p[2].y));
    const bool nan_free = minx == minx && maxx == maxx && miny == miny && maxy == maxy;
    const int x0 = (int)floorf(fminf(fmaxf(minx, 0.0f), W)), x1 = (int)floorf(fmaxf(fminf(maxx, W - 1.0f), -1.0f));
    const int y0 = (int)floorf(fminf(fmaxf(miny, 0.0f), H)), y1 = (int)floorf(fmaxf(fminf(maxy, H - 1.0f), -1.0f));
    if (!(nan_free && x0 <= x1 && y0 <= y1)) return false;
    rec.cov.box = make_ushort4((unsigned short)x0, (unsigned short)x1, (unsigned short)y0, (unsigned short)y1);
    const float inv_det = 1.0f / det;
    const bool front = det < 0.0f;
    const float2 nv[3] = {p[0], det < 0.0f ? p[2] : p[1], det < 0.0f ? p[1] : p[2]};
    uint32_t flags = (front ? 8u : 0u) | (kind << 4);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float2 a = nv[i], b = nv[(i + 1) % 3];
        const float dx = b.x - a.x, dy = b.y - a.y;
        if (dy < 0.0f || (dy == 0.0f && dx > 0.0f)) flags |= 1u << i;
        const bool flip = !(a.x < b.x || (a.x == b.x && a.y < b.y));
        const float2 el = flip ? b : a, eh = flip ? a : b;
        const float sg = flip ? -1.0f : 1.0f;
        rec.cov.lo_x[i] = el.x;
        rec.cov.lo_y[i] = el.y;
        rec.cov.bx[i] = (eh.x - el.x) * sg;
        rec.cov.nay[i] = -(eh.y - el.y) * sg;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) { // selects, not branches on the run-time n_attr: those made the compiler index the record in scratch memory (40 B per lane,
        // and every scratch access is a vector-memory operation that waits for the record stores in flight)
        const float da1 = attr[1][a] - attr[0][a], da2 = attr[2][a] - attr[0][a];
        const bool on = a < n_attr;
        rec.frag.a0[a] = on ? attr[0][a] : 0.0f;
        rec.frag.gx[a] = on ? (da1 * d2y - da2 * d1y) * inv_det : 0.0f;
        rec.frag.gy[a] = on ? (da2 * d1x - da1 * d2x) * inv_det : 0.0f;
    }
    if (kind == EK_COVER_TRI) { // color_cover: (rgb * a, a), shaders.wgsl:304-309
        const float* color = ctx.col;
        rec.frag.a0[0] = color[0] * color[3], rec.frag.a0[1] = color[1] * color[3], rec.frag.a0[2] = color[2] * color[3], rec.frag.a0[3] = color[3];
    }
    rec.frag.v0x = p[0].x;
    rec.frag.v0y = p[0].y;
    rec.frag.flat_u = flat_u;
    rec.frag.end_y = end_y;
    rec.cov.flags = flags;
    rec.cov.desc = desc;
    return true;
}

// ---------------------------------------------------------------------------------------------- pair stage
// (tile, position in the tile's list, key) triples of one wavefront, staged in LDS and written out in blocks with coalesced stores. The
// position comes from the returning atomic on the tile's counter (addresses spread over the frame), so k_scatter needs no atomics. The
// pair stream is cut into kSubStreams regions with a cursor each — a wavefront's blocks go to the regions in turn: ONE cursor for
// the whole frame serialises ~10^4 same-address atomics in L2 and was measured to cost more than all the binning arithmetic.
constexpr uint32_t kStage = 512;
constexpr uint32_t kSubStreams = 64;
struct Stage {
    uint32_t* tile;
    uint32_t* pos;
    uint32_t* key;
    uint32_t used;
    uint32_t sub; // the sub-stream of the wavefront's next block
    uint32_t cap; // entries the wavefront's stage holds (flushed when fewer than 64 are free)
    uint32_t at;  // k_bin_flat: where the wavefront's next block goes in the pair stream — its share of the range the workgroup reserved (0xFFFFFFFF: dropped)
};
// direct tile lists (RasterParams::direct): the staged entries go where they belong
CRH_D void stage_flush_direct(Stage& st, const RasterParams& r, uint32_t lane) {
    for (uint32_t i = lane; i < st.used; i += 64u) {
        const uint32_t t = st.tile[i], p = st.pos[i], base = r.tile_base[t];
        if (p < r.tile_base[t + 1u] - base && base + p < r.pair_capacity) r.tile_list[base + p] = st.key[i]; // (places computed on the device may run beyond the buffer: seen here, drawn again)
        else r.overflow[0] = 1u; // the tile has outgrown the place the previous frame left it
    }
    __builtin_amdgcn_wave_barrier();
    st.used = 0u;
}
CRH_D void stage_flush(Stage& st, const RasterParams& r, uint32_t lane) {
    if (st.used == 0u) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (r.direct) return stage_flush_direct(st, r, lane);
    const uint32_t region = r.pair_capacity / kSubStreams;
    uint32_t base = 0;
    if (lane == 0u) {
#ifdef CRH_ABLATE
        if (r.debug & 262144u) base = (st.sub * 7919u) % (region / 2u); else
#endif
        base = atomicAdd(&r.pair_cursor[st.sub], st.used);
        if (base + st.used > region) r.overflow[5] = 1u; // this region is full: the host grows the stream and runs the pass again
    }
    base = __shfl(base, 0, 64);
    const uint32_t first = st.sub * region;
    for (uint32_t i = lane; i < st.used; i += 64u)
        if (base + i < region) {
            r.pair_tile[first + base + i] = st.tile[i];
            r.pair_pos[first + base + i] = st.pos[i];
            r.pair_key[first + base + i] = st.key[i];
        }
    __builtin_amdgcn_wave_barrier();
    st.used = 0u;
    st.sub = (st.sub + 7u) % kSubStreams; // the next block goes to another region: one huge Shape must not fill a single region
}
// k_bin_flat: the wavefront's block goes to the range reserved for it (no atomic, no wait: coalesced stores only)
CRH_D void stage_flush_reserved(Stage& st, const RasterParams& r, uint32_t lane) {
    if (st.used == 0u) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (r.direct) return stage_flush_direct(st, r, lane);
#ifdef CRH_ABLATE
    if (r.debug & 2097152u) st.at = 0xFFFFFFFFu; // tools/ablate_flat.sh: no pair stores
#endif
    if (st.at != 0xFFFFFFFFu) {
        for (uint32_t i = lane; i < st.used; i += 64u) {
            r.pair_tile[st.at + i] = st.tile[i];
            r.pair_pos[st.at + i] = st.pos[i];
            r.pair_key[st.at + i] = st.key[i];
        }
        st.at += st.used;
    }
    __builtin_amdgcn_wave_barrier();
    st.used = 0u;
}
CRH_D uint32_t lanes_below(unsigned long long ballot, uint32_t lane) { return (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull)); }
// the lanes of `ballot` append one entry each
template <bool RESERVED = false>
CRH_D void stage_append(Stage& st, const RasterParams& r, uint32_t lane, unsigned long long ballot, uint32_t tile, uint32_t pos, uint32_t key) {
#ifdef CRH_ABLATE
    if (r.debug & 512u) return;
#endif
    if ((ballot >> lane) & 1ull) {
        const uint32_t at = st.used + lanes_below(ballot, lane);
        st.tile[at] = tile;
        st.pos[at] = pos;
        st.key[at] = key;
    }
    st.used += (uint32_t)__popcll(ballot);
    if (st.used > st.cap - 64u) {
        if (RESERVED) stage_flush_reserved(st, r, lane); else stage_flush(st, r, lane);
    }
}

CRH_D bool accepts(float e, uint32_t tl) { return e > 0.0f || (e == 0.0f && tl != 0u); }

// ---------------------------------------------------------------------------------------------- k_bin_edges
struct BinEdge { // one boundary edge of the item, canonical orientation
    float lo_x, lo_y, hi_x, hi_y, bx, nay, ymin, ymax;
    uint32_t tl, hull;
    int sigma, down;
    bool valid;
    float strip_det; // hull chain: det of the strip triangle that starts at this position (0: none, degenerate or not finite)
};
// boundary chain of a zig-zag strip (vertex.rs:28-35): the edge owned by strip position `pos` runs to position `target`
//   pos 0 -> 1;  even pos >= 2 -> pos - 2;  odd pos -> pos + 2, or — at the end of the strip — to the other one of the last two positions
// Edge i of an item: i < n_fe the fill chain(s), then the hull chain (n_hull_chain edges: 0 when the hull is drawn as triangles).
// `n_fe` fill chain edges, then `n_hull_chain` hull chain edges. An endpoint that is not finite on the frame (finite vertices times a
// finite matrix can overflow) sets *broken: the chain is not closed any more, so its backdrops mean nothing — the frame is then drawn by
// the triangle pass, which skips exactly the strip triangles with a non-finite determinant as the reference's rasterizer would.
CRH_D BinEdge load_edge(const SceneDev& s, const RasterParams& r, const ItemCtx& ctx, uint32_t n_fe, uint32_t n_hull_chain, uint32_t i) {
    BinEdge e = {};
    e.valid = false;
    if (i >= n_fe + n_hull_chain) return e;
    float2 a, b;
    if (i < n_fe) {
        // Everything the edge may need is requested at once — the flags of the neighbours and the four vertices the chain can run to —
        // instead of flag -> neighbour's flag -> target vertex one after the other (the item's wavefront spent 40 % of its time in this
        // chain of dependent loads). Indices are clamped to the item's own vertices; what is selected always exists.
        const uint32_t sv0 = ctx.sv0, g = sv0 + i, g_last = sv0 + n_fe - 1u;
        const uint32_t gm1 = i >= 1u ? g - 1u : g, gm2 = i >= 2u ? g - 2u : g, gp1 = min(g + 1u, g_last), gp2 = min(g + 2u, g_last);
        const uint32_t f = s.solid_flag[g], f_prev = s.solid_flag[gm1], f_next = s.solid_flag[gp1];
        const Vertex0 va = s.solid_v[g], vm1 = s.solid_v[gm1], vm2 = s.solid_v[gm2], vp1 = s.solid_v[gp1], vp2 = s.solid_v[gp2];
        const bool odd = (f & 1u) != 0u, last = (f & 2u) != 0u;
        const bool first = !odd && (i == 0u || (f_prev & 2u) != 0u);
        Vertex0 vb;
        if (first) {
            if (last) return e; // a strip of one vertex
            vb = vp1;
        } else if (!odd) {
            vb = vm2;
        } else {
            vb = last ? vm1 : ((f_next & 2u) ? vp1 : vp2);
        }
        a = make_float2(va.x, va.y), b = make_float2(vb.x, vb.y);
    } else {
        const uint32_t pos = i - n_fe, n = n_hull_chain, hull0 = ctx.hull0;
        uint32_t target;
        if (pos == 0u)
            target = 1u;
        else if ((pos & 1u) == 0u)
            target = pos - 2u;
        else
            target = pos + 1u == n ? pos - 1u : (pos + 2u == n ? pos + 1u : pos + 2u);
        const Vertex0 va = s.hull_v[hull0 + pos], vb = s.hull_v[hull0 + target];
        a = make_float2(va.x, va.y), b = make_float2(vb.x, vb.y);
        e.hull = 1u;
        if (pos + 2u < n) { // strip triangle `pos` = (pos, pos + 1, pos + 2), odd ones with the last two swapped: which way does it face?
            const float W = (float)r.width, H = (float)r.height;
            const Vertex0 v1 = s.hull_v[hull0 + ((pos & 1u) ? pos + 2u : pos + 1u)], v2 = s.hull_v[hull0 + ((pos & 1u) ? pos + 1u : pos + 2u)];
            const float2 p0 = to_framebuffer(ctx, W, H, va.x, va.y), p1 = to_framebuffer(ctx, W, H, v1.x, v1.y), p2 = to_framebuffer(ctx, W, H, v2.x, v2.y);
            const float d1x = p1.x - p0.x, d1y = p1.y - p0.y, d2x = p2.x - p0.x, d2y = p2.y - p0.y;
            const float det = d1x * d2y - d2x * d1y; // setup_plain_triangle's det
            e.strip_det = (det == det && is_finite(det)) ? det : 0.0f;
        }
    }
    const float W = (float)r.width, H = (float)r.height;
    a = to_framebuffer(ctx, W, H, a.x, a.y);
    b = to_framebuffer(ctx, W, H, b.x, b.y);
    if (!(is_finite(a.x) && is_finite(a.y) && is_finite(b.x) && is_finite(b.y))) {
        r.overflow[7] = 1u; // (see above; the host draws the frame again with the triangle pass)
        return e;
    }
    if (a.x == b.x && a.y == b.y) return e;
    const bool flip = !(a.x < b.x || (a.x == b.x && a.y < b.y)); // canonical (lexicographic) endpoint order
    const float2 lo = flip ? b : a, hi = flip ? a : b;
    e.lo_x = lo.x, e.lo_y = lo.y, e.hi_x = hi.x, e.hi_y = hi.y;
    e.bx = hi.x - lo.x;
    e.nay = -(hi.y - lo.y);
    const float dx = hi.x - lo.x, dy = hi.y - lo.y;
    e.tl = (dy < 0.0f || (dy == 0.0f && dx > 0.0f)) ? 1u : 0u;
    e.down = dy > 0.0f ? 1 : 0;
    e.sigma = flip ? 1 : -1; // -1: the chain runs in the canonical direction
    e.ymin = fminf(lo.y, hi.y), e.ymax = fmaxf(lo.y, hi.y);
    e.valid = true;
    return e;
}
CRH_D float wave_min(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
CRH_D float wave_max(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
CRH_D uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}

// The exact tile test of a set-up triangle: its best tile corner per edge decides (an edge function is monotone in x and y under fmaf).
// A conservative superset of "some sample of the tile is covered"; the raster kernel decides per sample.
struct TileTest {
    float bx[3], nay[3], lo_x[3], lo_y[3];
    float s_lo, s_hi;  // extreme sample offsets inside a tile
    uint32_t tl;       // bits 0-2: top-left per edge
    CRH_D void set(const PrimCoverage& cov, float lo, float hi) {
#pragma unroll
        for (int i = 0; i < 3; ++i) bx[i] = cov.bx[i], nay[i] = cov.nay[i], lo_x[i] = cov.lo_x[i], lo_y[i] = cov.lo_y[i];
        s_lo = lo, s_hi = hi;
        tl = cov.flags & 7u;
    }
    CRH_D bool hit(uint32_t tx, uint32_t ty) const {
        const float tx0 = (float)(tx * kTile), ty0 = (float)(ty * kTile);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) { // the best corner of the tile for this edge (selected here: two registers per edge less to carry)
            const float best_x = nay[i] > 0.0f ? s_hi : s_lo, best_y = bx[i] > 0.0f ? s_hi : s_lo;
            const float e = fmaf(best_y, bx[i], fmaf(best_x, nay[i], bx[i] * (ty0 - lo_y[i]) + nay[i] * (tx0 - lo_x[i])));
            ok = ok && accepts(e, (tl >> i) & 1u);
        }
        return ok;
    }
};
// Bins up to 64 set-up triangles (lane = triangle): every lane walks the tiles of ITS OWN pixel box — a few for a curve or stroke
// triangle; a triangle over more than kBigRect tiles is walked by the whole wavefront instead (lane = tile), one such triangle at a time.
#ifndef CRH_BIN_WAVES
#define CRH_BIN_WAVES 4 // measured 4, 5, 6: the same within noise (the kernel waits for memory, not for issue slots); 4 needs no spills
#endif
constexpr uint32_t kBigRect = 32;
constexpr uint32_t kRectLds = 256; // tiles of an item's rectangle whose backdrops fit the LDS table of the lane = edge path
CRH_D void bin_triangles(Stage& st, const RasterParams& r, uint32_t lane, bool drawn, const PrimCoverage& cov, uint32_t key, float s_lo, float s_hi) {
    TileTest test;
    test.set(cov, s_lo, s_hi);
    // (the tile rows of the pass' slab only, crh_frame_set_tile_rows: the raster kernels draw no others)
    const uint32_t bx0 = cov.box.x / kTile, bx1 = cov.box.y / kTile, by0 = max((uint32_t)cov.box.z / kTile, r.slab_ty0), by1 = min((uint32_t)cov.box.w / kTile + 1u, r.slab_ty1); // [by0, by1)
    const uint32_t nx = bx1 - bx0 + 1u, nt = (drawn && by0 < by1) ? nx * (by1 - by0) : 0u;
    const bool big = nt > kBigRect || (nt != 0u && (r.debug & 2u) != 0u); // debug bit 1 (tests): every triangle takes the wide path
    const uint32_t mine = big ? 0u : nt, longest = wave_max_u32(mine);
    uint32_t tx = bx0, ty = by0;
    for (uint32_t i = 0; i < longest; ++i) {
        const bool hit = i < mine && test.hit(tx, ty);
        const unsigned long long ballot = __ballot(hit);
        if (ballot) {
            const uint32_t tile = ty * r.tiles_x + tx;
            uint32_t pos = 0;
            if (hit) pos = atomicAdd(&r.tile_count[tile], 1u);
            stage_append(st, r, lane, ballot, tile, pos, key);
        }
        if (++tx > bx1) tx = bx0, ++ty;
    }
    unsigned long long todo = __ballot(big);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        TileTest wide;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            wide.bx[i] = __shfl(test.bx[i], src, 64), wide.nay[i] = __shfl(test.nay[i], src, 64), wide.lo_x[i] = __shfl(test.lo_x[i], src, 64);
            wide.lo_y[i] = __shfl(test.lo_y[i], src, 64);
        }
        wide.s_lo = test.s_lo, wide.s_hi = test.s_hi;
        wide.tl = (uint32_t)__shfl((int)test.tl, src, 64);
        const uint32_t wx0 = (uint32_t)__shfl((int)bx0, src, 64), wy0 = (uint32_t)__shfl((int)by0, src, 64), wnx = (uint32_t)__shfl((int)nx, src, 64);
        const uint32_t wnt = (uint32_t)__shfl((int)nt, src, 64), wkey = (uint32_t)__shfl((int)key, src, 64);
        for (uint32_t base = 0; base < wnt; base += 64u) {
            const uint32_t q = base + lane, qy = q / wnx, qx = q - qy * wnx;
            const bool hit = q < wnt && wide.hit(wx0 + qx, wy0 + qy);
            const unsigned long long ballot = __ballot(hit);
            if (!ballot) continue;
            const uint32_t tile = (wy0 + qy) * r.tiles_x + (wx0 + qx);
            uint32_t pos = 0;
            if (hit) pos = atomicAdd(&r.tile_count[tile], 1u);
            stage_append(st, r, lane, ballot, tile, pos, wkey);
        }
    }
}

// The same for a chunk of triangles whose common tile rectangle fits an LDS table (the usual case): the lanes walk their boxes twice — once
// counting per tile (LDS), once emitting with positions from LDS cursors — and in between lane = tile reserves the positions with ONE
// returning atomic per tile. (bin_triangles pays a round trip to L2 per step of the walk: 60 % of that wavefront's time.)
CRH_D bool bin_triangles_counted(Stage& st, const RasterParams& r, uint32_t lane, bool drawn, const PrimCoverage& cov, uint32_t key, float s_lo, float s_hi,
                                 uint32_t* cursor) {
    const uint32_t bx0 = cov.box.x / kTile, bx1 = cov.box.y / kTile, by0 = cov.box.z / kTile, by1 = cov.box.w / kTile;
    // the chunk's rectangle
    uint32_t rx0 = drawn ? bx0 : 0xFFFFFFFFu, ry0 = drawn ? by0 : 0xFFFFFFFFu, rx1 = drawn ? bx1 : 0u, ry1 = drawn ? by1 : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        rx0 = min(rx0, (uint32_t)__shfl_xor((int)rx0, d, 64)), ry0 = min(ry0, (uint32_t)__shfl_xor((int)ry0, d, 64));
        rx1 = max(rx1, (uint32_t)__shfl_xor((int)rx1, d, 64)), ry1 = max(ry1, (uint32_t)__shfl_xor((int)ry1, d, 64));
    }
    if (rx0 == 0xFFFFFFFFu) return true; // nothing drawn
    const uint32_t nx = rx1 - rx0 + 1u, n_rect = nx * (ry1 - ry0 + 1u);
    if (n_rect > kRectLds || (r.debug & 2u) != 0u) return false; // the caller takes the walk with an atomic per step (debug bit 1: always)
    TileTest test;
    test.set(cov, s_lo, s_hi);
    const uint32_t bnx = bx1 - bx0 + 1u, nt = drawn ? bnx * (by1 - by0 + 1u) : 0u, longest = wave_max_u32(nt);
    for (uint32_t q = lane; q < n_rect; q += 64u) cursor[q] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t tx = bx0, ty = by0;
    for (uint32_t i = 0; i < longest; ++i) { // count
        if (i < nt && test.hit(tx, ty)) atomicAdd(&cursor[(ty - ry0) * nx + (tx - rx0)], 1u);
        if (++tx > bx1) tx = bx0, ++ty;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t base = 0; base < n_rect; base += 64u) { // reserve
        const uint32_t q = base + lane, qy = q / nx, qx = q - qy * nx;
        const uint32_t n = q < n_rect ? cursor[q] : 0u;
#ifdef CRH_ABLATE
        if (r.debug & 131072u) { if (n) cursor[q] = q & 7u; } else
#endif
        if (n) cursor[q] = atomicAdd(&r.tile_count[(ry0 + qy) * r.tiles_x + rx0 + qx], n);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    tx = bx0, ty = by0;
    for (uint32_t i = 0; i < longest; ++i) { // emit
        const bool hit = i < nt && test.hit(tx, ty);
        const unsigned long long ballot = __ballot(hit);
        if (ballot) {
            uint32_t pos = 0;
            if (hit) pos = atomicAdd(&cursor[(ty - ry0) * nx + (tx - rx0)], 1u);
            stage_append(st, r, lane, ballot, ty * r.tiles_x + tx, pos, key);
        }
        if (++tx > bx1) tx = bx0, ++ty;
    }
    return true;
}

#ifdef CRH_ABLATE // tools/bin_phases.py: where does a wavefront of k_bin_edges spend its time? (cycle counter deltas summed in overflow[80 ...])
#define CRH_PHASE(k)                                                                                                   \
    if (r.debug & 65536u) {                                                                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                  \
        if (lane == 0u) atomicAdd(reinterpret_cast<unsigned long long*>(r.overflow + 80) + (k) + 8u * wave, now_ - phase_t); \
        phase_t = __builtin_amdgcn_s_memtime();                                                                        \
    }
#else
#define CRH_PHASE(k)
#endif
// One workgroup per draw item. Wavefront 0: the stroke and curve triangles (bin_triangles). Wavefront 1: the boundary edges, transposed —
// lane = tile of the item's rectangle (64 per pass), uniform loop over the edges (staged in LDS): every lane accumulates the backdrops of
// its tile and the bit mask of the edges that matter inside it, then emits its entries.
// QUEUED: the items are those k_bin_flat handed on (r.bin_queue, their number in r.overflow[6]) — the ones too large for its batches.
template <int S, bool QUEUED>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(CRH_BIN_WAVES))) void k_bin_edges(SceneDev s, RasterParams r) {
    __shared__ uint32_t stage_tile[2][kStage], stage_pos[2][kStage], stage_key[2][kStage];
    __shared__ float4 edge_a[64], edge_b[64];
    __shared__ int rect_bd[kRectLds], rect_hbd[kRectLds];    // lane = edge path: backdrops of the tiles of the item's rectangle ...
    __shared__ uint32_t rect_hull_touch[kRectLds / 32u];      // ... whether a hull edge matters inside the tile ...
    __shared__ uint32_t rect_cursor[kRectLds];                // ... and the count, then the next list position, of the edges that matter there
    __shared__ uint32_t rect_cursor_tri[kRectLds];            // the same for the triangle wavefront (its own rectangle, per chunk of 64 triangles)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    Stage st = {stage_tile[wave], stage_pos[wave], stage_key[wave], 0u, ((2u * blockIdx.x + wave) * 2654435761u) >> 26, kStage, 0u}; // (a hash: consecutive wavefronts start in unrelated sub-streams)
    // a workgroup takes items blockIdx.x, blockIdx.x + gridDim.x, ...: the pair stage carries over from one item to the next (fewer, fuller
    // flushes), and the two wavefronts never synchronise with each other
    const uint32_t n_work = QUEUED ? min(r.overflow[6], r.n_items) : r.n_items;
    for (uint32_t work = blockIdx.x; work < n_work; work += gridDim.x) {
    __builtin_amdgcn_wave_barrier(); // (the previous item's readers of the LDS tables are through)
    const uint32_t item = QUEUED ? r.bin_queue[work] : work;
    const DrawItem it = item_of(r, item);
    const ItemSlots k = item_slots(s, it);
    const ItemCtx ctx = item_ctx(s, r, it, k.cb);
    const uint32_t slot0 = r.slot_begin[item];
    if (slot0 + k.total > r.slot_capacity) continue; // cannot happen: the capacity is the scan's total
    if (r.item_elsewhere && r.item_elsewhere[item] != 0u) continue; // (a pass with a slab: no tile row of the item's box is in it)
#ifdef CRH_ABLATE
    unsigned long long phase_t = __builtin_amdgcn_s_memtime();
#endif
    const float ry_first = S == 1 ? 0.5f : 0.125f, r_last = (float)(kTile - 1) + (S == 1 ? 0.5f : 0.875f); // extreme sample offsets inside a tile
#ifdef CRH_ABLATE
    if ((r.debug & 1024u) && wave == 0u) continue;
    if ((r.debug & 2048u) && wave == 1u) continue;
#endif
    if (wave == 0u) {
        // ---------------- triangles: 64 at a time, lane = triangle
        for (uint32_t t0 = 0; t0 < k.n_tri; t0 += 64u) {
            const uint32_t t = t0 + lane;
            PrimRec rec = {};
            bool drawn = false;
            CRH_PHASE(0) // item data
            if (t < k.n_tri) {
                const uint32_t c = t < k.cb[1] ? t : t - k.cb[1] + k.cb[2]; // the Shape's candidate numbering without the solid strips
                drawn = setup_plain_triangle(s, r, ctx, c, rec);
                if (drawn) *reinterpret_cast<PrimRec*>(r.slots + (size_t)(slot0 + 4u * t) * 32u) = rec;
            }
            CRH_PHASE(1) // triangle set-up
            if (!bin_triangles_counted(st, r, lane, drawn, rec.cov, slot0 + 4u * t, ry_first, r_last, rect_cursor_tri))
                bin_triangles(st, r, lane, drawn, rec.cov, slot0 + 4u * t, ry_first, r_last);
            CRH_PHASE(2) // walk
        }
    } else {
        // ---------------- boundary edges: fill chain(s) then hull chain
        const uint32_t fe_slot0 = slot0 + k.fe0, synth_a = slot0 + k.synth_a, hull_slot0 = slot0 + k.hull0, synth_b = slot0 + k.synth_b;
        const float* item_color = ctx.col;
        const bool opaque_item = item_color[3] == 1.0f && is_finite(item_color[0]) && is_finite(item_color[1]) && is_finite(item_color[2]) &&
                                 r.occlude != 0u && (r.debug & 32768u) == 0u; // debug bit 15 (tests, A/B runs): no tile is ever treated as replaced
        if (lane < 13u + kCoverOpaque) { // 4 backdrop + 27 COVER slots (the COVER ones carry the premultiplied source colour, shaders.wgsl:304-309)
            SynthRec sr = {};
            sr.flags = (EK_SYNTH << 4) | (lane << 8);
            sr.first_slot = slot0, sr.synth_a = synth_a;
            if (lane >= 4u) sr.r = item_color[0] * item_color[3], sr.g = item_color[1] * item_color[3], sr.b = item_color[2] * item_color[3], sr.a = item_color[3];
            *reinterpret_cast<SynthRec*>(r.slots + (size_t)(lane < 4u ? synth_a + lane : synth_b + lane - 4u) * 32u) = sr;
        }
        // Do all triangles of the hull strip face the same way? Then the cover — the UNION of those triangles (renderer.rs:340-354) — is
        // where the winding number of the strip's boundary chain is not zero, and the chain is binned. A strip that folds over itself
        // (andrew() decides turns with an absolute margin, convex_hull.rs:17-20: under f32 cancellation its output is not always convex)
        // is drawn as the reference draws it, triangle by triangle.
        uint32_t n_hull_chain = k.n_hull, n_edges = k.n_fe + n_hull_chain;
        unsigned long long faces_front = 0, faces_back = 0;
        // one chunk of (up to 64) edges -> LDS table (+ the heap records the first time)
        float minx = INFINITY, maxx = -INFINITY, miny = INFINITY, maxy = -INFINITY;
        BinEdge kept = {}; // the lane's edge of the first chunk (most items have no other)
        auto stage_chunk = [&](uint32_t i0, bool write_records) {
            const uint32_t i = i0 + lane;
            const BinEdge e = load_edge(s, r, ctx, k.n_fe, n_hull_chain, i);
            if (write_records && i0 == 0u) kept = e;
            const uint32_t flags = (EK_EDGE << 4) | (e.tl ? kEdgeTl : 0u) | (e.sigma > 0 ? kEdgeSigmaPos : 0u) | (e.hull ? kEdgeHull : 0u);
            if (e.valid && write_records) {
                EdgeRec er;
                er.flags = flags, er.pad0 = 0u;
                er.lo_x = e.lo_x, er.lo_y = e.lo_y, er.hi_x = e.hi_x, er.hi_y = e.hi_y, er.bx = e.bx, er.nay = e.nay;
                *reinterpret_cast<EdgeRec*>(r.slots + (size_t)(i < k.n_fe ? fe_slot0 + i : hull_slot0 + (i - k.n_fe)) * 32u) = er;
            }
            __builtin_amdgcn_wave_barrier(); // the previous chunk's readers are done
            edge_a[lane] = make_float4(e.lo_x, e.lo_y, e.bx, e.nay);
            edge_b[lane] = make_float4(e.ymin, e.ymax, e.hi_x, __uint_as_float(flags | (e.down ? 8u : 0u) | (e.valid ? 0x100u : 0u)));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (e.valid) {
                minx = fminf(minx, e.lo_x), maxx = fmaxf(maxx, e.hi_x);
                miny = fminf(miny, e.ymin), maxy = fmaxf(maxy, e.ymax);
            }
            if (write_records) { // (a strip triangle is judged at its first position whether or not that position's chain edge is valid)
                faces_front |= __ballot(e.strip_det < 0.0f);
                faces_back |= __ballot(e.strip_det > 0.0f);
            }
        };
        CRH_PHASE(0) // item data, synthetic slots
        for (uint32_t i0 = 0; i0 < n_edges; i0 += 64u) stage_chunk(i0, true); // records + the box of every vertex (one chunk: the table stays)
        CRH_PHASE(1) // edge records
        const bool hull_as_triangles = k.n_hull != 0u && ((faces_front != 0ull && faces_back != 0ull) || (r.debug & 4u) != 0u); // debug bit 2 (tests): always
        if (hull_as_triangles) { // the fill chain alone (rare: the staging is simply done again)
            n_hull_chain = 0u, n_edges = k.n_fe;
            minx = INFINITY, maxx = -INFINITY, miny = INFINITY, maxy = -INFINITY;
            for (uint32_t i0 = 0; i0 < n_edges; i0 += 64u) stage_chunk(i0, false);
        }
        const bool single = n_edges <= 64u && (r.debug & 1u) == 0u; // debug bit 0 (tests): the chunked path even for short chains
        minx = wave_min(minx), maxx = wave_max(maxx), miny = wave_min(miny), maxy = wave_max(maxy);
        const float W = (float)r.width, H = (float)r.height;
        const int px0 = (int)floorf(fminf(fmaxf(minx, 0.0f), W)), px1 = (int)floorf(fmaxf(fminf(maxx, W - 1.0f), -1.0f));
        const int py0 = (int)floorf(fminf(fmaxf(miny, 0.0f), H)), py1 = (int)floorf(fmaxf(fminf(maxy, H - 1.0f), -1.0f));
#ifdef CRH_ABLATE
        if (r.debug & 8192u) n_edges = 0u;
#endif
        // (... of the pass' slab of tile rows, crh_frame_set_tile_rows: every tile row's backdrops and entries are its own, so the others are simply left out)
        const uint32_t tx_a = (uint32_t)max(px0, 0) / kTile, tx_b = (uint32_t)max(px1, 0) / kTile, ty_a = max((uint32_t)max(py0, 0) / kTile, r.slab_ty0);
        const uint32_t ty_b_end = min((uint32_t)max(py1, 0) / kTile + 1u, r.slab_ty1), ty_b = ty_b_end - 1u; // (only used when ty_a < ty_b_end)
        const uint32_t nx = tx_b - tx_a + 1u, n_rect = ty_a < ty_b_end ? nx * (ty_b_end - ty_a) : 0u;
        const bool in_frame = n_edges != 0u && minx <= maxx && px0 <= px1 && py0 <= py1 && ty_a < ty_b_end;
        if (in_frame && n_rect <= kRectLds && (r.debug & 1u) == 0u) {
            // ---------------- lane = EDGE (the common case: the rectangle's backdrops fit the LDS table). The transposed loop below costs
            // edges x tiles of the rectangle; here every edge visits the tiles of its OWN box and the tile rows whose backdrop line it
            // crosses, the backdrops being summed in LDS.
            // Three passes: (1) every edge counts, per tile of its own box, whether it matters there (LDS counters) and adds its backdrop
            // terms; (2) lane = tile: ONE returning atomic on the tile's global counter reserves the positions of the item's entries in the
            // tile's list, the synthetic entries are emitted; (3) the edges walk their boxes again and take their positions from the LDS
            // cursors. (A returning global atomic per (edge, tile) visit made the wavefront wait for a round trip to L2 per step of the walk.)
            for (uint32_t q = lane; q < n_rect; q += 64u) rect_bd[q] = 0, rect_hbd[q] = 0, rect_cursor[q] = 0u;
            if (lane < kRectLds / 32u) rect_hull_touch[lane] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool one_chunk = k.n_fe + k.n_hull <= 64u;
            auto edge_of = [&](uint32_t i) {
                BinEdge e;
                if (one_chunk) { // still in registers (a hull chain that is drawn as triangles drops out: i >= n_edges)
                    e = kept;
                    e.valid = e.valid && i < n_edges;
                } else {
                    e = load_edge(s, r, ctx, k.n_fe, n_hull_chain, i);
                }
                return e;
            };
            // the tiles of the edge's own box (a conservative integer range; the exact test decides tile by tile), walked by all lanes together
            auto walk = [&](const BinEdge& e, auto&& visit) {
                uint32_t bx0 = tx_a, bx1 = tx_a, by0 = ty_a, nt = 0;
                if (e.valid) {
                    const int x_lo = (int)ceilf((e.lo_x - r_last) * (1.0f / (float)kTile) - 0.01f), x_hi = (int)floorf(e.hi_x * (1.0f / (float)kTile) + 0.01f);
                    const int y_lo = (int)ceilf((e.ymin - r_last) * (1.0f / (float)kTile) - 0.01f), y_hi = (int)floorf((e.ymax - ry_first) * (1.0f / (float)kTile) + 0.01f);
                    const int cx0 = max(x_lo, (int)tx_a), cx1 = min(x_hi, (int)tx_b), cy0 = max(y_lo, (int)ty_a), cy1 = min(y_hi, (int)ty_b);
                    if (cx0 <= cx1 && cy0 <= cy1) bx0 = (uint32_t)cx0, bx1 = (uint32_t)cx1, by0 = (uint32_t)cy0, nt = (uint32_t)((cx1 - cx0 + 1) * (cy1 - cy0 + 1));
                }
                const bool up = e.nay > 0.0f; // E grows with ry (bx >= 0) and with rx iff nay > 0
                const uint32_t longest = wave_max_u32(nt);
                uint32_t tx = bx0, ty = by0;
                for (uint32_t w = 0; w < longest; ++w) {
                    const float tx0 = (float)(tx * kTile), ty0 = (float)(ty * kTile), q0y = ty0 + ry_first;
                    const float c = e.bx * (ty0 - e.lo_y) + e.nay * (tx0 - e.lo_x);
                    const bool gmax = accepts(fmaf(r_last, e.bx, fmaf(up ? r_last : 0.0f, e.nay, c)), e.tl), gmin = accepts(fmaf(ry_first, e.bx, fmaf(up ? 0.0f : r_last, e.nay, c)), e.tl);
                    const bool hit = w < nt && gmax != gmin && e.ymin <= ty0 + r_last && e.ymax >= q0y && e.lo_x <= tx0 + r_last && e.hi_x >= tx0;
                    visit(hit, tx, ty);
                    if (++tx > bx1) tx = bx0, ++ty;
                }
            };
            CRH_PHASE(2) // rectangle, LDS table cleared
            for (uint32_t i0 = 0; i0 < n_edges; i0 += 64u) { // ---- pass 1
                const BinEdge e = edge_of(i0 + lane);
                // backdrop rows: the tile rows whose line q0y lies in the edge's half-open y range; a conservative integer range first
                uint32_t row = ty_a, rows_mine = 0;
                if (e.valid) {
                    const int lo = (int)ceilf((e.ymin - ry_first) * (1.0f / (float)kTile) - 0.01f), hi = (int)floorf((e.ymax - ry_first) * (1.0f / (float)kTile) + 0.01f);
                    const int first = max(lo, (int)ty_a), last = min(hi, (int)ty_b);
                    if (first <= last) row = (uint32_t)first, rows_mine = (uint32_t)(last - first + 1);
                }
                const uint32_t most_rows = wave_max_u32(rows_mine);
                for (uint32_t rr = 0; rr < most_rows; ++rr) {
                    const uint32_t ty = row + rr;
                    const float ty0 = (float)(ty * kTile), q0y = ty0 + ry_first;
                    const bool crosses = rr < rows_mine && e.ymin <= q0y && q0y < e.ymax; // Y_e at the backdrop row
                    if (!__any(crosses)) continue;
                    int* const table = (e.hull ? rect_hbd : rect_bd) + (ty - ty_a) * nx;
                    for (uint32_t cx = 0; cx < nx; ++cx) {
                        const float tx0 = (float)((tx_a + cx) * kTile);
                        const float c = e.bx * (ty0 - e.lo_y) + e.nay * (tx0 - e.lo_x);
                        const bool gq0 = accepts(fmaf(ry_first, e.bx, fmaf(0.0f, e.nay, c)), e.tl);
                        const int term = e.sigma * ((gq0 ? 1 : 0) - e.down); // sigma * Y(q0) * (g(q0) - down)
                        if (crosses && term != 0) atomicAdd(&table[cx], term);
                    }
                }
                walk(e, [&](bool hit, uint32_t tx, uint32_t ty) {
                    if (hit) {
                        const uint32_t q = (ty - ty_a) * nx + (tx - tx_a);
                        atomicAdd(&rect_cursor[q], 1u);
                        if (e.hull) atomicOr(&rect_hull_touch[q >> 5], 1u << (q & 31u));
                    }
                });
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            CRH_PHASE(3) // pass 1
            for (uint32_t base = 0; base < n_rect; base += 64u) { // ---- pass 2, lane = tile (the COVER entry carries one unit of either backdrop)
                const uint32_t q = base + lane, qy = q / nx, qx = q - qy * nx;
                const bool active = q < n_rect;
                const uint32_t tile = (ty_a + qy) * r.tiles_x + tx_a + qx;
                const int bd = active ? rect_bd[q] : 0, hbd = active ? rect_hbd[q] : 0;
                const uint32_t n_touching = active ? rect_cursor[q] : 0u;
                const bool hull_touch = active && ((rect_hull_touch[q >> 5] >> (q & 31u)) & 1u) != 0u;
                const uint32_t abd = (uint32_t)(bd < 0 ? -bd : bd), ahbd = (uint32_t)(hbd < 0 ? -hbd : hbd);
                uint32_t n_cover = (active && n_hull_chain != 0u && (hbd != 0 || hull_touch)) ? 1u : 0u; // the tile is inside the hull or its boundary crosses it
                const int cbd = bd > 0 ? 1 : (bd < 0 ? -1 : 0), chbd = hbd > 0 ? 1 : (hbd < 0 ? -1 : 0);
                const bool hull_over_tile = n_cover != 0u && hbd != 0 && !hull_touch;
                const bool replaces_tile = hull_over_tile && opaque_item && n_touching == 0u && (bd & (int)r.winding_mask) != 0;
                const uint32_t cover_key = synth_b + (uint32_t)(cbd + 1) + 3u * (uint32_t)(chbd + 1) + (replaces_tile ? kCoverOpaque : (hull_over_tile ? kCoverHull : 0u));
                if (const unsigned long long opaque = __ballot(replaces_tile)) // the host's statistic: are there tiles to start late in? (overflow[4])
                    if (lane == 0u) atomicAdd(&r.overflow[4], (uint32_t)__popcll(opaque));
                uint32_t n_bd = n_cover ? (abd ? abd - 1u : 0u) : abd;
                uint32_t n_hbd = n_cover ? (ahbd ? ahbd - 1u : 0u) : 0u;
                const uint32_t bd_key = synth_a + (bd > 0 ? 0u : 1u), hbd_key = synth_a + (hbd > 0 ? 2u : 3u);
                uint32_t left = n_cover + n_bd + n_hbd, pos = 0;
#ifdef CRH_ABLATE
                if (r.debug & 131072u) pos = tile & 7u; else
#endif
                if (left + n_touching) pos = atomicAdd(&r.tile_count[tile], left + n_touching);
                if (active) rect_cursor[q] = pos + left; // where the edges' entries go
                for (;;) {
                    const unsigned long long ballot = __ballot(left != 0u);
                    if (!ballot) break;
                    uint32_t key = 0;
                    if (left) {
                        if (n_cover)
                            n_cover = 0, key = cover_key;
                        else if (n_bd)
                            --n_bd, key = bd_key;
                        else
                            --n_hbd, key = hbd_key;
                    }
                    stage_append(st, r, lane, ballot, tile, pos, key);
                    if (left) ++pos, --left;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            CRH_PHASE(4) // pass 2
            for (uint32_t i0 = 0; i0 < n_edges; i0 += 64u) { // ---- pass 3
                const uint32_t i = i0 + lane;
                const BinEdge e = edge_of(i);
                const uint32_t key = i < k.n_fe ? fe_slot0 + i : hull_slot0 + (i - k.n_fe);
                walk(e, [&](bool hit, uint32_t tx, uint32_t ty) {
                    const unsigned long long ballot = __ballot(hit);
                    if (ballot) {
                        uint32_t pos = 0;
                        if (hit) pos = atomicAdd(&rect_cursor[(ty - ty_a) * nx + (tx - tx_a)], 1u);
                        stage_append(st, r, lane, ballot, ty * r.tiles_x + tx, pos, key);
                    }
                });
            }
        } else if (in_frame) {
            for (uint32_t base = 0; base < n_rect; base += 64u) {
                const uint32_t q = base + lane, qy = q / nx, qx = q - qy * nx;
                const bool active = q < n_rect;
                const uint32_t tx = tx_a + qx, ty = ty_a + qy, tile = ty * r.tiles_x + tx;
                const float tx0 = (float)(tx * kTile), ty0 = (float)(ty * kTile), q0y = ty0 + ry_first;
                int bd = 0, hbd = 0;
                bool hull_touch = false;
                for (uint32_t i0 = 0; i0 < n_edges; i0 += 64u) {
                    if (!single) stage_chunk(i0, false);
                    const uint32_t count = min(64u, n_edges - i0);
                    unsigned long long mask = 0;
#ifdef CRH_ABLATE
                    if (r.debug & 16384u) continue;
#endif
                    for (uint32_t j = 0; j < count; ++j) {
                        const float4 A = edge_a[j], B = edge_b[j];
                        const uint32_t flags = __builtin_amdgcn_readfirstlane(__float_as_uint(B.w));
                        if (!(flags & 0x100u)) continue;
                        const uint32_t tl = flags & kEdgeTl;
                        const float c = A.z * (ty0 - A.y) + A.w * (tx0 - A.x);
                        const bool gq0 = accepts(fmaf(ry_first, A.z, fmaf(0.0f, A.w, c)), tl);
                        const bool y0_in = B.x <= q0y && q0y < B.y; // Y_e at the backdrop row
                        const int sigma = (flags & kEdgeSigmaPos) ? 1 : -1, down = (flags >> 3) & 1;
                        const int term = y0_in ? sigma * ((gq0 ? 1 : 0) - down) : 0; // sigma * Y(q0) * (g(q0) - down)
                        const bool up = A.w > 0.0f; // E grows with ry (bx >= 0) and with rx iff nay > 0
                        const bool gmax = accepts(fmaf(r_last, A.z, fmaf(up ? r_last : 0.0f, A.w, c)), tl), gmin = accepts(fmaf(ry_first, A.z, fmaf(up ? 0.0f : r_last, A.w, c)), tl);
                        const bool touch = active && gmax != gmin && B.x <= ty0 + r_last && B.y >= q0y && A.x <= tx0 + r_last && B.z >= tx0;
                        if (flags & kEdgeHull) {
                            hbd += term;
                            hull_touch = hull_touch || touch;
                        } else {
                            bd += term;
                        }
                        mask |= touch ? (1ull << j) : 0ull;
                    }
                    // ---- this chunk's entries; the last chunk's go together with the item's synthetic entries
                    const bool last_chunk = i0 + 64u >= n_edges;
                    const uint32_t abd = (uint32_t)(bd < 0 ? -bd : bd), ahbd = (uint32_t)(hbd < 0 ? -hbd : hbd);
                    uint32_t n_cover = 0, n_bd = 0, n_hbd = 0, cover_key = 0;
                    if (last_chunk && active) {
                        n_cover = (n_hull_chain != 0u && (hbd != 0 || hull_touch)) ? 1u : 0u; // the tile is inside the hull or its boundary crosses it
                        const int cbd = bd > 0 ? 1 : (bd < 0 ? -1 : 0), chbd = hbd > 0 ? 1 : (hbd < 0 ? -1 : 0);
                        cover_key = synth_b + (uint32_t)(cbd + 1) + 3u * (uint32_t)(chbd + 1) + ((n_cover != 0u && hbd != 0 && !hull_touch) ? kCoverHull : 0u);
                        n_bd = n_cover ? (abd ? abd - 1u : 0u) : abd; // the COVER entry carries one unit of either backdrop
                        n_hbd = n_cover ? (ahbd ? ahbd - 1u : 0u) : 0u;
                    }
                    const uint32_t bd_key = synth_a + (bd > 0 ? 0u : 1u), hbd_key = synth_a + (hbd > 0 ? 2u : 3u);
                    const uint32_t n_mine = (uint32_t)__popcll(mask) + n_cover + n_bd + n_hbd;
                    uint32_t pos = 0;
                    if (n_mine) pos = atomicAdd(&r.tile_count[tile], n_mine);
                    uint32_t left = n_mine;
                    for (;;) {
                        const unsigned long long ballot = __ballot(left != 0u);
                        if (!ballot) break;
                        uint32_t key = 0;
                        if (left) {
                            if (mask) {
                                const uint32_t i = i0 + (uint32_t)(__ffsll((long long)mask) - 1);
                                mask &= mask - 1ull;
                                key = i < k.n_fe ? fe_slot0 + i : hull_slot0 + (i - k.n_fe);
                            } else if (n_cover) {
                                n_cover = 0, key = cover_key;
                            } else if (n_bd) {
                                --n_bd, key = bd_key;
                            } else {
                                --n_hbd, key = hbd_key;
                            }
                        }
                        stage_append(st, r, lane, ballot, tile, pos, key);
                        if (left) ++pos, --left;
                    }
                }
            }
        }
        if (hull_as_triangles) { // the hull strip, triangle by triangle, as cover triangles (keys behind the fill chain and the backdrop slots)
            for (uint32_t t0 = 0; t0 + 2u < k.n_hull; t0 += 64u) {
                const uint32_t t = t0 + lane;
                PrimRec rec = {};
                bool drawn = false;
                if (t + 2u < k.n_hull) {
                    drawn = setup_plain_triangle(s, r, ctx, k.cb[6] + t, rec);
                    if (drawn) *reinterpret_cast<PrimRec*>(r.slots + (size_t)(hull_slot0 + 4u * t) * 32u) = rec;
                }
                bin_triangles(st, r, lane, drawn, rec.cov, hull_slot0 + 4u * t, ry_first, r_last);
            }
        }
    }
    CRH_PHASE(5) // pass 3 (edges) / nothing (triangles)
    } // items
    stage_flush(st, r, lane);
}


// ---------------------------------------------------------------------------------------------- k_bin_flat
// The same binning with the lanes packed ACROSS draw items. k_bin_edges gives every item a wavefront per role and each of them walks
// item -> ranges -> vertices -> records -> LDS passes -> returning atomics alone: 20 000 wavefronts of 33 us for the benchmark scene, a
// third to two thirds of their lanes idle (18 triangles, 40 edges per item), and that sum of wavefront lifetimes, not arithmetic, was the
// kernel's time. Here a workgroup (kFlatThreads lanes) takes a batch of up to kFlatThreads / 8 consecutive items at once:
//   0  lane = item: the item's record (ItemCtx, slot ranges, counts) into LDS — ONE round of dependent loads for the whole batch;
//   A  lane = triangle / lane = edge over the batch (prefix sums of the items' counts in LDS, five-step search): set-up records written
//      to the heap, the primitive kept in registers, the item's pixel box and the facing of its hull strip gathered with LDS atomics;
//   B  lane = item: the item's tile rectangle and its share of a pool of per-tile tables in LDS;
//   C  (pass 1) every edge adds its backdrop terms and counts, per tile of its own box, whether it matters there; every triangle
//      counts the tiles it reaches — the triangles use the item's table too, so no primitive needs a global atomic of its own;
//   D  (pass 2) lane = tile of the pool: ONE returning atomic on the tile's global counter reserves the list positions of everything the
//      item has there, the synthetic entries (COVER / backdrop units) are emitted;
//   E  (pass 3) edges and triangles walk again and take their positions from the LDS cursors.
// A hull strip that folds (k_bin_edges) has its triangles binned the old way at the end. What does not fit a batch — more than 256
// triangles or 512 edges in ONE item, a rectangle beyond the pool — is queued for k_bin_edges<S, true>, which has no such limits.
// Slots, records and keys are exactly those of k_bin_edges; the order of a tile's entries in memory differs, the raster kernel sorts.
#ifndef CRH_FLAT_WAVES
#define CRH_FLAT_WAVES 3 // 168 registers: no scratch memory (at 4 waves = 128 registers the triangle set-up spills, and a scratch access is a vector-memory
                         // operation that can only be waited for together with the record stores in flight); three workgroups per CU
#endif
#ifndef CRH_FLAT_ROUNDS
#define CRH_FLAT_ROUNDS 3
#endif
#ifndef CRH_FLAT_STAGE
#define CRH_FLAT_STAGE 128
#endif
#ifndef CRH_FLAT_THREADS
#define CRH_FLAT_THREADS 128 // threads of a workgroup of k_bin_flat: 256 (four wavefronts, a batch of up to 32 items), 128 or 64 (ONE wavefront, a quarter of every table).
                             // Round 5 (tools/r05b_flat_shape.sh): alone the kernel is fastest with 256 (S10k 0.112 ms; 128: 0.124; 64: 0.200) — but it runs in the gap between
                             // two raster kernels, where a workgroup starts as soon as ALL its wavefronts find registers and LDS on one CU, and a two-wave workgroup finds
                             // them earlier behind the draining raster grid: pipelined step S10k 0.3175 -> 0.3013 ms (64: 0.349), glyphs 0.716 -> 0.696 (0.708), S100k 2.07 -> 2.05 (1.94)
#endif
constexpr uint32_t kFlatThreads = CRH_FLAT_THREADS, kFlatWaveCount = kFlatThreads / 64u;
static_assert(kFlatThreads == 64u || kFlatThreads == 128u || kFlatThreads == kFlatThreads, "CRH_FLAT_THREADS");
#ifndef CRH_FLAT_POOL
#define CRH_FLAT_POOL 6 // tile cells of the batch's rectangles per thread
#endif
#ifndef CRH_FLAT_BATCH
#define CRH_FLAT_BATCH (CRH_FLAT_THREADS / 8) // items of a batch at most
#endif
// kFlatItems: entries of the index tables (find_item searches 32); kFlatBatch: items a batch holds (their LDS records)
constexpr uint32_t kFlatItems = 32, kFlatBatch = CRH_FLAT_BATCH, kFlatTris = kFlatThreads, kFlatEdgeRounds = CRH_FLAT_ROUNDS, kFlatEdges = kFlatThreads * kFlatEdgeRounds, kFlatPool = CRH_FLAT_POOL * kFlatThreads, kFlatStage = CRH_FLAT_STAGE; // 256 threads: 37 KB of LDS
static_assert(kFlatBatch >= 1u && kFlatBatch <= kFlatItems, "CRH_FLAT_BATCH");
constexpr uint32_t kFiOpaque = 1u, kFiSkip = 2u, kFiHullTris = 4u, kFiQueue = 8u; // kFiQueue: not binned here but by k_bin_edges (handed on when the item's turn is over)
struct FlatItem {
    ItemCtx ctx;
    uint32_t slot0, fe_slot0, synth_a, hull_slot0, synth_b; // absolute slot numbers of the item's regions
    uint32_t n_tri, n_fe, n_hull, n_hull_chain;
    uint32_t flags;  // kFi*
    int box[4];      // ordered-int images of the float box of everything the item draws: min x, min y, max x, max y (LDS atomics)
    uint32_t faces;  // bit 0: a strip triangle of the hull faces front, bit 1: one faces back
    uint32_t tx_a, ty_a, tx_b, ty_b, nx, n_rect;
};
CRH_D int ordered_int(float f) { // monotone float -> int (finite values): atomicMin / atomicMax on LDS integers
    const int i = __float_as_int(f);
    return i ^ ((i >> 31) & 0x7FFFFFFF);
}
CRH_D float ordered_float(int i) { return __int_as_float(i ^ ((i >> 31) & 0x7FFFFFFF)); }
CRH_D uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}
// the largest j < 32 with begin[j] <= x (begin is non-decreasing, entries behind the batch hold 0xFFFFFFFF)
CRH_D uint32_t find_item(const uint32_t* begin, uint32_t x) {
    uint32_t j = 0;
#pragma unroll
    for (uint32_t step = 16; step > 0; step >>= 1)
        if (begin[j + step] <= x) j += step;
    return j;
}
// A workgroup barrier that orders LDS accesses only. __syncthreads() also waits for the wavefront's global stores (s_waitcnt vmcnt(0):
// gfx950 counts loads and stores with one counter), and the record stores of phase A — 46 MB per frame of the benchmark scene — took
// 50 us to drain at the first barrier behind them, every workgroup waiting at once. Nothing here reads global memory another wavefront
// of the workgroup wrote, so the stores may stay in flight across the barriers.
CRH_D void lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
// A boundary edge between the passes of k_bin_flat: five registers instead of BinEdge's fourteen (the rest is recomputed with the very
// expressions load_edge used, so the values are the same bits).
struct PackedEdge {
    float lo_x, lo_y, hi_x, hi_y;
    uint32_t flags; // bit 0 valid, 1 top-left, 2 hull, 3 sigma > 0, 4 down; bits 8-12: the item of the batch (in the LDS table of k_bin_flat)
};
CRH_D PackedEdge pack_edge(const BinEdge& e) {
    return PackedEdge{e.lo_x, e.lo_y, e.hi_x, e.hi_y, (e.valid ? 1u : 0u) | (e.tl ? 2u : 0u) | (e.hull ? 4u : 0u) | (e.sigma > 0 ? 8u : 0u) | (e.down ? 16u : 0u)};
}
CRH_D BinEdge unpack_edge(const PackedEdge& p) {
    BinEdge e = {};
    e.lo_x = p.lo_x, e.lo_y = p.lo_y, e.hi_x = p.hi_x, e.hi_y = p.hi_y;
    e.bx = p.hi_x - p.lo_x;
    e.nay = -(p.hi_y - p.lo_y);
    e.ymin = fminf(p.lo_y, p.hi_y), e.ymax = fmaxf(p.lo_y, p.hi_y);
    e.valid = (p.flags & 1u) != 0u, e.tl = (p.flags >> 1) & 1u, e.hull = (p.flags >> 2) & 1u, e.sigma = (p.flags & 8u) ? 1 : -1, e.down = (int)((p.flags >> 4) & 1u);
    return e;
}
struct FlatTri { // a set-up triangle between the passes: its tile test and tile box
    TileTest test;
    uint32_t bx0, bx1, by0, nt, key, item;
};
#ifdef CRH_ABLATE // tools/bin_phases.py: cycles wavefront 0 of every workgroup of k_bin_flat spends per phase (summed in overflow[80 ...])
#define CRH_FLAT_PHASE(k)                                                                                              \
    if ((r.debug & 65536u) && tid == 0u) {                                                                             \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                  \
        atomicAdd(reinterpret_cast<unsigned long long*>(r.overflow + 80) + (k), now_ - phase_t);                       \
        phase_t = __builtin_amdgcn_s_memtime();                                                                        \
    }                                                                                                                  \
    if ((r.debug >> 24) == (k) + 1u) return; /* tools/ablate_flat.sh: the kernel up to and including phase k */
#else
#define CRH_FLAT_PHASE(k)
#endif
// what a batch holds follows from the workgroup's lanes (the tables of k_bin_flat<S, THREADS>; the host cuts its runs by the same numbers)
struct FlatShape {
    uint32_t threads, batch, tris, edges, pool;
};
constexpr FlatShape flat_shape(uint32_t threads) { return FlatShape{threads, threads / 8u, threads, threads * CRH_FLAT_ROUNDS, CRH_FLAT_POOL * threads}; }
template <int S, uint32_t THREADS>
#ifdef CRH_FLAT_VGPRS
#define CRH_FLAT_BUDGET __attribute__((amdgpu_num_vgpr(CRH_FLAT_VGPRS)))
#else
#define CRH_FLAT_BUDGET
#endif
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(CRH_FLAT_WAVES))) CRH_FLAT_BUDGET void k_bin_flat(SceneDev s, RasterParams r, uint32_t items_per_group) {
    // (the namespace's constants of these names describe the default shape; inside the kernel they are this instantiation's)
    constexpr uint32_t kFlatThreads = THREADS, kFlatWaveCount = THREADS / 64u, kFlatBatch = THREADS / 8u, kFlatTris = THREADS, kFlatEdges = THREADS * kFlatEdgeRounds, kFlatPool = CRH_FLAT_POOL * THREADS;
    static_assert(kFlatBatch >= 1u && kFlatBatch <= kFlatItems, "a batch's index tables hold 32 items");
    __shared__ uint32_t stage_tile[kFlatWaveCount][kFlatStage], stage_pos[kFlatWaveCount][kFlatStage], stage_key[kFlatWaveCount][kFlatStage];
    __shared__ FlatItem items[kFlatBatch];
    __shared__ uint32_t tri_begin[kFlatItems + 1], edge_begin[kFlatItems + 1], pool_begin[kFlatItems + 1];
    __shared__ int pool_bd[kFlatPool], pool_hbd[kFlatPool];
    __shared__ uint32_t pool_cursor[kFlatPool]; // pass 1: the entries the item's edges and triangles have in the tile (bits 0-19; bits 20-31: the hull edges among them), then the next list position
    __shared__ uint32_t batch[6];               // items in the batch, its triangles, its edges, tiles of its pool, items of the batch that are binned in this turn, (edge, tile row) pairs
    __shared__ uint32_t wave_opaque[kFlatWaveCount];         // opaque whole-tile covers every wavefront found in pass 2
    __shared__ uint32_t wave_entries[2u * kFlatWaveCount];        // entries every wavefront appends in pass 3 ([0..3]) and in pass 2 ([4..7]); then where its share of the pair stream begins
    __shared__ PackedEdge edge_table[kFlatEdges]; // the batch's boundary edges: the walks are balanced over (edge, tile row) pairs, whoever loaded the edge
    __shared__ uint32_t row_begin[kFlatEdges + 1]; // exclusive prefix of the tile rows every edge walks
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    Stage st = {stage_tile[wave], stage_pos[wave], stage_key[wave], 0u, 0u, kFlatStage, 0xFFFFFFFFu};
    uint32_t turn = 0; // batches this workgroup has binned (the pair sub-stream of a batch follows from it)
    const float ry_first = S == 1 ? 0.5f : 0.125f, r_last = (float)(kTile - 1) + (S == 1 ? 0.5f : 0.875f); // extreme sample offsets inside a tile
    const float W = (float)r.width, H = (float)r.height;
    const uint32_t first_item = r.bin_batches ? r.bin_batches[2u * blockIdx.x] : blockIdx.x * items_per_group;
    const uint32_t last_item = r.bin_batches ? r.bin_batches[2u * blockIdx.x + 1u] : min(r.n_items, first_item + items_per_group);
    const bool all_queued = (r.debug & (1u | 2u | 4194304u)) != 0u; // debug bits 0, 1 (tests of k_bin_edges' own code paths), 22: every item takes that kernel
#ifdef CRH_ABLATE
    unsigned long long phase_t = __builtin_amdgcn_s_memtime();
    const unsigned long long born_t = phase_t; // (tools/bin_phases.py: the longest-lived workgroup, overflow[120..121], and the sum, [122..123])
    uint32_t dump_items = 0, dump_tris = 0, dump_edges = 0, dump_pool = 0, dump_work = 0, dump_walk = 0; // (... and what every workgroup held, CRH_BIN_DUMP)
#endif
    for (uint32_t next = first_item; next < last_item;) {
        lds_barrier(); // (the previous batch's readers of the LDS records are through)
        // ---------------- 0: the records of the next items
        const uint32_t n_cand = min(kFlatBatch, last_item - next);
        if (tid < n_cand) {
            const uint32_t item = next + tid;
            const DrawItem it = item_of(r, item);
            const bool elsewhere = r.item_elsewhere && r.item_elsewhere[item] != 0u; // (a pass with a slab: the item has no tile row in it — neither binned here nor queued)
            const ItemSlots k = item_slots(s, it);
            FlatItem& fi = items[tid];
            fi.ctx = item_ctx(s, r, it, k.cb);
            const uint32_t slot0 = r.slot_begin[item];
            fi.slot0 = slot0, fi.fe_slot0 = slot0 + k.fe0, fi.synth_a = slot0 + k.synth_a, fi.hull_slot0 = slot0 + k.hull0, fi.synth_b = slot0 + k.synth_b;
            fi.n_tri = k.n_tri, fi.n_fe = k.n_fe, fi.n_hull = k.n_hull, fi.n_hull_chain = k.n_hull;
            const float* c = fi.ctx.col;
            const bool opaque = c[3] == 1.0f && is_finite(c[0]) && is_finite(c[1]) && is_finite(c[2]) && r.occlude != 0u && (r.debug & 32768u) == 0u;
            const bool oversize = k.n_tri > kFlatTris || k.n_fe + k.n_hull > kFlatEdges || all_queued || slot0 + k.total > r.slot_capacity;
            fi.flags = (opaque ? kFiOpaque : 0u) | ((oversize || elsewhere) ? kFiSkip : 0u) | ((oversize && !elsewhere && slot0 + k.total <= r.slot_capacity) ? kFiQueue : 0u);
            fi.box[0] = fi.box[1] = 0x7FFFFFFF, fi.box[2] = fi.box[3] = (int)0x80000000;
            fi.faces = 0u;
            fi.n_rect = 0u;
        }
        lds_barrier();
        if (wave == 0u) { // the batch: the longest run of items whose triangles and edges fit the lanes (an oversize item counts as empty)
            const FlatItem& mine_ = items[min(lane, kFlatBatch - 1u)];
            const bool real = lane < n_cand && lane < kFlatBatch, counted = real && (mine_.flags & kFiSkip) == 0u;
            const uint32_t nt = counted ? mine_.n_tri : 0u, ne = counted ? mine_.n_fe + mine_.n_hull : 0u;
            const uint32_t pt = wave_inclusive_scan(nt, lane), pe = wave_inclusive_scan(ne, lane);
            const unsigned long long fits = __ballot(real && pt <= kFlatTris && pe <= kFlatEdges);
            const uint32_t n_batch = (uint32_t)__builtin_ctzll(~fits); // leading lanes that fit (>= 1: one item alone always does)
            if (lane <= kFlatItems) {
                tri_begin[lane] = lane <= n_batch ? pt - nt : 0xFFFFFFFFu;
                edge_begin[lane] = lane <= n_batch ? pe - ne : 0xFFFFFFFFu;
            }
            const uint32_t total_t = (uint32_t)__shfl((int)pt, (int)n_batch - 1, 64), total_e = (uint32_t)__shfl((int)pe, (int)n_batch - 1, 64);
            if (lane == 0u) {
                batch[0] = n_batch, batch[1] = total_t, batch[2] = total_e;
                tri_begin[n_batch] = 0xFFFFFFFFu, edge_begin[n_batch] = 0xFFFFFFFFu; // (searches stop in front of it; the totals are in batch[])
            }
        }
        lds_barrier();
        const uint32_t n_batch = batch[0], n_tris = batch[1], n_edges = batch[2];
        CRH_FLAT_PHASE(0) // item records + batch
        // ---------------- A: set-up, lane = triangle and lane = edge (two edges per lane), records to the heap, boxes and hull facing to LDS.
        // All loads first, all stores last: gfx950 counts vector loads and stores with ONE counter, in order — a load issued behind the 128-byte
        // record stores can only be waited for together with them, and those take tens of microseconds to drain when every workgroup
        // writes its records at once.
        PackedEdge kept[kFlatEdgeRounds];
        float strip_det[kFlatEdgeRounds];
        uint32_t edge_item[kFlatEdgeRounds], edge_key[kFlatEdgeRounds];
#pragma unroll
        for (int k = 0; k < (int)kFlatEdgeRounds; ++k) {
            const uint32_t e = tid + kFlatThreads * (uint32_t)k;
            kept[k] = PackedEdge{0.0f, 0.0f, 0.0f, 0.0f, 0u};
            strip_det[k] = 0.0f;
            edge_item[k] = 0u, edge_key[k] = 0u;
            if (e < n_edges) {
                const uint32_t j = find_item(edge_begin, e), i = e - edge_begin[j];
                const FlatItem& fi = items[j];
                const BinEdge loaded = load_edge(s, r, fi.ctx, fi.n_fe, fi.n_hull, i);
                kept[k] = pack_edge(loaded), strip_det[k] = loaded.strip_det;
                edge_item[k] = j;
                edge_key[k] = i < fi.n_fe ? fi.fe_slot0 + i : fi.hull_slot0 + (i - fi.n_fe);
            }
        }
        CRH_FLAT_PHASE(1) // edge set-up
        FlatTri tri;
        tri.nt = 0u, tri.item = 0u, tri.key = 0u, tri.bx0 = tri.bx1 = tri.by0 = 0u;
        tri.test = TileTest{};
        PrimRec rec = {};
        bool tri_drawn = false;
        if (tid < n_tris) {
            const uint32_t j = find_item(tri_begin, tid), t = tid - tri_begin[j];
            const FlatItem& fi = items[j];
            const ItemCtx& ctx = fi.ctx;
            const uint32_t c = t < ctx.cb[1] ? t : t - ctx.cb[1] + ctx.cb[2]; // the Shape's candidate numbering without the solid strips
            tri_drawn = setup_plain_triangle(s, r, ctx, c, rec);
            tri.key = fi.slot0 + 4u * t, tri.item = j;
        }
        // ---- stores and LDS
#ifdef CRH_ABLATE
        const bool store_records = (r.debug & 8388608u) == 0u; // tools/ablate_flat.sh: no record stores
#else
        constexpr bool store_records = true;
#endif
        if (tri_drawn) {
            if (store_records) *reinterpret_cast<PrimRec*>(r.slots + (size_t)tri.key * 32u) = rec;
            tri.test.set(rec.cov, ry_first, r_last);
            tri.bx0 = rec.cov.box.x / kTile, tri.bx1 = rec.cov.box.y / kTile, tri.by0 = rec.cov.box.z / kTile;
            tri.nt = (tri.bx1 - tri.bx0 + 1u) * (rec.cov.box.w / kTile - tri.by0 + 1u);
            FlatItem& fi = items[tri.item];
            atomicMin(&fi.box[0], ordered_int((float)rec.cov.box.x)), atomicMin(&fi.box[1], ordered_int((float)rec.cov.box.z));
            atomicMax(&fi.box[2], ordered_int((float)rec.cov.box.y)), atomicMax(&fi.box[3], ordered_int((float)rec.cov.box.w));
        }
#pragma unroll
        for (int k = 0; k < (int)kFlatEdgeRounds; ++k) {
            if (tid + kFlatThreads * (uint32_t)k < n_edges) {
                FlatItem& fi = items[edge_item[k]];
                const BinEdge e = unpack_edge(kept[k]);
                if (e.valid) {
                    EdgeRec er;
                    er.flags = (EK_EDGE << 4) | (e.tl ? kEdgeTl : 0u) | (e.sigma > 0 ? kEdgeSigmaPos : 0u) | (e.hull ? kEdgeHull : 0u), er.pad0 = 0u;
                    er.lo_x = e.lo_x, er.lo_y = e.lo_y, er.hi_x = e.hi_x, er.hi_y = e.hi_y, er.bx = e.bx, er.nay = e.nay;
                    if (store_records) *reinterpret_cast<EdgeRec*>(r.slots + (size_t)edge_key[k] * 32u) = er;
                    atomicMin(&fi.box[0], ordered_int(e.lo_x)), atomicMin(&fi.box[1], ordered_int(e.ymin));
                    atomicMax(&fi.box[2], ordered_int(e.hi_x)), atomicMax(&fi.box[3], ordered_int(e.ymax));
                }
                // (a strip triangle is judged at its first position whether or not that position's chain edge is valid)
                if (strip_det[k] != 0.0f) atomicOr(&fi.faces, strip_det[k] < 0.0f ? 1u : 2u);
            }
        }
        CRH_FLAT_PHASE(2) // triangle set-up, record stores
        for (uint32_t q = tid; q < 31u * n_batch; q += kFlatThreads) { // the items' 4 backdrop + 27 COVER slots (the COVER ones carry the premultiplied source colour, shaders.wgsl:304-309)
            const uint32_t j = q / 31u, l = q - 31u * j;
            const FlatItem& fi = items[j];
            if (fi.flags & kFiSkip) continue;
            SynthRec sr = {};
            sr.flags = (EK_SYNTH << 4) | (l << 8);
            sr.first_slot = fi.slot0, sr.synth_a = fi.synth_a;
            const float* c = fi.ctx.col;
            if (l >= 4u) sr.r = c[0] * c[3], sr.g = c[1] * c[3], sr.b = c[2] * c[3], sr.a = c[3];
            if (store_records) *reinterpret_cast<SynthRec*>(r.slots + (size_t)(l < 4u ? fi.synth_a + l : fi.synth_b + l - 4u) * 32u) = sr;
        }
        lds_barrier();
        CRH_FLAT_PHASE(3) // synthetic records + barrier
        // ---------------- B: lane = item: tile rectangle, its tables in the pool; does the hull strip fold?
        if (wave == 0u) {
            uint32_t n_rect = 0;
            const bool mine = lane < n_batch;
            FlatItem& fi = items[min(lane, kFlatBatch - 1u)];
            if (mine && (fi.flags & kFiSkip) == 0u) {
                const float minx = ordered_float(fi.box[0]), miny = ordered_float(fi.box[1]), maxx = ordered_float(fi.box[2]), maxy = ordered_float(fi.box[3]);
                if (fi.box[0] <= fi.box[2]) { // something is drawn
                    const int px0 = (int)floorf(fminf(fmaxf(minx, 0.0f), W)), px1 = (int)floorf(fmaxf(fminf(maxx, W - 1.0f), -1.0f));
                    const int py0 = (int)floorf(fminf(fmaxf(miny, 0.0f), H)), py1 = (int)floorf(fmaxf(fminf(maxy, H - 1.0f), -1.0f));
                    // (the tile rows of the pass' slab only, crh_frame_set_tile_rows: a tile row's backdrops, counts and entries are its own; an item
                    // without a row in the slab has no rectangle and is not binned)
                    const uint32_t row_a = max((uint32_t)max(py0, 0) / kTile, r.slab_ty0), row_end = min((uint32_t)max(py1, 0) / kTile + 1u, r.slab_ty1);
                    if (px0 <= px1 && py0 <= py1 && row_a < row_end) {
                        fi.tx_a = (uint32_t)px0 / kTile, fi.tx_b = (uint32_t)px1 / kTile, fi.ty_a = row_a, fi.ty_b = row_end - 1u;
                        fi.nx = fi.tx_b - fi.tx_a + 1u;
                        n_rect = fi.nx * (fi.ty_b - fi.ty_a + 1u);
                    }
                }
                if (fi.n_hull != 0u && (fi.faces == 3u || (r.debug & 4u) != 0u)) fi.flags |= kFiHullTris, fi.n_hull_chain = 0u; // debug bit 2 (tests): always
            }
            // (a verified pass: what the item takes of a batch — the host sizes later passes' batches with it)
            if (r.item_cost && mine) {
                const bool skipped = (fi.flags & kFiSkip) != 0u, too_wide = n_rect > kFlatPool; // (too wide: queued once it is the first of a batch — a run of its own)
                r.item_cost[2u * (next + lane)] = skipped ? 0u : (too_wide ? 0xFFFFFFFFu : n_rect);
                r.item_cost[2u * (next + lane) + 1u] = (skipped || too_wide) ? 0x80000000u : (fi.n_tri | ((fi.n_fe + fi.n_hull) << 9) | ((fi.flags & kFiHullTris) ? 1u << 29 : 0u));
            }
            // Pool shares in item order. Items from the first one that does not fit are left to the workgroup's next turn (their records are
            // written again then); an item that does not fit the pool even alone goes to k_bin_edges.
            uint32_t end = wave_inclusive_scan(n_rect, lane);
            uint32_t n_fit = (uint32_t)__builtin_ctzll(~__ballot(mine && end <= kFlatPool)); // leading items that fit
            if (n_fit == 0u) {
                if (lane == 0u) fi.flags |= kFiSkip | kFiQueue;
                n_fit = 1u;
                if (lane == 0u) n_rect = 0u, end = 0u;
            }
            const uint32_t pool_total = (uint32_t)__shfl((int)end, (int)n_fit - 1, 64); // (every lane takes part in the shuffle)
            if (lane >= n_fit) n_rect = 0u, end = pool_total;
            if (mine) fi.n_rect = n_rect;
            if (lane <= kFlatItems) pool_begin[lane] = lane < n_batch ? end - n_rect : (lane == n_batch ? pool_total : 0xFFFFFFFFu);
            if (lane == 0u) batch[3] = pool_total, batch[4] = n_fit;
            // (handed on exactly once: a candidate that is not part of this turn is looked at again in the next one)
            if (lane < n_fit && (fi.flags & kFiQueue) != 0u) r.bin_queue[atomicAdd(&r.overflow[6], 1u)] = next + lane;
        }
        lds_barrier();
        const uint32_t n_pool = batch[3], n_turn = batch[4]; // (items n_turn .. n_batch - 1 are set up but not binned: n_rect == 0)
        for (uint32_t q = tid; q < n_pool; q += kFlatThreads) pool_bd[q] = 0, pool_hbd[q] = 0, pool_cursor[q] = 0u;
        lds_barrier();
        CRH_FLAT_PHASE(4) // rectangles, pool cleared
        // ---------------- the edges' walks are balanced over (edge, tile row) pairs. A lane that walked ITS edge kept its wavefront in the loop
        // for as long as the longest edge among 64 took (a hull edge across a 256-pixel Shape reaches 40 tiles, a polygon edge 3): nine
        // tenths of the lane-steps of the walks were idle. So every edge goes to an LDS table with the number of tile rows of its own box
        // (clamped to its item's rectangle), the counts are summed, and lane w of a pass takes pair w: the edge by a ten-step search in
        // the prefix sums, then the row. Within its row an edge visits the columns its LINE can reach — a float estimate with a margin
        // (2 px + 4e-6 of the largest coordinate: the distance at which f32 rounding can still flip an edge function is about 1e-7 of it),
        // clamped to the box; whether the edge matters in a tile is decided by the exact test as ever. Coordinates beyond 1e6 and horizontal
        // edges take the whole width of the box.
        struct EdgeBox {
            int bx0, bx1, by0, by1;
        };
        auto edge_box = [&](const BinEdge& e, const FlatItem& fi) {
            const int x_lo = (int)ceilf((e.lo_x - r_last) * (1.0f / (float)kTile) - 0.01f), x_hi = (int)floorf(e.hi_x * (1.0f / (float)kTile) + 0.01f);
            const int y_lo = (int)ceilf((e.ymin - r_last) * (1.0f / (float)kTile) - 0.01f), y_hi = (int)floorf((e.ymax - ry_first) * (1.0f / (float)kTile) + 0.01f);
            return EdgeBox{max(x_lo, (int)fi.tx_a), min(x_hi, (int)fi.tx_b), max(y_lo, (int)fi.ty_a), min(y_hi, (int)fi.ty_b)};
        };
#pragma unroll
        for (int k = 0; k < (int)kFlatEdgeRounds; ++k) {
            const uint32_t e = tid + kFlatThreads * (uint32_t)k;
            if (e < n_edges) {
                PackedEdge pe = kept[k];
                const FlatItem& fi = items[edge_item[k]];
                // the edge takes part: valid, its item is drawn here, and — a hull edge — its hull is binned as a chain
                bool live = (pe.flags & 1u) != 0u && fi.n_rect != 0u && !((pe.flags & 4u) != 0u && (fi.flags & kFiHullTris) != 0u);
                uint32_t rows = 0;
                if (live) {
                    const EdgeBox box = edge_box(unpack_edge(pe), fi);
                    live = box.by0 <= box.by1; // (an edge left of the rectangle — a Shape that sticks out of the frame — has no tile to walk but still crosses the rows' backdrop lines)
                    if (live) rows = (uint32_t)(box.by1 - box.by0 + 1);
                }
                pe.flags = (pe.flags & (live ? ~0u : ~1u)) | (edge_item[k] << 8);
                edge_table[e] = pe;
                row_begin[e] = rows;
            }
        }
        lds_barrier();
        if (wave == 0u) { // exclusive prefix of the row counts: twelve consecutive edges per lane
            const uint32_t per = (n_edges + 63u) / 64u, first = lane * per, last = min(n_edges, first + per);
            uint32_t sum = 0;
            for (uint32_t e = first; e < last; ++e) sum += row_begin[e];
            uint32_t run = wave_inclusive_scan(sum, lane) - sum;
            for (uint32_t e = first; e < last; ++e) {
                const uint32_t n = row_begin[e];
                row_begin[e] = run;
                run += n;
            }
            const uint32_t total = (uint32_t)__shfl((int)run, 63, 64);
            if (lane == 0u) row_begin[n_edges] = total, batch[5] = total;
        }
        lds_barrier();
        const uint32_t n_work = batch[5];
        // visit(active, edge, its item, the tile row, the columns x0 .. x1 of that row worth a test, the edge's slot number)
        auto for_edge_rows = [&](auto&& visit) {
            for (uint32_t w0 = 0; w0 < n_work; w0 += kFlatThreads) {
                const uint32_t w = w0 + tid;
                const bool active = w < n_work;
                uint32_t ei = 0; // the largest edge index with row_begin <= w (edges without rows share their successor's entry and are passed over)
#ifdef CRH_ABLATE
                if (r.debug & 4096u) ei = w % max(n_edges, 1u); else // what does the search cost? (the wrong edges: timing only)
#endif
#pragma unroll
                for (uint32_t step = kFlatEdges >= 512u ? 512u : (kFlatEdges >= 256u ? 256u : 128u); step > 0u; step >>= 1)
                    if (ei + step < n_edges && row_begin[ei + step] <= w) ei += step;
                const PackedEdge pe = edge_table[active ? ei : 0u];
                const BinEdge e = unpack_edge(pe);
                const uint32_t j = (pe.flags >> 8) & 31u;
                const FlatItem& fi = items[j];
                const EdgeBox box = edge_box(e, fi);
                const int ty = box.by0 + (int)(w - row_begin[ei]);
                int x0 = box.bx0, x1 = box.bx1;
                const float dy = e.hi_y - e.lo_y, largest = fmaxf(fmaxf(fabsf(e.lo_x), fabsf(e.hi_x)), fmaxf(fabsf(e.lo_y), fabsf(e.hi_y)));
                if (dy != 0.0f && largest < 1.0e6f) {
                    const float xs = (e.hi_x - e.lo_x) / dy, pad = 2.0f + 4.0e-6f * largest;
                    const float ty0 = (float)(ty * kTile), ya = ty0 + ry_first - pad, yb = ty0 + r_last + pad;
                    const float xa = e.lo_x + (ya - e.lo_y) * xs, xb = e.lo_x + (yb - e.lo_y) * xs;
                    const float lo = fminf(xa, xb) - pad - r_last, hi = fmaxf(xa, xb) + pad;
                    // (clamped as floats first: the estimates of a steep row can be far outside anything an int holds)
                    x0 = (int)floorf(fmaxf(lo * (1.0f / (float)kTile), (float)box.bx0));
                    x1 = (int)floorf(fminf(hi * (1.0f / (float)kTile), (float)box.bx1));
                }
                const uint32_t i = ei - edge_begin[j];
                visit(active, e, j, fi, ty, x0, x1, i < fi.n_fe ? fi.fe_slot0 + i : fi.hull_slot0 + (i - fi.n_fe));
            }
        };
        // does the edge matter in tile (tx, ty)? (the exact test of k_bin_edges)
        auto matters = [&](const BinEdge& e, int tx, int ty) {
            const bool up = e.nay > 0.0f; // E grows with ry (bx >= 0) and with rx iff nay > 0
            const float tx0 = (float)(tx * kTile), ty0 = (float)(ty * kTile), q0y = ty0 + ry_first;
            const float c = e.bx * (ty0 - e.lo_y) + e.nay * (tx0 - e.lo_x);
            const bool gmax = accepts(fmaf(r_last, e.bx, fmaf(up ? r_last : 0.0f, e.nay, c)), e.tl), gmin = accepts(fmaf(ry_first, e.bx, fmaf(up ? 0.0f : r_last, e.nay, c)), e.tl);
            return gmax != gmin && e.ymin <= ty0 + r_last && e.ymax >= q0y && e.lo_x <= tx0 + r_last && e.hi_x >= tx0;
        };
        auto walk_tri = [&](const FlatTri& t, bool live, auto&& visit) { // (the triangle's tile box, cut to the rows of its item's rectangle: the slab of a tile split)
            const FlatItem& of = items[t.item];
            const uint32_t nxt = t.bx1 - t.bx0 + 1u, row_a = max(t.by0, of.ty_a), row_end = min(t.by0 + t.nt / nxt, of.ty_b + 1u);
            const uint32_t nt = (live && row_a < row_end) ? nxt * (row_end - row_a) : 0u, longest = wave_max_u32(nt);
            uint32_t tx = t.bx0, ty = row_a;
            for (uint32_t w = 0; w < longest; ++w) {
                visit(w < nt && t.test.hit(tx, ty), tx, ty);
                if (++tx > t.bx1) tx = t.bx0, ++ty;
            }
        };
        // ---------------- C: pass 1
        uint32_t my_entries = 0; // what this lane will append in pass 3
        for_edge_rows([&](bool active, const BinEdge& e, uint32_t j, const FlatItem& fi, int ty, int x0, int x1, uint32_t) {
            const uint32_t base = pool_begin[j] + ((uint32_t)ty - fi.ty_a) * fi.nx, nx = fi.nx, tx_a = fi.tx_a;
            // Along a tile row the backdrop term of an edge that crosses the row's line is 0 left of the edge and +-1 from some column on (the
            // edge function is monotone in x under fmaf, so the very predicate a tile would evaluate switches once): that column is found
            // by bisection — five exact evaluations instead of one per column — and the unit goes there alone; pass 2 sums along the row.
            const float ty0 = (float)(ty * kTile), q0y = ty0 + ry_first;
            const bool crosses = active && e.ymin <= q0y && q0y < e.ymax; // Y_e at the backdrop row
            const uint32_t widest = wave_max_u32(crosses ? nx : 0u);
            if (widest) {
                uint32_t lo = 0, hi = crosses ? nx : 0u; // the first column with a non-zero term lies in [lo, hi]
                for (uint32_t span = widest; span > 0u; span >>= 1) { // (ceil(log2(widest + 1)) steps settle every lane)
                    const uint32_t mid = (lo + hi) >> 1;
                    const float tx0 = (float)((tx_a + mid) * kTile);
                    const float c = e.bx * (ty0 - e.lo_y) + e.nay * (tx0 - e.lo_x);
                    const bool gq0 = accepts(fmaf(ry_first, e.bx, fmaf(0.0f, e.nay, c)), e.tl);
                    const bool nonzero = ((gq0 ? 1 : 0) - e.down) != 0; // sigma * Y(q0) * (g(q0) - down)
                    if (lo < hi) {
                        if (nonzero) hi = mid; else lo = mid + 1u;
                    }
                }
                if (crosses && lo < nx) atomicAdd(&(e.hull ? pool_hbd : pool_bd)[base + lo], e.down ? -e.sigma : e.sigma);
            }
            const uint32_t cols = wave_max_u32(active && x0 <= x1 ? (uint32_t)(x1 - x0 + 1) : 0u);
            for (uint32_t c = 0; c < cols; ++c) {
                const int tx = x0 + (int)c;
                if (active && tx <= x1 && matters(e, tx, ty)) atomicAdd(&pool_cursor[base + ((uint32_t)tx - tx_a)], e.hull ? 0x00100001u : 1u), ++my_entries;
            }
        });
        if (n_tris) {
            const FlatItem& fi = items[tri.item];
            const bool live = tri.nt != 0u && fi.n_rect != 0u;
            const uint32_t base = pool_begin[tri.item], nx = fi.nx, tx_a = fi.tx_a, ty_a = fi.ty_a;
            walk_tri(tri, live, [&](bool hit, uint32_t tx, uint32_t ty) {
                if (hit) atomicAdd(&pool_cursor[base + (ty - ty_a) * nx + (tx - tx_a)], 1u), ++my_entries;
            });
        }
        lds_barrier();
        CRH_FLAT_PHASE(5) // pass 1
        // ---------------- D: pass 2, lane = tile of the pool (the COVER entry carries one unit of either backdrop).
        // Pass 1 left every edge's backdrop unit at the first column it applies to: summed along the rows first, in place (the lane of a row's
        // first tile walks the row). 2a: what the item has in the tile — COVER entry, backdrop units — and ONE returning atomic on the tile's
        // global counter for all of it plus the edges' and triangles' entries; the atomics of all the lane's tiles are in flight together,
        // and the tile's verdict replaces its backdrops in the LDS tables. Then the workgroup reserves its range of the pair stream with one
        // atomic (every wavefront knows what it will append), and 2b emits the synthetic entries.
        constexpr uint32_t kChunks = kFlatPool / kFlatThreads;
#pragma unroll 1
        for (uint32_t ch = 0; ch * kFlatThreads < n_pool; ++ch) {
            const uint32_t p = ch * kFlatThreads + tid;
            if (p < n_pool) {
                const uint32_t j = find_item(pool_begin, p);
                const uint32_t q = p - pool_begin[j], nx = items[j].nx;
                if (q % nx == 0u)
                    for (uint32_t c = 1; c < nx; ++c) pool_bd[p + c] += pool_bd[p + c - 1u], pool_hbd[p + c] += pool_hbd[p + c - 1u];
            }
        }
        lds_barrier();
        uint32_t my_synth = 0, my_opaque = 0;
        {
            uint32_t reserved[kChunks], lefts[kChunks];
#pragma unroll
            for (uint32_t ch = 0; ch < kChunks; ++ch) { // 2a
                reserved[ch] = 0u, lefts[ch] = 0u;
                const uint32_t p = ch * kFlatThreads + tid;
                if (p >= n_pool) continue;
                const uint32_t j = find_item(pool_begin, p);
                const FlatItem& fi = items[j];
                const uint32_t q = p - pool_begin[j], qy = q / fi.nx, qx = q - qy * fi.nx;
                const uint32_t tile = (fi.ty_a + qy) * r.tiles_x + fi.tx_a + qx;
                const int bd = pool_bd[p], hbd = pool_hbd[p];
                const uint32_t counted = pool_cursor[p];
                const uint32_t n_touching = counted & 0x000FFFFFu; // entries of the item's edges and triangles in this tile
                const bool hull_touch = (counted >> 20) != 0u;
                const uint32_t abd = (uint32_t)(bd < 0 ? -bd : bd), ahbd = (uint32_t)(hbd < 0 ? -hbd : hbd);
                const bool cover = fi.n_hull_chain != 0u && (hbd != 0 || hull_touch); // the tile is inside the hull or its boundary crosses it
                const int cbd = bd > 0 ? 1 : (bd < 0 ? -1 : 0), chbd = hbd > 0 ? 1 : (hbd < 0 ? -1 : 0);
                const bool hull_over_tile = cover && hbd != 0 && !hull_touch;
                const bool replaces_tile = hull_over_tile && (fi.flags & kFiOpaque) != 0u && n_touching == 0u && (bd & (int)r.winding_mask) != 0;
                const uint32_t cover_key = fi.synth_b + (uint32_t)(cbd + 1) + 3u * (uint32_t)(chbd + 1) + (replaces_tile ? kCoverOpaque : (hull_over_tile ? kCoverHull : 0u));
                const uint32_t n_bd = cover ? (abd ? abd - 1u : 0u) : abd, n_hbd = cover ? (ahbd ? ahbd - 1u : 0u) : 0u;
                const uint32_t left = (cover ? 1u : 0u) + n_bd + n_hbd;
                lefts[ch] = left;
                my_synth += left;
                my_opaque += replaces_tile ? 1u : 0u;
                if (left + n_touching) reserved[ch] = atomicAdd(&r.tile_count[tile], left + n_touching);
                // the verdict, for 2b: left (bits 0-11), backdrop units (12-23), COVER entry (24), bd > 0 (25), hbd > 0 (26) | the COVER key
                pool_bd[p] = (int)(left | (n_bd << 12) | (cover ? 1u << 24 : 0u) | (bd > 0 ? 1u << 25 : 0u) | (hbd > 0 ? 1u << 26 : 0u));
                pool_hbd[p] = (int)cover_key;
            }
#pragma unroll
            for (uint32_t ch = 0; ch < kChunks; ++ch)
                if (ch * kFlatThreads + tid < n_pool) pool_cursor[ch * kFlatThreads + tid] = reserved[ch] + lefts[ch]; // where the edges' and triangles' entries go
        }
        {
            uint32_t a = my_entries, b = my_synth | (my_opaque << 22); // (6 cells per lane, < 4096 entries each: a wavefront's sums stay below 2^22 and 2^10)
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) a += (uint32_t)__shfl_xor((int)a, d, 64), b += (uint32_t)__shfl_xor((int)b, d, 64);
            if (lane == 0u) wave_entries[wave] = a, wave_entries[kFlatWaveCount + wave] = b & 0x003FFFFFu, wave_opaque[wave] = b >> 22;
        }
        lds_barrier();
        if (tid == 0u) { // the workgroup's range of the pair stream: wavefront w's pass-2 entries, then its pass-3 entries
            uint32_t total = 0;
            for (uint32_t w = 0; w < 2u * kFlatWaveCount; ++w) total += wave_entries[w];
            uint32_t opaque = 0;
            for (uint32_t w = 0; w < kFlatWaveCount; ++w) opaque += wave_opaque[w];
            if (opaque) atomicAdd(&r.overflow[4], opaque); // the host's statistic: are there tiles to start late in?
            const uint32_t sub = ((blockIdx.x + 40503u * turn) * 2654435761u) >> 26, region = r.pair_capacity / kSubStreams;
            uint32_t begin = 0xFFFFFFFFu;
            if (r.direct) {
                begin = 0u; // (no pair stream: the entries go straight into the lists)
            } else if (total) {
                const uint32_t got = atomicAdd(&r.pair_cursor[sub], total);
                if (got + total > region)
                    r.overflow[5] = 1u; // this region is full: the host grows the stream and runs the pass again (nothing of this batch is written)
                else
                    begin = sub * region + got;
            }
            for (uint32_t w = 0; w < kFlatWaveCount; ++w) {
                const uint32_t mine = wave_entries[kFlatWaveCount + w] + wave_entries[w];
                wave_entries[w] = begin;
                if (begin != 0xFFFFFFFFu) begin += mine;
            }
        }
        lds_barrier();
        st.at = wave_entries[wave];
#pragma unroll 1
        for (uint32_t ch = 0; ch * kFlatThreads < n_pool; ++ch) { // 2b
            const uint32_t p = ch * kFlatThreads + tid;
            const bool active = p < n_pool;
            const uint32_t j = active ? find_item(pool_begin, p) : 0u;
            const FlatItem& fi = items[j];
            const uint32_t q = active ? p - pool_begin[j] : 0u, row_len = max(fi.nx, 1u), qy = q / row_len, qx = q - qy * row_len;
            const uint32_t tile = (fi.ty_a + qy) * r.tiles_x + fi.tx_a + qx;
            const uint32_t verdict = active ? (uint32_t)pool_bd[p] : 0u, cover_key = active ? (uint32_t)pool_hbd[p] : 0u;
            const uint32_t bd_key = fi.synth_a + ((verdict >> 25) & 1u ? 0u : 1u), hbd_key = fi.synth_a + ((verdict >> 26) & 1u ? 2u : 3u);
            uint32_t left = verdict & 4095u, n_bd = (verdict >> 12) & 4095u, pos = active ? pool_cursor[p] - left : 0u;
            bool cover = (verdict >> 24) & 1u;
            for (;;) {
                const unsigned long long ballot = __ballot(left != 0u);
                if (!ballot) break;
                uint32_t key = 0;
                if (left) {
                    if (cover)
                        cover = false, key = cover_key;
                    else if (n_bd)
                        --n_bd, key = bd_key;
                    else
                        key = hbd_key;
                }
                stage_append<true>(st, r, lane, ballot, tile, pos, key);
                if (left) ++pos, --left;
            }
        }
        lds_barrier(); // (pass 3 moves the cursors 2b has just read)
        CRH_FLAT_PHASE(6) // pass 2
        // ---------------- E: pass 3: the same walk, every entry with its place in the tile's list
        for_edge_rows([&](bool active, const BinEdge& e, uint32_t j, const FlatItem& fi, int ty, int x0, int x1, uint32_t key) {
            const uint32_t base = pool_begin[j] + ((uint32_t)ty - fi.ty_a) * fi.nx, tx_a = fi.tx_a;
            const uint32_t cols = wave_max_u32(active && x0 <= x1 ? (uint32_t)(x1 - x0 + 1) : 0u);
            for (uint32_t c = 0; c < cols; ++c) {
                const int tx = x0 + (int)c;
                const bool hit = active && tx <= x1 && matters(e, tx, ty);
                const unsigned long long ballot = __ballot(hit);
                if (ballot) {
                    uint32_t pos = 0;
                    if (hit) pos = atomicAdd(&pool_cursor[base + ((uint32_t)tx - tx_a)], 1u);
                    stage_append<true>(st, r, lane, ballot, (uint32_t)ty * r.tiles_x + (uint32_t)tx, pos, key);
                }
            }
        });
        if (n_tris) {
            const FlatItem& fi = items[tri.item];
            const bool live = tri.nt != 0u && fi.n_rect != 0u;
            const uint32_t base = pool_begin[tri.item], nx = fi.nx, tx_a = fi.tx_a, ty_a = fi.ty_a;
            walk_tri(tri, live, [&](bool hit, uint32_t tx, uint32_t ty) {
                const unsigned long long ballot = __ballot(hit);
                if (ballot) {
                    uint32_t pos = 0;
                    if (hit) pos = atomicAdd(&pool_cursor[base + (ty - ty_a) * nx + (tx - tx_a)], 1u);
                    stage_append<true>(st, r, lane, ballot, ty * r.tiles_x + tx, pos, tri.key);
                }
            });
        }
        stage_flush_reserved(st, r, lane); // (the batch's range is used up exactly)
        CRH_FLAT_PHASE(7) // pass 3
        // ---------------- hull strips that fold: triangle by triangle, as cover triangles (keys behind the fill chain and the backdrop slots);
        // their entries take the ordinary way into the pair stream (a cursor atomic per block)
        bool any_folded = false;
        for (uint32_t j = 0; j < n_turn; ++j) any_folded = any_folded || (items[j].flags & (kFiHullTris | kFiSkip)) == kFiHullTris;
        if (any_folded) { // (uniform: LDS values)
            st.sub = ((kFlatWaveCount * blockIdx.x + wave + 977u * turn) * 2654435761u) >> 26;
            for (uint32_t j = 0; j < n_turn; ++j) {
                const FlatItem& fi = items[j];
                if ((fi.flags & (kFiHullTris | kFiSkip)) != kFiHullTris) continue;
                const ItemCtx& ctx = fi.ctx;
                for (uint32_t t0 = 64u * wave; t0 + 2u < fi.n_hull; t0 += kFlatThreads) { // 64 triangles per wavefront and turn
                    const uint32_t t = t0 + lane;
                    PrimRec rec = {};
                    bool drawn = false;
                    if (t + 2u < fi.n_hull) {
                        drawn = setup_plain_triangle(s, r, ctx, ctx.cb[6] + t, rec);
                        if (drawn) *reinterpret_cast<PrimRec*>(r.slots + (size_t)(fi.hull_slot0 + 4u * t) * 32u) = rec;
                    }
                    bin_triangles(st, r, lane, drawn, rec.cov, fi.hull_slot0 + 4u * t, ry_first, r_last);
                }
            }
            stage_flush(st, r, lane);
        }
#ifdef CRH_ABLATE
        dump_items += n_turn, dump_tris += n_tris, dump_edges += n_edges, dump_pool += n_pool, dump_work += n_work, dump_walk += wave_max_u32(tri.nt);
#endif
        next += n_turn;
        ++turn;
        CRH_FLAT_PHASE(8) // folded hulls
    }
    // (a run the host cut for one turn needed more: the costs it was cut by are stale — the host counts these and measures again, api.hip)
    if (r.bin_batches && turn > 1u && tid == 0u) atomicAdd(&r.overflow[kExtraTurnsWord], turn - 1u);
#ifdef CRH_ABLATE
    if ((r.debug & 65536u) && tid == 0u) {
        const unsigned long long life = __builtin_amdgcn_s_memtime() - born_t;
        atomicMax(reinterpret_cast<unsigned long long*>(r.overflow + 120), life);
        atomicAdd(reinterpret_cast<unsigned long long*>(r.overflow + 122), life);
        atomicAdd(r.overflow + 124, 1u);
    }
    if ((r.debug & 65536u) && r.item_cost && lane == 0u && wave == 0u) { // CRH_BIN_DUMP: the workgroup's record behind the items' costs
        uint32_t* rec = r.item_cost + 2u * (size_t)(r.n_items + 1u) + 8u * (size_t)blockIdx.x;
        rec[0] = (uint32_t)(__builtin_amdgcn_s_memtime() - born_t), rec[1] = turn, rec[2] = dump_items, rec[3] = dump_tris, rec[4] = dump_edges, rec[5] = dump_pool, rec[6] = dump_work, rec[7] = dump_walk;
    }
#endif
}

__global__ __launch_bounds__(256) void k_scatter(RasterParams r) {
    if (r.overflow[0] | r.overflow[5]) return;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, region = r.pair_capacity / kSubStreams;
    const uint32_t sub = i / region, at = i - sub * region;
    if (sub >= kSubStreams || at >= r.pair_cursor[sub]) return;
    r.tile_list[r.tile_offset[r.pair_tile[i]] + r.pair_pos[i]] = r.pair_key[i];
}

// ---------------------------------------------------------------------------------------------- k_raster_edges
// A 16-bit row mask -> the lane masks of the lane's sample slots. Lane (px, rq) owns row bit rq + 4b in slot b (msaa 1) or 4 rq + q in
// slot q (msaa 4); a slot's lane mask repeats each of its four row bits over a 16-lane group. Scalar unit only: s_bitreplicate doubles
// every bit of its 32-bit operand, a tree of 8 of them turns 16 bits into 4 x 64.
struct SlotMasks {
    unsigned long long m[4];
};
CRH_D unsigned long long bit_double(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long out;
    asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(out) : "s"(v));
    return out;
#else
    return v;
#endif
}
template <int S>
CRH_D SlotMasks slot_masks(uint32_t rows16) {
    if (S == 4) { // gather every sample's four row bits: bit 4 q + rq <- bit 4 rq + q (a 4 x 4 bit transpose)
        const uint32_t t = rows16;
        rows16 = (t & 0x8421u) | ((t & 0x0842u) << 3) | ((t & 0x0084u) << 6) | ((t & 0x0008u) << 9) | ((t >> 3) & 0x0842u) | ((t >> 6) & 0x0084u) | ((t >> 9) & 0x0008u);
    }
    const unsigned long long x2 = bit_double(rows16 & 0xFFFFu), x4 = bit_double((uint32_t)x2);
    const unsigned long long lo = bit_double((uint32_t)x4), hi = bit_double((uint32_t)(x4 >> 32));
    SlotMasks k;
    k.m[0] = bit_double((uint32_t)lo), k.m[1] = bit_double((uint32_t)(lo >> 32)), k.m[2] = bit_double((uint32_t)hi), k.m[3] = bit_double((uint32_t)(hi >> 32));
    return k;
}

#ifndef CRH_EDGE_TILE_WAVES
#define CRH_EDGE_TILE_WAVES 5 // measured 4: 0.321, 5: 0.322, 6: 0.331 (spills), 8: 0.421 ms on the benchmark scene
#endif
// (A second argument — the maximum — was tried to leave register room for the binning kernels beside this one: at a cap of four the
// compiler takes 106 registers and nothing else fits either; and a maximum above the hardware's eight makes LLVM drop the whole attribute.)
#ifndef CRH_COL_LDS_SAMPLES
#define CRH_COL_LDS_SAMPLES 2 // msaa 4: samples of a lane whose colours live in LDS between the covers (k_raster_edges); dashed scene, raster alone: 0 (round 4) 1.60, 2: 1.545, 3: 1.572, 4 (four workgroups per CU): 1.80 - 1.87 ms
#endif
#ifndef CRH_STROKE_TILE_WAVES
#define CRH_STROKE_TILE_WAVES 5 // dashed strokes, msaa 4 (ms): 1 (145 registers): 2.96, 4: 2.34, 5: 2.24, 6: 2.83, 8: 5.05 — the kernel waits, it does not issue
#endif
// One workgroup per 16x16 tile, laid out exactly as k_raster_tile (raster.hip): msaa 1 = one wavefront, four pixel rows per lane;
// msaa 4 = four wavefronts, one pixel row x four samples per lane. Per sample the lane keeps the winding counter, the hull winding of
// the item being drawn and the colour; entries are walked in key order (= draw order).
// LONG: the late start also for lists of several chunks (selected by the host for frames whose tiles hold many entries on average).
template <int S, int ROWS, bool STROKES, bool LONG>
__global__ __launch_bounds__(64 * (4 / ROWS)) __attribute__((amdgpu_waves_per_eu((STROKES || S == 4) ? CRH_STROKE_TILE_WAVES : CRH_EDGE_TILE_WAVES))) void k_raster_edges(SceneDev s, RasterParams r) {
    const uint32_t bid = blockIdx.x; // the workgroup's place in the frame's tile order
    extern __shared__ uint32_t sort_buffer[];
    __shared__ float4 entry_buffer[4 / ROWS][64 * 3];
    __shared__ uint8_t compact_table[STROKES ? 4 / ROWS : 1][STROKES ? 256 : 4]; // (lane, slot) codes of the samples a stroke triangle has to decide
    // msaa 4 (round 5): a lane's sixteen colour floats — read and written at COVER entries only — live in LDS between the covers (a private
    // 64-byte place per lane: no barrier, no bank conflict), so that the stroke fragment stages get the register file (the dashed scene's
    // kernel kept 124 bytes per lane in scratch memory, 30 frames' worth of scratch stores per launch)
    // (... of kColLdsSamples of its four samples: with all four a workgroup takes 33.7 KB of LDS at the usual sort capacity, four instead of
    // five fit a CU, and the kernel is slower than with its spills — 1.87 against 1.60 ms on the dashed scene; with two, 1.545: the kernel is
    // bound by the vector instructions of the dash evaluation, not by its scratch traffic)
    constexpr bool kColLds = S == 4;
    constexpr int kColLdsSamples = CRH_COL_LDS_SAMPLES;
    __shared__ float4 col_lds[kColLds ? 4 / ROWS : 1][kColLds ? ROWS * kColLdsSamples * 64 : 1];
    constexpr uint32_t kB = CRH_XCD_BLOCK_LOG2, kBlock = 1u << kB;
    const uint32_t turn = bid >> 3;
    const uint32_t blocks_x = (r.tiles_x + kBlock - 1u) >> kB, block = (turn >> (2u * kB)) * 8u + (bid & 7u);
    uint32_t tx = (block % blocks_x) * kBlock + (turn & (kBlock - 1u)), ty = (block / blocks_x) * kBlock + ((turn >> kB) & (kBlock - 1u));
    if (r.tile_order) { // the host's order for this frame: every XCD's heavy tiles first (api.hip order_tiles_heavy_first)
        const uint32_t mine = r.tile_order[bid];
        if (mine == 0xFFFFFFFFu) return;
        ty = mine / r.tiles_x, tx = mine - ty * r.tiles_x;
    }
    if (tx >= r.tiles_x || ty >= r.tiles_y || ty < r.slab_ty0 || ty >= r.slab_ty1) return; // (beyond the frame, or not in this pass' slab of tile rows)
    const uint32_t tile = ty * r.tiles_x + tx;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* __restrict__ keys = sort_buffer + wave * r.sort_capacity;
    const uint32_t px = lane & 15u, rq = lane >> 4;
    const uint32_t first_row = 4u * ROWS * wave;
    const uint32_t gx = tx * kTile + px;
    const float tx0 = (float)(tx * kTile), ty0 = (float)(ty * kTile);
    const int tpx = (int)(tx * kTile), tpy = (int)(ty * kTile);
    float sx[S], sy0[S];
    if (S == 1) {
        sx[0] = (float)px + 0.5f;
        sy0[0] = (float)(first_row + rq) + 0.5f;
    } else {
        const float ox[4] = {0.375f, 0.875f, 0.125f, 0.625f}, oy[4] = {0.125f, 0.375f, 0.625f, 0.875f};
#pragma unroll
        for (int q = 0; q < S; ++q) {
            sx[q] = (float)px + ox[q & 3];
            sy0[q] = (float)(first_row + rq) + oy[q & 3];
        }
    }
    const uint32_t row_shift = first_row + rq; // triangles: bit row_shift + 4b of a 16-bit pixel-row mask

    // bit  bit_of(b, q)  of a 16-bit mask shifted right by edge_shift
    int winding[ROWS][S], hullw[ROWS][S];
    float col[ROWS][S][4];
#pragma unroll
    for (int b = 0; b < ROWS; ++b)
#pragma unroll
        for (int q = 0; q < S; ++q) {
            winding[b][q] = 0;
            hullw[b][q] = 0;
            col[b][q][0] = col[b][q][1] = col[b][q][2] = col[b][q][3] = 0.0f;
        }
    if (r.load_existing) {
#pragma unroll
        for (int b = 0; b < ROWS; ++b) {
            const uint32_t gy = ty * kTile + first_row + 4u * b + rq;
            if (gx < r.width && gy < r.height) {
                const float4 d = load_pixel(r, gx, gy);
#pragma unroll
                for (int q = 0; q < S; ++q) col[b][q][0] = d.x, col[b][q][1] = d.y, col[b][q][2] = d.z, col[b][q][3] = d.w;
            }
        }
    }
    auto col_fetch = [&]() { // the lane's colours: LDS -> registers (in front of what blends or stores them)
        if constexpr (kColLds) {
#pragma unroll
            for (int b = 0; b < ROWS; ++b)
#pragma unroll
                for (int q = 0; q < kColLdsSamples; ++q) {
                    const float4 v = col_lds[threadIdx.x >> 6][(uint32_t)(b * kColLdsSamples + q) * 64u + (threadIdx.x & 63u)];
                    col[b][q][0] = v.x, col[b][q][1] = v.y, col[b][q][2] = v.z, col[b][q][3] = v.w;
                }
        }
    };
    auto col_keep = [&]() { // registers -> LDS (the registers are free again until the next cover)
        if constexpr (kColLds) {
#pragma unroll
            for (int b = 0; b < ROWS; ++b)
#pragma unroll
                for (int q = 0; q < kColLdsSamples; ++q) col_lds[threadIdx.x >> 6][(uint32_t)(b * kColLdsSamples + q) * 64u + (threadIdx.x & 63u)] = make_float4(col[b][q][0], col[b][q][1], col[b][q][2], col[b][q][3]);
        }
    };
    col_keep();
    const uint32_t list_begin = r.direct ? r.tile_base[tile] : r.tile_offset[tile];
    uint32_t n = (r.overflow[0] | r.overflow[5]) ? 0u : (r.direct ? r.tile_count[tile] : r.tile_offset[tile + 1] - list_begin);
    constexpr uint32_t kLdsSortMax = kSortBytesMax / (4u * (4u / ROWS));
    if (n > r.sort_capacity && n <= kLdsSortMax) { // the host grows the sort buffer (overflow[3] = the longest list) and runs the frame again
        if (r.direct && threadIdx.x == 0u) atomicMax(&r.overflow[3], n); // (otherwise the scan of the counts has published it)
        n = 0;
    } else if (n > kLdsSortMax && r.direct && threadIdx.x == 0u) {
        atomicMax(&r.overflow[3], n); // (sorted in place below; the host keeps the longest list it has heard of: crh_frame::longest_list)
    }
#ifdef CRH_ABLATE
    if (r.debug & 64u) n = 0;
    if ((r.debug & 524288u) && n > 64u) n = 0;   // only the lists that fit one chunk
    if ((r.debug & 1048576u) && n <= 64u) n = 0; // only the longer ones
#endif
    uint32_t my_key = 0xFFFFFFFFu;
    const bool sorted_in_place = n > kLdsSortMax;
    uint32_t* const segment = r.tile_list + list_begin;
    if (sorted_in_place) { // as raster.hip: a normalised bitonic network over the tile's segment of the list, in global memory
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
        for (uint32_t kk = 2; kk <= padded; kk <<= 1) {
            const uint32_t half = kk >> 1;
            for (uint32_t p = tid; p < (padded >> 1); p += n_threads) {
                const uint32_t blk = p / half, t = p - blk * half;
                exchange(blk * kk + t, blk * kk + kk - 1u - t);
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
        if (n > 1u) {
            const uint32_t depth = n <= 8u ? 8u : (n <= 16u ? 16u : (n <= 32u ? 32u : 64u)); // a network as deep as the list needs (the keys sit in the leading lanes)
#pragma unroll
            for (uint32_t kk = 2; kk <= 64u; kk <<= 1) {
                if (kk > depth) break;
#pragma unroll
                for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                    const uint32_t other = __shfl_xor(my_key, j, 64);
                    const bool keep_min = ((lane & j) == 0) == ((lane & kk) == 0);
                    my_key = keep_min ? min(my_key, other) : max(my_key, other);
                }
            }
        }
    } else {
        uint32_t padded = 128;
        while (padded < n) padded <<= 1;
        for (uint32_t i = lane; i < padded; i += 64u) keys[i] = i < n ? r.tile_list[list_begin + i] : 0xFFFFFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t kk = 2; kk <= padded; kk <<= 1)
            for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                for (uint32_t i = lane; i < padded; i += 64u) {
                    const uint32_t partner = i ^ j;
                    if (partner > i) {
                        const uint32_t a = keys[i], b = keys[partner];
                        if (((i & kk) == 0) ? (a > b) : (a < b)) {
                            keys[i] = b;
                            keys[partner] = a;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
    }

    const uint8_t* slots = r.slots;
    const int wmask = (int)r.winding_mask;
    // Cooperative row evaluation of an edge entry: lane j < 16 evaluates the wave's sample row j (msaa 1: pixel row j; msaa 4: local
    // row j / 4, sample j % 4) at the left tile boundary, lane 16 the backdrop row q0 of the tile (k_bin_edges); one ballot per question.
    const uint32_t row_j = lane & 15u;
    const float ry_row = lane == 16u ? (S == 1 ? 0.5f : 0.125f)
                                     : (S == 1 ? (float)row_j + 0.5f : (float)(first_row + (row_j >> 2)) + ((float)(row_j & 3u) * 0.25f + 0.125f));
    const float sy_row = ty0 + ry_row;
    // The late start (see "Occlusion" below) of a list longer than one chunk is found before the walk: the chunks' keys and the first word
    // of their slots from the END of the list — X, the last opaque cover over the whole tile none of whose item's triangles are in the
    // list (keys ascend: a binary search), then R, the last cover before X that resets the whole tile. The walk begins behind R.
    auto key_of = [&](uint32_t i) -> uint32_t { return sorted_in_place ? __hip_atomic_load(segment + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : keys[i]; };
    uint32_t walk_from = 0, verify_entry = 0xFFFFFFFFu; // absolute positions in the list
    constexpr bool kLongLateStart = LONG; // (a variant of its own: compiled into the kernel of the 10 000 path scene — few tiles beyond one chunk — it cost 8 %: 0.234 -> 0.255 ms)
    if (kLongLateStart && n > 64u && !r.load_existing) {
        uint32_t x_at = 0xFFFFFFFFu;
        for (uint32_t c0 = ((n - 1u) >> 6) << 6;; c0 -= 64u) {
            const uint32_t k = c0 + lane < n ? key_of(c0 + lane) : 0xFFFFFFFFu;
            uint32_t code = 0;
            if (c0 + lane < n) {
                const uint32_t flags = *reinterpret_cast<const uint32_t*>(slots + (size_t)k * 32u);
                code = ((flags >> 4) & 15u) == EK_SYNTH ? (flags >> 8) & 31u : 0u;
            }
            unsigned long long resets = __builtin_amdgcn_ballot_w64(code >= 4u + kCoverHull);
            if (x_at == 0xFFFFFFFFu) {
                unsigned long long candidates = __builtin_amdgcn_ballot_w64(code >= 4u + kCoverOpaque);
                while (candidates) {
                    const uint32_t at = 63u - (uint32_t)__builtin_clzll(candidates);
                    const SynthRec sr = load_uniform(reinterpret_cast<const SynthRec*>(slots + (size_t)__builtin_amdgcn_readlane(k, at) * 32u));
                    uint32_t lo = 0, hi = n; // the number of keys below the item's synthetic slots
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (key_of(mid) < sr.synth_a) lo = mid + 1u;
                        else hi = mid;
                    }
                    if (lo == 0u || key_of(lo - 1u) < sr.first_slot) {
                        x_at = c0 + at;
                        resets &= (1ull << at) - 1ull;
                        break;
                    }
                    candidates &= ~(1ull << at);
                }
            }
            if (x_at != 0xFFFFFFFFu && resets) {
                walk_from = c0 + 64u - (uint32_t)__builtin_clzll(resets); // behind the last whole-tile reset before X
                verify_entry = x_at;
                break;
            }
            if (c0 == 0u) break;
        }
    }
    uint32_t first_j = walk_from & 63u;
    bool again_from_the_top = false;
    for (uint32_t q0 = kLongLateStart ? walk_from & ~63u : 0u; q0 < n; q0 += 64u) {
        if (sorted_in_place)
            my_key = q0 + lane < n ? __hip_atomic_load(segment + q0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
        else if (n > 64u)
            my_key = q0 + lane < n ? keys[q0 + lane] : 0xFFFFFFFFu;
        const uint32_t count = min(64u, n - q0);
        // ---- entry setup, vectorised across the chunk: lane j prepares entry j
        float4 e0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), e1 = e0, e2 = e0;
        bool hull_over_tile = false, replaces_tile = false; // this lane's entry is a cover that resets every sample / an opaque cover over the whole tile
        uint32_t item_first = 0, item_synth_a = 0;
        if (lane < count) {
            const uint8_t* slot = slots + (size_t)my_key * 32u;
            const uint32_t flags = *reinterpret_cast<const uint32_t*>(slot);
            const uint32_t kind = (flags >> 4) & 15u;
            if (kind == EK_EDGE) {
                const EdgeRec er = *reinterpret_cast<const EdgeRec*>(slot);
                const float c = er.bx * (ty0 - er.lo_y) + er.nay * (tx0 - er.lo_x);
                const bool xr = er.lo_x <= tx0 && tx0 < er.hi_x; // the edge crosses the line of the left tile boundary
                e0 = make_float4(c, er.bx, er.nay, 0.0f);
                e1 = make_float4(fminf(er.lo_y, er.hi_y), fmaxf(er.lo_y, er.hi_y), 0.0f, __uint_as_float(flags | (xr ? 0x1000u : 0u)));
            } else if (kind == EK_SYNTH) {
                const SynthRec sr = *reinterpret_cast<const SynthRec*>(slot);
                e0 = make_float4(sr.r, sr.g, sr.b, sr.a);
                e1.w = __uint_as_float(flags);
                const uint32_t code_all = (flags >> 8) & 31u;
                hull_over_tile = code_all >= 4u + kCoverHull;
                replaces_tile = code_all >= 4u + kCoverOpaque;
                item_first = sr.first_slot, item_synth_a = sr.synth_a;
            } else {
                const PrimCoverage mine = *reinterpret_cast<const PrimCoverage*>(slot);
                const int bx0 = max((int)mine.box.x, tpx) - tpx, bx1 = min((int)mine.box.y, tpx + kTile - 1) - tpx;
                const int by0 = max((int)mine.box.z, tpy) - tpy, by1 = min((int)mine.box.w, tpy + kTile - 1) - tpy;
                const uint32_t col_bits = bx1 >= bx0 ? (2u << bx1) - (1u << bx0) : 0u, row_bits = by1 >= by0 ? (2u << by1) - (1u << by0) : 0u;
                float cc[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) cc[i] = mine.bx[i] * (ty0 - mine.lo_y[i]) + mine.nay[i] * (tx0 - mine.lo_x[i]);
                e0 = make_float4(cc[0], cc[1], cc[2], __uint_as_float(col_bits | (row_bits << 16)));
                e1 = make_float4(mine.bx[0], mine.bx[1], mine.bx[2], __uint_as_float(flags));
                e2 = make_float4(mine.nay[0], mine.nay[1], mine.nay[2], __uint_as_float(mine.desc));
            }
        }
        float4* __restrict__ entries = entry_buffer[wave];
        __builtin_amdgcn_wave_barrier();
        entries[lane * 3u + 0u] = e0;
        entries[lane * 3u + 1u] = e1;
        entries[lane * 3u + 2u] = e2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifdef CRH_ABLATE
        if (r.debug & 128u) continue;
#endif
        // sample (b, q) of this lane inside a set-up triangle? (k_raster_tile's coverage: packed edge functions, box bits)
        auto coverage = [&](const float4& ea4, const float4& eb4, const float4& ec4, uint32_t flags, bool (&inside)[ROWS][S]) {
            const uint32_t bits = __float_as_uint(ea4.w);
            const float c0 = ea4.x, c1 = ea4.y, c2 = ea4.z, bx_0 = eb4.x, bx_1 = eb4.y, bx_2 = eb4.z, nay_0 = ec4.x, nay_1 = ec4.y, nay_2 = ec4.z;
            const uint32_t lane_rows = ((bits >> px) & 1u) ? (bits >> 16) >> row_shift : 0u;
            const int thr0 = 1 - (int)(flags & 1u), thr1 = 1 - (int)((flags >> 1) & 1u), thr2 = 1 - (int)((flags >> 2) & 1u);
            float ha[S], hb[S], hc[S];
#pragma unroll
            for (int q = 0; q < S; ++q) {
                ha[q] = fmaf(sx[q], nay_0, c0);
                hb[q] = fmaf(sx[q], nay_1, c1);
                hc[q] = fmaf(sx[q], nay_2, c2);
            }
#pragma unroll
            for (int cmb = 0; cmb < ROWS * S; cmb += 2) {
                const int b0 = cmb / S, k0 = cmb % S, b1 = (cmb + 1) / S, k1 = (cmb + 1) % S;
                const f32x2 y = {sy0[k0] + (float)(4 * b0), sy0[k1] + (float)(4 * b1)};
                const f32x2 ea = fma2(y, splat2(bx_0), f32x2{ha[k0], ha[k1]});
                const f32x2 eb = fma2(y, splat2(bx_1), f32x2{hb[k0], hb[k1]});
                const f32x2 ec = fma2(y, splat2(bx_2), f32x2{hc[k0], hc[k1]});
                inside[b0][k0] = (__float_as_int(ea[0]) >= thr0) & (__float_as_int(eb[0]) >= thr1) & (__float_as_int(ec[0]) >= thr2) & ((lane_rows & (1u << (4 * b0))) != 0u);
                inside[b1][k1] = (__float_as_int(ea[1]) >= thr0) & (__float_as_int(eb[1]) >= thr1) & (__float_as_int(ec[1]) >= thr2) & ((lane_rows & (1u << (4 * b1))) != 0u);
            }
        };
        // Two nested loops over the chunk's entries: the inner one runs over what only changes the winding counters (triangles, edges,
        // backdrops) and stops at a cover entry, which the outer loop body handles — the ONE place where the colour registers are
        // written. (With the colour updated inside a multi-way dispatch, every iteration ended with two dozen register copies.)
        // Occlusion. If the list holds an opaque cover over the whole tile (X, the last one; none of its item's triangles in the tile), what was
        // drawn before X only matters through the winding a sample may still carry when X tests it (an item whose hull boundary crosses the
        // tile can leave one just outside its hull). The winding is zero everywhere behind a cover that resets the whole tile (R, the last one
        // before X): the list is started behind R with cleared counters — exact for the winding; the colours it misses are all overwritten by
        // X unless some sample fails X's stencil test, and then (verify_at) the tile is done again from the top.
        uint32_t j = kLongLateStart ? first_j : 0u, verify_at = (kLongLateStart && verify_entry - q0 < 64u) ? verify_entry - q0 : 0xFFFFFFFFu; // (a list of several chunks: found above)
        first_j = 0;
        if (n <= 64u && !r.load_existing) { // (the whole list is in this chunk, and the tile starts from a known colour)
            unsigned long long candidates = __builtin_amdgcn_ballot_w64(replaces_tile);
            const unsigned long long resets = __builtin_amdgcn_ballot_w64(hull_over_tile);
            while (candidates) {
                const uint32_t at = 63u - (uint32_t)__builtin_clzll(candidates);
                const uint32_t first_slot = __builtin_amdgcn_readlane(item_first, at), synth_a = __builtin_amdgcn_readlane(item_synth_a, at);
                // the bin kernel vouched for the item's edges; its triangles are binned by another wavefront: one of them in this tile has a key
                // in [first_slot, synth_a), and keys ascend
                const uint32_t below = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(lane < count && my_key < synth_a));
                if (below == 0u || __builtin_amdgcn_readlane(my_key, below - 1u) < first_slot) {
                    const unsigned long long before = resets & ((1ull << at) - 1ull);
                    j = before ? 64u - (uint32_t)__builtin_clzll(before) : 0u; // behind the last whole-tile reset before X
                    verify_at = j ? at : 0xFFFFFFFFu;
                    break;
                }
                candidates &= ~(1ull << at);
            }
        }
        while (j < count) {
        for (; j < count; ++j) {
            const uint32_t prim = __builtin_amdgcn_readlane(my_key, j);
            const float4 ea4 = entries[j * 3u + 0u], eb4 = entries[j * 3u + 1u], ec4 = entries[j * 3u + 2u];
            const uint32_t flags = __builtin_amdgcn_readfirstlane(__float_as_uint(eb4.w));
            const uint32_t kind = (flags >> 4) & 15u;
            if (kind == EK_COVER_TRI || (kind == EK_SYNTH && ((flags >> 8) & 31u) >= 4u)) break; // a cover: the outer loop's business
#ifdef CRH_ABLATE // tools/ablate_edges.sh: what does each class of entries cost?
            if ((r.debug & 256u) && lane == 0u) atomicAdd(&r.overflow[8 + (kind == EK_EDGE ? ((flags & kEdgeHull) ? 1 : 0) : (kind == EK_SYNTH ? 2 : 3))], 1u);
            if ((r.debug & 8u) && kind == EK_EDGE) continue;
            if ((r.debug & 16u) && kind == EK_SYNTH) continue;
            if ((r.debug & 32u) && kind != EK_EDGE && kind != EK_SYNTH) continue;
#endif
            // Every class updates the state it owns in place, unconditionally per sample (a select with the old value where nothing
            // changes): state that is modified under a wave-uniform inner branch, or merged behind the dispatch, costs the compiler a
            // dozen register copies per entry at the join.
            if (kind == EK_EDGE) {
                // w(p) += sigma * [ Y_k g(p) + A_k ],  A_k = xr (g(q_k) - g(q_0)) - Y_k g(q_k)   (see the header of this file)
                const float c0 = ea4.x, ebx = ea4.y, enay = ea4.z, ymin = eb4.x, ymax = eb4.y;
                const int thr = 1 - (int)(flags & kEdgeTl);
                // the 16 sample rows of the wave + q0, one row per lane: g at the left tile boundary and the half-open y range
                const float eq = fmaf(ry_row, ebx, fmaf(0.0f, enay, c0));
                const unsigned long long gq_all = __builtin_amdgcn_ballot_w64(__float_as_int(eq) >= thr);
                const unsigned long long y_all = __builtin_amdgcn_ballot_w64((ymin <= sy_row) & (sy_row < ymax));
                const uint32_t gq = (uint32_t)gq_all & 0xFFFFu, ym = (uint32_t)y_all & 0xFFFFu;
                const uint32_t g0 = ((uint32_t)gq_all >> 16) & 1u ? 0xFFFFu : 0u, xr = (flags & 0x1000u) ? 0xFFFFu : 0u;
                const uint32_t a_plus = xr & gq & ~g0 & ~ym;                      // A = +1
                const uint32_t a_minus = (xr & ~gq & g0) | (ym & gq & (~xr | g0)); // A = -1
                const int unit = (flags & kEdgeSigmaPos) ? 1 : -1;               // sigma
                if (ROWS == 1 && (ym | a_plus | a_minus) == 0u) continue;        // msaa 4: none of this wavefront's four pixel rows is in reach
                const SlotMasks ys = slot_masks<S>(ym);
                float h[S];
#pragma unroll
                for (int q = 0; q < S; ++q) h[q] = fmaf(sx[q], enay, c0);
                int d[ROWS][S];
#pragma unroll
                for (int cmb = 0; cmb < ROWS * S; cmb += 2) {
                    const int b0 = cmb / S, k0 = cmb % S, b1 = (cmb + 1) / S, k1 = (cmb + 1) % S;
                    const f32x2 y = {sy0[k0] + (float)(4 * b0), sy0[k1] + (float)(4 * b1)};
                    const f32x2 ev = fma2(y, splat2(ebx), f32x2{h[k0], h[k1]});
                    if constexpr (LONG) { // the y range per sample as a bit of ym on the VECTOR unit: this variant's counters say the scalar unit is the busier one (70 % against 63 %)
                        const uint32_t y_shift = S == 1 ? rq : 4u * rq;
                        d[b0][k0] = (__float_as_int(ev[0]) >= thr ? unit : 0) & __builtin_amdgcn_sbfe((int)ym, y_shift + (S == 1 ? 4u * b0 : (uint32_t)k0), 1u);
                        d[b1][k1] = (__float_as_int(ev[1]) >= thr ? unit : 0) & __builtin_amdgcn_sbfe((int)ym, y_shift + (S == 1 ? 4u * b1 : (uint32_t)k1), 1u);
                    } else {
                    d[b0][k0] = ((__float_as_int(ev[0]) >= thr) & __builtin_amdgcn_inverse_ballot_w64(ys.m[S == 1 ? b0 : k0])) ? unit : 0;
                    d[b1][k1] = ((__float_as_int(ev[1]) >= thr) & __builtin_amdgcn_inverse_ballot_w64(ys.m[S == 1 ? b1 : k1])) ? unit : 0;
                    }
                }
                if (a_plus | a_minus) { // the edge crosses the left tile boundary (or runs left of it): row constants (on the deltas, not on the state)
                    const uint32_t lane_shift = S == 1 ? rq : 4u * rq; // sample (b, q) of lane (px, rq) is row bit rq + 4b (msaa 1) / 4 rq + q (msaa 4)
                    const uint32_t psh = (unit > 0 ? a_plus : a_minus) >> lane_shift, nsh = (unit > 0 ? a_minus : a_plus) >> lane_shift;
#pragma unroll
                    for (int b = 0; b < ROWS; ++b)
#pragma unroll
                        for (int q = 0; q < S; ++q) {
                            const int p = S == 1 ? 4 * b : q;
                            d[b][q] += (int)((psh >> p) & 1u) - (int)((nsh >> p) & 1u);
                        }
                }
                { // both sets of counters, each with its (scalar) share of the delta — one multiply-add per counter: a branch on the edge's
                  // kind around the additions left the compiler a dozen register copies per entry where the two sides meet
                    const int to_hull = (flags & kEdgeHull) ? 1 : 0, to_fill = 1 - to_hull;
#pragma unroll
                    for (int b = 0; b < ROWS; ++b)
#pragma unroll
                        for (int q = 0; q < S; ++q) {
                            hullw[b][q] = __mul24(d[b][q], to_hull) + hullw[b][q];
                            winding[b][q] = __mul24(d[b][q], to_fill) + winding[b][q];
                        }
                }
            } else if (kind == EK_SYNTH) { // a whole-tile backdrop of the fill (codes 0, 1) or hull (2, 3) winding
                const uint32_t code = (flags >> 8) & 31u;
                const int vw = code < 2u ? ((code & 1u) ? -1 : 1) : 0, vh = code < 2u ? 0 : ((code & 1u) ? -1 : 1);
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int q = 0; q < S; ++q) {
                        winding[b][q] += vw;
                        hullw[b][q] += vh;
                    }
            } else {
            // ---- curve and stroke triangles, as k_raster_tile
            if (ROWS == 1) { // msaa 4: this wavefront owns four of the tile's sixteen pixel rows — a sliver's box seldom reaches them all
                const uint32_t row_bits = __builtin_amdgcn_readfirstlane(__float_as_uint(ea4.w)) >> 16;
                if (((row_bits >> first_row) & 15u) == 0u) continue;
            }
            const PrimFragment frag = load_uniform(reinterpret_cast<const PrimFragment*>(slots + (size_t)prim * 32u + 64u));
            bool inside[ROWS][S];
            coverage(ea4, eb4, ec4, flags, inside);
            {
            const int delta = (flags & 8u) ? 1 : -1; // front (ccw on screen) increments, back decrements (renderer.rs:577-582)
            const float dx0 = tx0 - frag.v0x, dy0 = ty0 - frag.v0y;
            if (kind <= KIND_RC) { // the four implicit-curve tests (shaders.wgsl:236-266), one straight-line variant per kind
                auto curve = [&](auto kind_tag) {
                    constexpr uint32_t K = decltype(kind_tag)::value;
                    constexpr int NA = K == KIND_IQ ? 2 : (K == KIND_RC ? 4 : 3); // attributes the kind interpolates
                    float hx[NA][S]; // attribute planes, tile relative: the row-independent inner fma
#pragma unroll
                    for (int t = 0; t < NA; ++t) {
                        const float ac = fmaf(dy0, frag.gy[t], fmaf(dx0, frag.gx[t], frag.a0[t]));
#pragma unroll
                        for (int q = 0; q < S; ++q) hx[t][q] = fmaf(sx[q], frag.gx[t], ac);
                    }
#pragma unroll
                    for (int b = 0; b < ROWS; ++b) {
                        bool row_touched = false;
#pragma unroll
                        for (int q = 0; q < S; ++q) row_touched = row_touched | inside[b][q];
                        if (!__any(row_touched)) continue; // curve triangles are small: most rows of the tile are not touched
#pragma unroll
                        for (int q = 0; q < S; ++q) {
                            const float y = sy0[q] + (float)(4 * b);
                            float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                            for (int t = 0; t < NA; ++t) a[t] = fmaf(y, frag.gy[t], hx[t][q]);
                            const float lhs = (K == KIND_IQ || K == KIND_RQ) ? a[0] * a[0] : a[0] * a[0] * a[0];
                            const float rhs = K == KIND_IQ ? a[1] : (K == KIND_RC ? a[1] * a[2] * a[3] : a[1] * a[2]);
                            winding[b][q] += (inside[b][q] && lhs - rhs <= 0.0f) ? delta : 0;
                        }
                    }
                };
                if (kind == KIND_IQ)
                    curve(std::integral_constant<uint32_t, KIND_IQ>{});
                else if (kind == KIND_IC)
                    curve(std::integral_constant<uint32_t, KIND_IC>{});
                else if (kind == KIND_RQ)
                    curve(std::integral_constant<uint32_t, KIND_RQ>{});
                else
                    curve(std::integral_constant<uint32_t, KIND_RC>{});
            } else if (STROKES) { // KIND_LINE / KIND_JOINT: the stroke fragment stages (shaders.wgsl:268-300)
                // A stroke triangle is a sliver: a few percent of the wave's 256 samples are inside it, and what is decided per sample
                // (dash pattern walk, caps, the joint's atan2) is long and branchy. So the samples to decide — inside the triangle, stencil
                // still zero: Equal(0) -> IncrementWrap, both faces (renderer.rs:571-576) — are COMPACTED: their (lane, slot) codes go to
                // an LDS table in dense order, lane d decides the d-th of them, and the verdicts return through a ballot.
                unsigned long long cand[ROWS * S];
                uint32_t base[ROWS * S], total = 0;
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int q = 0; q < S; ++q) {
                        cand[b * S + q] = __builtin_amdgcn_ballot_w64(inside[b][q] && (winding[b][q] & wmask) == 0);
                        base[b * S + q] = total;
                        total += (uint32_t)__popcll(cand[b * S + q]);
                    }
#ifdef CRH_ABLATE
                if (r.debug & 512u) total = 0u; // tools/ablate_edges.sh: the stroke triangles' coverage without their fragment stages
#endif
                if (total != 0u) {
                    uint8_t* const table = compact_table[wave];
                    uint32_t rank[ROWS * S];
                    __builtin_amdgcn_wave_barrier(); // the previous entry's readers of the table are through
#pragma unroll
                    for (int k = 0; k < ROWS * S; ++k) {
                        rank[k] = base[k] + __builtin_amdgcn_mbcnt_hi((uint32_t)(cand[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cand[k], 0u));
                        if ((cand[k] >> lane) & 1ull) table[rank[k]] = (uint8_t)(lane | ((uint32_t)k << 6));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const crh_dynamic_stroke_descriptor dsc = load_uniform(&s.descriptors[__builtin_amdgcn_readfirstlane(__float_as_uint(ec4.w))]);
                    const uint32_t caps = dsc.caps, count_dashed_join = dsc.count_dashed_join;
                    const uint32_t flat_u = frag.flat_u;
                    const float end_y = frag.end_y;
                    const bool dashed = (count_dashed_join & 4u) != 0u;
                    float ac[3]; // the attribute planes, tile relative
#pragma unroll
                    for (int t = 0; t < 3; ++t) ac[t] = fmaf(dy0, frag.gy[t], fmaf(dx0, frag.gx[t], frag.a0[t]));
                    for (uint32_t first = 0; first < total; first += 64u) {
                        const uint32_t d = first + lane;
                        const bool live = d < total;
                        const uint32_t code = live ? table[d] : 0u;
                        const uint32_t src = code & 63u, slot = code >> 6; // whose sample: lane (px, rq), slot (b, q)
                        const uint32_t spx = src & 15u, srq = src >> 4;
                        float x, y; // exactly sx[q] and sy0[q] + 4 b of that lane
                        if (S == 1) {
                            x = (float)spx + 0.5f;
                            y = ((float)(first_row + srq) + 0.5f) + (float)(4u * slot);
                        } else {
                            const float ox = slot == 0u ? 0.375f : (slot == 1u ? 0.875f : (slot == 2u ? 0.125f : 0.625f));
                            const float oy = slot == 0u ? 0.125f : (slot == 1u ? 0.375f : (slot == 2u ? 0.625f : 0.875f));
                            x = (float)spx + ox;
                            y = (float)(first_row + srq) + oy;
                        }
                        const float a0 = fmaf(y, frag.gy[0], fmaf(x, frag.gx[0], ac[0])), a1 = fmaf(y, frag.gy[1], fmaf(x, frag.gx[1], ac[1]));
                        const float a2 = fmaf(y, frag.gy[2], fmaf(x, frag.gx[2], ac[2]));
                        bool fill = false;
                        if (live) {
                            if (kind == KIND_LINE) {
                                if (dashed)
                                    fill = stroke_dashed(dsc, a0, a1);
                                else if ((flat_u & 65536u) != 0u)
                                    fill = cap_test(a0, a1 - end_y, caps >> 4);
                                else if (a1 < 0.0f)
                                    fill = cap_test(a0, -a1, caps);
                                else
                                    fill = true;
                            } else {
                                const float radius = sqrtf(a0 * a0 + a1 * a1);
                                const uint32_t join = count_dashed_join & 3u;
                                fill = join == 1u ? (flat_u & 65536u) != 0u : (join == 2u ? radius <= 0.5f : true);
                                if (fill && dashed) fill = stroke_dashed_joint(dsc, radius, a0, a1, a2);
                            }
                        }
                        const unsigned long long filled = __builtin_amdgcn_ballot_w64(fill);
#pragma unroll
                        for (int b = 0; b < ROWS; ++b)
#pragma unroll
                            for (int q = 0; q < S; ++q) {
                                const uint32_t at = rank[b * S + q] - first; // < 64 iff this pass decided the sample
                                const bool mine = ((cand[b * S + q] >> lane) & 1ull) != 0ull && at < 64u;
                                winding[b][q] += (mine && ((filled >> (at & 63u)) & 1ull) != 0ull) ? 1 : 0;
                            }
                    }
                }
            }
            } // curve / stroke triangles
            } // triangles
        } // run of entries that only change the winding counters
        if (j >= count) break;
        {
            // ---- entry j is a cover: color_cover (renderer.rs:340-354, 736-754) — blend where the winding is not zero, zero the winding
            //      of the covered samples; premultiplied "over" (shaders.wgsl:304-309, blending of examples/showcase/main.rs:32-43)
            const uint32_t prim = __builtin_amdgcn_readlane(my_key, j);
            const float4 ea4 = entries[j * 3u + 0u], eb4 = entries[j * 3u + 1u], ec4 = entries[j * 3u + 2u];
            const uint32_t flags = __builtin_amdgcn_readfirstlane(__float_as_uint(eb4.w));
            const uint32_t kind = (flags >> 4) & 15u;
            ++j;
#ifdef CRH_ABLATE
            if ((r.debug & 256u) && lane == 0u) atomicAdd(&r.overflow[8 + (kind == EK_SYNTH ? 2 : 3)], 1u);
            if ((r.debug & 16u) && kind == EK_SYNTH) continue;
            if ((r.debug & 32u) && kind != EK_SYNTH) continue;
#endif
            bool blend[ROWS][S];
            float cs0, cs1, cs2, cs3;
            col_fetch();
            if (kind == EK_SYNTH) { // COVER over the samples inside the hull, one unit of both backdrops folded in
                const uint32_t code = 4u + (((flags >> 8) & 31u) - 4u) % 9u;
                const int bd = (int)((code - 4u) % 3u) - 1, hbd = (int)((code - 4u) / 3u) - 1;
                cs0 = ea4.x, cs1 = ea4.y, cs2 = ea4.z, cs3 = ea4.w;
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int q = 0; q < S; ++q) {
                        const int w = winding[b][q] + bd;
                        const bool in_hull = hullw[b][q] + hbd != 0;
                        blend[b][q] = in_hull && (w & wmask) != 0;
                        winding[b][q] = in_hull ? 0 : w;
                        hullw[b][q] = 0;
                    }
                if (j - 1u == verify_at) { // X of the late start: does it overwrite every sample? (it does unless a sample inherited a winding)
                    bool every = true;
#pragma unroll
                    for (int b = 0; b < ROWS; ++b)
#pragma unroll
                        for (int q = 0; q < S; ++q) every = every && blend[b][q];
                    verify_at = 0xFFFFFFFFu;
                    if (__builtin_amdgcn_ballot_w64(every) != ~0ull || (r.debug & 33554432u) != 0u) { // no (or debug bit 25, tests: never trusted): the colours behind it matter — the whole list, from cleared state
#pragma unroll
                        for (int b = 0; b < ROWS; ++b)
#pragma unroll
                            for (int q = 0; q < S; ++q) {
                                winding[b][q] = 0, hullw[b][q] = 0;
                                col[b][q][0] = col[b][q][1] = col[b][q][2] = col[b][q][3] = 0.0f;
                            }
                        col_keep();
                        j = 0;
                        if (kLongLateStart && n > 64u) { // the first chunk has to be set up again
                            verify_entry = 0xFFFFFFFFu;
                            again_from_the_top = true;
                            break;
                        }
                        continue;
                    }
                }
            } else { // a triangle of a folded hull strip, drawn as the reference draws it
                const PrimFragment frag = load_uniform(reinterpret_cast<const PrimFragment*>(slots + (size_t)prim * 32u + 64u));
                bool inside[ROWS][S];
                coverage(ea4, eb4, ec4, flags, inside);
                cs0 = frag.a0[0], cs1 = frag.a0[1], cs2 = frag.a0[2], cs3 = frag.a0[3];
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int q = 0; q < S; ++q) {
                        blend[b][q] = inside[b][q] && (winding[b][q] & wmask) != 0;
                        winding[b][q] = inside[b][q] ? 0 : winding[b][q];
                    }
            }
            // an opaque source over finite colours: src + dst * (1 - 1) is the source (r.occlude: every colour of the pass is tame; a source
            // component that is -0 would come out as +0 through the arithmetic, so it takes the arithmetic): a select per channel, no multiply-add
            const bool replace = r.occlude != 0u && cs3 == 1.0f && __float_as_uint(cs0) != 0x80000000u && __float_as_uint(cs1) != 0x80000000u && __float_as_uint(cs2) != 0x80000000u;
            if (replace) {
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int q = 0; q < S; ++q) {
                        col[b][q][0] = blend[b][q] ? cs0 : col[b][q][0];
                        col[b][q][1] = blend[b][q] ? cs1 : col[b][q][1];
                        col[b][q][2] = blend[b][q] ? cs2 : col[b][q][2];
                        col[b][q][3] = blend[b][q] ? cs3 : col[b][q][3];
                    }
            } else {
            const float one_minus_a = 1.0f - cs3;
#pragma unroll
            for (int b = 0; b < ROWS; ++b)
#pragma unroll
                for (int q = 0; q < S; ++q) { // unconditional, in place (a select with the old value where nothing blends)
                    const float n0 = cs0 + col[b][q][0] * one_minus_a, n1 = cs1 + col[b][q][1] * one_minus_a;
                    const float n2 = cs2 + col[b][q][2] * one_minus_a, n3 = cs3 + col[b][q][3] * one_minus_a;
                    col[b][q][0] = blend[b][q] ? n0 : col[b][q][0];
                    col[b][q][1] = blend[b][q] ? n1 : col[b][q][1];
                    col[b][q][2] = blend[b][q] ? n2 : col[b][q][2];
                    col[b][q][3] = blend[b][q] ? n3 : col[b][q][3];
                }
            }
            if (r.format == CRH_FORMAT_RGBA8_ATTACHMENT) { // an Rgba8Unorm attachment keeps 8 bits of what the blender writes (idempotent on the others)
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int q = 0; q < S; ++q)
#pragma unroll
                        for (int ch = 0; ch < 4; ++ch) col[b][q][ch] = attachment_unorm8(col[b][q][ch]);
            }
            col_keep();
        }
        } // entries of the chunk
        if (kLongLateStart && again_from_the_top) { // (X of the late start did not overwrite every sample: the whole list, chunk 0 first)
            again_from_the_top = false;
            q0 = 0u - 64u;
        }
    }
    // ---- MSAA resolve (box average) + RGBA8 unorm store
    col_fetch();
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
                for (int q = 0; q < S; ++q) sum = sum + col[b][q][ch];
                avg[ch] = sum * inv;
            }
            store_pixel(r, gx, gy, avg[0], avg[1], avg[2], avg[3]);
        }
    }
}
// (A RESIDENT grid — as many workgroups as stay on the machine at four wavefronts per SIMD, each looping over the places bid, bid + gridDim.x, ...
// of a falling tile order, so that the next frame's binning and tessellation workgroups find room beside it — was measured in round 4: dealt
// statically, the tiles balance worse than under the hardware's dynamic dispatch (0.29 - 0.31 against 0.22 ms), the loop around the kernel
// body costs 5 - 12 % by itself, and the front lanes gain nothing. Not kept.)

// ---------------------------------------------------------------------------------------------- k_raster_fill (round 5)
// The per-sample kernel of fill scenes at msaa 1 (no stroke triangles), with the walk re-stated (renderer.rs:304-318, 340-354, 565-582;
// shaders.wgsl:233-266, 304-309; vertex.rs:28-35): what k_raster_edges<1, 4, false, *> draws, entry for entry and bit for bit, but
//   * CLASS-BATCHED: a chunk's sorted entries are cut into groups  N* C  (N: edges, backdrop units, curve triangles; C: one cover). Inside a
//     group the N entries only add integers to the winding counters, so they commute: one ballot per class for the whole chunk, then per
//     group a homogeneous loop per class over the class mask (scalar bit scans) — no per-entry read of the entry's flags through LDS into a
//     scalar register and no chain of uniform branches per entry (round 4's ablation priced that dispatch at 15 vector + 45 scalar
//     instructions per walked entry, half of the kernel's scalar work). A group's backdrop units are summed on the scalar unit (population
//     counts of their masks) and applied with one add per sample row.
//   * PACKED COUNTERS: a sample keeps  fill + 65536 * hull  in ONE register (k_raster_rows' cell; exact while a tile's list has fewer than
//     16 384 entries — every entry moves a counter by two at most —, the host keeps frames with longer lists on k_raster_edges). A fill
//     edge or a curve triangle adds its accept bit with ONE add-with-carry (the compare's lane mask is the carry), four registers less per lane.
//   * The sample rows an edge's y range or a triangle's box leaves out are folded into the compare as a sign bit (two cheap vector
//     instructions per row) instead of lane masks built on the scalar unit (fifteen scalar instructions per entry: by the measured issue
//     rates, tools/valu_rate.hip, a scalar instruction costs a SIMD as much as a compare); a triangle's three edge tests are one saturating
//     subtract each, a three-way minimum and ONE compare.
// LONG: as k_raster_edges (the late start found across the chunks of a long list).
CRH_D int add_lane_bit(int v, unsigned long long mask) { // v + (this lane's bit of mask): the mask is the carry-in (in place: no copy at a loop's back edge)
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long carry;
    asm("v_addc_co_u32 %0, %1, %0, 0, %2" : "+v"(v), "=s"(carry) : "s"(mask));
#endif
    return v;
}
CRH_D int sub_lane_bit(int v, unsigned long long mask) { // v - (this lane's bit of mask)
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long carry;
    asm("v_subb_co_u32 %0, %1, %0, 0, %2" : "+v"(v), "=s"(carry) : "s"(mask));
#endif
    return v;
}
// One compare-exchange step of a sorting network on the 64 lanes' u32 keys, partner inside the 16-lane row: the partner's key comes in as a DPP
// operand of v_min_u32 / v_max_u32 themselves (no LDS permute, no address arithmetic); keep_min: the lanes that keep the smaller key.
// (s_nop 1: a DPP operand written by the VALU instruction in front needs two wait states, and the assembler does not see into the asm.)
#if defined(__HIP_DEVICE_COMPILE__)
#define CRH_CX_DPP(key_, keep_min_, ctrl_)                                                                                                  \
    {                                                                                                                                       \
        uint32_t lo_, hi_;                                                                                                                  \
        asm("s_nop 1\n\tv_min_u32_dpp %0, %2, %2 " ctrl_ " row_mask:0xf bank_mask:0xf\n\tv_max_u32_dpp %1, %2, %2 " ctrl_ " row_mask:0xf bank_mask:0xf" \
            : "=&v"(lo_), "=&v"(hi_)                                                                                                        \
            : "v"(key_));                                                                                                                   \
        key_ = __builtin_amdgcn_inverse_ballot_w64(keep_min_) ? lo_ : hi_;                                                                   \
    }
#else
#define CRH_CX_DPP(key_, keep_min_, ctrl_) { (void)(keep_min_); }
#endif
// ... partner = lane ^ 4 = quad mirror of the half-row mirror: one DPP move, then as above
CRH_D uint32_t cx_xor4(uint32_t key, unsigned long long keep_min) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x141, 0xF, 0xF, false); // row_half_mirror
    uint32_t lo, hi;
    asm("s_nop 1\n\tv_min_u32_dpp %0, %2, %3 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\tv_max_u32_dpp %1, %2, %3 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf"
        : "=&v"(lo), "=&v"(hi)
        : "v"(t), "v"(key));
    return __builtin_amdgcn_inverse_ballot_w64(keep_min) ? lo : hi;
#else
    return key;
#endif
}
// ... partner in another row (lane ^ 16, ^ 31, ^ 32, ^ 63): through the LDS crossbar
CRH_D uint32_t cx_far(uint32_t key, uint32_t lane, uint32_t xor_mask, unsigned long long keep_min) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t other = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane ^ xor_mask) << 2), (int)key);
    return __builtin_amdgcn_inverse_ballot_w64(keep_min) ? min(key, other) : max(key, other);
#else
    return key;
#endif
}
// The normalised bitonic network (every merge begins with a mirror step, so all exchanges keep the minimum in the lower lane) over the first
// `depth` lanes — 8, 16, 32 or 64 (wave uniform) —, ascending; lanes without a key hold 0xFFFFFFFF.
CRH_D uint32_t sort_keys_in_lanes(uint32_t key, uint32_t lane, uint32_t depth) {
    constexpr unsigned long long kBit0 = 0x5555555555555555ull, kBit1 = 0x3333333333333333ull, kBit2 = 0x0F0F0F0F0F0F0F0Full, kBit3 = 0x00FF00FF00FF00FFull,
                                 kBit4 = 0x0000FFFF0000FFFFull, kBit5 = 0x00000000FFFFFFFFull; // lanes whose bit b is clear
    CRH_CX_DPP(key, kBit0, "quad_perm:[1,0,3,2]") // kk = 2
    CRH_CX_DPP(key, kBit1, "quad_perm:[3,2,1,0]") // kk = 4: mirror, 1
    CRH_CX_DPP(key, kBit0, "quad_perm:[1,0,3,2]")
    CRH_CX_DPP(key, kBit2, "row_half_mirror")     // kk = 8: mirror, 2, 1
    CRH_CX_DPP(key, kBit1, "quad_perm:[2,3,0,1]")
    CRH_CX_DPP(key, kBit0, "quad_perm:[1,0,3,2]")
    if (depth > 8u) {
        CRH_CX_DPP(key, kBit3, "row_mirror")      // kk = 16: mirror, 4, 2, 1
        key = cx_xor4(key, kBit2);
        CRH_CX_DPP(key, kBit1, "quad_perm:[2,3,0,1]")
        CRH_CX_DPP(key, kBit0, "quad_perm:[1,0,3,2]")
    }
    if (depth > 16u) {
        key = cx_far(key, lane, 31u, kBit4);      // kk = 32: mirror, 8, 4, 2, 1
        CRH_CX_DPP(key, kBit3, "row_ror:8")
        key = cx_xor4(key, kBit2);
        CRH_CX_DPP(key, kBit1, "quad_perm:[2,3,0,1]")
        CRH_CX_DPP(key, kBit0, "quad_perm:[1,0,3,2]")
    }
    if (depth > 32u) {
        key = cx_far(key, lane, 63u, kBit5);      // kk = 64: mirror, 16, 8, 4, 2, 1
        key = cx_far(key, lane, 16u, kBit4);
        CRH_CX_DPP(key, kBit3, "row_ror:8")
        key = cx_xor4(key, kBit2);
        CRH_CX_DPP(key, kBit1, "quad_perm:[2,3,0,1]")
        CRH_CX_DPP(key, kBit0, "quad_perm:[1,0,3,2]")
    }
    return key;
}
// "over" on the lanes of mask[b] only (sample row b), the others keep their colour: the blends run under the masks as the EXEC mask — no
// select per channel —, dst = src + dst * (1 - alpha) as a multiply and an add (Rust does not contract), or dst = src for an opaque source.
// One asm statement for the four rows: the colour registers are updated in place (per row, the compiler moved them to temporaries and back).
CRH_D void blend_rows(float (&c)[4][4], const unsigned long long (&mask)[4], float s0, float s1, float s2, float s3, float one_minus_a, bool replace) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long saved;
    if (replace) {
        asm volatile("s_mov_b64 %[saved], exec\n\t"
                 "s_mov_b64 exec, %[m0]\n\t"
                 "v_mov_b32 %[c00], %[s0]\n\t"
                 "v_mov_b32 %[c01], %[s1]\n\t"
                 "v_mov_b32 %[c02], %[s2]\n\t"
                 "v_mov_b32 %[c03], %[s3]\n\t"
                 "s_mov_b64 exec, %[m1]\n\t"
                 "v_mov_b32 %[c10], %[s0]\n\t"
                 "v_mov_b32 %[c11], %[s1]\n\t"
                 "v_mov_b32 %[c12], %[s2]\n\t"
                 "v_mov_b32 %[c13], %[s3]\n\t"
                 "s_mov_b64 exec, %[m2]\n\t"
                 "v_mov_b32 %[c20], %[s0]\n\t"
                 "v_mov_b32 %[c21], %[s1]\n\t"
                 "v_mov_b32 %[c22], %[s2]\n\t"
                 "v_mov_b32 %[c23], %[s3]\n\t"
                 "s_mov_b64 exec, %[m3]\n\t"
                 "v_mov_b32 %[c30], %[s0]\n\t"
                 "v_mov_b32 %[c31], %[s1]\n\t"
                 "v_mov_b32 %[c32], %[s2]\n\t"
                 "v_mov_b32 %[c33], %[s3]\n\t"
                 "s_mov_b64 exec, %[saved]"
                 : [c00] "+v"(c[0][0]), [c01] "+v"(c[0][1]), [c02] "+v"(c[0][2]), [c03] "+v"(c[0][3]), [c10] "+v"(c[1][0]), [c11] "+v"(c[1][1]), [c12] "+v"(c[1][2]), [c13] "+v"(c[1][3]), [c20] "+v"(c[2][0]), [c21] "+v"(c[2][1]), [c22] "+v"(c[2][2]), [c23] "+v"(c[2][3]), [c30] "+v"(c[3][0]), [c31] "+v"(c[3][1]), [c32] "+v"(c[3][2]), [c33] "+v"(c[3][3]), [saved] "=&s"(saved)
                 : [m0] "s"(mask[0]), [m1] "s"(mask[1]), [m2] "s"(mask[2]), [m3] "s"(mask[3]), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2), [s3] "v"(s3));
    } else {
        asm volatile("s_mov_b64 %[saved], exec\n\t"
                 "s_mov_b64 exec, %[m0]\n\t"
                 "v_mul_f32 %[c00], %[c00], %[oma]\n\t"
                 "v_mul_f32 %[c01], %[c01], %[oma]\n\t"
                 "v_mul_f32 %[c02], %[c02], %[oma]\n\t"
                 "v_mul_f32 %[c03], %[c03], %[oma]\n\t"
                 "v_add_f32 %[c00], %[s0], %[c00]\n\t"
                 "v_add_f32 %[c01], %[s1], %[c01]\n\t"
                 "v_add_f32 %[c02], %[s2], %[c02]\n\t"
                 "v_add_f32 %[c03], %[s3], %[c03]\n\t"
                 "s_mov_b64 exec, %[m1]\n\t"
                 "v_mul_f32 %[c10], %[c10], %[oma]\n\t"
                 "v_mul_f32 %[c11], %[c11], %[oma]\n\t"
                 "v_mul_f32 %[c12], %[c12], %[oma]\n\t"
                 "v_mul_f32 %[c13], %[c13], %[oma]\n\t"
                 "v_add_f32 %[c10], %[s0], %[c10]\n\t"
                 "v_add_f32 %[c11], %[s1], %[c11]\n\t"
                 "v_add_f32 %[c12], %[s2], %[c12]\n\t"
                 "v_add_f32 %[c13], %[s3], %[c13]\n\t"
                 "s_mov_b64 exec, %[m2]\n\t"
                 "v_mul_f32 %[c20], %[c20], %[oma]\n\t"
                 "v_mul_f32 %[c21], %[c21], %[oma]\n\t"
                 "v_mul_f32 %[c22], %[c22], %[oma]\n\t"
                 "v_mul_f32 %[c23], %[c23], %[oma]\n\t"
                 "v_add_f32 %[c20], %[s0], %[c20]\n\t"
                 "v_add_f32 %[c21], %[s1], %[c21]\n\t"
                 "v_add_f32 %[c22], %[s2], %[c22]\n\t"
                 "v_add_f32 %[c23], %[s3], %[c23]\n\t"
                 "s_mov_b64 exec, %[m3]\n\t"
                 "v_mul_f32 %[c30], %[c30], %[oma]\n\t"
                 "v_mul_f32 %[c31], %[c31], %[oma]\n\t"
                 "v_mul_f32 %[c32], %[c32], %[oma]\n\t"
                 "v_mul_f32 %[c33], %[c33], %[oma]\n\t"
                 "v_add_f32 %[c30], %[s0], %[c30]\n\t"
                 "v_add_f32 %[c31], %[s1], %[c31]\n\t"
                 "v_add_f32 %[c32], %[s2], %[c32]\n\t"
                 "v_add_f32 %[c33], %[s3], %[c33]\n\t"
                 "s_mov_b64 exec, %[saved]"
                 : [c00] "+v"(c[0][0]), [c01] "+v"(c[0][1]), [c02] "+v"(c[0][2]), [c03] "+v"(c[0][3]), [c10] "+v"(c[1][0]), [c11] "+v"(c[1][1]), [c12] "+v"(c[1][2]), [c13] "+v"(c[1][3]), [c20] "+v"(c[2][0]), [c21] "+v"(c[2][1]), [c22] "+v"(c[2][2]), [c23] "+v"(c[2][3]), [c30] "+v"(c[3][0]), [c31] "+v"(c[3][1]), [c32] "+v"(c[3][2]), [c33] "+v"(c[3][3]), [saved] "=&s"(saved)
                 : [m0] "s"(mask[0]), [m1] "s"(mask[1]), [m2] "s"(mask[2]), [m3] "s"(mask[3]), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2), [s3] "v"(s3), [oma] "v"(one_minus_a));
    }
#else
    (void)c, (void)mask, (void)s0, (void)s1, (void)s2, (void)s3, (void)one_minus_a, (void)replace;
#endif
}
// bit `bit` of `rows` clear -> the sign bit set in x (a sample row that is left out fails every  x >= 0 / x >= 1  test): v_lshlrev_b32 + v_and_or_b32
CRH_D int reject_unless_row(int x, uint32_t not_rows, int bit) { return (int)(((not_rows << (31 - bit)) & 0x80000000u) | (uint32_t)x); }
// WAVES: wavefronts per SIMD the build is held to. Round 6: SEVEN (72 registers, no scratch memory), for every frame. Until then the kernel was built twice —
// five waves (91 registers, no scratch) and, for frames of long lists, six (80 registers, 20 B of scratch: a pair of scratch stores per wavefront, as many
// bytes as the tile it draws) — because LLVM's SLP vectorizer kept the lane-invariant operands of its v_pk_* instructions duplicated in register pairs for
// the whole kernel. Without that pass (build.py FILE_FLAGS) the kernel needs 76 registers at any occupancy, 72 when held to seven waves, and spills from
// eight on (64 registers, 12 B). Same box, S10k @ 4096^2 / 100 000 paths @ 8192^2, raster kernel alone and pipelined step in ms (profiles/r06_experiments.txt):
//   round 5 (SLP, 5 / 6 waves)  0.166, 0.302 / 1.008, 1.950      no SLP, held to 5 or 6   0.150, 0.298 - 0.306 / 0.977, 1.93 - 1.95
//   no SLP, held to 7           0.142, 0.289 - 0.293 / 0.920, 1.877                      held to 8 (spills)       0.142, 0.299 / 0.976, 1.956
#ifndef CRH_FILL_TILE_WAVES
#define CRH_FILL_TILE_WAVES 7
#endif
template <bool LONG, int WAVES = CRH_FILL_TILE_WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES))) void k_raster_fill(SceneDev s, RasterParams r) {
    constexpr int ROWS = 4;
    const uint32_t bid = blockIdx.x; // the workgroup's place in the frame's tile order
    extern __shared__ uint32_t sort_buffer[];
    __shared__ float4 entry_buffer[64 * 3];
    constexpr uint32_t kB = CRH_XCD_BLOCK_LOG2, kBlock = 1u << kB;
    const uint32_t turn = bid >> 3;
    const uint32_t blocks_x = (r.tiles_x + kBlock - 1u) >> kB, block = (turn >> (2u * kB)) * 8u + (bid & 7u);
    uint32_t tx = (block % blocks_x) * kBlock + (turn & (kBlock - 1u)), ty = (block / blocks_x) * kBlock + ((turn >> kB) & (kBlock - 1u));
    if (r.tile_order) { // the host's order for this frame: every XCD's heavy tiles first (api.hip order_tiles_heavy_first)
        const uint32_t mine = r.tile_order[bid];
        if (mine == 0xFFFFFFFFu) return;
        ty = mine / r.tiles_x, tx = mine - ty * r.tiles_x;
    }
    if (tx >= r.tiles_x || ty >= r.tiles_y || ty < r.slab_ty0 || ty >= r.slab_ty1) return; // (beyond the frame, or not in this pass' slab of tile rows)
    const uint32_t tile = ty * r.tiles_x + tx;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* __restrict__ keys = sort_buffer;
    const uint32_t px = lane & 15u, rq = lane >> 4;
    const uint32_t gx = tx * kTile + px;
    const float tile_y0 = (float)(ty * kTile); // (the walk derives the tile's origin again per chunk, below)
    const int tpx = (int)(tx * kTile), tpy = (int)(ty * kTile);
    const float sx = (float)px + 0.5f, sy0 = (float)rq + 0.5f; // the lane's samples: column px, rows rq + 4 b
    int cell[ROWS];        // fill winding + 65536 * hull winding of sample row rq + 4 b
    float col[ROWS][4];
#pragma unroll
    for (int b = 0; b < ROWS; ++b) {
        cell[b] = 0;
        col[b][0] = col[b][1] = col[b][2] = col[b][3] = 0.0f;
    }
    if (r.load_existing) {
#pragma unroll
        for (int b = 0; b < ROWS; ++b) {
            const uint32_t gy = ty * kTile + 4u * b + rq;
            if (gx < r.width && gy < r.height) {
                const float4 d = load_pixel(r, gx, gy);
                col[b][0] = d.x, col[b][1] = d.y, col[b][2] = d.z, col[b][3] = d.w;
            }
        }
    }
    const uint32_t list_begin = r.direct ? r.tile_base[tile] : r.tile_offset[tile];
    uint32_t n = (r.overflow[0] | r.overflow[5]) ? 0u : (r.direct ? r.tile_count[tile] : r.tile_offset[tile + 1] - list_begin);
    constexpr uint32_t kLdsSortMax = kSortBytesMax / 4u;
    {   // a list the sort buffer cannot hold: the host grows the buffer (overflow[3] = the longest list; without lists in place the scan of the
        // counts has published it) and runs the frame again. (n is decided by selects on scalars only, and pinned to a scalar register: joined
        // behind the lane-0 branch of the report, the compiler took it — and with it every branch of the walk — for lane dependent.)
        const bool too_long = n > r.sort_capacity && n <= kLdsSortMax;
        if (r.direct != 0u && (too_long || n > kLdsSortMax) && threadIdx.x == 0u) atomicMax(&r.overflow[3], n);
        n = __builtin_amdgcn_readfirstlane(too_long ? 0u : n);
    }
#ifdef CRH_ABLATE
    if (r.debug & 64u) n = 0;
#endif
    uint32_t my_key = 0xFFFFFFFFu;
    const bool sorted_in_place = n > kLdsSortMax;
    uint32_t* const segment = r.tile_list + list_begin;
    if (sorted_in_place) { // as k_raster_edges: a normalised bitonic network over the tile's segment of the list, in global memory
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
        for (uint32_t kk = 2; kk <= padded; kk <<= 1) {
            const uint32_t half = kk >> 1;
            for (uint32_t p = lane; p < (padded >> 1); p += 64u) {
                const uint32_t blk = p / half, t = p - blk * half;
                exchange(blk * kk + t, blk * kk + kk - 1u - t);
            }
            __threadfence();
            __syncthreads();
            for (uint32_t j = half >> 1; j > 0; j >>= 1) {
                for (uint32_t p = lane; p < (padded >> 1); p += 64u) {
                    const uint32_t i = 2u * j * (p / j) + (p % j);
                    exchange(i, i + j);
                }
                __threadfence();
                __syncthreads();
            }
        }
    } else if (n <= 64u) {
        if (lane < n) my_key = r.tile_list[list_begin + lane];
        if (n > 1u) my_key = sort_keys_in_lanes(my_key, lane, n <= 8u ? 8u : (n <= 16u ? 16u : (n <= 32u ? 32u : 64u))); // a network as deep as the list needs (the keys sit in the leading lanes)
    } else {
        uint32_t padded = 128;
        while (padded < n) padded <<= 1;
        for (uint32_t i = lane; i < padded; i += 64u) keys[i] = i < n ? r.tile_list[list_begin + i] : 0xFFFFFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t kk = 2; kk <= padded; kk <<= 1)
            for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                for (uint32_t i = lane; i < padded; i += 64u) {
                    const uint32_t partner = i ^ j;
                    if (partner > i) {
                        const uint32_t a = keys[i], b = keys[partner];
                        if (((i & kk) == 0) ? (a > b) : (a < b)) {
                            keys[i] = b;
                            keys[partner] = a;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
    }

    const uint8_t* slots = r.slots;
    const uint32_t wmask = r.winding_mask;
    // Cooperative row evaluation of an edge entry: lane j < 16 evaluates sample row j at the left tile boundary, lane 16 the backdrop row q0
    const uint32_t row_j = lane & 15u;
    const float ry_row = lane == 16u ? 0.5f : (float)row_j + 0.5f;
    const float sy_row = tile_y0 + ry_row;
    auto key_of = [&](uint32_t i) -> uint32_t { return sorted_in_place ? __hip_atomic_load(segment + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : keys[i]; };
    uint32_t walk_from = 0, verify_entry = 0xFFFFFFFFu; // absolute positions in the list
    if (LONG && n > 64u && !r.load_existing) { // the late start of a list of several chunks (k_raster_edges: X, the last opaque whole-tile cover; R, the last whole-tile reset before it)
        uint32_t x_at = 0xFFFFFFFFu;
        for (uint32_t c0 = ((n - 1u) >> 6) << 6;; c0 -= 64u) {
            const uint32_t k = c0 + lane < n ? key_of(c0 + lane) : 0xFFFFFFFFu;
            uint32_t code = 0;
            if (c0 + lane < n) {
                const uint32_t flags = *reinterpret_cast<const uint32_t*>(slots + (size_t)k * 32u);
                code = ((flags >> 4) & 15u) == EK_SYNTH ? (flags >> 8) & 31u : 0u;
            }
            unsigned long long resets = __builtin_amdgcn_ballot_w64(code >= 4u + kCoverHull);
            if (x_at == 0xFFFFFFFFu) {
                unsigned long long candidates = __builtin_amdgcn_ballot_w64(code >= 4u + kCoverOpaque);
                while (candidates) {
                    const uint32_t at = 63u - (uint32_t)__builtin_clzll(candidates);
                    const SynthRec sr = load_uniform(reinterpret_cast<const SynthRec*>(slots + (size_t)__builtin_amdgcn_readlane(k, at) * 32u));
                    uint32_t lo = 0, hi = n; // the number of keys below the item's synthetic slots
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (key_of(mid) < sr.synth_a) lo = mid + 1u;
                        else hi = mid;
                    }
                    if (lo == 0u || key_of(lo - 1u) < sr.first_slot) {
                        x_at = c0 + at;
                        resets &= (1ull << at) - 1ull;
                        break;
                    }
                    candidates &= ~(1ull << at);
                }
            }
            if (x_at != 0xFFFFFFFFu && resets) {
                walk_from = c0 + 64u - (uint32_t)__builtin_clzll(resets); // behind the last whole-tile reset before X
                verify_entry = x_at;
                break;
            }
            if (c0 == 0u) break;
        }
    }
    uint32_t first_j = walk_from & 63u;
    bool again_from_the_top = false;
    for (uint32_t q0 = LONG ? walk_from & ~63u : 0u; q0 < n; q0 += 64u) {
        // The tile's origin as floats, derived from the scalar tile coordinates HERE, per chunk: as values of the whole kernel the compiler kept
        // them duplicated in vector register pairs (operands of packed adds) and, in the six-wave build, spilled those — 16 bytes per lane, one
        // pair of scratch stores per wavefront: as many bytes as the tile it draws (WRITE_SIZE 1.95 x the frame at 8192^2).
        uint32_t tx_here = tx, ty_here = ty;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(tx_here), "+s"(ty_here));
#endif
        const float tx0 = (float)(tx_here * kTile), ty0 = (float)(ty_here * kTile);
        // (loads at a clamped index and a select, not a load under a lane-dependent condition: the loop has no divergent branch, see the set-up)
        if (sorted_in_place) {
            const uint32_t k_at = __hip_atomic_load(segment + min(q0 + lane, n - 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            my_key = q0 + lane < n ? k_at : 0xFFFFFFFFu;
        } else if (n > 64u) {
            const uint32_t k_at = keys[min(q0 + lane, n - 1u)];
            my_key = q0 + lane < n ? k_at : 0xFFFFFFFFu;
        }
        const uint32_t count = min(64u, n - q0);
        // ---- entry set-up, vectorised across the chunk: lane j prepares entry j (the records of k_raster_edges) and says what class it is.
        // WITHOUT a divergent branch: the three kinds of record are all decoded from the same 64 bytes and the results selected (the paths of
        // a three-way branch run one after the other anyway when a chunk holds all kinds). With it, this loop is a region of uniform branches
        // only, which the compiler leaves as written — behind ONE divergent branch it linearised the whole walk into guarded blocks with a
        // dozen register copies per entry where they meet.
        const uint32_t first_key = __builtin_amdgcn_readfirstlane(my_key);
        const uint32_t safe_key = lane < count ? my_key : first_key; // (a slot that exists: entry 0's)
        const uint4* slot16 = reinterpret_cast<const uint4*>(slots + (size_t)safe_key * 32u);
        const uint4 w0 = slot16[0], w1 = slot16[1];
        const uint32_t flags = w0.x, kind = (flags >> 4) & 15u;
        const bool is_edge = kind == EK_EDGE, is_synth = kind == EK_SYNTH;
        const uint4* tail16 = slot16 + ((is_edge || is_synth) ? 0 : 2); // the second half of a triangle's coverage record (the others read their own line again)
        const uint4 w2 = tail16[0], w3 = tail16[1];
        auto f = [](uint32_t v) { return __uint_as_float(v); };
        // a boundary edge (EdgeRec: flags, -, lo_x, lo_y | hi_x, hi_y, bx, nay)
        const float e_c = f(w1.z) * (ty0 - f(w0.w)) + f(w1.w) * (tx0 - f(w0.z));
        const bool xr = f(w0.z) <= tx0 && tx0 < f(w1.x); // the edge crosses the line of the left tile boundary
        const float e_ymin = fminf(f(w0.w), f(w1.y)), e_ymax = fmaxf(f(w0.w), f(w1.y));
        // a backdrop unit or a COVER (SynthRec: flags, first_slot, r, g | b, a, synth_a, -); a COVER's two folded backdrop units decoded here, once:
        // (code - 4) % 9 = bd + 1 + 3 (hbd + 1) -> eflags bits 16-17 and 18-19
        // (... with multiplications: a division the compiler would put behind a lane-dependent branch; exact for these small operands)
        const uint32_t code_p5 = ((flags >> 8) & 31u) + 5u, code9 = code_p5 - 9u * ((code_p5 * 57u) >> 9); // (code + 5) % 9, code + 5 <= 36
        const uint32_t code9_div3 = (code9 * 11u) >> 5, code9_mod3 = code9 - 3u * code9_div3;               // code9 <= 8
        // a curve triangle (PrimCoverage: flags, desc, box | lo_x[3], lo_y[3], bx[3], nay[3])
        const int px0 = (int)(w0.z & 0xFFFFu), px1 = (int)(w0.z >> 16), py0 = (int)(w0.w & 0xFFFFu), py1 = (int)(w0.w >> 16);
        const int bx0 = max(px0, tpx) - tpx, bx1 = min(px1, tpx + kTile - 1) - tpx;
        const int by0 = max(py0, tpy) - tpy, by1 = min(py1, tpy + kTile - 1) - tpy;
        const uint32_t col_bits = bx1 >= bx0 ? (2u << bx1) - (1u << bx0) : 0u, row_bits = by1 >= by0 ? (2u << by1) - (1u << by0) : 0u;
        const float cc0 = f(w2.z) * (ty0 - f(w1.w)) + f(w3.y) * (tx0 - f(w1.x));
        const float cc1 = f(w2.w) * (ty0 - f(w2.x)) + f(w3.z) * (tx0 - f(w1.y));
        const float cc2 = f(w3.x) * (ty0 - f(w2.y)) + f(w3.w) * (tx0 - f(w1.z));
        const float4 e0 = make_float4(is_edge ? e_c : (is_synth ? f(w0.z) : cc0), is_edge ? f(w1.z) : (is_synth ? f(w0.w) : cc1), is_edge ? f(w1.w) : (is_synth ? f(w1.x) : cc2),
                                      is_synth ? f(w1.y) : f(col_bits | (row_bits << 16)));
        const float4 e1 = make_float4(is_edge ? e_ymin : f(w2.z), is_edge ? e_ymax : f(w2.w), f(w3.x), 0.0f);
        const float4 e2 = make_float4(f(w3.y), f(w3.z), f(w3.w), 0.0f);
        // (bit 20: an opaque source none of whose components is -0 — `replace` below —, decided here so that the walk's branch on it is on a scalar)
        const bool opaque_source = f(w1.y) == 1.0f && w0.z != 0x80000000u && w0.w != 0x80000000u && w1.x != 0x80000000u;
        const uint32_t eflags = lane < count ? flags | (is_edge ? (xr ? 0x1000u : 0u) : (is_synth ? (code9_mod3 << 16) | (code9_div3 << 18) | (opaque_source ? 1u << 20 : 0u) : 0u)) : 15u << 4; // (no entry: no class)
        const uint32_t item_first = w0.y, item_synth_a = w1.z; // (read at COVER entries only)
        float4* __restrict__ entries = entry_buffer;
        __builtin_amdgcn_wave_barrier();
        entries[lane * 3u + 0u] = e0;
        entries[lane * 3u + 1u] = e1;
        entries[lane * 3u + 2u] = e2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- the chunk's classes, one ballot each
        const uint32_t ekind = (eflags >> 4) & 15u, ecode = (eflags >> 8) & 31u;
        const bool synth = ekind == EK_SYNTH;
        const unsigned long long m_cover = __builtin_amdgcn_ballot_w64(ekind == EK_COVER_TRI || (synth && ecode >= 4u));
        const unsigned long long m_edge = __builtin_amdgcn_ballot_w64(ekind == EK_EDGE);
        const unsigned long long m_tri = __builtin_amdgcn_ballot_w64(ekind >= KIND_IQ && ekind <= KIND_RC);
        const unsigned long long m_units = __builtin_amdgcn_ballot_w64(synth && ecode < 4u);       // backdrop units: fill +1 / -1 (codes 0, 1), hull +1 / -1 (2, 3)
        const unsigned long long m_odd = __builtin_amdgcn_ballot_w64(synth && (ecode & 1u) != 0u); // ... the -1 ones
        const unsigned long long m_hullu = __builtin_amdgcn_ballot_w64(synth && (ecode & 2u) != 0u); // ... the hull's
#ifdef CRH_ABLATE
        if (r.debug & 128u) continue;
#endif
        // sample row b of this lane inside a set-up triangle? (k_raster_tile's coverage — E_i = fma(y, bx_i, fma(x, nay_i, c_i)) accepted as
        // an integer >= 1 - top_left_i — as the lane mask of the samples inside: a saturating subtract per edge, a three-way minimum, the box
        // rows folded in as a sign bit, ONE compare)
        struct TriCoverage {
            float ha, hb, hc, bxa, bxb, bxc;
            int thr0, thr1, thr2;
            uint32_t not_rows; // bit 4 b clear: the lane's sample row b is inside the triangle's pixel box
        };
        auto tri_coverage = [&](const float4& ea4, const float4& eb4, const float4& ec4, uint32_t flags) {
            const uint32_t bits = __float_as_uint(ea4.w);
            TriCoverage t;
            t.not_rows = ((bits >> px) & 1u) ? ~((bits >> 16) >> rq) : ~0u;
            t.thr0 = 1 - (int)(flags & 1u), t.thr1 = 1 - (int)((flags >> 1) & 1u), t.thr2 = 1 - (int)((flags >> 2) & 1u);
            t.ha = fmaf(sx, ec4.x, ea4.x), t.hb = fmaf(sx, ec4.y, ea4.y), t.hc = fmaf(sx, ec4.z, ea4.z);
            t.bxa = eb4.x, t.bxb = eb4.y, t.bxc = eb4.z;
            return t;
        };
        auto inside_row = [&](const TriCoverage& t, int b) -> unsigned long long {
            const float y = sy0 + (float)(4 * b);
            const int xa = __builtin_elementwise_sub_sat(__float_as_int(fmaf(y, t.bxa, t.ha)), t.thr0);
            const int xb = __builtin_elementwise_sub_sat(__float_as_int(fmaf(y, t.bxb, t.hb)), t.thr1);
            const int xc = __builtin_elementwise_sub_sat(__float_as_int(fmaf(y, t.bxc, t.hc)), t.thr2);
            return __builtin_amdgcn_ballot_w64(reject_unless_row(min(xa, min(xb, xc)), t.not_rows, 4 * b) >= 0);
        };
        // Occlusion (k_raster_edges): X = the last opaque cover over the whole tile none of whose item's triangles are in the list; the walk starts
        // behind R, the last whole-tile reset in front of it, and X verifies that it overwrites every sample.
        uint32_t j0 = LONG ? first_j : 0u, verify_at = (LONG && verify_entry - q0 < 64u) ? verify_entry - q0 : 0xFFFFFFFFu;
        first_j = 0;
        if (n <= 64u && !r.load_existing) { // (the whole list is in this chunk, and the tile starts from a known colour)
            unsigned long long candidates = __builtin_amdgcn_ballot_w64(synth && ecode >= 4u + kCoverOpaque);
            const unsigned long long resets = __builtin_amdgcn_ballot_w64(synth && ecode >= 4u + kCoverHull);
            while (candidates) {
                const uint32_t at = 63u - (uint32_t)__builtin_clzll(candidates);
                const uint32_t first_slot = __builtin_amdgcn_readlane(item_first, at), synth_a = __builtin_amdgcn_readlane(item_synth_a, at);
                const uint32_t below = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(lane < count && my_key < synth_a));
                if (below == 0u || __builtin_amdgcn_readlane(my_key, below - 1u) < first_slot) {
                    const unsigned long long before = resets & ((1ull << at) - 1ull);
                    j0 = before ? 64u - (uint32_t)__builtin_clzll(before) : 0u; // behind the last whole-tile reset before X
                    verify_at = j0 ? at : 0xFFFFFFFFu;
                    break;
                }
                candidates &= ~(1ull << at);
            }
        }
        const unsigned long long all = count >= 64u ? ~0ull : (1ull << count) - 1ull;
        unsigned long long rest = j0 >= 64u ? 0ull : all & (~0ull << j0);
        while (rest) {
            const unsigned long long covers = m_cover & rest;
            const uint32_t end = covers ? (uint32_t)__builtin_ctzll(covers) : 64u;
            const unsigned long long group = end >= 64u ? rest : rest & ((1ull << end) - 1ull); // the N entries in front of the next cover
            // ---- backdrop units of the group: summed on the scalar unit
            const unsigned long long units = group & m_units;
            if (units) {
                const int fill_units = (int)__popcll(units & ~m_hullu & ~m_odd) - (int)__popcll(units & ~m_hullu & m_odd);
                const int hull_units = (int)__popcll(units & m_hullu & ~m_odd) - (int)__popcll(units & m_hullu & m_odd);
                int add = fill_units + hull_units * 65536;
#ifdef CRH_ABLATE
                if (r.debug & 16u) add = 0;
#endif
#pragma unroll
                for (int b = 0; b < ROWS; ++b) cell[b] += add;
            }
            // ---- boundary edges:  w(p) += sigma * [ Y_k g(p) + A_k ],  A_k = xr (g(q_k) - g(q_0)) - Y_k g(q_k)   (see the header of this file)
            unsigned long long edges = group & m_edge;
#ifdef CRH_ABLATE
            if (r.debug & 8u) edges = 0ull;
#endif
            while (edges) {
                const uint32_t j = (uint32_t)__builtin_ctzll(edges);
                edges &= edges - 1ull;
                const float4 ea4 = entries[j * 3u + 0u];
                const float2 yr = *reinterpret_cast<const float2*>(&entries[j * 3u + 1u]);
                const uint32_t flags = __builtin_amdgcn_readlane(eflags, j);
                const float c0 = ea4.x, ebx = ea4.y, enay = ea4.z, ymin = yr.x, ymax = yr.y;
                const int thr = 1 - (int)(flags & kEdgeTl);
                // the 16 sample rows + q0, one row per lane: g at the left tile boundary and the half-open y range
                const float eq = fmaf(ry_row, ebx, fmaf(0.0f, enay, c0));
                const unsigned long long gq_all = __builtin_amdgcn_ballot_w64(__float_as_int(eq) >= thr);
                const unsigned long long y_all = __builtin_amdgcn_ballot_w64(ymin <= sy_row) & __builtin_amdgcn_ballot_w64(sy_row < ymax);
                const uint32_t gq = (uint32_t)gq_all & 0xFFFFu, ym = (uint32_t)y_all & 0xFFFFu;
                const bool g0 = (((uint32_t)gq_all >> 16) & 1u) != 0u, xr = (flags & 0x1000u) != 0u;
                // A_k = xr (g(q_k) - g(q_0)) - Y_k g(q_k), three cases:  !xr: -(ym & gq);  xr & !g0: +(gq & ~ym);  xr & g0: -(~gq | ym)
                const uint32_t ygq = ym & gq;
                const uint32_t a_plus = (xr && !g0) ? gq ^ ygq : 0u;
                const uint32_t a_minus = xr ? (g0 ? (gq ^ 0xFFFFu) | ym : 0u) : ygq;
                const bool positive = (flags & kEdgeSigmaPos) != 0u; // sigma
                const int weight = (flags & kEdgeHull) ? 65536 : 1;  // the half of the cell the edge's chain belongs to
                const int unit = positive ? weight : -weight;
                const float h = fmaf(sx, enay, c0);
                const f32x2 y01 = {sy0, sy0 + 4.0f}, y23 = {sy0 + 8.0f, sy0 + 12.0f};
                const f32x2 ev01 = fma2(y01, splat2(ebx), f32x2{h, h}), ev23 = fma2(y23, splat2(ebx), f32x2{h, h});
                // Straight-line code from here (an if / else per property of the edge — hull or fill, sigma, "spans all rows" — came back from the
                // compiler as guarded blocks with the four counters copied at every join): the lanes whose sample row b takes the edge's unit,
                // Y_k g(p), with the rows outside the y range failing the compare through their sign bit
                const uint32_t not_ym = ~ym >> rq; // bit 4 b clear: the lane's sample row b is inside the edge's y range
                cell[0] += (reject_unless_row(__float_as_int(ev01[0]), not_ym, 0) >= thr) ? unit : 0;
                cell[1] += (reject_unless_row(__float_as_int(ev01[1]), not_ym, 4) >= thr) ? unit : 0;
                cell[2] += (reject_unless_row(__float_as_int(ev23[0]), not_ym, 8) >= thr) ? unit : 0;
                cell[3] += (reject_unless_row(__float_as_int(ev23[1]), not_ym, 12) >= thr) ? unit : 0;
                if (a_plus | a_minus) { // the edge crosses the left tile boundary (or runs left of it): row constants A_k sigma. Two bits per row —
                                        // 01: +1, 11: -1 — in one scalar word; the lane shifts its rows down and takes signed two-bit fields
                    const uint32_t up = positive ? a_plus : a_minus, down = positive ? a_minus : a_plus;
                    const int consts = (int)((((uint32_t)bit_double(up) & 0x55555555u) | (uint32_t)bit_double(down)) >> (2u * rq));
#pragma unroll
                    for (int b = 0; b < ROWS; ++b) cell[b] = __mul24(__builtin_amdgcn_sbfe(consts, 8u * b, 2u), weight) + cell[b];
                }
            }
            // ---- curve triangles: the four implicit-curve tests (shaders.wgsl:236-266), one straight-line variant per kind
            unsigned long long tris = group & m_tri;
#ifdef CRH_ABLATE
            if (r.debug & 32u) tris = 0ull;
#endif
            while (tris) {
                const uint32_t j = (uint32_t)__builtin_ctzll(tris);
                tris &= tris - 1ull;
                const uint32_t prim = __builtin_amdgcn_readlane(my_key, j);
                const uint32_t flags = __builtin_amdgcn_readlane(eflags, j);
                const PrimFragment frag = load_uniform(reinterpret_cast<const PrimFragment*>(slots + (size_t)prim * 32u + 64u));
                const float4 ea4 = entries[j * 3u + 0u], eb4 = entries[j * 3u + 1u], ec4 = entries[j * 3u + 2u];
                const uint32_t kind = (flags >> 4) & 15u;
                const TriCoverage cov = tri_coverage(ea4, eb4, ec4, flags);
                const int back = (flags & 8u) ? 0 : -1; // front (ccw on screen) increments, back decrements (renderer.rs:577-582)
                const uint32_t box_rows = __builtin_amdgcn_readfirstlane(__float_as_uint(ea4.w)) >> 16;
                const float dx0 = tx0 - frag.v0x, dy0 = ty0 - frag.v0y;
                auto curve = [&](auto kind_tag) {
                    constexpr uint32_t K = decltype(kind_tag)::value;
                    constexpr int NA = K == KIND_IQ ? 2 : (K == KIND_RC ? 4 : 3); // attributes the kind interpolates
                    float hx[NA]; // attribute planes, tile relative: the row-independent inner fma
#pragma unroll
                    for (int t = 0; t < NA; ++t) hx[t] = fmaf(sx, frag.gx[t], fmaf(dy0, frag.gy[t], fmaf(dx0, frag.gx[t], frag.a0[t])));
#pragma unroll
                    for (int b = 0; b < ROWS; ++b) {
                        if (((box_rows >> (4 * b)) & 15u) == 0u) continue; // none of the four sample rows 4 b .. 4 b + 3 is in the triangle's box
                        const unsigned long long inside = inside_row(cov, b);
                        if (inside == 0ull) continue; // curve triangles are small: most rows of the tile are not touched
                        const float y = sy0 + (float)(4 * b);
                        float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int t = 0; t < NA; ++t) a[t] = fmaf(y, frag.gy[t], hx[t]);
                        const float lhs = (K == KIND_IQ || K == KIND_RQ) ? a[0] * a[0] : a[0] * a[0] * a[0];
                        const float rhs = K == KIND_IQ ? a[1] : (K == KIND_RC ? a[1] * a[2] * a[3] : a[1] * a[2]);
                        const unsigned long long filled = inside & __builtin_amdgcn_ballot_w64(lhs - rhs <= 0.0f);
                        cell[b] = add_lane_bit(cell[b] ^ back, filled) ^ back; // ~(~c + bit) = c - bit
                    }
                };
                if (kind == KIND_IQ)
                    curve(std::integral_constant<uint32_t, KIND_IQ>{});
                else if (kind == KIND_IC)
                    curve(std::integral_constant<uint32_t, KIND_IC>{});
                else if (kind == KIND_RQ)
                    curve(std::integral_constant<uint32_t, KIND_RQ>{});
                else
                    curve(std::integral_constant<uint32_t, KIND_RC>{});
            }
            if (end >= 64u) break; // (the group runs on into the next chunk)
            rest = end >= 63u ? 0ull : rest & (~0ull << (end + 1u));
            // ---- entry `end` is a cover: color_cover (renderer.rs:340-354, 736-754) — blend where the winding is not zero, zero the winding of
            //      the covered samples; premultiplied "over" (shaders.wgsl:304-309, blending of examples/showcase/main.rs:32-43)
            const uint32_t flags = __builtin_amdgcn_readlane(eflags, end);
            const uint32_t kind = (flags >> 4) & 15u;
#ifdef CRH_ABLATE
            if ((r.debug & 16u) && kind == EK_SYNTH) continue;
#endif
            unsigned long long blend[ROWS]; // lanes whose sample row b takes the source
            float cs0, cs1, cs2, cs3;
            // replace: an opaque source over finite colours — src + dst * (1 - 1) is the source (r.occlude: every colour of the pass is tame; a source
            // component that is -0 would come out as +0 through the arithmetic, so it takes the arithmetic)
            bool replace;
            if (kind == EK_SYNTH) { // COVER over the samples inside the hull, one unit of both backdrops folded in
                replace = r.occlude != 0u && (flags & (1u << 20)) != 0u;
                const float4 ea4 = entries[end * 3u + 0u];
                const uint32_t code_all = (flags >> 8) & 31u;
                const int add = ((int)((flags >> 16) & 3u) - 1) + ((int)((flags >> 18) & 3u) - 1) * 65536;
                cs0 = ea4.x, cs1 = ea4.y, cs2 = ea4.z, cs3 = ea4.w;
                if (code_all >= 4u + kCoverHull) { // the whole tile lies inside the hull (no hull edge in the list, hull backdrop not zero): every sample is tested, every winding zeroed
#pragma unroll
                    for (int b = 0; b < ROWS; ++b) {
                        blend[b] = __builtin_amdgcn_ballot_w64((((uint32_t)(cell[b] + add)) & wmask) != 0u);
                        cell[b] = 0;
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < ROWS; ++b) {
                        const int t = cell[b] + add;
                        const int fill = __builtin_amdgcn_sbfe(t, 0u, 16u);
                        const unsigned long long in_hull = __builtin_amdgcn_ballot_w64(t != fill); // (65536 x the hull winding is what is left, whatever the sign of the fill's)
                        blend[b] = in_hull & __builtin_amdgcn_ballot_w64(((uint32_t)fill & wmask) != 0u);
                        cell[b] = __builtin_amdgcn_inverse_ballot_w64(in_hull) ? 0 : fill;
                    }
                }
                if (end == verify_at) { // X of the late start: does it overwrite every sample? (it does unless a sample inherited a winding)
                    verify_at = 0xFFFFFFFFu;
                    if ((blend[0] & blend[1] & blend[2] & blend[3]) != ~0ull || (r.debug & 33554432u) != 0u) { // no (or debug bit 25, tests: never trusted): the colours behind it matter — the whole list, from cleared state
#pragma unroll
                        for (int b = 0; b < ROWS; ++b) {
                            cell[b] = 0;
                            col[b][0] = col[b][1] = col[b][2] = col[b][3] = 0.0f;
                        }
                        if (LONG && n > 64u) { // the first chunk has to be set up again
                            verify_entry = 0xFFFFFFFFu;
                            again_from_the_top = true;
                            break;
                        }
                        rest = all;
                        continue;
                    }
                }
            } else { // a triangle of a folded hull strip, drawn as the reference draws it
                const uint32_t prim = __builtin_amdgcn_readlane(my_key, end);
                const PrimFragment frag = load_uniform(reinterpret_cast<const PrimFragment*>(slots + (size_t)prim * 32u + 64u));
                const float4 ea4 = entries[end * 3u + 0u], eb4 = entries[end * 3u + 1u], ec4 = entries[end * 3u + 2u];
                const TriCoverage cov = tri_coverage(ea4, eb4, ec4, flags);
                cs0 = frag.a0[0], cs1 = frag.a0[1], cs2 = frag.a0[2], cs3 = frag.a0[3];
                replace = r.occlude != 0u && cs3 == 1.0f && __float_as_uint(cs0) != 0x80000000u && __float_as_uint(cs1) != 0x80000000u && __float_as_uint(cs2) != 0x80000000u;
#pragma unroll
                for (int b = 0; b < ROWS; ++b) {
                    const unsigned long long inside = inside_row(cov, b);
                    const int fill = __builtin_amdgcn_sbfe(cell[b], 0u, 16u);
                    blend[b] = inside & __builtin_amdgcn_ballot_w64(((uint32_t)fill & wmask) != 0u);
                    cell[b] = __builtin_amdgcn_inverse_ballot_w64(inside) ? cell[b] - fill : cell[b]; // (the fill winding to zero, the hull's — zero in such items — as it is)
                }
            }
            const float one_minus_a = 1.0f - cs3;
            blend_rows(col, blend, cs0, cs1, cs2, cs3, one_minus_a, replace);
            if (r.format == CRH_FORMAT_RGBA8_ATTACHMENT) { // an Rgba8Unorm attachment keeps 8 bits of what the blender writes (idempotent on the others)
#pragma unroll
                for (int b = 0; b < ROWS; ++b)
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) col[b][ch] = attachment_unorm8(col[b][ch]);
            }
        } // groups of the chunk
        if (LONG && again_from_the_top) { // (X of the late start did not overwrite every sample: the whole list, chunk 0 first)
            again_from_the_top = false;
            q0 = 0u - 64u;
        }
    }
    // ---- RGBA8 unorm / binary16 store (the lane's column derived again from the scalar tile coordinate: kept from the top of the kernel it was the six-wave
    //      build's last spilled register)
    uint32_t tx_end = tx, all_lanes = ~0u;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(tx_end), "+s"(all_lanes)); // (opaque operands: neither value is the one computed at the top of the kernel and kept alive since)
    const uint32_t lane_end = __builtin_amdgcn_mbcnt_hi(all_lanes, __builtin_amdgcn_mbcnt_lo(all_lanes, 0u)); // the lane's number (= threadIdx.x: one wavefront per workgroup)
#else
    const uint32_t lane_end = threadIdx.x;
#endif
    const uint32_t gx_end = tx_end * kTile + (lane_end & 15u);
#pragma unroll
    for (int b = 0; b < ROWS; ++b) {
        const uint32_t gy = ty * kTile + 4u * b + rq;
        if (gx_end < r.width && gy < r.height) store_pixel(r, gx_end, gy, col[b][0], col[b][1], col[b][2], col[b][3]);
    }
}

// ---------------------------------------------------------------------------------------------- k_raster_rows (round 4)
// The same pass with the winding numbers ACCUMULATED IN LDS and the lanes spread over (entry, sample row) instead of every lane evaluating
// every entry on its own samples (renderer.rs:304-318, 340-354, 565-582; vertex.rs:28-35; shaders.wgsl:233-266, 304-309). msaa 1, no strokes.
//
// Along a sample row the contribution of an edge entry is a STEP function of the column: g_e = accepts(fma(ry, bx, fma(rx, nay, c))) is monotone
// in rx (round-to-nearest is monotone, twice), so  sigma * [Y_k g(j) + A_k]  is a row constant plus one unit from some switch column on.
// A lane that owns (edge, row) finds the switch column with five evaluations of the EXACT predicate (a lower bound over 16 columns) and makes
// two deposits into a 16 x 16 grid of the tile in DELTA form — the row constant at column 0, +-1 at the switch column; whole-tile backdrop
// units deposit a constant per row; a curve triangle is evaluated by lanes laid over its own pixel box (one sample per lane, the per-sample
// expressions of k_raster_edges) and deposits +-1 at (row, column) and -+1 at column + 1. A cell holds the fill winding in its low and the
// hull winding in its high 16 bits as ONE integer (fill + 65536 * hull): deposits and prefix sums are linear, the two are taken apart after
// the sum (exact while both stay below 2^15 in magnitude: a tile list of fewer than 32 768 entries cannot exceed it, longer ones go to
// k_raster_edges). What is serial is only what the reference orders: a tile's entries are cut into GROUPS  N* C+  (N: edges, backdrop
// units, curve triangles; C: cover entries — every item's covers follow all of its other entries in key order, and the N entries of an item
// without a cover in this tile simply stay in the winding, exactly as the stencil buffer keeps them); the deposits of up to kRowSlots groups
// go to as many grids at once (lanes packed across items: an item has three entries per tile on average), then the groups' cover entries are
// committed in order: the pixel owners (lane = row, four consecutive columns) read their four cells, prefix-sum along the row (three adds
// and two DPP steps inside the quad), add the winding left over from earlier items, test, blend, zero.
// tools/proto_edges.cpp (mode 2) proves the row form against oracle/raster.hpp on the CPU; the GPU parity suite checks this kernel.
#ifndef CRH_ROW_SLOTS
#define CRH_ROW_SLOTS 2
#endif
#ifndef CRH_ROW_TILE_WAVES
#define CRH_ROW_TILE_WAVES 6
#endif
constexpr uint32_t kRowSlots = CRH_ROW_SLOTS; // a power of two
CRH_D int quad_from(int v, int ctrl_0012_or_0101, bool shift_two) { // lane c of a quad <- lane c - 1 (c - 2); lanes without a source get their own value (masked by the caller)
#if defined(__HIP_DEVICE_COMPILE__)
    return shift_two ? __builtin_amdgcn_mov_dpp(v, 0x44, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(v, 0x90, 0xF, 0xF, true);
#else
    return v;
#endif
}
// inclusive prefix sum over the wavefront with DPP modifiers (no LDS): Hillis-Steele inside the 16-lane rows (lanes without a source keep the
// identity 0), then lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3
#if defined(__HIP_DEVICE_COMPILE__)
#define CRH_DPP(v_, ctrl_, rows_) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v_), ctrl_, rows_, 0xF, false))
#else
#define CRH_DPP(v_, ctrl_, rows_) (v_)
#endif
CRH_D uint32_t dpp_scan_add(uint32_t v) {
    v += CRH_DPP(v, 0x111, 0xF), v += CRH_DPP(v, 0x112, 0xF), v += CRH_DPP(v, 0x114, 0xF), v += CRH_DPP(v, 0x118, 0xF);
    v += CRH_DPP(v, 0x142, 0xA), v += CRH_DPP(v, 0x143, 0xC);
    return v;
}
// the first of the 16 columns of sample row y (x_j = j + 0.5) at which  accepts(E_j) != inv  — E = fma(y, bx, fma(x, nay, c)) is monotone in x,
// so the predicate switches at most once —, 16 if there is none: five evaluations of the exact predicate
CRH_D uint32_t switch_column(float y, float c, float bx, float nay, int thr, bool inv) {
    float lo = 0.0f;
#pragma unroll
    for (int step = 8; step >= 1; step >>= 1) {
        const float x = lo + ((float)step - 0.5f);
        const bool g = __float_as_int(fmaf(y, bx, fmaf(x, nay, c))) >= thr;
        lo = (g != inv) ? lo : lo + (float)step;
    }
    const bool g15 = __float_as_int(fmaf(y, bx, fmaf(15.5f, nay, c))) >= thr;
    return (lo == 15.0f && g15 == inv) ? 16u : (uint32_t)(int)lo;
}
// the first of 16 indices (as a float, 0 .. 15) at which a predicate that is monotone (false ... false true ... true) holds, 16 if none: five evaluations
template <class P>
CRH_D uint32_t first_of_16(P holds) {
    float lo = 0.0f;
#pragma unroll
    for (int step = 8; step >= 1; step >>= 1) lo = holds(lo + (float)(step - 1)) ? lo : lo + (float)step;
    return (lo == 15.0f && !holds(15.0f)) ? 16u : (uint32_t)(int)lo;
}
template <bool LONG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CRH_ROW_TILE_WAVES))) void k_raster_rows(SceneDev s, RasterParams r) {
    const uint32_t bid = blockIdx.x;
    extern __shared__ uint32_t sort_buffer[];
    __shared__ float4 entry_buffer[64 * 3];
    __shared__ uint32_t grid[kRowSlots][16][16]; // [group slot][sample row][column], delta form, fill + 65536 * hull
    __shared__ float4 frag_buffer[64 * 3];       // a curve triangle's attribute planes relative to the tile: constants, x gradients, y gradients
    __shared__ uint8_t edge_list[64], tri_list[64], slot_of[64];
    __shared__ uint16_t pair_start[66];          // [rank of a triangle in the chunk] its first (triangle, row) pair; [number of triangles] all pairs
    __shared__ uint16_t edge_pair_start[66];     // the same for the boundary edges: their (edge, row) pairs — the rows of the edge's half-open y range
    __shared__ uint32_t vgrid[kRowSlots][16];    // [group slot][sample row] what begins at that row and holds for every column of it and of the rows below (delta form down the tile)
    constexpr uint32_t kB = CRH_XCD_BLOCK_LOG2, kBlock = 1u << kB;
    const uint32_t turn = bid >> 3;
    const uint32_t blocks_x = (r.tiles_x + kBlock - 1u) >> kB, block = (turn >> (2u * kB)) * 8u + (bid & 7u);
    uint32_t tx = (block % blocks_x) * kBlock + (turn & (kBlock - 1u)), ty = (block / blocks_x) * kBlock + ((turn >> kB) & (kBlock - 1u));
    if (r.tile_order) { // the host's order for this frame: every XCD's heavy tiles first (api.hip order_tiles_heavy_first)
        const uint32_t mine = r.tile_order[bid];
        if (mine == 0xFFFFFFFFu) return;
        ty = mine / r.tiles_x, tx = mine - ty * r.tiles_x;
    }
    if (tx >= r.tiles_x || ty >= r.tiles_y || ty < r.slab_ty0 || ty >= r.slab_ty1) return; // (beyond the frame, or not in this pass' slab of tile rows)
    const uint32_t tile = ty * r.tiles_x + tx;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* __restrict__ keys = sort_buffer;
    // pixel owners: lane = (row k, quad c) owns the pixels (4c .. 4c + 3, k) of the tile: 16 bytes of a frame row
    const uint32_t row_k = lane >> 2, quad_c = lane & 3u;
    const float tx0 = (float)(tx * kTile), ty0 = (float)(ty * kTile);
    const int tpx = (int)(tx * kTile), tpy = (int)(ty * kTile);
    const uint32_t gx0 = tx * kTile + 4u * quad_c, gy = ty * kTile + row_k;
    const int quad_m1 = quad_c >= 1u ? -1 : 0, quad_m2 = quad_c >= 2u ? -1 : 0;
    int left[4]; // the winding the stencil buffer still holds from earlier items (non-zero only where an item's fill pokes out of its hull)
    bool left_any = false; // ... on some sample of the tile
    float col[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        left[i] = 0;
        col[i][0] = col[i][1] = col[i][2] = col[i][3] = 0.0f;
    }
    if (r.load_existing && gy < r.height) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (gx0 + (uint32_t)i < r.width) {
                const float4 d = load_pixel(r, gx0 + (uint32_t)i, gy);
                col[i][0] = d.x, col[i][1] = d.y, col[i][2] = d.z, col[i][3] = d.w;
            }
    }
    uint4* const my_cells = reinterpret_cast<uint4*>(&grid[0][row_k][4u * quad_c]); // + slot * 64 (uint4 units)
#pragma unroll
    for (uint32_t g = 0; g < kRowSlots; ++g) my_cells[g * 64u] = make_uint4(0u, 0u, 0u, 0u);
    if (lane < kRowSlots * 16u) (&vgrid[0][0])[lane] = 0u;
    const uint32_t list_begin = r.direct ? r.tile_base[tile] : r.tile_offset[tile];
    uint32_t n = (r.overflow[0] | r.overflow[5]) ? 0u : (r.direct ? r.tile_count[tile] : r.tile_offset[tile + 1] - list_begin);
    constexpr uint32_t kLdsSortMax = kSortBytesMax / 4u;
    if (n > r.sort_capacity && n <= kLdsSortMax) { // the host grows the sort buffer (overflow[3] = the longest list) and runs the frame again
        if (r.direct && threadIdx.x == 0u) atomicMax(&r.overflow[3], n);
        n = 0;
    } else if (n > kLdsSortMax && r.direct && threadIdx.x == 0u) {
        atomicMax(&r.overflow[3], n); // (the host takes frames with lists of 16 384 entries and more away from this kernel: its cells hold two 16-bit fields)
    }
    uint32_t my_key = 0xFFFFFFFFu;
    const bool sorted_in_place = n > kLdsSortMax;
    uint32_t* const segment = r.tile_list + list_begin;
    if (sorted_in_place) { // as k_raster_edges: a normalised bitonic network over the tile's segment of the list, in global memory
        const uint32_t tid = threadIdx.x, n_threads = 64u;
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
        for (uint32_t kk = 2; kk <= padded; kk <<= 1) {
            const uint32_t half = kk >> 1;
            for (uint32_t p = tid; p < (padded >> 1); p += n_threads) {
                const uint32_t blk = p / half, t = p - blk * half;
                exchange(blk * kk + t, blk * kk + kk - 1u - t);
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
        // (a network as deep as the list needs: most tiles hold fewer than 32 entries)
        const uint32_t depth = n <= 2u ? 2u : (n <= 4u ? 4u : (n <= 8u ? 8u : (n <= 16u ? 16u : (n <= 32u ? 32u : 64u))));
        if (n > 1u) {
#pragma unroll
            for (uint32_t kk = 2; kk <= 64u; kk <<= 1) {
                if (kk > depth) break;
#pragma unroll
                for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                    const uint32_t other = __shfl_xor(my_key, j, 64);
                    const bool keep_min = ((lane & j) == 0) == ((lane & kk) == 0);
                    my_key = keep_min ? min(my_key, other) : max(my_key, other);
                }
            }
        }
    } else {
        uint32_t padded = 128;
        while (padded < n) padded <<= 1;
        for (uint32_t i = lane; i < padded; i += 64u) keys[i] = i < n ? r.tile_list[list_begin + i] : 0xFFFFFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t kk = 2; kk <= padded; kk <<= 1)
            for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                for (uint32_t i = lane; i < padded; i += 64u) {
                    const uint32_t partner = i ^ j;
                    if (partner > i) {
                        const uint32_t a = keys[i], b = keys[partner];
                        if (((i & kk) == 0) ? (a > b) : (a < b)) {
                            keys[i] = b;
                            keys[partner] = a;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
    }

    const uint8_t* slots = r.slots;
    const int wmask = (int)r.winding_mask;
    auto key_of = [&](uint32_t i) -> uint32_t { return sorted_in_place ? __hip_atomic_load(segment + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : keys[i]; };
    // the late start of a list of several chunks: as k_raster_edges<.., LONG>
    uint32_t walk_from = 0, verify_entry = 0xFFFFFFFFu;
    if (LONG && n > 64u && !r.load_existing) {
        uint32_t x_at = 0xFFFFFFFFu;
        for (uint32_t c0 = ((n - 1u) >> 6) << 6;; c0 -= 64u) {
            const uint32_t k = c0 + lane < n ? key_of(c0 + lane) : 0xFFFFFFFFu;
            uint32_t code = 0;
            if (c0 + lane < n) {
                const uint32_t flags = *reinterpret_cast<const uint32_t*>(slots + (size_t)k * 32u);
                code = ((flags >> 4) & 15u) == EK_SYNTH ? (flags >> 8) & 31u : 0u;
            }
            unsigned long long resets = __builtin_amdgcn_ballot_w64(code >= 4u + kCoverHull);
            if (x_at == 0xFFFFFFFFu) {
                unsigned long long candidates = __builtin_amdgcn_ballot_w64(code >= 4u + kCoverOpaque);
                while (candidates) {
                    const uint32_t at = 63u - (uint32_t)__builtin_clzll(candidates);
                    const SynthRec sr = load_uniform(reinterpret_cast<const SynthRec*>(slots + (size_t)__builtin_amdgcn_readlane(k, at) * 32u));
                    uint32_t lo = 0, hi = n;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (key_of(mid) < sr.synth_a) lo = mid + 1u;
                        else hi = mid;
                    }
                    if (lo == 0u || key_of(lo - 1u) < sr.first_slot) {
                        x_at = c0 + at;
                        resets &= (1ull << at) - 1ull;
                        break;
                    }
                    candidates &= ~(1ull << at);
                }
            }
            if (x_at != 0xFFFFFFFFu && resets) {
                walk_from = c0 + 64u - (uint32_t)__builtin_clzll(resets);
                verify_entry = x_at;
                break;
            }
            if (c0 == 0u) break;
        }
    }
    uint32_t first_j = walk_from & 63u;
    bool again_from_the_top = false;
    bool prev_cover = true; // the entry in front of the chunk's first one was a cover (or there was none): the first entry opens a group
    uint32_t group_base = 0; // the group of the chunk's first entry (groups are numbered along the list; slot = group % kRowSlots)
    for (uint32_t q0 = LONG ? walk_from & ~63u : 0u; q0 < n; q0 += 64u) {
        if (sorted_in_place)
            my_key = q0 + lane < n ? __hip_atomic_load(segment + q0 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
        else if (n > 64u)
            my_key = q0 + lane < n ? keys[q0 + lane] : 0xFFFFFFFFu;
        const uint32_t count = min(64u, n - q0);
        // ---- entry setup, vectorised across the chunk: lane j prepares entry j (as k_raster_edges)
        float4 e0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), e1 = e0, e2 = e0;
        bool hull_over_tile = false, replaces_tile = false;
        uint32_t item_first = 0, item_synth_a = 0;
        uint32_t cls = 0; // 1: edge-like (a boundary edge or a backdrop unit), 2: curve triangle, 3: cover
        float4 f0 = e0, f1 = e0, f2 = e0;
        uint32_t my_rows = 0; // a curve triangle: the sample rows of its box inside the tile; a boundary edge: the rows of its y range
        uint32_t v_row = 16u; // an edge-like entry: the row from which on it adds v_value to every sample of the tile (16: none)
        int v_value = 0;
        if (lane < count) {
            const uint8_t* slot = slots + (size_t)my_key * 32u;
            const uint32_t flags = *reinterpret_cast<const uint32_t*>(slot);
            const uint32_t kind = (flags >> 4) & 15u;
            if (kind == EK_EDGE) {
                const EdgeRec er = *reinterpret_cast<const EdgeRec*>(slot);
                const float c = er.bx * (ty0 - er.lo_y) + er.nay * (tx0 - er.lo_x);
                const bool xr = er.lo_x <= tx0 && tx0 < er.hi_x; // the edge crosses the line of the left tile boundary
                // The edge's contribution to sample (row k, column j), sigma * [Y_k g(k, j) + xr (g(q_k) - g(q_0)) - Y_k g(q_k)], in two parts:
                //   sigma Y_k (g(k, j) - g(q_k))   lives in the rows of the edge's half-open y range [ky0, ky1): one (edge, row) lane each, below;
                //   sigma xr (g(q_k) - g(q_0))     is the same for every column, and g(q_k) is monotone in k (bx >= 0): a unit from the row on at
                //                                  which the edge function at the left tile boundary switches — ONE deposit down the tile.
                const float ymin = fminf(er.lo_y, er.hi_y), ymax = fmaxf(er.lo_y, er.hi_y);
                const uint32_t ky0 = first_of_16([&](float k) { return ymin <= ty0 + (k + 0.5f); }), ky1 = first_of_16([&](float k) { return !(ty0 + (k + 0.5f) < ymax); });
                my_rows = ky1 > ky0 ? ky1 - ky0 : 0u;
                const int thr = 1 - (int)(flags & kEdgeTl);
                const float at_left = fmaf(0.0f, er.nay, c);
                const int sigma_field = ((flags & kEdgeSigmaPos) ? 1 : -1) * ((flags & kEdgeHull) ? 65536 : 1);
                if (xr && !(__float_as_int(fmaf(0.5f, er.bx, at_left)) >= thr)) { // g(q_0) = 0: the unit begins at the first row with g(q_k) = 1
                    v_row = first_of_16([&](float k) { return __float_as_int(fmaf(k + 0.5f, er.bx, at_left)) >= thr; });
                    v_value = sigma_field;
                }
                e0 = make_float4(c, er.bx, er.nay, 0.0f);
                e1 = make_float4(ymin, ymax, __uint_as_float(ky0), __uint_as_float(flags | (xr ? 0x1000u : 0u)));
                cls = 1u;
            } else if (kind == EK_SYNTH) {
                const SynthRec sr = *reinterpret_cast<const SynthRec*>(slot);
                e0 = make_float4(sr.r, sr.g, sr.b, sr.a);
                e1.w = __uint_as_float(flags);
                const uint32_t code_all = (flags >> 8) & 31u;
                hull_over_tile = code_all >= 4u + kCoverHull;
                replaces_tile = code_all >= 4u + kCoverOpaque;
                item_first = sr.first_slot, item_synth_a = sr.synth_a;
                cls = code_all >= 4u ? 3u : 1u;
                if (code_all < 4u) v_row = 0u, v_value = ((code_all & 1u) ? -1 : 1) * (code_all < 2u ? 1 : 65536); // a whole-tile backdrop unit of the fill (codes 0, 1) or hull (2, 3) winding
            } else {
                const PrimCoverage mine = *reinterpret_cast<const PrimCoverage*>(slot);
                const int bx0 = max((int)mine.box.x, tpx) - tpx, bx1 = min((int)mine.box.y, tpx + kTile - 1) - tpx;
                const int by0 = max((int)mine.box.z, tpy) - tpy, by1 = min((int)mine.box.w, tpy + kTile - 1) - tpy;
                const uint32_t col_bits = bx1 >= bx0 ? (2u << bx1) - (1u << bx0) : 0u, row_bits = by1 >= by0 ? (2u << by1) - (1u << by0) : 0u;
                float cc[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) cc[i] = mine.bx[i] * (ty0 - mine.lo_y[i]) + mine.nay[i] * (tx0 - mine.lo_x[i]);
                e0 = make_float4(cc[0], cc[1], cc[2], __uint_as_float(col_bits | (row_bits << 16)));
                e1 = make_float4(mine.bx[0], mine.bx[1], mine.bx[2], __uint_as_float(flags));
                e2 = make_float4(mine.nay[0], mine.nay[1], mine.nay[2], 0.0f);
                cls = kind == EK_COVER_TRI ? 3u : (kind <= KIND_RC ? 2u : 0u); // (stroke triangles: not in a pass this kernel draws)
                if (cls == 2u) { // the fragment half: attribute planes through vertex 0 -> relative to the tile (the expressions of k_raster_edges)
                    const PrimFragment* frag = reinterpret_cast<const PrimFragment*>(slot + 64);
                    const float4 a0 = *reinterpret_cast<const float4*>(frag->a0), gxv = *reinterpret_cast<const float4*>(frag->gx), gyv = *reinterpret_cast<const float4*>(frag->gy);
                    const float dx0 = tx0 - frag->v0x, dy0 = ty0 - frag->v0y;
                    f0 = make_float4(fmaf(dy0, gyv.x, fmaf(dx0, gxv.x, a0.x)), fmaf(dy0, gyv.y, fmaf(dx0, gxv.y, a0.y)), fmaf(dy0, gyv.z, fmaf(dx0, gxv.z, a0.z)),
                                     fmaf(dy0, gyv.w, fmaf(dx0, gxv.w, a0.w)));
                    f1 = gxv, f2 = gyv;
                    my_rows = col_bits ? (uint32_t)__popc(row_bits) : 0u;
                }
            }
        }
        float4* __restrict__ entries = entry_buffer;
        __builtin_amdgcn_wave_barrier();
        entries[lane * 3u + 0u] = e0;
        entries[lane * 3u + 1u] = e1;
        entries[lane * 3u + 2u] = e2;
        if (cls == 2u) frag_buffer[lane * 3u + 0u] = f0, frag_buffer[lane * 3u + 1u] = f1, frag_buffer[lane * 3u + 2u] = f2;
        // ---- where the walk starts (the late start of k_raster_edges, found the same way) ...
        uint32_t j_start = LONG ? first_j : 0u, verify_at = (LONG && verify_entry - q0 < 64u) ? verify_entry - q0 : 0xFFFFFFFFu;
        first_j = 0;
        if (n <= 64u && !r.load_existing) {
            unsigned long long candidates = __builtin_amdgcn_ballot_w64(replaces_tile);
            const unsigned long long resets = __builtin_amdgcn_ballot_w64(hull_over_tile);
            while (candidates) {
                const uint32_t at = 63u - (uint32_t)__builtin_clzll(candidates);
                const uint32_t first_slot = __builtin_amdgcn_readlane(item_first, at), synth_a = __builtin_amdgcn_readlane(item_synth_a, at);
                const uint32_t below = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(lane < count && my_key < synth_a));
                if (below == 0u || __builtin_amdgcn_readlane(my_key, below - 1u) < first_slot) {
                    const unsigned long long before = resets & ((1ull << at) - 1ull);
                    j_start = before ? 64u - (uint32_t)__builtin_clzll(before) : 0u;
                    verify_at = j_start ? at : 0xFFFFFFFFu;
                    break;
                }
                candidates &= ~(1ull << at);
            }
        }
        const unsigned long long edge_all = __builtin_amdgcn_ballot_w64(cls == 1u), tri_all = __builtin_amdgcn_ballot_w64(cls == 2u), cover_all = __builtin_amdgcn_ballot_w64(cls == 3u);
        { // the chunk's edge-like entries and its triangles in list order, and where each triangle's (triangle, row) pairs begin
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t edge_pairs_end = edge_all ? dpp_scan_add(cls == 1u ? my_rows : 0u) : 0u;
            if (cls == 1u) {
                const uint32_t rank = (uint32_t)__popcll(edge_all & below);
                edge_list[rank] = (uint8_t)lane;
                edge_pair_start[rank] = (uint16_t)(edge_pairs_end - my_rows);
            }
            if (lane == 63u) edge_pair_start[__popcll(edge_all)] = (uint16_t)edge_pairs_end;
            const uint32_t pairs_end = tri_all ? dpp_scan_add(cls == 2u ? my_rows : 0u) : 0u;
            if (cls == 2u) {
                const uint32_t rank = (uint32_t)__popcll(tri_all & below);
                tri_list[rank] = (uint8_t)lane;
                pair_start[rank] = (uint16_t)(pairs_end - my_rows);
            }
            if (lane == 63u) pair_start[__popcll(tri_all)] = (uint16_t)pairs_end;
        }
        bool chunk_again;
        do { // (run once; twice when X of the late start turns out not to overwrite every sample)
            chunk_again = false;
            // ---- ... which entries take part, their groups  N* C+  and the slot of each group's grid
            const unsigned long long live = j_start ? ~((1ull << j_start) - 1ull) : ~0ull; // (j_start < 64: a reset lies in front of it)
            const unsigned long long cover_mask = cover_all & live;
            const unsigned long long in_walk = (count == 64u ? ~0ull : (1ull << count) - 1ull) & live;
            // a group begins at an entry that is no cover and follows a cover (the chunk's first entry: follows the previous chunk's last)
            const unsigned long long after_cover = (cover_all << 1) | (((j_start == 0u && !prev_cover) ? 0ull : 1ull) << j_start);
            const unsigned long long begins = after_cover & ~cover_all & in_walk;
            {
                const unsigned long long upto = lane == 63u ? ~0ull : (2ull << lane) - 1ull;
                const uint32_t group = group_base + (uint32_t)__popcll(begins & upto);
                slot_of[lane] = (uint8_t)(group & (kRowSlots - 1u));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            uint32_t pos = j_start;
            while (pos < count) {
                // ---- a round: the entries of the next kRowSlots groups
                uint32_t round_end = count;
                {
                    unsigned long long later = pos >= 63u ? 0ull : begins & ~((2ull << pos) - 1ull); // groups that begin behind `pos`
#pragma unroll
                    for (uint32_t g = 1; g < kRowSlots; ++g) later &= later - 1ull; // drop kRowSlots - 1 of them (0 & -1 stays 0)
                    if (later) round_end = (uint32_t)__builtin_ctzll(later);
                }
                const unsigned long long below_pos = (1ull << pos) - 1ull, below_end = round_end == 64u ? ~0ull : (1ull << round_end) - 1ull;
#ifdef CRH_ABLATE // tools/count_rows.py: how often does each part of the kernel run?
                if ((r.debug & 256u) && lane == 0u) {
                    atomicAdd(&r.overflow[8], 1u);
                    atomicAdd(&r.overflow[10], (uint32_t)__popcll(edge_all & below_end & ~below_pos));
                    atomicAdd(&r.overflow[9], ((uint32_t)__popcll(edge_all & below_end & ~below_pos) + 3u) / 4u);
                    atomicAdd(&r.overflow[11], (uint32_t)__popcll(tri_all & below_end & ~below_pos));
                    atomicAdd(&r.overflow[13], (uint32_t)__popcll(cover_mask & below_end & ~below_pos));
                    if (pos == j_start && q0 == 0u) atomicAdd(&r.overflow[14], 1u);
                }
#endif
                // ---- deposits of the edge-like entries. (a) lane = entry: what an entry adds to whole rows (a backdrop unit; an edge's switch at the
                //      left tile boundary) goes down the tile in delta form; (b) lane = (edge, row of its y range), the pairs of the round's edges
                //      laid end to end: the switch column of the row, two deposits along it.
                if (v_row < 16u && ((below_end & ~below_pos) >> lane) & 1ull)
                    __hip_atomic_fetch_add(&vgrid[slot_of[lane]][v_row], (uint32_t)v_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t e_lo = (uint32_t)__popcll(edge_all & below_pos), e_hi = (uint32_t)__popcll(edge_all & below_end); // (pos >= j_start: ranks among all of the chunk's)
                if (e_lo < e_hi) {
                    const uint32_t p_lo = __builtin_amdgcn_readfirstlane((uint32_t)edge_pair_start[e_lo]), p_hi = __builtin_amdgcn_readfirstlane((uint32_t)edge_pair_start[e_hi]);
                    uint32_t first_step = 1;
                    while (first_step * 2u < e_hi - e_lo) first_step *= 2u;
                    for (uint32_t p0 = p_lo; p0 < p_hi; p0 += 64u) {
#ifdef CRH_ABLATE
                        if ((r.debug & 256u) && lane == 0u) atomicAdd(&r.overflow[9], 1u);
#endif
                        const uint32_t pair = p0 + lane;
                        if (pair < p_hi) {
                            uint32_t rank = e_lo; // the last edge of the round whose pairs begin at or before `pair`
                            for (uint32_t step = first_step; step >= 1u; step >>= 1) {
                                const uint32_t cand = rank + step;
                                if (cand < e_hi && (uint32_t)edge_pair_start[min(cand, 64u)] <= pair) rank = cand;
                            }
                            const uint32_t idx = edge_list[rank];
                            const float4 ea4 = entries[idx * 3u + 0u], eb4 = entries[idx * 3u + 1u];
                            const uint32_t flags = __float_as_uint(eb4.w);
                            const uint32_t row = (__float_as_uint(eb4.z) + (pair - (uint32_t)edge_pair_start[rank])) & 15u;
                            const float ry = (float)row + 0.5f;
                            const float c0 = ea4.x, ebx = ea4.y, enay = ea4.z;
                            const int thr = 1 - (int)(flags & kEdgeTl);
                            const int gqk = __float_as_int(fmaf(ry, ebx, fmaf(0.0f, enay, c0))) >= thr ? 1 : 0;
                            const bool inv = enay < 0.0f; // g falls along the row: search the first column where it is 0
                            const uint32_t sw = switch_column(ry, c0, ebx, enay, thr, inv);
                            const int g_left = ((sw == 0u) != inv) ? 1 : 0;
                            const int sigma_field = ((flags & kEdgeSigmaPos) ? 1 : -1) * ((flags & kEdgeHull) ? 65536 : 1);
                            uint32_t* const cells = &grid[slot_of[idx]][row][0];
                            if (g_left != gqk) __hip_atomic_fetch_add(cells, (uint32_t)((g_left - gqk) * sigma_field), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (sw > 0u && sw < 16u) __hip_atomic_fetch_add(cells + sw, (uint32_t)(inv ? -sigma_field : sigma_field), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
                // ---- deposits of the curve triangles: lane = (triangle, sample row of its box). The columns of the row that lie inside the triangle
                //      are an interval — each edge function switches once along the row —, found with three searches; the lane then walks the
                //      interval, decides every sample with the implicit-curve test (shaders.wgsl:236-266, the expressions of k_raster_edges)
                //      and deposits +-delta where a run of accepted samples begins and ends. (Measured against the samples of all intervals laid
                //      end to end as tasks, one per lane: fewer instructions on large triangles, more phases — slower where this kernel is used.)
                const uint32_t t_lo = (uint32_t)__popcll(tri_all & below_pos), t_hi = (uint32_t)__popcll(tri_all & below_end);
                if (t_lo < t_hi) {
                    const uint32_t p_lo = __builtin_amdgcn_readfirstlane((uint32_t)pair_start[t_lo]), p_hi = __builtin_amdgcn_readfirstlane((uint32_t)pair_start[t_hi]);
                    uint32_t first_step = 1;
                    while (first_step * 2u < t_hi - t_lo) first_step *= 2u;
                    for (uint32_t p0 = p_lo; p0 < p_hi; p0 += 64u) {
#ifdef CRH_ABLATE
                        if ((r.debug & 256u) && lane == 0u) atomicAdd(&r.overflow[12], 1u), atomicAdd(&r.overflow[15], min(64u, p_hi - p0));
#endif
                        const uint32_t pair = p0 + lane;
                        const bool valid = pair < p_hi;
                        uint32_t rank = t_lo; // the last triangle of the round whose pairs begin at or before `pair`
                        for (uint32_t step = first_step; step >= 1u; step >>= 1) {
                            const uint32_t cand = rank + step;
                            if (cand < t_hi && (uint32_t)pair_start[min(cand, 64u)] <= pair) rank = cand;
                        }
                        const uint32_t idx = tri_list[rank];
                        const float4 ea4 = entries[idx * 3u + 0u], eb4 = entries[idx * 3u + 1u], ec4 = entries[idx * 3u + 2u];
                        const uint32_t flags = __float_as_uint(eb4.w), bits = __float_as_uint(ea4.w);
                        const uint32_t col_bits = bits & 0xFFFFu, row_bits = bits >> 16;
                        const uint32_t row = ((uint32_t)__builtin_ctz(row_bits | 0x10000u) + (pair - (uint32_t)pair_start[rank])) & 15u;
                        const float y = (float)row + 0.5f;
                        uint32_t c_lo = (uint32_t)__builtin_ctz(col_bits | 0x10000u), c_hi = c_lo + (uint32_t)__popc(col_bits);
                        {
                            const bool i0 = ec4.x < 0.0f, i1 = ec4.y < 0.0f, i2 = ec4.z < 0.0f;
                            const uint32_t s0 = switch_column(y, ea4.x, eb4.x, ec4.x, 1 - (int)(flags & 1u), i0);
                            const uint32_t s1 = switch_column(y, ea4.y, eb4.y, ec4.y, 1 - (int)((flags >> 1) & 1u), i1);
                            const uint32_t s2 = switch_column(y, ea4.z, eb4.z, ec4.z, 1 - (int)((flags >> 2) & 1u), i2);
                            c_lo = max(c_lo, max(i0 ? 0u : s0, max(i1 ? 0u : s1, i2 ? 0u : s2)));
                            c_hi = min(c_hi, min(i0 ? s0 : 16u, min(i1 ? s1 : 16u, i2 ? s2 : 16u)));
                        }
                        if (!valid || c_lo >= c_hi) c_lo = 1u, c_hi = 0u; // nothing to walk
                        const float4 fc = frag_buffer[idx * 3u + 0u], fx = frag_buffer[idx * 3u + 1u], fy = frag_buffer[idx * 3u + 2u];
                        const uint32_t kind = (flags >> 4) & 15u;
                        const bool square = kind == KIND_IQ || kind == KIND_RQ;
                        const int delta = (flags & 8u) ? 1 : -1; // front (ccw on screen) increments, back decrements (renderer.rs:577-582)
                        uint32_t* const cells = &grid[slot_of[idx]][row][0];
                        uint32_t j = c_lo;
                        float x = (float)c_lo + 0.5f;
                        bool prev = false;
                        while (__builtin_amdgcn_ballot_w64(j <= c_hi) != 0ull) { // (one column past the interval: a run that reaches its end is closed there)
                            if (j <= c_hi) {
                                const float a0 = fmaf(y, fy.x, fmaf(x, fx.x, fc.x)), a1 = fmaf(y, fy.y, fmaf(x, fx.y, fc.y));
                                const float a2 = fmaf(y, fy.z, fmaf(x, fx.z, fc.z)), a3 = fmaf(y, fy.w, fmaf(x, fx.w, fc.w));
                                const float sq = a0 * a0, m12 = a1 * a2;
                                const float lhs = square ? sq : sq * a0;
                                const float rhs = kind == KIND_IQ ? a1 : (kind == KIND_RC ? m12 * a3 : m12);
                                const bool acc = j < c_hi && lhs - rhs <= 0.0f;
                                if (acc != prev && j < 16u) __hip_atomic_fetch_add(cells + j, (uint32_t)(acc ? delta : -delta), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                prev = acc;
                            }
                            j += 1u;
                            x += 1.0f;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // ---- the covers of the round, in order: commit the group's grid, color_cover (renderer.rs:340-354, 736-754; shaders.wgsl:304-309)
                unsigned long long covers = cover_mask & below_end & ~below_pos;
                // a cover finds deposits in its group's grid iff the entry in front of it is no cover (the group's later covers find it committed)
                const unsigned long long dirty_covers = cover_all & ((~cover_all << 1) | ((j_start == 0u && !prev_cover) ? 1ull : 0ull));
                bool restart = false;
                while (covers) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(covers);
                    covers &= covers - 1ull;
                    const float4 ea4 = entries[j * 3u + 0u], eb4 = entries[j * 3u + 1u], ec4 = entries[j * 3u + 2u];
                    const uint32_t flags = __builtin_amdgcn_readfirstlane(__float_as_uint(eb4.w));
                    const uint32_t kind = (flags >> 4) & 15u;
                    const bool dirty = ((dirty_covers >> j) & 1ull) != 0ull;
                    bool blend[4];
                    float cs0, cs1, cs2, cs3;
                    if (!dirty && !left_any && kind == EK_SYNTH) {
                        // Nothing was deposited for this cover and no sample carries a winding: fill = hull = 0 on the whole tile, so the hull test is
                        // "hbd != 0" and the stencil test "bd != 0 under the winding rule" for every sample alike (an item over the whole tile)
                        const uint32_t code = 4u + (((flags >> 8) & 31u) - 4u) % 9u;
                        const int bd = (int)((code - 4u) % 3u) - 1, hbd = (int)((code - 4u) / 3u) - 1;
                        const bool all = hbd != 0 && (bd & wmask) != 0;
                        if (j == verify_at) {
                            verify_at = 0xFFFFFFFFu;
                            if (!all || (r.debug & 33554432u) != 0u) {
                                restart = true;
                                break;
                            }
                        }
                        if (!all) continue; // (the winding stays zero either way)
                        cs0 = ea4.x, cs1 = ea4.y, cs2 = ea4.z, cs3 = ea4.w;
                        const bool replace = r.occlude != 0u && cs3 == 1.0f && __float_as_uint(cs0) != 0x80000000u && __float_as_uint(cs1) != 0x80000000u && __float_as_uint(cs2) != 0x80000000u;
                        const float one_minus_a = 1.0f - cs3;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            col[i][0] = replace ? cs0 : cs0 + col[i][0] * one_minus_a, col[i][1] = replace ? cs1 : cs1 + col[i][1] * one_minus_a;
                            col[i][2] = replace ? cs2 : cs2 + col[i][2] * one_minus_a, col[i][3] = replace ? cs3 : cs3 + col[i][3] * one_minus_a;
                        }
                        if (r.format == CRH_FORMAT_RGBA8_ATTACHMENT) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int ch = 0; ch < 4; ++ch) col[i][ch] = attachment_unorm8(col[i][ch]);
                        }
                        continue;
                    }
                    int p[4] = {0, 0, 0, 0};
                    if (dirty) {
                        uint4* const mine = my_cells + (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)slot_of[j]) * 64u;
                        const uint4 d = *mine;
                        *mine = make_uint4(0u, 0u, 0u, 0u);
                        uint32_t* const my_down = &vgrid[0][row_k] + (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)slot_of[j]) * 16u;
                        uint32_t down = *my_down; // what begins at this row for all columns: summed down the tile (the four lanes of a row hold the same)
                        *my_down = 0u;
                        down += CRH_DPP(down, 0x114, 0xF), down += CRH_DPP(down, 0x118, 0xF), down += CRH_DPP(down, 0x142, 0xA), down += CRH_DPP(down, 0x143, 0xC);
                        // prefix sum along the row: inside the lane, then over the four lanes of the row
                        const int total = (int)(d.x + d.y + d.z + d.w);
                        int scan = total + (quad_from(total, 0, false) & quad_m1);
                        scan = scan + (quad_from(scan, 0, true) & quad_m2);
                        p[0] = scan - total + (int)d.x + (int)down, p[1] = p[0] + (int)d.y, p[2] = p[1] + (int)d.z, p[3] = p[2] + (int)d.w;
                    }
                    if (kind == EK_SYNTH) { // COVER over the samples inside the hull, one unit of both backdrops folded in
                        const uint32_t code = 4u + (((flags >> 8) & 31u) - 4u) % 9u;
                        const int bd = (int)((code - 4u) % 3u) - 1, hbd = (int)((code - 4u) / 3u) - 1;
                        cs0 = ea4.x, cs1 = ea4.y, cs2 = ea4.z, cs3 = ea4.w;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int fill = __builtin_amdgcn_sbfe(p[i], 0u, 16u), hull = (p[i] - fill) >> 16;
                            const int w = left[i] + fill + bd;
                            const bool in_hull = hull + hbd != 0;
                            blend[i] = in_hull && (w & wmask) != 0;
                            left[i] = in_hull ? 0 : w;
                        }
                        if (j == verify_at) { // X of the late start: does it overwrite every sample? (it does unless a sample inherited a winding)
                            const bool every = blend[0] && blend[1] && blend[2] && blend[3];
                            verify_at = 0xFFFFFFFFu;
                            if (__builtin_amdgcn_ballot_w64(every) != ~0ull || (r.debug & 33554432u) != 0u) { // no (or debug bit 25): the colours behind it matter — the whole list, from cleared state
                                restart = true;
                                break;
                            }
                        }
                    } else { // a triangle of a folded hull strip, drawn as the reference draws it
                        const uint32_t prim = __builtin_amdgcn_readlane(my_key, j);
                        const PrimFragment frag = load_uniform(reinterpret_cast<const PrimFragment*>(slots + (size_t)prim * 32u + 64u));
                        const uint32_t bits = __builtin_amdgcn_readfirstlane(__float_as_uint(ea4.w));
                        const int thr0 = 1 - (int)(flags & 1u), thr1 = 1 - (int)((flags >> 1) & 1u), thr2 = 1 - (int)((flags >> 2) & 1u);
                        const float y = (float)row_k + 0.5f;
                        const bool row_in = ((bits >> (16u + row_k)) & 1u) != 0u;
                        cs0 = frag.a0[0], cs1 = frag.a0[1], cs2 = frag.a0[2], cs3 = frag.a0[3];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t px = 4u * quad_c + (uint32_t)i;
                            const float x = (float)px + 0.5f;
                            const float ea = fmaf(y, eb4.x, fmaf(x, ec4.x, ea4.x)), eb = fmaf(y, eb4.y, fmaf(x, ec4.y, ea4.y)), ec = fmaf(y, eb4.z, fmaf(x, ec4.z, ea4.z));
                            const bool inside = row_in && ((bits >> px) & 1u) != 0u && (__float_as_int(ea) >= thr0) & (__float_as_int(eb) >= thr1) & (__float_as_int(ec) >= thr2);
                            const int w = left[i] + __builtin_amdgcn_sbfe(p[i], 0u, 16u);
                            blend[i] = inside && (w & wmask) != 0;
                            left[i] = inside ? 0 : w;
                        }
                    }
                    left_any = __builtin_amdgcn_ballot_w64((left[0] | left[1] | left[2] | left[3]) != 0) != 0ull;
                    // an opaque source over finite colours: src + dst * (1 - 1) is the source (r.occlude: every colour of the pass is tame; a
                    // source component that is -0 would come out as +0 through the arithmetic, so it takes the arithmetic)
                    const bool replace = r.occlude != 0u && cs3 == 1.0f && __float_as_uint(cs0) != 0x80000000u && __float_as_uint(cs1) != 0x80000000u && __float_as_uint(cs2) != 0x80000000u;
                    if (replace) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            col[i][0] = blend[i] ? cs0 : col[i][0];
                            col[i][1] = blend[i] ? cs1 : col[i][1];
                            col[i][2] = blend[i] ? cs2 : col[i][2];
                            col[i][3] = blend[i] ? cs3 : col[i][3];
                        }
                    } else {
                        const float one_minus_a = 1.0f - cs3;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float n0 = cs0 + col[i][0] * one_minus_a, n1 = cs1 + col[i][1] * one_minus_a;
                            const float n2 = cs2 + col[i][2] * one_minus_a, n3 = cs3 + col[i][3] * one_minus_a;
                            col[i][0] = blend[i] ? n0 : col[i][0];
                            col[i][1] = blend[i] ? n1 : col[i][1];
                            col[i][2] = blend[i] ? n2 : col[i][2];
                            col[i][3] = blend[i] ? n3 : col[i][3];
                        }
                    }
                    if (r.format == CRH_FORMAT_RGBA8_ATTACHMENT) { // an Rgba8Unorm attachment keeps 8 bits of what the blender writes (idempotent on the others)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int ch = 0; ch < 4; ++ch) col[i][ch] = attachment_unorm8(col[i][ch]);
                    }
                }
                if (restart) { // (X of the late start did not overwrite every sample: everything again, from cleared state, the shortcut off)
                    left_any = false;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        left[i] = 0;
                        col[i][0] = col[i][1] = col[i][2] = col[i][3] = 0.0f;
                    }
#pragma unroll
                    for (uint32_t g = 0; g < kRowSlots; ++g) my_cells[g * 64u] = make_uint4(0u, 0u, 0u, 0u);
                    if (lane < kRowSlots * 16u) (&vgrid[0][0])[lane] = 0u;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    prev_cover = true, group_base = 0;
                    if (LONG && n > 64u) {
                        verify_entry = 0xFFFFFFFFu;
                        again_from_the_top = true;
                    } else {
                        j_start = 0;
                        chunk_again = true;
                    }
                    break;
                }
                pos = round_end;
            }
            if (!chunk_again && !again_from_the_top) {
                group_base += (uint32_t)__popcll(begins); // = the group of the chunk's last entry: it may go on in the next chunk
                prev_cover = ((cover_all >> (count - 1u)) & 1ull) != 0ull;
            }
            __builtin_amdgcn_wave_barrier(); // the lists and slots of this pass over the chunk are read: they may be rewritten
        } while (chunk_again);
        if (LONG && again_from_the_top) { // the whole list, chunk 0 first
            again_from_the_top = false;
            q0 = 0u - 64u;
        }
    }
    // ---- RGBA8 unorm / RGBA16F store: 16 (32) bytes of a frame row per lane
    if (gy < r.height) {
        if (r.format != CRH_FORMAT_RGBA16F && gx0 + 3u < r.width) {
            uint4 px;
            px.x = pack_unorm8(col[0][0], col[0][1], col[0][2], col[0][3]), px.y = pack_unorm8(col[1][0], col[1][1], col[1][2], col[1][3]);
            px.z = pack_unorm8(col[2][0], col[2][1], col[2][2], col[2][3]), px.w = pack_unorm8(col[3][0], col[3][1], col[3][2], col[3][3]);
            *reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(r.rgba8) + (size_t)gy * r.width + gx0) = px;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (gx0 + (uint32_t)i < r.width) store_pixel(r, gx0 + (uint32_t)i, gy, col[i][0], col[i][1], col[i][2], col[i][3]);
        }
    }
}
// ---------------------------------------------------------------------------------------------- launchers
void launch_scan_u32(const uint32_t* in, uint32_t* out, uint32_t* block_sum, uint32_t n, hipStream_t stream); // raster.hip

// slots per draw item (shapes in the plain pass) and their exclusive scan: slot_begin[n_items + 1]
void launch_slot_ranges(const SceneDev& s, const RasterParams& r, uint32_t n_items, uint32_t* item_nslots, uint32_t* slot_begin, uint32_t* scratch, hipStream_t stream) {
    if (n_items == 0) {
        (void)hipMemsetAsync(slot_begin, 0, 4, stream);
        return;
    }
    hipLaunchKernelGGL(k_item_nslots, dim3((n_items + 255u) / 256u), dim3(256), 0, stream, s, r, n_items, item_nslots);
    launch_scan_u32(item_nslots, slot_begin, scratch, n_items, stream);
}
// The plain pass' two per-Shape ranges at the end of a tessellation — contiguous primitive ids (triangle pass) and slot ranges (edge pass) —
// in three launches instead of six (counts of both, then the two scans side by side): small kernels on a lane that starves beside the
// binning and raster kernels pay for every launch.
// The tail of the tessellation lane — per Shape: candidate triangles and slots, then both prefixes — as SINGLE-WAVE workgroups (round 6). The lane of frame i + 1 runs beside
// the raster kernel of frame i, and frame i + 1's binning waits for its end: as 256-thread workgroups k_shape_counts (6 us of work) found no four free wave slots on one compute
// unit until the seven-wave raster grid had drained — 130 us in the rocprofv3 timeline (gpurun_out/r06_trace_steady.txt) —, and the binning started 37 us behind the raster kernel's end.
__global__ __launch_bounds__(64) void k_shape_counts(SceneDev s, uint32_t* shape_ncand, uint32_t* shape_nslots) {
    const uint32_t shape = blockIdx.x * 64u + threadIdx.x;
    if (shape >= s.n_shapes) return;
    uint32_t c[8];
    shape_ncand[shape] = shape_candidates(s, shape, c);
    RasterParams plain = {}; // items == nullptr: item i is Shape i, Stencil + Color
    shape_nslots[shape] = item_slots(s, item_of(plain, shape)).total;
}
// two exclusive prefixes of n items each (blockIdx.y picks the pair), 512 items per single-wave workgroup; out[n] = the total
struct WaveScan {
    const uint32_t* in;
    uint32_t* out;
    uint32_t* block_sum;
};
constexpr uint32_t kWaveScanItems = 8, kWaveScanBlock = 64 * kWaveScanItems;
__global__ __launch_bounds__(64) void k_wave_scan_local2(WaveScan a, WaveScan b, uint32_t n) {
    const WaveScan j = blockIdx.y ? b : a;
    const uint32_t lane = threadIdx.x, i0 = blockIdx.x * kWaveScanBlock + lane * kWaveScanItems;
    uint32_t v[kWaveScanItems], mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < kWaveScanItems; ++k) v[k] = i0 + k < n ? j.in[i0 + k] : 0u, mine += v[k];
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += up;
    }
    uint32_t run = incl - mine;
#pragma unroll
    for (uint32_t k = 0; k < kWaveScanItems; ++k) {
        if (i0 + k < n) j.out[i0 + k] = run;
        run += v[k];
    }
    if (lane == 63u) j.block_sum[blockIdx.x] = incl;
}
__global__ __launch_bounds__(64) void k_wave_scan_add2(WaveScan a, WaveScan b, uint32_t n, uint32_t blocks) {
    const WaveScan j = blockIdx.y ? b : a;
    const uint32_t lane = threadIdx.x;
    uint32_t sum = 0;
    for (uint32_t k = lane; k < blockIdx.x; k += 64u) sum += j.block_sum[k];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d, 64);
    const uint32_t i0 = blockIdx.x * kWaveScanBlock + lane * kWaveScanItems;
#pragma unroll
    for (uint32_t k = 0; k < kWaveScanItems; ++k)
        if (i0 + k < n) j.out[i0 + k] += sum;
    if (blockIdx.x + 1u == blocks && lane == 0u) j.out[n] = sum + j.block_sum[blockIdx.x];
}
// (scratch0 / scratch1: (n_shapes + 511) / 512 block sums each)
void launch_plain_ranges(const SceneDev& s, uint32_t* shape_ncand, uint32_t* shape_prim_begin, uint32_t* shape_nslots, uint32_t* shape_slot_begin, uint32_t* scratch0, uint32_t* scratch1, hipStream_t stream) {
    if (s.n_shapes == 0) {
        (void)hipMemsetAsync(shape_prim_begin, 0, 4, stream);
        (void)hipMemsetAsync(shape_slot_begin, 0, 4, stream);
        return;
    }
    hipLaunchKernelGGL(k_shape_counts, dim3((s.n_shapes + 63u) / 64u), dim3(64), 0, stream, s, shape_ncand, shape_nslots);
    const uint32_t blocks = (s.n_shapes + kWaveScanBlock - 1u) / kWaveScanBlock;
    const WaveScan a = {shape_ncand, shape_prim_begin, scratch0}, b = {shape_nslots, shape_slot_begin, scratch1};
    hipLaunchKernelGGL(k_wave_scan_local2, dim3(blocks, 2), dim3(64), 0, stream, a, b, s.n_shapes);
    hipLaunchKernelGGL(k_wave_scan_add2, dim3(blocks, 2), dim3(64), 0, stream, a, b, s.n_shapes, blocks);
}
// The items of a pass cut into runs of consecutive items that fill ONE batch of k_bin_flat each (cost: what a verified pass wrote to
// RasterParams::item_cost). A workgroup's life is a chain of barrier-separated phases per batch, whatever the batch holds: with equal
// NUMBERS of items per workgroup the ones with large Shapes took two to four batches (10 000 Shapes of 16-256 px: the longest workgroup
// lived 1.84 x the mean, and the kernel lasts as long as that one); with one full batch per workgroup every workgroup lives one chain and
// the hardware's dispatch balances the rest. runs[2 k], runs[2 k + 1] = the first item of run k and the one behind its last.
// The lanes of k_bin_flat's workgroups for a pass of n_items items: 128 (alone the kernel is fastest with 256, in the gap between two raster kernels
// with 128: §4.3 of DESIGN.md), 64 from 65 536 items on (100 000 paths @ 8192^2: pipelined step 2.05 -> 1.94 ms; 50 000 glyphs and the metric's
// 10 000 paths are slower that way). CRH_BIN_FLAT_THREADS pins it (64 / 128; read per pass).
uint32_t flat_threads_for(uint32_t n_items) {
    if (const char* e = getenv("CRH_BIN_FLAT_THREADS")) return atoi(e) == 64 ? 64u : kFlatThreads;
    return (n_items >= 65536u && kFlatThreads != 64u) ? 64u : kFlatThreads;
}
bool bin_itemwise(const RasterParams& r) { // (read per launch: tests and A/B runs switch it inside one process)
    const FlatShape shape = flat_shape(flat_threads_for(r.n_items));
    return getenv("CRH_BIN_ITEMWISE") != nullptr ||
           (getenv("CRH_BIN_FLAT") == nullptr && r.n_items != 0u && (r.hint_tris / r.n_items > shape.tris / 2u || r.hint_edges / r.n_items > shape.edges / 2u));
}
void flat_batches(const uint32_t* cost, uint32_t n_items, std::vector<uint32_t>& runs) {
    const FlatShape shape = flat_shape(flat_threads_for(n_items));
    const uint32_t kFlatBatch = shape.batch, kFlatTris = shape.tris, kFlatEdges = shape.edges, kFlatPool = shape.pool; // (of the kernel that will take these runs)
    struct Run {
        uint32_t first, last;
        float ticks;
    };
    std::vector<Run> all;
    const float cap = getenv("CRH_BIN_BATCH_TICKS") ? (float)atof(getenv("CRH_BIN_BATCH_TICKS")) : 0.0f; // A/B runs: close a run at this predicted life as well
    const uint32_t most = getenv("CRH_BIN_BATCH_ITEMS") ? (uint32_t)std::max(1, atoi(getenv("CRH_BIN_BATCH_ITEMS"))) : kFlatBatch; // ... or at this many items
    uint32_t n = 0, tris = 0, edges = 0, cells = 0, widest = 0, folded = 0, first = 0;
    bool closed = false;
    // a workgroup's life in shader clocks, fitted to the lifetimes tools/bin_phases.py dumps (10 000 Shapes / 50 000 glyphs): the chain of
    // phases, then what grows with the batch — the walks over (edge, tile row) pairs, the pool's cells, the longest triangle's tile box
    // (it goes with the widest rectangle), hull strips that fold (item by item)
    auto ticks = [&]() { return 76000.0f + 2000.0f * (float)n + 40.0f * (float)tris + 250.0f * (float)edges + 60.0f * (float)cells + 45.0f * (float)widest + 8000.0f * (float)folded; };
    for (uint32_t i = 0; i < n_items; ++i) {
        const uint32_t c = cost[2u * i], w = cost[2u * i + 1u];
        const bool alone = c == 0xFFFFFFFFu; // wider than the pool: k_bin_flat hands it on when it is the first of a batch
        const uint32_t t = (w >> 31) ? 0u : (w & 0x1FFu), e = (w >> 31) ? 0u : ((w >> 9) & 0x3FFu);
        if (n != 0u && (alone || closed || n >= std::min(most, kFlatBatch) || tris + t > kFlatTris || edges + e > kFlatEdges || cells + c > kFlatPool || (cap > 0.0f && ticks() > cap))) {
            all.push_back(Run{first, i, ticks()});
            n = tris = edges = cells = widest = folded = 0u, first = i, closed = false;
        }
        n += 1u, tris += t, edges += e, cells += alone ? 0u : c, widest = std::max(widest, alone ? 0u : c), folded += (w >> 29) & 1u;
        if (alone) closed = true; // (the next item opens a run)
    }
    if (n_items) all.push_back(Run{first, n_items, ticks()});
    // the long runs first: the hardware starts workgroups in grid order, and a long one that starts late ends after everything else
    const bool in_order = getenv("CRH_BIN_BATCH_ORDER") != nullptr; // A/B runs: the runs in item order
    if (!in_order) std::stable_sort(all.begin(), all.end(), [](const Run& a, const Run& b) { return a.ticks > b.ticks; });
    runs.clear();
    for (const Run& run : all) runs.push_back(run.first), runs.push_back(run.last);
}
void flat_batch_limits(uint32_t n_items, uint32_t limits[4]) {
    const FlatShape shape = flat_shape(flat_threads_for(n_items));
    limits[0] = shape.batch, limits[1] = shape.tris, limits[2] = shape.edges, limits[3] = shape.pool;
}
void launch_bin_edges(const SceneDev& s, const RasterParams& r, uint32_t samples, hipStream_t stream, void (*mark)(void*, const char*, uint64_t), void* ctx, hipEvent_t after_bin) {
    // tile_count and, right behind it, the overflow words (overflow[8 ...] are the cursors of the pair sub-streams): one memset (tile_cursor, in front, is the triangle pass')
    (void)hipMemsetAsync(r.tile_count, 0, sizeof(uint32_t) * r.n_tiles + 32 + 4 * kSubStreams + 32, stream); // (... and kExtraTurnsWord behind them)
    // Items per workgroup. One is best while the grid is small (S10k: 0.169 ms; two: 0.189, four: 0.21 — an item is a chain of dependent
    // memory operations, and a wavefront that takes a second item doubles it); tens of thousands of small items are bound by workgroup
    // turnover instead (50 000 glyphs: one 0.45, two 0.31, four 0.31, eight 0.33 ms). So: about 12 000 workgroups.
    const uint32_t pinned = getenv("CRH_BIN_ITEMS") ? max(1, atoi(getenv("CRH_BIN_ITEMS"))) : 0u; // (read per launch: tests and A/B runs switch it inside one process)
    // k_bin_edges for every item: CRH_BIN_ITEMWISE (A/B runs, tests), or a pass whose AVERAGE item is beyond what a batch of k_bin_flat holds
    // (the dashed strokes of config 5: a thousand line triangles per Shape) — every item would be queued anyway
    const bool itemwise = bin_itemwise(r);
    if (r.n_items && itemwise) {
        const uint32_t items_per_group = pinned ? pinned : min(8u, max(1u, (r.n_items + 12287u) / 12288u));
        const uint32_t bin_grid = (r.n_items + items_per_group - 1u) / items_per_group;
        if (samples == 4)
            hipLaunchKernelGGL((k_bin_edges<4, false>), dim3(bin_grid), dim3(128), 0, stream, s, r);
        else
            hipLaunchKernelGGL((k_bin_edges<1, false>), dim3(bin_grid), dim3(128), 0, stream, s, r);
    } else if (r.n_items) {
        // k_bin_flat: a batch of items per 256-thread workgroup. A workgroup lives about as long whether it holds two items or twenty (the
        // same chain of phases), so the grid is sized to ONE round of resident workgroups — three per CU — as long as that leaves a batch
        // within the kernel's 32 items; what a batch cannot hold is queued and binned item by item behind it.
        const FlatShape shape = flat_shape(flat_threads_for(r.n_items));
        const uint32_t resident = (getenv("CRH_BIN_CUS") ? (uint32_t)max(1, atoi(getenv("CRH_BIN_CUS"))) : 256u) * (4u * CRH_FLAT_WAVES) / (shape.threads / 64u); // workgroups the CUs of the binning lane hold at once (four SIMDs of CRH_FLAT_WAVES wavefronts each)
        // ... and within what the lanes of a batch hold (threads triangles, three times as many edges): an item that does not fit is left to the
        // workgroup's next turn, which doubles the workgroup's life — the averages of the scene keep a batch nine tenths full
        const uint32_t by_tris = r.hint_tris ? (uint32_t)((uint64_t)shape.tris * 9u / 10u * r.n_items / r.hint_tris) : shape.batch;
        const uint32_t by_edges = r.hint_edges ? (uint32_t)((uint64_t)shape.edges * 9u / 10u * r.n_items / r.hint_edges) : shape.batch;
        const uint32_t fitting = max(1u, min(shape.batch, min(by_tris, by_edges)));
        const uint32_t items_per_group = pinned ? min(pinned, shape.batch) : min(fitting, max(1u, (r.n_items + resident - 1u) / resident));
        const uint32_t flat_grid = r.bin_batches ? r.n_bin_batches : (r.n_items + items_per_group - 1u) / items_per_group, queue_grid = min(r.n_items, 4096u);
        if (samples == 4) {
            if (shape.threads == 64u) hipLaunchKernelGGL((k_bin_flat<4, 64u>), dim3(flat_grid), dim3(64), 0, stream, s, r, items_per_group);
            else hipLaunchKernelGGL((k_bin_flat<4, kFlatThreads>), dim3(flat_grid), dim3(kFlatThreads), 0, stream, s, r, items_per_group);
            if (!r.skip_queue) hipLaunchKernelGGL((k_bin_edges<4, true>), dim3(queue_grid), dim3(128), 0, stream, s, r);
        } else {
            if (shape.threads == 64u) hipLaunchKernelGGL((k_bin_flat<1, 64u>), dim3(flat_grid), dim3(64), 0, stream, s, r, items_per_group);
            else hipLaunchKernelGGL((k_bin_flat<1, kFlatThreads>), dim3(flat_grid), dim3(kFlatThreads), 0, stream, s, r, items_per_group);
            if (!r.skip_queue) hipLaunchKernelGGL((k_bin_edges<1, true>), dim3(queue_grid), dim3(128), 0, stream, s, r);
        }
    }
    if (after_bin) (void)hipEventRecord(after_bin, stream);
    if (mark) mark(ctx, "raster_bin", 0);
    if (!r.direct) launch_scan_tiles(r, stream);
    if (mark) mark(ctx, "raster_tile_scan", 0);
}
void launch_scatter(const RasterParams& r, hipStream_t stream, void (*mark)(void*, const char*, uint64_t), void* ctx) {
    if (r.pair_capacity && !r.direct) hipLaunchKernelGGL(k_scatter, dim3((r.pair_capacity + 255u) / 256u), dim3(256), 0, stream, r);
    if (mark) mark(ctx, "raster_scatter", 0);
}
// the places of the next frames' lists: caps[t] = count[t] + count[t] / 2 + 64, summed into tile_base[0 .. n_tiles] (tile_base[n_tiles] = all of them)
// (+ 64: a Shape whose boundary moves into an empty tile brings a dozen or two entries at once — with + 16, rounds 3 and 4, a zoom of 1 % per
// frame outgrew some list every few frames)
constexpr uint32_t kListSlack = 64u;
// ... of the LONGEST list within `radius` tiles (round 5): a camera that moves shifts the content by whole tiles between two passes into the same target —
// a zoom of 1 % per frame about the centre of 4096^2 moves the border by 40 pixels from one pass into a target to the next —, so a tile's next
// list resembles a neighbour's, not its own. 49 counts per tile out of L2, beside the raster kernel.
// SINGLE-WAVE workgroups, all three kernels of the chain (round 6): they run beside the raster kernel of the pass whose counts they read, and behind them on the same stream
// wait the next pass' instance copies and its binning. With k_raster_fill at seven waves per SIMD (504 of a SIMD's 512 registers) a 256-thread workgroup — four wavefronts that
// must find room on ONE compute unit at the same moment — got no slot until that grid had drained: k_tile_caps lasted 138 us, exactly as long as the raster kernel beside it
// (rocprofv3 timeline, gpurun_out/r06_trace_moved.txt), and a scene that moves paid 0.32 - 0.34 ms per step where round 5's five-wave raster kernel had left it 0.30.
// One wavefront takes the first slot that frees.
__global__ __launch_bounds__(64) void k_tile_caps(const uint32_t* count, uint32_t* caps, uint32_t n, uint32_t tiles_x, uint32_t radius) {
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= n) return;
    uint32_t longest = count[t];
    if (radius) {
        const uint32_t tiles_y = n / tiles_x, ty = t / tiles_x, tx = t - ty * tiles_x;
        const uint32_t x0 = tx > radius ? tx - radius : 0u, x1 = min(tiles_x - 1u, tx + radius), y0 = ty > radius ? ty - radius : 0u, y1 = min(tiles_y - 1u, ty + radius);
        for (uint32_t y = y0; y <= y1; ++y)
            for (uint32_t x = x0; x <= x1; ++x) longest = max(longest, count[y * tiles_x + x]);
    }
    caps[t] = longest + (longest >> 1) + kListSlack;
}
// exclusive prefix of caps -> tile_base[0 .. n], tile_base[n] = the total: 512 items per single-wave workgroup (eight a lane), then every workgroup adds the sums in front of it
constexpr uint32_t kBaseItems = 8, kBaseBlock = 64 * kBaseItems;
__global__ __launch_bounds__(64) void k_tile_base_local(const uint32_t* caps, uint32_t* base, uint32_t* block_sum, uint32_t n) {
    const uint32_t lane = threadIdx.x, i0 = blockIdx.x * kBaseBlock + lane * kBaseItems;
    uint32_t v[kBaseItems], mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < kBaseItems; ++k) v[k] = i0 + k < n ? caps[i0 + k] : 0u, mine += v[k];
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= (uint32_t)d) incl += up;
    }
    uint32_t run = incl - mine;
#pragma unroll
    for (uint32_t k = 0; k < kBaseItems; ++k) {
        if (i0 + k < n) base[i0 + k] = run;
        run += v[k];
    }
    if (lane == 63u) block_sum[blockIdx.x] = incl;
}
__global__ __launch_bounds__(64) void k_tile_base_add(uint32_t* base, const uint32_t* block_sum, uint32_t n, uint32_t blocks) {
    const uint32_t lane = threadIdx.x;
    uint32_t sum = 0;
    for (uint32_t k = lane; k < blockIdx.x; k += 64u) sum += block_sum[k];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d, 64);
    const uint32_t i0 = blockIdx.x * kBaseBlock + lane * kBaseItems;
#pragma unroll
    for (uint32_t k = 0; k < kBaseItems; ++k)
        if (i0 + k < n) base[i0 + k] += sum;
    if (blockIdx.x + 1u == blocks && lane == 0u) base[n] = sum + block_sum[blockIdx.x];
}
// (scratch: (n_tiles + 511) / 512 block sums)
void launch_tile_bases(const uint32_t* tile_count, uint32_t* caps, uint32_t* tile_base, uint32_t* scratch, uint32_t n_tiles, uint32_t tiles_x, uint32_t radius, hipStream_t stream) {
    const uint32_t blocks = (n_tiles + kBaseBlock - 1u) / kBaseBlock;
    hipLaunchKernelGGL(k_tile_caps, dim3((n_tiles + 63u) / 64u), dim3(64), 0, stream, tile_count, caps, n_tiles, tiles_x, radius);
    hipLaunchKernelGGL(k_tile_base_local, dim3(blocks), dim3(64), 0, stream, caps, tile_base, scratch, n_tiles);
    hipLaunchKernelGGL(k_tile_base_add, dim3(blocks), dim3(64), 0, stream, tile_base, scratch, n_tiles, blocks);
}
void launch_raster_edges(const SceneDev& s, const RasterParams& r, uint32_t samples, hipStream_t stream, void (*mark)(void*, const char*, uint64_t), void* ctx,
                         uint64_t raster_bytes, bool has_stroke) {
    constexpr uint32_t kBlock = 1u << CRH_XCD_BLOCK_LOG2;
    const uint32_t blocks = ((r.tiles_x + kBlock - 1u) / kBlock) * ((r.tiles_y + kBlock - 1u) / kBlock);
    const dim3 grid((r.tile_order && r.order_places) ? r.order_places : ((blocks + 7u) / 8u) * kBlock * kBlock * 8u); // the places of the tile order (a multiple of 8: place b is drawn on XCD b mod 8)
    // fill scenes at msaa 1: the class-batched walk with packed counters (k_raster_fill); CRH_FILL_KERNEL=0 keeps k_raster_edges<1, 4, false, *> (A/B runs, tests: the two are bit-equal)
    const char* fill_env = getenv("CRH_FILL_KERNEL"); // (read per launch: tests switch it inside one process)
    const bool fill_kernel = !(fill_env && fill_env[0] == '0') && r.winding_mask <= 0xFFFFu && r.fill_cells != 0u;
#define CRH_LAUNCH_EDGES(S_, ROWS_, STROKES_, LONG_) \
    hipLaunchKernelGGL((k_raster_edges<S_, ROWS_, STROKES_, LONG_>), grid, dim3(64 * (4 / ROWS_)), (4 / ROWS_) * r.sort_capacity * 4u, stream, s, r)
    if (samples == 1 && !has_stroke && r.rows) { // the row-span kernel: winding numbers accumulated in LDS, lanes over (entry, sample row)
        if (r.long_lists)
            hipLaunchKernelGGL((k_raster_rows<true>), grid, dim3(64), r.sort_capacity * 4u, stream, s, r);
        else
            hipLaunchKernelGGL((k_raster_rows<false>), grid, dim3(64), r.sort_capacity * 4u, stream, s, r);
    } else if (samples == 4) {
        if (has_stroke) CRH_LAUNCH_EDGES(4, 1, true, false); else CRH_LAUNCH_EDGES(4, 1, false, false);
    } else if (has_stroke) {
        CRH_LAUNCH_EDGES(1, 4, true, false);
    } else if (fill_kernel) {
        // (always the variant that looks for its late start across the chunks of a long list: measured on the 10 000 path scene — few lists
        // beyond one chunk — it is as fast as the one without, 0.1655 against 0.168 ms, and it is the build without scratch memory)
        // (... and ONE build for every frame: round 5's six-wave build for frames of long lists is gone with its reason, see CRH_FILL_TILE_WAVES)
        hipLaunchKernelGGL((k_raster_fill<true>), grid, dim3(64), r.sort_capacity * 4u, stream, s, r);
    } else if (r.long_lists) {
        CRH_LAUNCH_EDGES(1, 4, false, true);
    } else {
        CRH_LAUNCH_EDGES(1, 4, false, false);
    }
#undef CRH_LAUNCH_EDGES
    if (mark) mark(ctx, (samples == 1 && !has_stroke && r.rows) ? "raster_rows" : "raster_tiles", raster_bytes);
}

} // namespace crh
