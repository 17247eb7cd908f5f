// csrc/tessellate.hip — the GPU side of Shape::from_paths (renderer.rs:177-215) for a whole scene at once.
//
//   k_count   one lane per path element: how many records of each stream will it emit  -> workgroup-local exclusive scan
//   k_scan    one workgroup: exclusive scan of the per-workgroup totals                 -> global offsets
//   k_emit    one lane per path element: recompute and write at the scanned offsets     (fill.rs, stroke.rs)
//   k_stroke_lengths  one lane per stroked path: replay the f32 running length sum in reference order (stroke.rs:156,307,111)
//   k_hull_*  one wavefront per Shape: sort proto_hull (registers / LDS) + monotone chain (convex_hull.rs, vertex.rs:28-35)
//
// Two-phase emission reproduces the sequential `start_index` bookkeeping of the reference (fill.rs:361-365,
// stroke.rs:95,108,126-129) exactly: offsets are exclusive prefix sums in element order.
//
//   k_tess_runs   (round 5) the three of them in ONE launch per frame when no Shape has more elements than a workgroup has lanes:
//                 Shape-aligned workgroups (a run of consecutive Shapes each), the ten-channel scan in LDS, the runs' bases and every
//                 element's counts kept from the first tessellation of the paths — see "k_tess_runs" below.
#include "fill.hpp"
#include "scene.hpp"
#include "stroke.hpp"

namespace crh {

// ------------------------------------------------------------------------------------------------ sinks
struct CountSink {
    uint32_t curve_n = 0, solid_n = 0, hull_n = 0;
    CRH_D void curve(float2, const float*) { curve_n += 1; }
    CRH_D void solid(float2) { solid_n += 1; }
    CRH_D void hull(float2) { hull_n += 1; }
};

// Where a filled path's polygon vertices go: triangle_fan_to_strip (vertex.rs:28-35) applied on the fly.
struct SolidCursor {
    const SceneDev* s;
    uint32_t vertex_base; // global index of the path's first strip vertex
    uint32_t index_base;  // global index-stream position of the path's first entry
    uint32_t first_value; // index value of the path's first vertex, relative to its shape (fill.rs:361)
    uint32_t n;           // vertices of the path
    uint32_t j;           // position in path_solid_vertices of the next vertex
    CRH_D void push(float2 v) {
        const uint32_t pos = (j < ((n + 1) >> 1)) ? 2u * j : 2u * (n - 1u - j) + 1u;
        s->solid_v[vertex_base + pos] = {v.x, v.y};
        s->solid_i[index_base + pos] = (uint16_t)(first_value + pos);
        s->solid_flag[vertex_base + pos] = (uint8_t)((pos & 1u) | (pos + 1u == n ? 2u : 0u));
        j += 1;
    }
};

struct HullCursor {
    const SceneDev* s;
    uint32_t at;
    uint32_t path;
    CRH_D void push(float2 v) { // SafeFloat::from, safe_float.rs:111-120
        if (!is_finite(v.x) || !is_finite(v.y)) raise_error(*s, path, CRH_ERR_NON_FINITE);
        s->hull_cand[at] = {v.x == 0.0f ? 0.0f : v.x, v.y == 0.0f ? 0.0f : v.y};
        at += 1;
    }
};

template <bool RATIONAL>
struct CubicEmitSink {
    SolidCursor solid_cursor;
    HullCursor hull_cursor;
    uint32_t curve_at;
    CRH_D void curve(float2 v, const float* w) {
        if constexpr (RATIONAL)
            solid_cursor.s->rc_v[curve_at] = {v.x, v.y, w[0], w[1], w[2], w[3]};
        else
            solid_cursor.s->ic_v[curve_at] = {v.x, v.y, w[0], w[1], w[2]};
        curve_at += 1;
    }
    CRH_D void solid(float2 v) { solid_cursor.push(v); }
    CRH_D void hull(float2 v) { hull_cursor.push(v); }
};

// The start point of a filled segment is `path_solid_vertices.last()` (fill.rs:292,300,329,338): the END of the previous
// segment as it was pushed to the polygon. Lines, quadratics and integral cubics push their end point verbatim (x / 1 == x), but a
// rational cubic pushes point_to_vec(weighted point) = ((x * w) / w, (y * w) / w) (fill.rs:248, utils.rs:106-118), which may be
// one ulp away from x. The lane re-derives that round trip from the previous record instead of carrying sequential state.
CRH_D float2 fill_start_point(const SceneDev& s, uint32_t e) {
    const float* q = s.pool + s.elem_off[e - 1u];
    if (s.elem_type[e - 1u] == ELEM_RC) {
        const float w = q[3];
        return make_float2((q[8] * w) / w, (q[9] * w) / w);
    }
    const float* p = s.pool + s.elem_off[e];
    return make_float2(p[-2], p[-1]);
}

CRH_D void load_cubic(const SceneDev& s, uint32_t e, bool rational, Pt cp[4]) {
    const float* p = s.pool + s.elem_off[e];
    const float2 start = fill_start_point(s, e);
    if (rational) { // fill.rs:337-342
        cp[0] = weighted_vec_to_point(p[0], start.x, start.y);
        cp[1] = weighted_vec_to_point(p[1], p[4], p[5]);
        cp[2] = weighted_vec_to_point(p[2], p[6], p[7]);
        cp[3] = weighted_vec_to_point(p[3], p[8], p[9]);
    } else { // fill.rs:299-304
        cp[0] = vec_to_point(start.x, start.y);
        cp[1] = vec_to_point(p[0], p[1]);
        cp[2] = vec_to_point(p[2], p[3]);
        cp[3] = vec_to_point(p[4], p[5]);
    }
}

// ------------------------------------------------------------------------------------------------ k_count
CRH_D void count_fill_element(const SceneDev& s, uint32_t e, uint32_t type, uint32_t path, uint32_t cnt[NCH]) {
    switch (type) {
        case ELEM_MOVE: // fill.rs:271-272
        case ELEM_LINE: // fill.rs:280-284
            cnt[CH_SOLID_V] = 1;
            cnt[CH_HULL] = 1;
            break;
        case ELEM_IQ: // fill.rs:285-296
            cnt[CH_SOLID_V] = 1;
            cnt[CH_HULL] = 2;
            cnt[CH_IQ] = 1;
            break;
        case ELEM_RQ: // fill.rs:321-333
            cnt[CH_SOLID_V] = 1;
            cnt[CH_HULL] = 2;
            cnt[CH_RQ] = 1;
            break;
        case ELEM_IC:
        case ELEM_RC: {
            Pt cp[4];
            load_cubic(s, e, type == ELEM_RC, cp);
            CountSink sink;
            uint32_t err = 0;
            cubic_fill(cp, type == ELEM_IC, sink, err);
            if (err) raise_error(s, path, err);
            cnt[CH_SOLID_V] = sink.solid_n;
            cnt[CH_HULL] = sink.hull_n;
            cnt[type == ELEM_IC ? CH_IC_V : CH_RC_V] = sink.curve_n;
            break;
        }
        default: // ELEM_END: fill.rs:361-365 appends one restart marker per path
            cnt[CH_SOLID_END] = 1;
            break;
    }
}

// STROKES = false: the Scene has no stroked path (the host knows) — the kernel is built without the stroke code: its root arrays are what
// sends 80 / 176 B per lane of the general kernels to scratch memory, and the fill code alone needs 60 registers less
template <bool STROKES = true>
CRH_D void count_element(const SceneDev& s, uint32_t e, uint32_t cnt[NCH]) {
    const uint32_t type = s.elem_type[e];
    const uint32_t path = s.elem_path[e];
    const int32_t stroke = STROKES ? s.path_stroke[path] : -1;
    if (!STROKES || stroke < 0)
        count_fill_element(s, e, type, path, cnt);
    else
        count_stroke_element(s, e, type, path, s.stroke_options[stroke], cnt);
}

#ifdef CRH_TESS_WAVES
#define CRH_TESS_OCCUPANCY __attribute__((amdgpu_waves_per_eu(CRH_TESS_WAVES)))
#else
#define CRH_TESS_OCCUPANCY
#endif
__global__ __launch_bounds__(kTessBlock) CRH_TESS_OCCUPANCY void k_count(SceneDev s) {
    __shared__ uint32_t wave_total[kTessBlock / 64][NCH];
    const uint32_t e = blockIdx.x * kTessBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t cnt[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) cnt[c] = 0;
    if (e < s.n_elems) count_element(s, e, cnt);
    uint32_t excl[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        uint32_t v = cnt[c];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(v, d, 64);
            if (lane >= (uint32_t)d) v += up;
        }
        excl[c] = v - cnt[c];
        if (lane == 63) wave_total[wave][c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        uint32_t base = 0;
        for (uint32_t w = 0; w < wave; ++w) base += wave_total[w][c];
        excl[c] += base;
    }
    if (e < s.n_elems) {
        ElemScan out;
#pragma unroll
        for (int c = 0; c < NCH; ++c) out.v[c] = excl[c];
        s.elem_scan[e] = out;
    }
    if (threadIdx.x == kTessBlock - 1) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) s.wg_total[blockIdx.x * NCH + c] = excl[c] + cnt[c];
    }
}

// ------------------------------------------------------------------------------------------------ k_scan
// The scan over the workgroups' totals (rows) in two small launches of SINGLE-WAVE workgroups: k_scan_rows — a wavefront per group of 64
// rows: exclusive prefix inside the group (all ten channels, wave shuffles), group totals; k_scan_groups — one wavefront: exclusive prefix of
// the groups, scene totals, sentinel rows. gscan() adds group_base + wg_base + elem_scan. (One 1024-thread workgroup did it in one launch — 45 us
// for the 5300 rows of the 50 000 glyph scene — but sixteen wavefronts need sixteen free slots on ONE compute unit, and next to the raster kernel of
// the frame before they waited for its whole grid to drain: 1.3 ms in the run on the 100 000 path scene.)
__global__ __launch_bounds__(64) void k_scan_rows(SceneDev s) {
    const uint32_t lane = threadIdx.x, w = blockIdx.x * 64u + lane;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint32_t mine = w < s.n_wg ? s.wg_total[w * NCH + c] : 0u;
        uint32_t v = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(v, d, 64);
            if (lane >= (uint32_t)d) v += up;
        }
        if (w < s.n_wg) s.wg_base[w * NCH + c] = v - mine;
        if (lane == 63) s.group_base[blockIdx.x * NCH + c] = v; // the group's total: k_scan_groups turns it into the group's base
    }
}
__global__ __launch_bounds__(64) void k_scan_groups(SceneDev s) {
    const uint32_t lane = threadIdx.x, n_groups = (s.n_wg + 63u) / 64u;
    uint32_t running[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) running[c] = 0;
    for (uint32_t first = 0; first < n_groups; first += 64u) {
        const uint32_t g = first + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint32_t mine = g < n_groups ? s.group_base[g * NCH + c] : 0u;
            uint32_t v = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(v, d, 64);
                if (lane >= (uint32_t)d) v += up;
            }
            if (g < n_groups) s.group_base[g * NCH + c] = running[c] + v - mine;
            running[c] += (uint32_t)__shfl((int)v, 63, 64);
        }
    }
    if (lane < NCH) {
        const uint32_t c = lane;
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < NCH; ++k) t = c == (uint32_t)k ? running[k] : t;
        s.totals[c] = t;
    }
}
// where every Shape's records begin and end in the ten streams (one lane per Shape; an empty Shape begins where it ends)
__global__ __launch_bounds__(256) void k_shape_rows(SceneDev s) {
    const uint32_t shape = blockIdx.x * 256u + threadIdx.x;
    if (shape >= s.n_shapes) return;
    // (streams that do not hold these totals — new paths uploaded over the capacities of the ones before, api.hip crh_scene::optimistic — get no
    // records from k_emit: the rows of the run before stay, so that a pass drawn before the host has sized the streams reads inside them)
    if (s.capacity[0] | s.capacity[1] | s.capacity[2] | s.capacity[3] | s.capacity[4] | s.capacity[5] | s.capacity[6] | s.capacity[7] | s.capacity[8] | s.capacity[9]) {
        bool ok = true;
#pragma unroll
        for (int c = 0; c < NCH; ++c) ok = ok && s.totals[c] <= s.capacity[c];
        if (!ok) return;
    }
    const uint32_t e0 = s.shape_elem_begin[shape], e1 = s.shape_elem_begin[shape + 1u];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        s.shape_base[shape * kShapeRow + c] = e0 < s.n_elems ? gscan(s, e0, c) : s.totals[c];
        s.shape_base[shape * kShapeRow + NCH + c] = e1 < s.n_elems ? gscan(s, e1, c) : s.totals[c];
    }
}

// ------------------------------------------------------------------------------------------------ k_emit
CRH_D bool fits(const SceneDev& s) {
    bool ok = true;
#pragma unroll
    for (int c = 0; c < NCH; ++c) ok = ok && s.totals[c] <= s.capacity[c];
    return ok;
}

struct GlobalScan { // the two-pass path: group base + row base + the element's prefix inside its row
    const SceneDev* s;
    CRH_D uint32_t operator()(uint32_t e, int ch) const { return gscan(*s, e, ch); }
};
constexpr uint32_t kScanPitch = NCH + 1; // (odd: lanes of consecutive elements reading one channel spread over the LDS banks)
struct LocalScan { // k_tess_runs: rows [0, n] of the workgroup's run in LDS, the run's reserved bases already added
    const uint32_t* rows;
    uint32_t first;
    CRH_D uint32_t operator()(uint32_t e, int ch) const { return rows[(e - first) * kScanPitch + (uint32_t)ch]; }
};

template <class Scan>
CRH_D void emit_fill_element(const SceneDev& s, uint32_t e, uint32_t type, uint32_t path, const uint32_t g[NCH], const Scan& scan) {
    const uint32_t shape = s.path_shape[path];
    const uint32_t shape_first = s.shape_elem_begin[shape];
    const uint32_t move = s.path_elem_begin[path];
    const uint32_t end = s.path_elem_begin[path + 1] - 1u;
    const uint32_t shape_solid = scan(shape_first, CH_SOLID_V), shape_ends = scan(shape_first, CH_SOLID_END);
    const uint32_t move_solid = scan(move, CH_SOLID_V), move_ends = scan(move, CH_SOLID_END);
    SolidCursor solid;
    solid.s = &s;
    solid.vertex_base = move_solid;
    solid.index_base = move_solid + move_ends;
    solid.first_value = move_solid - shape_solid;
    solid.n = scan(end, CH_SOLID_V) - move_solid;
    solid.j = g[CH_SOLID_V] - move_solid;
    (void)shape_ends;
    HullCursor hull = {&s, g[CH_HULL], path};
    const float* p = s.pool + s.elem_off[e];
    switch (type) {
        case ELEM_MOVE:
        case ELEM_LINE: {
            const float2 v = make_float2(p[0], p[1]);
            hull.push(v);
            solid.push(v);
            break;
        }
        case ELEM_IQ: { // fill.rs:285-296
            const uint32_t at = 3u * g[CH_IQ];
            s.iq_v[at + 0] = {p[2], p[3], 1.0f, 1.0f};
            s.iq_v[at + 1] = {p[0], p[1], 0.5f, 0.0f};
            const float2 start = fill_start_point(s, e);
            s.iq_v[at + 2] = {start.x, start.y, 0.0f, 0.0f};
            hull.push(make_float2(p[0], p[1]));
            hull.push(make_float2(p[2], p[3]));
            solid.push(make_float2(p[2], p[3]));
            break;
        }
        case ELEM_RQ: { // fill.rs:321-333
            const uint32_t at = 3u * g[CH_RQ];
            const float weight = 1.0f / p[0];
            s.rq_v[at + 0] = {p[3], p[4], 1.0f, 1.0f, 1.0f};
            s.rq_v[at + 1] = {p[1], p[2], 0.5f * weight, 0.0f, weight};
            const float2 start = fill_start_point(s, e);
            s.rq_v[at + 2] = {start.x, start.y, 0.0f, 0.0f, 1.0f};
            hull.push(make_float2(p[1], p[2]));
            hull.push(make_float2(p[3], p[4]));
            solid.push(make_float2(p[3], p[4]));
            break;
        }
        case ELEM_IC: {
            Pt cp[4];
            load_cubic(s, e, false, cp);
            CubicEmitSink<false> sink = {solid, hull, g[CH_IC_V]};
            uint32_t err = 0;
            cubic_fill(cp, true, sink, err);
            break;
        }
        case ELEM_RC: {
            Pt cp[4];
            load_cubic(s, e, true, cp);
            CubicEmitSink<true> sink = {solid, hull, g[CH_RC_V]};
            uint32_t err = 0;
            cubic_fill(cp, false, sink, err);
            break;
        }
        default: { // ELEM_END: the restart marker that replaces the last index (fill.rs:363-364)
            s.solid_i[solid.index_base + solid.n] = 0xFFFFu;
            break;
        }
    }
}

__global__ __launch_bounds__(kTessBlock) CRH_TESS_OCCUPANCY void k_emit(SceneDev s) {
    const uint32_t e = blockIdx.x * kTessBlock + threadIdx.x;
    if (e >= s.n_elems) return;
    if (!fits(s)) {
        if (e == 0) raise_error(s, 0, CRH_ERR_UNSUPPORTED + 0x80u); // capacity overflow: the host reallocates and re-runs
        return;
    }
    uint32_t g[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) g[c] = gscan(s, e, c);
    const uint32_t type = s.elem_type[e];
    const uint32_t path = s.elem_path[e];
    const int32_t stroke = s.path_stroke[path];
    const GlobalScan scan = {&s};
    if (stroke < 0)
        emit_fill_element(s, e, type, path, g, scan);
    else
        emit_stroke_element(s, e, type, path, s.stroke_options[stroke], g, scan);
}

// ------------------------------------------------------------------------------------------------ k_tess_runs
// One launch per frame instead of k_count + two scan launches + k_emit (fill.rs:263-367, stroke.rs:205-465 and the loop renderer.rs:187-196
// for a whole scene), for Scenes whose Shapes fit a workgroup. Workgroup b takes the Shapes [tess_run[b], tess_run[b + 1]) — at most
// kTessBlock elements, lane = element. HOW MANY records an element emits is a property of the uploaded paths, not of the frame: the first
// tessellation of new paths (the one whose totals the host needs anyway, to size the streams) counts once —
//   k_tess_count_runs  the counting sinks; every element's ten counts packed into one word (elem_cnt), the run's totals;
//   k_scan_runs        exclusive prefix over the runs (one small workgroup): run_base, the scene's totals —
// and every tessellation, that one included, is ONE kernel that analyses each element ONCE:
//   k_tess_runs        the lane unpacks its counts, ten-channel exclusive scan over the workgroup (wave shuffles + LDS), the rows go to LDS
//                      with the run's base added — every offset the emission asks for (its own, its path's MOVE and END, its Shape's first
//                      element) is a row of this workgroup: no scan across workgroups, no look-back, no elem_scan[] in memory —, lanes over
//                      the run's Shapes write their shape_base rows, the lane emits with the writing sinks.
// (A first cut reserved the workgroups' ranges with one atomic per channel and workgroup instead of run_base — Shapes in arbitrary order
// inside the streams. Bit-equal Shape by Shape, and slower than the two passes: 6 300 workgroups of the glyph scene queue up on the ten
// cursors, 0.27 ms against 0.14)
// Stroked elements: HULL = LINE_V + the joins' extra candidates; LINE_V in 24 bits (the host rejects step counts beyond 2^20).
CRH_D uint32_t pack_counts(const uint32_t cnt[NCH], bool stroked, bool& fits_word) {
    if (stroked) {
        fits_word = cnt[CH_LINE_V] < (1u << 24);
        return cnt[CH_LINE_V] | (cnt[CH_LINE_CUT] << 24) | (cnt[CH_JOINT] << 26) | ((cnt[CH_HULL] - cnt[CH_LINE_V]) << 28);
    }
    fits_word = true; // (one segment: at most 6 polygon vertices, 3 hull candidates, 12 curve vertices)
    return cnt[CH_SOLID_V] | (cnt[CH_HULL] << 3) | (cnt[CH_SOLID_END] << 5) | (cnt[CH_IQ] << 6) | (cnt[CH_RQ] << 7) | (cnt[CH_IC_V] << 8) | (cnt[CH_RC_V] << 12);
}
CRH_D void unpack_counts(uint32_t w, bool stroked, uint32_t cnt[NCH]) {
    const uint32_t line_v = w & 0xFFFFFFu;
    cnt[CH_LINE_V] = stroked ? line_v : 0u;
    cnt[CH_LINE_CUT] = stroked ? (w >> 24) & 3u : 0u;
    cnt[CH_JOINT] = stroked ? (w >> 26) & 3u : 0u;
    cnt[CH_HULL] = stroked ? line_v + (w >> 28) : (w >> 3) & 3u;
    cnt[CH_SOLID_V] = stroked ? 0u : w & 7u;
    cnt[CH_SOLID_END] = stroked ? 0u : (w >> 5) & 1u;
    cnt[CH_IQ] = stroked ? 0u : (w >> 6) & 1u;
    cnt[CH_RQ] = stroked ? 0u : (w >> 7) & 1u;
    cnt[CH_IC_V] = stroked ? 0u : (w >> 8) & 15u;
    cnt[CH_RC_V] = stroked ? 0u : (w >> 12) & 15u;
}

// BLOCK: lanes of a workgroup = the elements a run holds at most (SceneDev::run_block: 256, or 128 for scenes of many small Shapes)
template <bool STROKES, int BLOCK>
__global__ __launch_bounds__(BLOCK) CRH_TESS_OCCUPANCY void k_tess_count_runs(SceneDev s) {
    __shared__ uint32_t wave_total[BLOCK / 64][NCH];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t shape0 = s.tess_run[blockIdx.x], shape1 = s.tess_run[blockIdx.x + 1u];
    const uint32_t first = s.shape_elem_begin[shape0], n = s.shape_elem_begin[shape1] - first; // (<= BLOCK: the host cut the runs)
    const uint32_t e = first + tid;
    uint32_t cnt[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) cnt[c] = 0;
    if (tid < n) {
        count_element<STROKES>(s, e, cnt);
        const uint32_t path = s.elem_path[e];
        bool fits_word;
        s.elem_cnt[e] = pack_counts(cnt, STROKES && s.path_stroke[path] >= 0, fits_word);
        if (!fits_word) raise_error(s, path, CRH_ERR_UNSUPPORTED);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        uint32_t v = cnt[c];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d, 64);
        if (lane == 0u) wave_total[wave][c] = v;
    }
    __syncthreads();
    if (tid < NCH) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < (uint32_t)BLOCK / 64u; ++w) total += wave_total[w][tid];
        s.run_base[blockIdx.x * NCH + tid] = total;
    }
}

// the runs' totals -> where every run begins (in place), the totals behind the last run and in totals[]. ONE WAVEFRONT, every lane a stretch of
// consecutive runs: sum them, one scan over the lanes' sums (shuffles: no LDS, no barrier), write the stretch's prefixes. (Until round 6: a 256-thread
// workgroup, 256 runs a turn, three barriers and a round trip to memory per turn — the 6 300 runs of the 50 000 glyph scene took 25 turns, 0.28 ms
// under the profiler; VERDICT r05 item 8. The first rewrite used 1 024 lanes: beside the raster kernel of the frame in front — new paths every frame — a
// sixteen-wave workgroup waits for sixteen free slots on ONE compute unit, i.e. for that grid to drain: the step of tools/r05_reupload_calls.py went from
// 0.48 to 0.97 ms. A single wavefront takes the first slot that frees. A scan that looks back across workgroups is not needed: 100 000 runs are 1 600 rows a lane.)
constexpr uint32_t kScanRunsThreads = 64;
__global__ __launch_bounds__(kScanRunsThreads) void k_scan_runs(SceneDev s) {
    const uint32_t lane = threadIdx.x;
    const uint32_t per = (s.n_runs + kScanRunsThreads - 1u) / kScanRunsThreads, first = min(s.n_runs, lane * per), last = min(s.n_runs, first + per);
    uint32_t sum[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) sum[c] = 0u;
    for (uint32_t run = first; run < last; ++run)
#pragma unroll
        for (int c = 0; c < NCH; ++c) sum[c] += s.run_base[run * NCH + c];
    uint32_t begin[NCH], total[NCH]; // where the lane's stretch begins; the scene's totals
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        uint32_t v = sum[c];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(v, d, 64);
            if (lane >= (uint32_t)d) v += up;
        }
        begin[c] = v - sum[c];
        total[c] = (uint32_t)__shfl((int)v, 63, 64);
    }
    for (uint32_t run = first; run < last; ++run)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint32_t mine = s.run_base[run * NCH + c];
            s.run_base[run * NCH + c] = begin[c];
            begin[c] += mine;
        }
    if (lane == 0u) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) s.run_base[s.n_runs * NCH + c] = total[c], s.totals[c] = total[c];
    }
}

template <bool STROKES, int BLOCK>
__global__ __launch_bounds__(BLOCK) CRH_TESS_OCCUPANCY void k_tess_runs(SceneDev s) {
    __shared__ uint32_t rows[(BLOCK + 1) * kScanPitch];
    __shared__ uint32_t wave_total[BLOCK / 64][NCH];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t shape0 = s.tess_run[blockIdx.x], shape1 = s.tess_run[blockIdx.x + 1u];
    const uint32_t first = s.shape_elem_begin[shape0], n = s.shape_elem_begin[shape1] - first;
    const uint32_t e = first + tid;
    const uint32_t* const scene_totals = s.run_base + s.n_runs * NCH;
    if (blockIdx.x == 0u && tid < NCH) s.totals[tid] = scene_totals[tid]; // (this set's copy: what the kernels behind this one and the host read)
    bool ok = true;
#pragma unroll
    for (int c = 0; c < NCH; ++c) ok = ok && scene_totals[c] <= s.capacity[c];
    if (!ok) { // (uniform) stale capacities: the host sizes the streams from totals[] and runs again
        if (blockIdx.x == 0u && tid == 0u) raise_error(s, 0, CRH_ERR_UNSUPPORTED + 0x80u);
        return;
    }
    uint32_t type = ELEM_MOVE, path = 0;
    int32_t stroke = -1;
    uint32_t cnt[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) cnt[c] = 0;
    if (tid < n) {
        type = s.elem_type[e], path = s.elem_path[e];
        if (STROKES) stroke = s.path_stroke[path];
        unpack_counts(s.elem_cnt[e], stroke >= 0, cnt);
    }
    uint32_t g[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        uint32_t v = cnt[c];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(v, d, 64);
            if (lane >= (uint32_t)d) v += up;
        }
        g[c] = v - cnt[c];
        if (lane == 63) wave_total[wave][c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        uint32_t base = s.run_base[blockIdx.x * NCH + c];
#pragma unroll
        for (uint32_t w = 0; w + 1u < (uint32_t)BLOCK / 64u; ++w) base += w < wave ? wave_total[w][c] : 0u;
        g[c] += base;
        if (tid < n) rows[tid * kScanPitch + c] = g[c];
        if (tid == 0u) rows[n * kScanPitch + c] = s.run_base[(blockIdx.x + 1u) * NCH + c];
    }
    __syncthreads();
    for (uint32_t shape = shape0 + tid; shape < shape1; shape += BLOCK) {
        const uint32_t* b0 = rows + (s.shape_elem_begin[shape] - first) * kScanPitch;
        const uint32_t* b1 = rows + (s.shape_elem_begin[shape + 1u] - first) * kScanPitch;
#pragma unroll
        for (int c = 0; c < NCH; ++c) s.shape_base[shape * kShapeRow + c] = b0[c], s.shape_base[shape * kShapeRow + NCH + c] = b1[c];
    }
    if (tid >= n) return;
    const LocalScan scan = {rows, first};
    if (!STROKES || stroke < 0) {
        emit_fill_element(s, e, type, path, g, scan);
    } else {
        if (type == ELEM_MOVE) { // what k_stroke_lengths walks: the path's vertex pairs and its first join
            const uint32_t end = s.path_elem_begin[path + 1u] - 1u;
            s.path_scan[3u * path] = g[CH_LINE_V] >> 1, s.path_scan[3u * path + 1u] = scan(end + 1u, CH_LINE_V) >> 1, s.path_scan[3u * path + 2u] = g[CH_JOINT];
        }
        emit_stroke_element(s, e, type, path, s.stroke_options[stroke], g, scan);
    }
}

// ------------------------------------------------------------------------------------------------ k_hull
// convex_hull::andrew (convex_hull.rs:7-40) + triangle_fan_to_strip (renderer.rs:197). The candidates are sorted in SafeFloat's
// lexicographic order (safe_float.rs:163-173; equal keys are bit-identical after -0 canonicalisation, so stability is moot), then the
// monotone chain is walked.
//   k_hull_small  Shapes with <= 64 candidates (the common case), kHullBatch Shapes per single-wavefront workgroup. Phase 1: the wavefront
//                 sorts one Shape at a time (lane = candidate, bitonic network on registers) into LDS. Phase 2: a lane pair per Shape —
//                 the Shapes walk their (inherently serial) monotone chains simultaneously, the stack being a byte index per entry, its
//                 two topmost points kept in registers. Larger Shapes are queued.
//   k_hull_large  Shapes with 65..2048 candidates, taken from the queue by a fixed grid: bitonic sort + chain in LDS.
//   k_hull_huge   Shapes beyond that: the same in global memory, one workgroup per Shape.
#ifndef CRH_ABLATE_HULL
#define CRH_ABLATE_HULL 0
#endif
constexpr uint32_t kHullSmall = 64;
constexpr uint32_t kHullMid = 256;
constexpr uint32_t kHullMax = 2048; // candidates per Shape that fit the large LDS sort (16 KiB of points + 8 KiB of chain indices)
#ifndef CRH_HULL_PACK
#define CRH_HULL_PACK 2 // glyph scene, step / latency / kernel alone in ms — 1: 0.632 / 0.879 / 0.105, 2: 0.624 / 0.890, 4: 0.617 - 0.628 / 0.916 / 0.141, 8: 0.655 / 0.962 (profiles/r06_experiments.txt)
#endif
constexpr uint32_t kHullPack = CRH_HULL_PACK; // Shapes of 65 .. kHullMid candidates per wavefront of k_hull_large

CRH_D bool lex_less(float2 a, float2 b) { return a.x < b.x || (a.x == b.x && a.y < b.y); }
CRH_D float turn(float2 a, float2 b, float2 c) { return triple(vec_to_point(a.x, a.y), vec_to_point(b.x, b.y), vec_to_point(c.x, c.y)); }

// the strip order of triangle_fan_to_strip (vertex.rs:28-35): [0, h-1, 1, h-2, ...]
CRH_D uint32_t fan_to_strip_source(uint32_t i, uint32_t h) { return (i & 1u) == 0 ? (i >> 1) : h - 1u - (i >> 1); }

// One wavefront per workgroup: next to the previous frame's raster kernel (single-wave workgroups refilling every slot the moment it
// frees) a four-wave workgroup never found four free slots at once and waited for the raster grid to drain (0.28 ms instead of 0.03).
constexpr uint32_t kHullWaves = 1;
#ifndef CRH_HULL_BATCH
#define CRH_HULL_BATCH 4
#endif
constexpr uint32_t kHullBatch = CRH_HULL_BATCH;                  // Shapes per workgroup of k_hull_small (phase 1: one after the other; phase 2: a lane pair each)
constexpr uint32_t kHullRow = kHullSmall + 1;       // LDS row pitch in float2 (odd: lane-per-Shape accesses spread over the banks)

// The value of lane (lane ^ J). J < 16 stays inside the 16-lane row and is a DPP operand modifier (quad permute, row rotate); only 16 and 32
// go through ds_bpermute. (With __shfl_xor for every step of the sorting network the 50 000 glyph scene kept the LDS crossbar busier with
// the 42 permutes per sort than the VALUs with the compares.)
template <uint32_t J>
CRH_D float lane_xor(float v, uint32_t lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int x = __float_as_int(v);
    if (J == 1) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));  // quad_perm [1, 0, 3, 2]
    if (J == 2) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));  // quad_perm [2, 3, 0, 1]
    if (J == 8) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false)); // row_ror:8
    if (J == 4) { // row_ror:4 brings lane - 4, row_ror:12 lane + 4 (mod 16)
        const int down = __builtin_amdgcn_update_dpp(x, x, 0x124, 0xF, 0xF, false), up = __builtin_amdgcn_update_dpp(x, x, 0x12C, 0xF, 0xF, false);
        return __int_as_float((lane & 4u) ? down : up);
    }
    return __shfl_xor(v, (int)J, 64);
#else
    (void)lane;
    return v;
#endif
}
// one compare-exchange step of the bitonic network on (x, y) keys in lexicographic order
template <uint32_t J>
CRH_D void bitonic_step(float2& p, uint32_t lane, uint32_t k) {
    const float2 q = make_float2(lane_xor<J>(p.x, lane), lane_xor<J>(p.y, lane));
    const bool keep_min = ((lane & J) == 0) == ((lane & k) == 0);
    const bool q_less = q.x < p.x || (q.x == p.x && q.y < p.y);
    p = (keep_min == q_less) ? q : p; // min keeps q when q < p, max keeps q when !(q < p); equal keys are identical
}
template <uint32_t K>
CRH_D void bitonic_stage(float2& p, uint32_t lane) {
    if (K >= 64) bitonic_step<32>(p, lane, K);
    if (K >= 32) bitonic_step<16>(p, lane, K);
    if (K >= 16) bitonic_step<8>(p, lane, K);
    if (K >= 8) bitonic_step<4>(p, lane, K);
    if (K >= 4) bitonic_step<2>(p, lane, K);
    bitonic_step<1>(p, lane, K);
}

// The Shapes beyond k_hull_small's 64 candidates, listed for the kernels behind it: queue 0 up to kHullMid candidates (small LDS footprint), queue 1
// up to kHullMax, queue 2 global memory. One lane per Shape, ONE returning atomic per queue and block of 256 Shapes: until the end of round 5
// k_hull_small queued its large Shapes itself, an atomic each on one counter — returning atomics on one address are served at the memory side,
// ≈ 25 ns apiece, and the 6 000 large Shapes of the 50 000 glyph scene made that queue half of the kernel's 0.12 ms (without its sorts the
// kernel took 0.117 ms, without its chain walks 0.078: neither was what it waited for).
__global__ __launch_bounds__(256) void k_hull_queues(SceneDev s) {
    __shared__ uint32_t wave_count[4][3], block_base[3];
    if (!fits(s)) return;
    const uint32_t shape = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t n = 0;
    if (shape < s.n_shapes) n = s.shape_base[shape * kShapeRow + NCH + CH_HULL] - s.shape_base[shape * kShapeRow + CH_HULL];
    const uint32_t queue = n > kHullMax ? 2u : (n > kHullMid ? 1u : (n > kHullSmall ? 0u : 3u)); // 3: k_hull_small's own
    unsigned long long members[3];
#pragma unroll
    for (uint32_t q = 0; q < 3u; ++q) {
        members[q] = __ballot(queue == q);
        if (lane == 0u) wave_count[wave][q] = (uint32_t)__popcll(members[q]);
    }
    __syncthreads();
    if (threadIdx.x < 3u) {
        const uint32_t q = threadIdx.x, total = wave_count[0][q] + wave_count[1][q] + wave_count[2][q] + wave_count[3][q];
        block_base[q] = total ? atomicAdd(s.hull_large_count + q, total) : 0u;
    }
    __syncthreads();
    if (queue < 3u) {
        uint32_t at = block_base[queue];
        for (uint32_t w = 0; w < wave; ++w) at += wave_count[w][queue];
        const unsigned long long mine = queue == 0u ? members[0] : (queue == 1u ? members[1] : members[2]);
        at += (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
        s.hull_large_list[queue * s.n_shapes + at] = shape;
    }
}

__global__ __launch_bounds__(64 * kHullWaves) void k_hull_small(SceneDev s) {
    __shared__ float2 sorted[kHullBatch][kHullRow];         // sorted candidates of the batch's Shapes
    __shared__ uint8_t stack[kHullBatch][2 * kHullSmall];   // the monotone chain as indices into `sorted`
    __shared__ uint32_t count[kHullBatch];                  // candidates per Shape; 0 = nothing to do here (empty, queued or rejected)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (!fits(s)) return;
    const uint32_t first_shape = blockIdx.x * kHullBatch;
    // ---- phase 1: every wavefront sorts kHullBatch / kHullWaves Shapes (lane = candidate, bitonic network on registers); the dependent
    // global loads (range, then candidates) of all its Shapes are issued together before the first sort
    constexpr uint32_t kPerWave = kHullBatch / kHullWaves;
    uint32_t n_of[kPerWave], base_of[kPerWave];
#pragma unroll
    for (uint32_t u = 0; u < kPerWave; ++u) {
        const uint32_t shape = first_shape + wave + kHullWaves * u;
        n_of[u] = 0;
        base_of[u] = 0;
        if (shape < s.n_shapes) {
            base_of[u] = s.shape_base[shape * kShapeRow + CH_HULL];
            n_of[u] = s.shape_base[shape * kShapeRow + NCH + CH_HULL] - base_of[u];
        }
    }
    const float inf = __uint_as_float(0x7f800000u);
    float2 p_of[kPerWave];
#pragma unroll
    for (uint32_t u = 0; u < kPerWave; ++u) {
        p_of[u] = make_float2(inf, inf);
        if (n_of[u] <= kHullSmall && lane < n_of[u]) p_of[u] = make_float2(s.hull_cand[base_of[u] + lane].x, s.hull_cand[base_of[u] + lane].y);
    }
#pragma unroll
    for (uint32_t u = 0; u < kPerWave; ++u) {
        const uint32_t slot = wave + kHullWaves * u, shape = first_shape + slot;
        uint32_t n = n_of[u];
        if (shape < s.n_shapes) {
            if (n > kHullSmall) { // (k_hull_queues has put it on the list of a kernel behind this one)
                n = 0;
            } else if (n == 0) {
                if (lane == 0) s.hull_count[shape] = 0;
            } else {
                float2 p = p_of[u];
                if (n >= 3) { // fewer are returned as they are (convex_hull.rs:9-11)
                    bitonic_stage<2>(p, lane);
                    bitonic_stage<4>(p, lane);
                    bitonic_stage<8>(p, lane);
                    bitonic_stage<16>(p, lane);
                    bitonic_stage<32>(p, lane);
                    bitonic_stage<64>(p, lane);
                }
                sorted[slot][lane] = p;
            }
        }
        if (lane == 0) count[slot] = n;
    }
    __syncthreads();
    // ---- phase 2: two lanes per Shape; the lower and the upper chain of Andrew's scan are independent (the upper one starts from the last
    //      point on a stack floor of its own, convex_hull.rs:24-33), so the even lane walks the lower and the odd lane the upper chain
#if CRH_ABLATE_HULL == 1
    return;
#endif
    if (wave != 0 || lane >= 2u * kHullBatch) return;
    const uint32_t slot = lane >> 1, upper = lane & 1u;
    const uint32_t n = count[slot];
    const uint32_t shape = first_shape + slot;
    if (n == 0) return; // both lanes of the pair
    const float2* pts = sorted[slot];
    uint8_t* chain = stack[slot] + upper * kHullSmall;
    uint32_t m = 0;
    if (n < 3) {
        if (!upper)
            for (uint32_t i = 0; i < n; ++i) chain[i] = (uint8_t)i;
        m = upper ? 1u : n + 1u; // so that the hull below is pts[0 .. n)
    } else {
        float2 a = make_float2(0.0f, 0.0f), b = a; // chain[m - 2], chain[m - 1]
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t i = upper ? n - 1u - k : k;
            const float2 c = pts[i];
            while (m > 1 && turn(a, b, c) <= kErrorMargin) {
                m -= 1;
                b = a;
                if (m >= 2) a = pts[chain[m - 2]];
            }
            chain[m++] = (uint8_t)i;
            a = b;
            b = c;
        }
    }
    // the hull: lower[0 .. ml - 1) ++ upper[0 .. mu - 1) — each chain without its last point, which opens the other one
    const uint32_t ml = (uint32_t)__shfl((int)m, (int)(lane & ~1u), 64) - 1u, mu = (uint32_t)__shfl((int)m, (int)(lane | 1u), 64) - 1u;
    const uint32_t h = ml + mu;
    const uint8_t* lower_chain = stack[slot];
    const uint8_t* upper_chain = stack[slot] + kHullSmall;
    const uint32_t base = s.shape_base[shape * kShapeRow + CH_HULL];
    for (uint32_t i = upper; i < h; i += 2u) { // triangle_fan_to_strip order (vertex.rs:28-35); the pair shares the writes
        const uint32_t c = fan_to_strip_source(i, h);
        const float2 q = pts[c < ml ? lower_chain[c] : upper_chain[c - ml]];
        s.hull_v[base + i] = {q.x, q.y};
    }
    if (!upper) s.hull_count[shape] = h;
}

// PACK Shapes per wavefront (round 6): the sorts one after the other, all lanes on one Shape; then the chain walks of ALL of them at once, a lane pair per
// Shape — the walk is serial and two lanes wide, and a wavefront's instruction is issued whether two lanes or eight take part: with one Shape per wavefront
// the 6 000 Shapes of the 50 000 glyph scene beyond 64 candidates spent four fifths of their instructions on it (a hundred points a Shape, forty instructions a
// point). The chains hold 16-bit indices into the sorted points (k_hull_small's bytes), not the points: 3 KB of LDS per Shape instead of 6.
template <uint32_t CAP, uint32_t QUEUE, uint32_t PACK>
__global__ __launch_bounds__(64) void k_hull_large(SceneDev s) {
    __shared__ float2 pts[PACK][CAP];
    __shared__ uint16_t chain[PACK][2][CAP]; // [.][0]: lower chain, [.][1]: upper chain
    __shared__ uint32_t chain_m[PACK][2], count[PACK], first_v[PACK], shape_of[PACK];
    if (!fits(s)) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t queued = s.hull_large_count[QUEUE];
    for (uint32_t q0 = blockIdx.x * PACK; q0 < queued; q0 += gridDim.x * PACK) {
        const float inf = __uint_as_float(0x7f800000u);
        for (uint32_t k = 0; k < PACK; ++k) { // ---- the sorts
            uint32_t n = 0, base = 0, shape = 0;
            if (q0 + k < queued) {
                shape = s.hull_large_list[QUEUE * s.n_shapes + q0 + k];
                base = s.shape_base[shape * kShapeRow + CH_HULL];
                n = s.shape_base[shape * kShapeRow + NCH + CH_HULL] - base;
            }
            if (lane == 0u) count[k] = n, first_v[k] = base, shape_of[k] = shape;
            if (n == 0u) continue; // (uniform)
            uint32_t padded = 1;
            while (padded < n) padded <<= 1;
            float2* const p = pts[k];
            for (uint32_t i = lane; i < padded; i += 64) p[i] = i < n ? make_float2(s.hull_cand[base + i].x, s.hull_cand[base + i].y) : make_float2(inf, inf);
            __syncthreads();
            for (uint32_t kk = 2; kk <= padded; kk <<= 1) {
                for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                    for (uint32_t i = lane; i < padded; i += 64) {
                        const uint32_t partner = i ^ j;
                        if (partner > i) {
                            const float2 a = p[i], b = p[partner];
                            const bool ascending = (i & kk) == 0;
                            if (ascending ? lex_less(b, a) : lex_less(a, b)) {
                                p[i] = b;
                                p[partner] = a;
                            }
                        }
                    }
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        if (lane < 2u * PACK) { // ---- the walks: the two monotone chains of Andrew's scan are independent — the upper one starts from the last point on a
            // stack floor of its own (convex_hull.rs:24-33) — so the even lane of a pair builds the lower and the odd lane the upper chain, each with its two top
            // points in registers (LDS is read only when a point is popped). The hull is lower[0 .. ml-1) ++ upper[0 .. mu-1).
            const uint32_t k = lane >> 1, upper = lane & 1u, n = count[k];
            const float2* const p = pts[k];
            uint16_t* const mine = chain[k][upper];
            uint32_t m = 0;
            float2 below = make_float2(0.0f, 0.0f), top = make_float2(0.0f, 0.0f); // p[mine[m - 2]], p[mine[m - 1]]
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t at = upper ? n - 1u - i : i;
                const float2 c = p[at];
                while (m > 1 && turn(below, top, c) <= kErrorMargin) {
                    m -= 1;
                    top = below;
                    if (m > 1) below = p[mine[m - 2]];
                }
                mine[m++] = (uint16_t)at;
                below = top;
                top = c;
            }
            chain_m[k][upper] = n ? m - 1u : 0u; // without the last point, which opens the other chain
        }
        __syncthreads();
        for (uint32_t k = 0; k < PACK; ++k) { // ---- the hulls, in the strip order of triangle_fan_to_strip
            if (count[k] == 0u) continue;
            const uint32_t ml = chain_m[k][0], h = ml + chain_m[k][1], base = first_v[k];
            for (uint32_t i = lane; i < h; i += 64) {
                const uint32_t c = fan_to_strip_source(i, h);
                const float2 v = pts[k][c < ml ? chain[k][0][c] : chain[k][1][c - ml]];
                s.hull_v[base + i] = {v.x, v.y};
            }
            if (lane == 0) s.hull_count[shape_of[k]] = h;
        }
        __syncthreads();
    }
}

// Shapes with more hull candidates than the LDS kernels hold (a paragraph of text as one Shape, a long dashed stroke): the same algorithm
// with the sort buffer and the chain in global memory, one 256-thread workgroup per Shape. Rare, so simple: O(n log^2 n) bitonic passes
// separated by workgroup barriers, then the serial chain on one lane.
__global__ __launch_bounds__(256) void k_hull_huge(SceneDev s) {
    __shared__ uint32_t chain_n;
    if (!fits(s)) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t queued = s.hull_large_count[2];
    for (uint32_t q = blockIdx.x; q < queued; q += gridDim.x) {
        const uint32_t shape = s.hull_large_list[2u * s.n_shapes + q];
        const uint32_t base = s.shape_base[shape * kShapeRow + CH_HULL];
        const uint32_t n = s.shape_base[shape * kShapeRow + NCH + CH_HULL] - base;
        float2* pts = s.hull_sort + 2u * (size_t)base;   // next_pow2(n) <= 2n slots
        float2* chain = s.hull_chain + 2u * (size_t)base; // the chain never holds more than 2n points
        uint32_t padded = 1;
        while (padded < n) padded <<= 1;
        const float inf = __uint_as_float(0x7f800000u);
        for (uint32_t i = tid; i < padded; i += 256u) pts[i] = i < n ? make_float2(s.hull_cand[base + i].x, s.hull_cand[base + i].y) : make_float2(inf, inf);
        __threadfence_block();
        __syncthreads();
        for (uint32_t k = 2; k <= padded; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < padded; i += 256u) {
                    const uint32_t partner = i ^ j;
                    if (partner > i) {
                        const float2 a = pts[i], b = pts[partner];
                        const bool ascending = (i & k) == 0;
                        if (ascending ? lex_less(b, a) : lex_less(a, b)) {
                            pts[i] = b;
                            pts[partner] = a;
                        }
                    }
                }
                __threadfence_block();
                __syncthreads();
            }
        }
        if (tid == 0) {
            uint32_t m = 0;
            for (uint32_t i = 0; i < n; ++i) {
                const float2 p = pts[i];
                while (m > 1 && turn(chain[m - 2], chain[m - 1], p) <= kErrorMargin) m -= 1;
                chain[m++] = p;
            }
            m -= 1;
            const uint32_t t = m + 1;
            for (uint32_t i = n; i-- > 0;) {
                const float2 p = pts[i];
                while (m > t && turn(chain[m - 2], chain[m - 1], p) <= kErrorMargin) m -= 1;
                chain[m++] = p;
            }
            m -= 1;
            chain_n = m;
        }
        __threadfence_block();
        __syncthreads();
        const uint32_t h = chain_n;
        for (uint32_t i = tid; i < h; i += 256u) {
            const float2 p = chain[fan_to_strip_source(i, h)];
            s.hull_v[base + i] = {p.x, p.y};
        }
        if (tid == 0) s.hull_count[shape] = h;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ self test
__global__ void k_fmath(int fn, const float* a, const float* b, float* out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    switch (fn) {
        case 0: out[i] = crh_atan2f(a[i], b[i]); break;
        case 1: out[i] = crh_acosf(a[i]); break;
        case 2: crh_sincosf(a[i], &s, &c); out[i] = s; break;
        case 3: crh_sincosf(a[i], &s, &c); out[i] = c; break;
        case 4: out[i] = crh_powf(a[i], b[i]); break;
        case 6: out[i] = sqrtf(a[i]); break;
        case 7: out[i] = a[i] / b[i]; break;
        case 8: out[i] = 1.0f / sqrtf(a[i] * a[i] + b[i] * b[i]); break;
        case 9: out[i] = a[i] * b[i] - 4.0f * a[i] * b[i] * b[i]; break;
        default: out[i] = crh_wgsl_mod(a[i], b[i]); break;
    }
}
void launch_fmath(int fn, const float* a, const float* b, float* out, uint64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(k_fmath, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, fn, a, b, out, n);
}

// ------------------------------------------------------------------------------------------------ launchers
void launch_stroke_records(const SceneDev& s, hipStream_t stream, void (*mark)(void*, const char*, uint64_t), void* ctx) {
    if (s.n_elems == 0) return;
    hipLaunchKernelGGL(k_stroke_records, dim3((s.n_paths + 63) / 64), dim3(64), 0, stream, s);
    if (mark) mark(ctx, "stroke_records", 0);
}
// need_totals: the capacities of the streams are not known yet (new paths). The two-pass path counts and scans in any case; the one-pass kernel
// counts (k_tess_count_runs, k_scan_runs) only then: per upload, not per frame.
// ---------------------------------------------------------------------------------------------- the element stream of an upload (round 6)
// Every Path becomes  MOVE, segment, ..., END;  the pool holds a path's start point followed by its segments' records (path.rs:15-52 layouts), so the
// point in front of a record is the end of the segment before it. Element numbers and pool offsets follow from the batch's index arrays and the
// prefix of the segments' float counts:   path p with segments [g0, g1):  MOVE = element g0 + 2 p at pool offset prefix[g0] + 2 p;  segment g = element
// g + 2 p + 1 at prefix[g] + 2 (p + 1);  END = element g1 + 2 p + 1 at prefix[g1] + 2 (p + 1) — what the host loop of rounds 1 - 5 counted out one by one.
CRH_D uint32_t last_not_above(const uint32_t* begin, uint32_t n, uint32_t x) { // the largest i <= n - 1 with begin[i] <= x (begin non-decreasing, begin[0] <= x)
    uint32_t lo = 0, hi = n; // invariant: begin[lo] <= x, (hi == n or begin[hi] > x)
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (begin[mid] <= x) lo = mid;
        else hi = mid;
    }
    return lo;
}
// ONE launch: lane i builds segment i, path i and Shape i, whichever exist. (A first version ran the segments' prefix as a device scan and the three parts as
// kernels of their own: five launches on the upload stream, each waiting for wave slots beside the raster kernel of the frame in front — new paths every
// frame went from 0.44 to 0.51 ms per step. The copy engine had never had to queue for compute units; one launch does so once.)
__global__ __launch_bounds__(256) void k_upload_build(UploadBuild u) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < u.n_segments) {
        const uint32_t g = i;
        const uint32_t p = u.seg_path[g];
        const uint32_t e = g + 2u * p + 1u, from = u.seg_prefix[g], n = u.seg_prefix[g + 1u] - from, off = from + 2u * (p + 1u);
        u.elem_type[e] = u.types[g], u.elem_off[e] = off, u.elem_prev_off[e] = off - 2u, u.elem_path[e] = p;
        if (u.elem_off_again) u.elem_off_again[e] = off;
        for (uint32_t k = 0; k < n; ++k) u.pool[off + k] = u.control[from + k];
    }
    if (i < u.n_paths) {
        const uint32_t p = i, g0 = u.path_seg[p], g1 = u.path_seg[p + 1u];
        const uint32_t e0 = g0 + 2u * p, at0 = u.seg_prefix[g0] + 2u * p, e1 = g1 + 2u * p + 1u, at1 = u.seg_prefix[g1] + 2u * (p + 1u);
        u.elem_type[e0] = ELEM_MOVE, u.elem_off[e0] = at0, u.elem_prev_off[e0] = at0 >= 2u ? at0 - 2u : 0u, u.elem_path[e0] = p;
        u.elem_type[e1] = ELEM_END, u.elem_off[e1] = at1, u.elem_prev_off[e1] = at1 - 2u, u.elem_path[e1] = p;
        if (u.elem_off_again) u.elem_off_again[e0] = at0, u.elem_off_again[e1] = at1;
        u.pool[at0] = u.start[2u * p], u.pool[at0 + 1u] = u.start[2u * p + 1u];
        u.path_elem_begin[p] = e0;
        if (p + 1u == u.n_paths) u.path_elem_begin[u.n_paths] = u.n_elems;
        u.path_shape[p] = last_not_above(u.shape_path, u.n_shapes, p);
    }
    if (i <= u.n_shapes) {
        const uint32_t p0 = i < u.n_shapes ? u.shape_path[i] : u.n_paths;
        u.shape_elem_begin[i] = p0 < u.n_paths ? u.path_seg[p0] + 2u * p0 : u.n_elems; // (a Shape without paths begins where the next path's elements do)
    }
    if (i == 0u && u.n_paths == 0u) u.path_elem_begin[0] = u.n_elems;
}
void launch_build_elements(const UploadBuild& u, hipStream_t stream) {
    const uint32_t lanes = max(max(u.n_segments, u.n_paths), u.n_shapes + 1u);
    hipLaunchKernelGGL(k_upload_build, dim3((lanes + 255u) / 256u), dim3(256), 0, stream, u);
}

void launch_tessellate(const SceneDev& s, hipStream_t stream, void (*mark)(void*, const char*, uint64_t), void* ctx, const uint64_t bytes[4], bool has_stroke, bool need_totals) {
    if (s.n_elems == 0) return;
    if (has_stroke) launch_stroke_records(s, stream, mark, ctx);
    if (s.n_runs) {
        if (!need_totals) return;
        if (s.run_block == 128u) {
            if (has_stroke) hipLaunchKernelGGL((k_tess_count_runs<true, 128>), dim3(s.n_runs), dim3(128), 0, stream, s);
            else hipLaunchKernelGGL((k_tess_count_runs<false, 128>), dim3(s.n_runs), dim3(128), 0, stream, s);
        } else {
            if (has_stroke) hipLaunchKernelGGL((k_tess_count_runs<true, kTessBlock>), dim3(s.n_runs), dim3(kTessBlock), 0, stream, s);
            else hipLaunchKernelGGL((k_tess_count_runs<false, kTessBlock>), dim3(s.n_runs), dim3(kTessBlock), 0, stream, s);
        }
        if (mark) mark(ctx, "tess_count", bytes[0]);
        hipLaunchKernelGGL(k_scan_runs, dim3(1), dim3(kScanRunsThreads), 0, stream, s);
        if (mark) mark(ctx, "tess_scan", bytes[1]);
        return;
    }
    hipLaunchKernelGGL(k_count, dim3(s.n_wg), dim3(kTessBlock), 0, stream, s);
    if (mark) mark(ctx, "tess_count", bytes[0]);
    hipLaunchKernelGGL(k_scan_rows, dim3((s.n_wg + 63u) / 64u), dim3(64), 0, stream, s);
    hipLaunchKernelGGL(k_scan_groups, dim3(1), dim3(64), 0, stream, s);
    hipLaunchKernelGGL(k_shape_rows, dim3((s.n_shapes + 255u) / 256u), dim3(256), 0, stream, s);
    if (mark) mark(ctx, "tess_scan", bytes[1]);
}
// hull_queued: what the three queues behind k_hull_small held when these paths were tessellated before ([0] <= 256 candidates, [1] <= 2048, [2] beyond), or nullptr: not known yet
void launch_emit(const SceneDev& s, hipStream_t stream, void (*mark)(void*, const char*, uint64_t), void* ctx, const uint64_t bytes[4], bool has_stroke, bool big_shapes, const uint32_t* hull_queued) {
    if (s.n_elems == 0) return;
    if (s.n_runs) {
        if (s.run_block == 128u) {
            if (has_stroke) hipLaunchKernelGGL((k_tess_runs<true, 128>), dim3(s.n_runs), dim3(128), 0, stream, s);
            else hipLaunchKernelGGL((k_tess_runs<false, 128>), dim3(s.n_runs), dim3(128), 0, stream, s);
        } else {
            if (has_stroke) hipLaunchKernelGGL((k_tess_runs<true, kTessBlock>), dim3(s.n_runs), dim3(kTessBlock), 0, stream, s);
            else hipLaunchKernelGGL((k_tess_runs<false, kTessBlock>), dim3(s.n_runs), dim3(kTessBlock), 0, stream, s);
        }
        if (mark) mark(ctx, "tess_fused", bytes[2]);
    } else {
        hipLaunchKernelGGL(k_emit, dim3(s.n_wg), dim3(kTessBlock), 0, stream, s);
        if (mark) mark(ctx, "tess_emit", bytes[2]);
    }
    if (has_stroke) {
        hipLaunchKernelGGL(k_stroke_lengths, dim3((s.n_paths + 63) / 64), dim3(64), 0, stream, s);
        if (mark) mark(ctx, "stroke_lengths", 0);
    }
    if (has_stroke || big_shapes) { // (some Shape may have more than 64 hull candidates)
        (void)hipMemsetAsync(s.hull_large_count, 0, 16, stream);
        hipLaunchKernelGGL(k_hull_queues, dim3((s.n_shapes + 255u) / 256u), dim3(256), 0, stream, s);
    }
    hipLaunchKernelGGL(k_hull_small, dim3((s.n_shapes + kHullBatch - 1u) / kHullBatch), dim3(64 * kHullWaves), 0, stream, s);
    if (mark) mark(ctx, "tess_hull", bytes[3]);
    if (has_stroke || big_shapes) { // some Shape may have more than 64 hull candidates: drain the queue
        // (a kernel whose queue is known to be empty is not launched: it would execute nothing, but wait for wave slots beside the raster grid)
        // (kHullPack Shapes of up to 256 candidates per wavefront, 3 KB of LDS each)
        if (!hull_queued || hull_queued[0]) hipLaunchKernelGGL((k_hull_large<kHullMid, 0, kHullPack>), dim3(min(((hull_queued ? hull_queued[0] : s.n_shapes) + kHullPack - 1u) / kHullPack, 16384u)), dim3(64), 0, stream, s);
        if (!hull_queued || hull_queued[1]) hipLaunchKernelGGL((k_hull_large<kHullMax, 1, 1>), dim3(min(hull_queued ? hull_queued[1] : s.n_shapes, 1024u)), dim3(64), 0, stream, s);
        if (!hull_queued || hull_queued[2]) hipLaunchKernelGGL(k_hull_huge, dim3(min(hull_queued ? hull_queued[2] : s.n_shapes, 256u)), dim3(256), 0, stream, s);
        if (mark) mark(ctx, "tess_hull_large", 0);
    }
}

} // namespace crh
