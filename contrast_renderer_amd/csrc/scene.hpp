// csrc/scene.hpp — HBM layout of a scene (a batch of Shapes) shared by the tessellation kernels, the tile rasterizer
// and the C ABI. See DESIGN.md "Data layout in HBM".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/contrast_hip.h"

namespace crh {

// Element stream: every Path becomes  MOVE, segment, segment, ..., END  — one lane per element.
enum : uint8_t { ELEM_LINE = 0, ELEM_IQ = 1, ELEM_IC = 2, ELEM_RQ = 3, ELEM_RC = 4, ELEM_MOVE = 5, ELEM_END = 6 };

// crh_scene_upload, round 6: the element stream is built ON THE DEVICE from the caller's struct-of-arrays batch (the host copies the batch's arrays —
// the floats canonicalised on the way, SafeFloat::from — and checks its structure; what used to be a loop over every element on the calling thread is
// one small kernel on the upload stream: csrc/tessellate.hip launch_build_elements).
struct UploadBuild {
    // the batch as the host staged it (device copies)
    const float* control;        // [n_control_floats] crh_path_batch::control_data, -0 canonicalised
    const float* start;          // [n_paths][2] crh_path_batch::path_start, -0 canonicalised
    const uint8_t* types;        // [n_segments]
    const uint32_t* path_seg;    // [n_paths + 1] crh_path_batch::path_segment_begin
    const uint32_t* shape_path;  // [n_shapes + 1] crh_path_batch::shape_path_begin
    uint32_t n_segments, n_paths, n_shapes, n_elems;
    const uint32_t* seg_prefix;  // [n_segments + 1] where a segment's record begins in control (the host sums the segments' float counts anyway, to check them against the batch's total)
    const uint32_t* seg_path;    // [n_segments] the path of every segment (from the same loop)
    // what the tessellation reads (SceneDev's pointers of the same names)
    uint8_t* elem_type;
    uint32_t *elem_off, *elem_off_again /* the second copy k_stroke_records rewrites, or nullptr */, *elem_prev_off, *elem_path;
    float* pool;
    uint32_t *path_elem_begin, *path_shape, *shape_elem_begin;
};

// Scan channels: how many records of each output stream an element emits.
enum {
    CH_LINE_V = 0,   // stroke line vertices (Vertex2f1i, 20 B)
    CH_HULL = 1,     // hull candidates (SafeFloat<f32,2>, 8 B)
    CH_LINE_CUT = 2, // 0xFFFF restart markers in the line index stream
    CH_JOINT = 3,    // stroke joins (5 x Vertex3f1i + 6 x u16 each)
    CH_SOLID_V = 4,  // fill polygon vertices (Vertex0, 8 B)
    CH_SOLID_END = 5,// filled paths closed (one restart marker each)
    CH_IQ = 6,       // integral quadratic segments (3 x Vertex2f)
    CH_IC_V = 7,     // integral cubic vertices (Vertex3f)
    CH_RQ = 8,       // rational quadratic segments (3 x Vertex3f)
    CH_RC_V = 9,     // rational cubic vertices (Vertex4f)
    NCH = 10
};

// vertex.rs:1-26
struct Vertex0 {
    float x, y;
};
struct Vertex2f {
    float x, y, u, v;
};
struct Vertex2f1i {
    float x, y, u, v;
    uint32_t i;
};
struct Vertex3f {
    float x, y, u, v, w;
};
struct Vertex3f1i {
    float x, y, u, v, w;
    uint32_t i;
};
struct Vertex4f {
    float x, y, k, l, m, n;
};

// Per-element exclusive prefix inside its kTessBlock-element workgroup (global prefix = wg_base[e >> kTessBlockShift] + this).
struct ElemScan {
    uint32_t v[NCH];
};

#ifndef CRH_TESS_BLOCK
#define CRH_TESS_BLOCK 256
#endif
constexpr int kTessBlock = CRH_TESS_BLOCK; // elements per workgroup of k_count / k_emit
constexpr int kTessBlockShift = kTessBlock == 256 ? 8 : (kTessBlock == 128 ? 7 : 6);
static_assert(kTessBlock == 256 || kTessBlock == 128 || kTessBlock == 64, "kTessBlock");

struct SceneDev {
    // ---- inputs (written once by crh_scene_upload) ----
    uint32_t n_elems, n_paths, n_shapes, n_wg;
    const uint8_t* elem_type;          // [n_elems]
    const uint32_t* elem_off0;         // [n_elems] float offset of the element's OWN record in `pool` (as uploaded)
    uint32_t* elem_off;                // [n_elems] record the element is processed with: == elem_off0 except behind a skipped curve
                                       //           segment of a stroked path (k_stroke_records: the reference's stale typed iterator)
    uint32_t* elem_prev_off;           // [n_elems] float offset of previous_control_point when the element is processed
    const uint32_t* elem_path;         // [n_elems]
    const float* pool;                 // per path: start.x start.y then the segment records (contrast_hip.h layouts)
    const uint32_t* path_elem_begin;   // [n_paths + 1] index of the path's MOVE element
    const uint32_t* path_shape;        // [n_paths]
    const int32_t* path_stroke;        // [n_paths] index into stroke_options or -1
    const uint32_t* shape_elem_begin;  // [n_shapes + 1]
    const uint32_t* shape_dyn_begin;   // [n_shapes + 1] into descriptors
    const crh_stroke_options* stroke_options;
    crh_dynamic_stroke_descriptor* descriptors; // 48 B records (renderer.rs:20-27)
    // ---- scan state ----
    ElemScan* elem_scan;   // [n_elems]
    uint32_t* wg_total;    // [n_wg][NCH]
    uint32_t* wg_base;     // [n_wg][NCH]: exclusive prefix of the row inside its group of 64 rows (k_scan_rows)
    uint32_t* group_base;  // [n_wg / 64 + 2][NCH]: exclusive prefix of the group (k_scan_groups); its first NCH words double as the group totals before that
    uint32_t* totals;      // [NCH]
    uint32_t* shape_base;  // [n_shapes][2][NCH] where each Shape's records begin and end in every stream (kShapeRow words per Shape)
    // ---- the one-pass tessellation (k_tess_runs): Shape-aligned workgroups, no scan across workgroups ----
    uint32_t n_runs;            // 0: the two-pass path (some Shape has more elements than a workgroup has lanes)
    uint32_t run_block;         // elements a run holds at most = lanes of a workgroup of k_tess_runs: kTessBlock, or 128 when that makes more than 4 096 runs of Shapes that fit 128
    const uint32_t* tess_run;   // [n_runs + 1] first Shape of every run of consecutive Shapes with <= run_block elements in total
    uint32_t* elem_cnt;         // [n_elems] what every element emits, packed (pack_counts): a property of the paths, counted once per upload
    uint32_t* run_base;         // [n_runs + 1][NCH] where every run's records begin in the ten streams (row n_runs: the totals); per upload as well
    uint32_t* path_scan;        // [n_paths][3] a stroked path's first vertex pair, the pair behind its last one, its first join (k_stroke_lengths)
    // ---- outputs ----
    uint32_t capacity[NCH];
    Vertex2f1i* line_v;
    Vertex3f1i* joint_v;
    Vertex0* solid_v;
    Vertex2f* iq_v;
    Vertex3f* ic_v;
    Vertex3f* rq_v;
    Vertex4f* rc_v;
    Vertex0* hull_cand; // proto_hull, in reference order
    Vertex0* hull_v;    // per shape at hull_cand offset: andrew() output in strip order
    uint32_t* hull_count; // [n_shapes]
    uint32_t* hull_large_list;  // [3][n_shapes] Shapes with 65..256 / 257..2048 / more hull candidates, queued by k_hull_small for k_hull_large / k_hull_huge
    uint32_t* hull_large_count; // [3] (+1 pad)
    float2* hull_sort;          // [2 * hull candidates] global-memory sort buffer of k_hull_huge: a Shape with candidates [b, b + n) owns [2b, 2b + 2n)
    float2* hull_chain;         // [2 * hull candidates] its monotone chain
    uint16_t* line_i;   // line_v + line_cut entries
    uint16_t* joint_i;  // 6 per join
    uint16_t* solid_i;  // solid_v + solid_end entries
    uint8_t* solid_flag; // per solid vertex: bit0 = parity of the position in its strip, bit1 = last vertex of the strip
    uint8_t* line_pair_cut; // per line vertex pair: 1 = the strip is cut after this pair (written only by the cutting lane)
    uint8_t* line_pair_mode; // per line vertex pair: PAIR_* (how k_stroke_lengths treats the pair)
    float* line_inc;    // per line vertex pair: length increment replayed by k_stroke_lengths
    uint32_t* status;   // atomicMin of (path_index << 8 | code); 0xFFFFFFFF = ok
};

__device__ __forceinline__ void raise_error(const SceneDev& s, uint32_t path, uint32_t code) { atomicMin(s.status, (path << 8) | code); }

constexpr uint32_t kShapeRow = 2u * NCH; // words of a Shape's row of shape_base: begin[NCH], end[NCH]

// global exclusive prefix of channel ch at element e
__device__ __forceinline__ uint32_t gscan(const SceneDev& s, uint32_t e, int ch) {
    const uint32_t row = e >> kTessBlockShift;
    return s.group_base[(row >> 6) * NCH + ch] + s.wg_base[row * NCH + ch] + s.elem_scan[e].v[ch];
}

} // namespace crh
