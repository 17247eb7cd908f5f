// csrc/stroke.hpp — one lane per path element for StrokeBuilder::add_path (stroke.rs:205-465).
//
// The reference strokes a path with a sequential state machine (previous point / previous tangent / first tangent /
// running length / pending strip). Here every segment lane rebuilds the state it needs from its neighbours:
//   * its start point is the end point of the previous element (the record just before its own in the pool),
//   * the "previous tangent" is the end tangent of the nearest previous non-skipped segment (look-back),
//   * strip cuts, restart markers and u16 index values follow from the scanned counts,
//   * the f32 running length is NOT associative, so lanes only store per-pair increments and a one-lane-per-path
//     kernel (k_stroke_lengths) replays the additions in the reference's order and patches texcoord.y / .z.
// The path's MOVE element emits nothing; its END element emits the closing line + joins or the end cap.
//
// Skipped (NaN-tangent, stroke.rs:267-269) segments: the reference `continue`s without updating previous_control_point and —
// for curve segments — without advancing the typed iterator it peek()ed (stroke.rs:229,238,248,257 vs :318,337,357,376), so the
// next segment of that type is stroked with the stale record. That is the one truly sequential dependency on *which data* a
// segment sees; k_stroke_records resolves it with one lane per stroked path (a walk that only evaluates end tangents) and
// publishes, per element, the record and the previous point it is processed with. Everything after that is per-element again.
#pragma once
#include "ga.hpp"
#include "scene.hpp"

namespace crh {

enum : uint8_t { PAIR_NORMAL = 0, PAIR_FIXED = 1, PAIR_AFTER_JOIN = 2, PAIR_END_CAP = 3 };

struct SegGeom {
    Pt p0, p1; // start / end as unweighted points
    Pl ts, te; // start / end tangents (stroke.rs:222-265, unweighted control points even for rational segments)
    bool skip;
};

CRH_D void quadratic_tangents(Pt a, Pt b, Pt c, Pl& ts, Pl& te) { // stroke.rs:179-187
    ts = signum(join(a, b));
    te = signum(join(b, c));
    if (is_nan(ts.c) || is_nan(te.c)) {
        ts = signum(join(a, c));
        te = ts;
    }
}
CRH_D void cubic_tangents(Pt a, Pt b, Pt c, Pt d, Pl& ts, Pl& te) { // stroke.rs:189-202
    ts = signum(join(a, b));
    if (is_nan(ts.c)) ts = signum(join(a, c));
    te = signum(join(c, d));
    if (is_nan(te.c)) te = signum(join(b, d));
    if (is_nan(ts.c) || is_nan(te.c)) te = signum(join(a, d));
}

CRH_D SegGeom segment_geometry(const SceneDev& s, uint32_t e, uint32_t type) {
    const float* p = s.pool + s.elem_off[e];
    const float* q = s.pool + s.elem_prev_off[e];
    SegGeom g;
    g.p0 = vec_to_point(q[0], q[1]);
    switch (type) {
        case ELEM_LINE:
            g.p1 = vec_to_point(p[0], p[1]);
            g.ts = signum(join(g.p0, g.p1));
            g.te = g.ts;
            break;
        case ELEM_IQ:
            g.p1 = vec_to_point(p[2], p[3]);
            quadratic_tangents(g.p0, vec_to_point(p[0], p[1]), g.p1, g.ts, g.te);
            break;
        case ELEM_IC:
            g.p1 = vec_to_point(p[4], p[5]);
            cubic_tangents(g.p0, vec_to_point(p[0], p[1]), vec_to_point(p[2], p[3]), g.p1, g.ts, g.te);
            break;
        case ELEM_RQ:
            g.p1 = vec_to_point(p[3], p[4]);
            quadratic_tangents(g.p0, vec_to_point(p[1], p[2]), g.p1, g.ts, g.te);
            break;
        default: // ELEM_RC
            g.p1 = vec_to_point(p[8], p[9]);
            cubic_tangents(g.p0, vec_to_point(p[4], p[5]), vec_to_point(p[6], p[7]), g.p1, g.ts, g.te);
            break;
    }
    g.skip = is_nan(g.ts.c) || is_nan(g.te.c);
    return g;
}

struct PrevState {
    bool first; // no non-skipped segment before
    Pt point;   // previous_control_point
    Pl tangent; // previous_tangent
};
// state of the sequential loop just before element e (e may be the END element)
CRH_D PrevState previous_state(const SceneDev& s, uint32_t e, uint32_t move) {
    PrevState st;
    for (uint32_t j = e - 1u; j > move; --j) {
        const SegGeom g = segment_geometry(s, j, s.elem_type[j]);
        if (!g.skip) {
            st.first = false;
            st.point = g.p1;
            st.tangent = g.te;
            return st;
        }
    }
    const float* p = s.pool + s.elem_off[move];
    st.first = true;
    st.point = vec_to_point(p[0], p[1]);
    st.tangent = {0.0f, 0.0f, 0.0f};
    return st;
}
CRH_D Pl first_tangent_of_path(const SceneDev& s, uint32_t move, uint32_t end) {
    for (uint32_t j = move + 1u; j < end; ++j) {
        const SegGeom g = segment_geometry(s, j, s.elem_type[j]);
        if (!g.skip) return g.ts;
    }
    return {0.0f, 0.0f, 0.0f};
}

CRH_D Pt offset_control_point(Pt cp, Pl tangent, float offset) { // stroke.rs:18-22
    return {cp.w + 0.0f * offset, cp.x + tangent.x * offset, cp.y + tangent.y * offset};
}

// emit_stroke_join (stroke.rs:53-121), geometry only
struct JoinGeom {
    bool active;
    int n_hull;
    float dot, side_sign;
    Pt v[5];
};
CRH_D JoinGeom join_geometry(const crh_stroke_options& so, Pt cp, Pl prev_t, Pl next_t) {
    JoinGeom j;
    j.dot = dot(prev_t, next_t);
    j.active = !(fabsf(j.dot - 1.0f) <= kErrorMargin);
    j.n_hull = 0;
    if (!j.active) return j;
    j.side_sign = f32_signum(meet(prev_t, next_t).w);
    const float miter_clip = so.width * so.miter_clip;
    const float side_offset = (so.offset - j.side_sign * 0.5f) * so.width;
    const Pt prev_edge_vertex = offset_control_point(cp, prev_t, side_offset);
    const Pt next_edge_vertex = offset_control_point(cp, next_t, side_offset);
    const Pl prev_edge_tangent = contract(contract(prev_t, prev_edge_vertex), prev_edge_vertex);
    const Pl next_edge_tangent = contract(contract(next_t, next_edge_vertex), next_edge_vertex);
    const Pt intersection = line_line_intersection(prev_edge_tangent, next_edge_tangent);
    j.v[0] = cp;
    j.v[1] = prev_edge_vertex;
    j.v[2] = next_edge_vertex;
    j.v[3] = intersection;
    j.v[4] = intersection;
    const bool anti_parallel = fabsf(j.dot + 1.0f) <= kErrorMargin;
    if (anti_parallel || mag(join(cp, intersection)) > miter_clip) {
        const Pl mid = anti_parallel ? neg(rotate_cw(prev_t)) : signum(prev_t + next_t);
        const Pt clipping_vertex = offset_control_point(cp, mid, -j.side_sign * miter_clip);
        const Pl clipping_plane = contract(contract(mid, clipping_vertex), clipping_vertex);
        j.v[3] = line_line_intersection(prev_edge_tangent, clipping_plane);
        j.v[4] = line_line_intersection(clipping_plane, next_edge_tangent);
        j.n_hull = 2;
    } else {
        j.n_hull = 1;
    }
    return j;
}

// ---- parameter counts (no solving needed: only the number of angle steps matters) ----------------------------
CRH_D uint32_t angle_steps(Pl st, Pl et, float angle_step, Cx& polar_start, Cx& polar_range) { // curve.rs:230-233
    polar_start = {st.x, st.y};
    const Cx polar_end = {et.x, et.y};
    polar_range = cdiv(polar_end, polar_start);
    const float f = fabsf(crh_atan2f(polar_range.im, polar_range.re) / angle_step) + 0.5f;
    if (!(f == f) || f <= 0.0f) return 0u; // `as usize` saturates: NaN -> 0
    return f >= 4294967040.0f ? 0xFFFFFFFFu : (uint32_t)f;
}

struct CubicIntervals { // at most four intervals [a_k, b_k]; no arrays: an index known at run time only would send them to scratch memory
    int n;
    float a0, b0, a1, b1, a2, b2, a3, b3;
    CRH_D float a(int k) const { return k == 0 ? a0 : (k == 1 ? a1 : (k == 2 ? a2 : a3)); }
    CRH_D float b(int k) const { return k == 0 ? b0 : (k == 1 ? b1 : (k == 2 ? b2 : b3)); }
};
// the split at the inflection points of cubic_uniform_tangent_angle! (curve.rs:257-286)
CRH_D CubicIntervals cubic_intervals(const Pt pb[4], bool integral) {
    float d[4];
    inflection_coefficients(pb, integral, d);
    Root roots[3];
    const float discriminant = integral ? integral_inflection_points(d, false, roots) : rational_inflection_points(d, false, roots);
    // the roots inside [0, 1], in the order found ...
    float v[3];
    bool f[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float t = roots[k].re / roots[k].den;
        f[k] = roots[k].den != 0.0f && t >= 0.0f && t <= 1.0f;
        v[k] = t;
    }
    int ns = (f[0] ? 1 : 0) + (f[1] ? 1 : 0) + (f[2] ? 1 : 0);
    float c0 = f[0] ? v[0] : (f[1] ? v[1] : v[2]);
    float c1 = f[0] ? (f[1] ? v[1] : v[2]) : v[2];
    float c2 = v[2];
    // ... sorted (a stable insertion sort of at most three: the same compares in the same order) ...
    if (ns >= 2 && c1 < c0) {
        const float s = c0;
        c0 = c1, c1 = s;
    }
    if (ns == 3 && c2 < c1) {
        const float s = c2;
        c2 = c1;
        if (s < c0) c1 = c0, c0 = s;
        else c1 = s;
    }
    // ... without near-duplicates: `i = 1; while i < ns { if split[i] - split[i - 1] < margin { remove split[i] } else { i += 1 } }` takes two turns at most
    {
        int i = 1;
#pragma unroll
        for (int turn = 0; turn < 2; ++turn) {
            if (i < ns) {
                const float hi = i == 1 ? c1 : c2, lo = i == 1 ? c0 : c1;
                if (hi - lo < kErrorMargin) {
                    if (i == 1) c1 = c2;
                    ns -= 1;
                } else {
                    i += 1;
                }
            }
        }
    }
    CubicIntervals iv;
    const bool shrink = fabsf(discriminant) < kErrorMargin; // (intervals stop short of an inflection point of a curve with a cusp)
    const float lo_of = shrink ? kEpsilon : 0.0f;
    auto end_of = [&](float s) { return shrink ? s - kEpsilon : s; };
    auto begin_behind = [&](float s) { return shrink ? s + kEpsilon : s; };
    (void)lo_of;
    iv.n = ns + 1;
    iv.a0 = 0.0f, iv.b0 = ns >= 1 ? end_of(c0) : 1.0f;
    iv.a1 = begin_behind(c0), iv.b1 = ns >= 2 ? end_of(c1) : 1.0f;
    iv.a2 = begin_behind(c1), iv.b2 = ns >= 3 ? end_of(c2) : 1.0f;
    iv.a3 = begin_behind(c2), iv.b3 = 1.0f;
    return iv;
}

CRH_D void stroke_power_basis(const SceneDev& s, uint32_t e, uint32_t type, Pt pb[4]) { // stroke.rs:319-383
    const float* p = s.pool + s.elem_off[e];
    const float* q = s.pool + s.elem_prev_off[e];
    const Pt prev = vec_to_point(q[0], q[1]);
    Pt cp[4];
    switch (type) {
        case ELEM_IQ:
            cp[0] = prev;
            cp[1] = vec_to_point(p[0], p[1]);
            cp[2] = vec_to_point(p[2], p[3]);
            quadratic_power_basis(cp, pb);
            pb[3] = pb[2] * 0.0f; // (never read for a quadratic; left unset, the four points stayed in scratch memory)
            break;
        case ELEM_RQ:
            cp[0] = prev;
            cp[1] = weighted_vec_to_point(p[0], p[1], p[2]);
            cp[2] = vec_to_point(p[3], p[4]);
            quadratic_power_basis(cp, pb);
            pb[3] = pb[2] * 0.0f;
            break;
        case ELEM_IC:
            cp[0] = prev;
            cp[1] = vec_to_point(p[0], p[1]);
            cp[2] = vec_to_point(p[2], p[3]);
            cp[3] = vec_to_point(p[4], p[5]);
            cubic_power_basis(cp, pb);
            break;
        default: {
            const float2 pv = point_to_vec(prev);
            cp[0] = weighted_vec_to_point(p[0], pv.x, pv.y);
            cp[1] = weighted_vec_to_point(p[1], p[4], p[5]);
            cp[2] = weighted_vec_to_point(p[2], p[6], p[7]);
            cp[3] = weighted_vec_to_point(p[3], p[8], p[9]);
            cubic_power_basis(cp, pb);
            break;
        }
    }
}

// number of curve parameters emit_curve_stroke! will visit (stroke.rs:138-141)
CRH_D uint32_t curve_parameter_count(const SceneDev& s, uint32_t e, uint32_t type, const crh_stroke_options& so, const SegGeom& g) {
    if (so.curve_approximation == CRH_CURVE_UNIFORMLY_SPACED_PARAMETERS) return so.steps;
    Cx ps, pr;
    if (type == ELEM_IQ || type == ELEM_RQ) { // curve.rs:306-322, :355-380
        const uint32_t steps = angle_steps(g.ts, g.te, so.angle_step, ps, pr);
        return (steps >= 2u ? steps - 1u : 0u) + 1u;
    }
    Pt pb[4];
    stroke_power_basis(s, e, type, pb);
    const CubicIntervals iv = cubic_intervals(pb, type == ELEM_IC);
    uint32_t n = 0;
    for (int k = 0; k < iv.n; ++k) {
        const Pl st = signum(cubic_tangent(pb, iv.a(k)));
        const Pl et = signum(cubic_tangent(pb, iv.b(k)));
        const uint32_t steps = angle_steps(st, et, so.angle_step, ps, pr);
        n += (steps >= 2u ? steps - 1u : 0u) + 1u;
    }
    return n;
}

// ---- which record / previous point every element of a stroked path is processed with (see the header comment) ----------------
__global__ __launch_bounds__(64) void k_stroke_records(SceneDev s) {
    const uint32_t path = blockIdx.x * 64u + threadIdx.x;
    if (path >= s.n_paths || s.path_stroke[path] < 0) return;
    const uint32_t move = s.path_elem_begin[path], end = s.path_elem_begin[path + 1] - 1u;
    uint32_t head_iq = ~0u, head_ic = ~0u, head_rq = ~0u, head_rc = ~0u; // element whose record is at the head of each typed iterator
    uint32_t prev_off = s.elem_off0[move];                                // previous_control_point = path.start (stroke.rs:207)
    for (uint32_t e = move + 1u; e < end; ++e) {
        const uint32_t type = s.elem_type[e];
        uint32_t head = type == ELEM_IQ ? head_iq : (type == ELEM_IC ? head_ic : (type == ELEM_RQ ? head_rq : head_rc)); // (a copy: a pointer to one of the four would keep them in scratch memory)
        uint32_t src = e; // lines are taken with next() (stroke.rs:223): always their own record
        if (type != ELEM_LINE) {
            if (head == ~0u) head = e;
            src = head;
        }
        const uint32_t rec = s.elem_off0[src];
        s.elem_off[e] = rec;
        s.elem_prev_off[e] = prev_off;
        const SegGeom g = segment_geometry(s, e, type);
        if (!g.skip && type != ELEM_LINE) {
            uint32_t j = head + 1u;
            while (j < end && s.elem_type[j] != type) ++j;
            head = j;
        }
        if (type == ELEM_IQ) head_iq = head;
        else if (type == ELEM_IC) head_ic = head;
        else if (type == ELEM_RQ) head_rq = head;
        else if (type != ELEM_LINE) head_rc = head;
        if (g.skip) continue; // stroke.rs:267-269: neither the iterator nor previous_control_point move
        const uint32_t end_point = type == ELEM_LINE ? 0u : (type == ELEM_IQ ? 2u : (type == ELEM_IC ? 4u : (type == ELEM_RQ ? 3u : 8u)));
        prev_off = rec + end_point;
    }
    s.elem_off[end] = s.elem_off0[end];
    s.elem_prev_off[end] = prev_off;
}

// ---- count ---------------------------------------------------------------------------------------------------------
CRH_D void count_tail(const SceneDev& s, uint32_t e, uint32_t path, const crh_stroke_options& so, uint32_t cnt[NCH]) {
    const uint32_t move = s.path_elem_begin[path];
    const PrevState st = previous_state(s, e, move);
    const bool any = !st.first;
    uint32_t verts = 0, cuts = 0, joints = 0, hull = 0;
    if (so.closed) {
        const float* sp = s.pool + s.elem_off[move];
        const Pt start = vec_to_point(sp[0], sp[1]);
        const Pl line_segment = join(st.point, start);
        const float length = mag(line_segment);
        const Pl first_tangent = first_tangent_of_path(s, move, e);
        bool pending = any;
        if (length > 0.0f) {
            const Pl segment_tangent = line_segment * (1.0f / length);
            const JoinGeom j1 = join_geometry(so, st.point, st.tangent, segment_tangent);
            if (j1.active) {
                joints += 1;
                hull += j1.n_hull;
                cuts += pending ? 1u : 0u;
                verts += 2;
            }
            verts += 2;
            const JoinGeom j2 = join_geometry(so, start, segment_tangent, first_tangent);
            if (j2.active) {
                joints += 1;
                hull += j2.n_hull;
                cuts += 1;
                verts += 2;
            }
            pending = true;
        } else {
            const JoinGeom j = join_geometry(so, start, st.tangent, first_tangent);
            if (j.active) {
                joints += 1;
                hull += j.n_hull;
                cuts += pending ? 1u : 0u;
                verts += 2;
                pending = true;
            }
        }
        cuts += pending ? 1u : 0u;
    } else {
        cuts += any ? 1u : 0u;
        verts += 4;
        cuts += 1;
    }
    cnt[CH_LINE_V] = verts;
    cnt[CH_LINE_CUT] = cuts;
    cnt[CH_JOINT] = joints;
    cnt[CH_HULL] = hull + verts;
}

CRH_D void count_stroke_element(const SceneDev& s, uint32_t e, uint32_t type, uint32_t path, const crh_stroke_options& so, uint32_t cnt[NCH]) {
    if (type == ELEM_MOVE) return;
    if (type == ELEM_END) {
        count_tail(s, e, path, so, cnt);
        return;
    }
    const SegGeom g = segment_geometry(s, e, type);
    if (g.skip) return;
    const uint32_t move = s.path_elem_begin[path];
    const PrevState st = previous_state(s, e, move);
    uint32_t verts = 0, cuts = 0, joints = 0, hull = 0;
    if (st.first) {
        if (!so.closed) verts += 2;
        if (so.closed || type != ELEM_LINE) verts += 2;
    } else {
        const JoinGeom j = join_geometry(so, g.p0, st.tangent, g.ts);
        if (j.active) {
            joints += 1;
            hull += j.n_hull;
            cuts += 1;
            verts += 2;
        }
    }
    if (type == ELEM_LINE)
        verts += 2;
    else
        verts += 2u * curve_parameter_count(s, e, type, so, g);
    cnt[CH_LINE_V] = verts;
    cnt[CH_LINE_CUT] = cuts;
    cnt[CH_JOINT] = joints;
    cnt[CH_HULL] = hull + verts;
}

// ---- emit ----------------------------------------------------------------------------------------------------------
struct StrokeCursor {
    const SceneDev* s;
    const crh_stroke_options* so;
    uint32_t path;
    uint32_t vertex_at;   // global line vertex index of the next vertex
    uint32_t cuts_at;     // global count of cut markers before the next index entry
    uint32_t shape_vertex; // global line vertex index of the shape's first line vertex
    uint32_t joint_at;    // global join index of the next join
    uint32_t shape_joint;
    uint32_t hull_at;

    CRH_D void hull(float2 v) {
        if (!is_finite(v.x) || !is_finite(v.y)) raise_error(*s, path, CRH_ERR_NON_FINITE);
        s->hull_cand[hull_at++] = {v.x == 0.0f ? 0.0f : v.x, v.y == 0.0f ? 0.0f : v.y};
    }
    // emit_stroke_vertices (stroke.rs:28-51). texcoord.y is patched by k_stroke_lengths unless mode == PAIR_FIXED.
    CRH_D void pair(Pt point, Pl tangent, uint32_t flags, float increment, uint8_t mode, float fixed_y) {
        const float w = so->width, o = so->offset;
        const float2 a = point_to_vec(offset_control_point(point, tangent, (o - 0.5f) * w));
        const float2 b = point_to_vec(offset_control_point(point, tangent, (o + 0.5f) * w));
        const uint32_t u = so->dynamic_stroke_options_group | flags;
        s->line_v[vertex_at] = {a.x, a.y, -0.5f, fixed_y, u};
        s->line_v[vertex_at + 1] = {b.x, b.y, 0.5f, fixed_y, u};
        const uint32_t pos = vertex_at + cuts_at;
        s->line_i[pos] = (uint16_t)(vertex_at - shape_vertex);
        s->line_i[pos + 1] = (uint16_t)(vertex_at + 1u - shape_vertex);
        s->line_inc[vertex_at >> 1] = increment;
        s->line_pair_mode[vertex_at >> 1] = mode;
        hull(a); // cut_stroke_polygon appends every line vertex to proto_hull (stroke.rs:125)
        hull(b);
        vertex_at += 2;
    }
    CRH_D void cut() { // cut_stroke_polygon with a non-empty pending strip (stroke.rs:123-132)
        s->line_i[vertex_at + cuts_at] = 0xFFFFu;
        s->line_pair_cut[(vertex_at >> 1) - 1u] = 1;
        cuts_at += 1;
    }
    // emit_stroke_join; returns false when the tangents are parallel (stroke.rs:62-65)
    CRH_D bool join_at(Pt cp, Pl prev_t, Pl next_t, bool pending) {
        const JoinGeom j = join_geometry(*so, cp, prev_t, next_t);
        if (!j.active) return false;
        hull(point_to_vec(j.v[3]));
        if (j.n_hull == 2) hull(point_to_vec(j.v[4]));
        const Pl scaled_tangent = prev_t * (1.0f / -so->width);
        for (int k = 0; k < 5; ++k) {
            const float2 p = point_to_vec(j.v[k]);
            s->joint_v[5u * joint_at + k] = {p.x, p.y, j.side_sign * join(j.v[k], scaled_tangent), dot(join(j.v[k], cp), scaled_tangent), 0.0f,
                                             so->dynamic_stroke_options_group};
            s->joint_i[6u * joint_at + k] = (uint16_t)(5u * (joint_at - shape_joint) + k);
        }
        s->joint_i[6u * joint_at + 5u] = 0xFFFFu;
        joint_at += 1;
        const float increment = crh_acosf(j.dot) / (3.14159265358979323846f * 2.0f) * so->width;
        if (pending) cut();
        pair(cp, next_t, 0u, increment, PAIR_AFTER_JOIN, 0.0f);
        return true;
    }
};

// interpolate_normal! (curve.rs:228-252): writes steps-1 parameters to out[], returns how many
template <class Solve>
CRH_D uint32_t interpolate_normal(Pl st, Pl et, float angle_step, float* out, Solve solve) {
    Cx polar_start, polar_range;
    const uint32_t steps = angle_steps(st, et, angle_step, polar_start, polar_range);
    if (steps < 2u) return 0u;
    const Cx polar_step = cpowf(polar_range, 1.0f / (float)steps);
    for (uint32_t i = 1; i < steps; ++i) {
        const Cx interpolated = cmul(polar_start, cpowi(polar_step, i));
        const Pl normal = {0.0f, interpolated.re, interpolated.im};
        Root r[4];
        int n = 0;
        solve(normal, r, n);
        float parameter = 0.0f;
        bool found = false; // (the first root inside [0, 1]; four fixed turns: r[] stays in registers — a loop to a run-time n sent it to scratch memory)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < n && !found && r[k].den != 0.0f) {
                const float t = r[k].re / r[k].den;
                if (t >= 0.0f && t <= 1.0f) parameter = t, found = true;
            }
        }
        out[i - 1u] = parameter;
    }
    return steps - 1u;
}
CRH_D void insertion_sort(float* v, uint32_t n) {
    for (uint32_t i = 1; i < n; ++i) {
        const float t = v[i];
        uint32_t j = i;
        while (j > 0 && t < v[j - 1]) {
            v[j] = v[j - 1];
            --j;
        }
        v[j] = t;
    }
}

// the four *_uniform_tangent_angle samplers (curve.rs:306-418) writing into `out` (the lane's own output slots)
CRH_D uint32_t sample_parameters(uint32_t type, const Pt pb[4], const SegGeom& g, float angle_step, float* out) {
    if (type == ELEM_IQ) {
        const Pl planes0 = {pb[1].w, pb[1].x, pb[1].y};
        const Pl planes1 = Pl{pb[2].w, pb[2].x, pb[2].y} * 2.0f;
        const uint32_t n = interpolate_normal(g.ts, g.te, angle_step, out, [&](Pl normal, Root* r, int& rn) {
            solve_linear(dot(normal, planes0), dot(normal, planes1), r, rn);
        });
        out[n] = 1.0f;
        return n + 1u;
    }
    if (type == ELEM_RQ) {
        const Pl planes0 = join(pb[1], pb[0]);
        const Pl planes1 = join(pb[2], pb[0]) * 2.0f;
        const Pl planes2 = join(pb[2], pb[1]);
        const uint32_t n = interpolate_normal(g.ts, g.te, angle_step, out, [&](Pl normal_in, Root* r, int& rn) {
            const Pl normal = rotate_cw(normal_in);
            solve_quadratic(dot(normal, planes0), dot(normal, planes1), dot(normal, planes2), r, rn);
        });
        out[n] = 1.0f;
        return n + 1u;
    }
    const bool integral = type == ELEM_IC;
    const CubicIntervals iv = cubic_intervals(pb, integral);
    uint32_t total = 0;
    for (int k = 0; k < iv.n; ++k) {
        const float a = iv.a(k), b = iv.b(k);
        Pt trimmed[4];
        reparametrize_cubic(pb, a, b, trimmed);
        const Pl st = signum(cubic_tangent(pb, a));
        const Pl et = signum(cubic_tangent(pb, b));
        float* slice = out + total;
        uint32_t n;
        if (integral) {
            const Pl p0 = {trimmed[1].w, trimmed[1].x, trimmed[1].y};
            const Pl p1 = Pl{trimmed[2].w, trimmed[2].x, trimmed[2].y} * 2.0f;
            const Pl p2 = Pl{trimmed[3].w, trimmed[3].x, trimmed[3].y} * 3.0f;
            n = interpolate_normal(st, et, angle_step, slice, [&](Pl normal, Root* r, int& rn) {
                solve_quadratic(dot(normal, p0), dot(normal, p1), dot(normal, p2), r, rn);
            });
        } else {
            const Pl p0 = join(trimmed[1], trimmed[0]);
            const Pl p1 = join(trimmed[2], trimmed[0]) * 2.0f;
            const Pl p2 = join(trimmed[2], trimmed[1]) + join(trimmed[3], trimmed[0]) * 3.0f;
            const Pl p3 = join(trimmed[3], trimmed[1]) * 2.0f;
            const Pl p4 = join(trimmed[3], trimmed[2]);
            n = interpolate_normal(st, et, angle_step, slice, [&](Pl normal_in, Root* r, int& rn) {
                const Pl normal = rotate_cw(normal_in);
                const float cf[5] = {dot(normal, p0), dot(normal, p1), dot(normal, p2), dot(normal, p3), dot(normal, p4)};
                solve_quartic(cf, r, rn);
            });
        }
        for (uint32_t i = 0; i < n; ++i) slice[i] = a + (b - a) * slice[i];
        insertion_sort(slice, n);
        slice[n] = b;
        total += n + 1u;
    }
    return total;
}

CRH_D void emit_tail(const SceneDev& s, uint32_t e, uint32_t path, const crh_stroke_options& so, StrokeCursor& cur) {
    const uint32_t move = s.path_elem_begin[path];
    const PrevState st = previous_state(s, e, move);
    const bool any = !st.first;
    if (so.closed) {
        const float* sp = s.pool + s.elem_off[move];
        const Pt start = vec_to_point(sp[0], sp[1]);
        const Pl line_segment = join(st.point, start);
        const float length = mag(line_segment);
        const Pl first_tangent = first_tangent_of_path(s, move, e);
        bool pending = any;
        if (length > 0.0f) {
            const Pl segment_tangent = line_segment * (1.0f / length);
            if (cur.join_at(st.point, st.tangent, segment_tangent, pending)) pending = true;
            cur.pair(start, segment_tangent, 0u, length, PAIR_NORMAL, 0.0f);
            pending = true;
            cur.join_at(start, segment_tangent, first_tangent, true);
        } else {
            if (cur.join_at(start, st.tangent, first_tangent, pending)) pending = true;
        }
        if (pending) cur.cut();
    } else {
        if (any) cur.cut();
        cur.pair(st.point, st.tangent, 0x10000u, 0.0f, PAIR_NORMAL, 0.0f);
        const Pl normal = rotate_cw(st.tangent);
        cur.pair(offset_control_point(st.point, normal, -0.5f * fabsf(so.width)), st.tangent, 0x10000u, 0.0f, PAIR_END_CAP, 0.0f);
        cur.cut();
    }
}

// `scan(e, ch)`: the exclusive prefix of channel ch at element e — gscan() behind the two-pass scan, the workgroup's LDS rows in k_tess_runs
template <class Scan>
CRH_D void emit_stroke_element(const SceneDev& s, uint32_t e, uint32_t type, uint32_t path, const crh_stroke_options& so, const uint32_t g[NCH], const Scan& scan) {
    if (type == ELEM_MOVE) return;
    const uint32_t shape_first = s.shape_elem_begin[s.path_shape[path]];
    StrokeCursor cur;
    cur.s = &s;
    cur.so = &so;
    cur.path = path;
    cur.vertex_at = g[CH_LINE_V];
    cur.cuts_at = g[CH_LINE_CUT];
    cur.shape_vertex = scan(shape_first, CH_LINE_V);
    cur.joint_at = g[CH_JOINT];
    cur.shape_joint = scan(shape_first, CH_JOINT);
    cur.hull_at = g[CH_HULL];
    if (type == ELEM_END) {
        emit_tail(s, e, path, so, cur);
        return;
    }
    const SegGeom sg = segment_geometry(s, e, type);
    if (sg.skip) return;
    const uint32_t move = s.path_elem_begin[path];
    const PrevState st = previous_state(s, e, move);
    if (st.first) { // stroke.rs:270-293
        if (!so.closed) {
            const Pl normal = rotate_cw(sg.ts);
            cur.pair(offset_control_point(sg.p0, normal, 0.5f * fabsf(so.width)), sg.ts, 0u, 0.0f, PAIR_FIXED, (0.0f - 0.5f * so.width) / so.width);
        }
        if (so.closed || type != ELEM_LINE) cur.pair(sg.p0, sg.ts, 0u, 0.0f, PAIR_NORMAL, 0.0f);
    } else {
        cur.join_at(sg.p0, st.tangent, sg.ts, true);
    }
    if (type == ELEM_LINE) { // stroke.rs:306-316
        cur.pair(sg.p1, sg.te, 0u, mag(join(sg.p0, sg.p1)), PAIR_NORMAL, 0.0f);
        return;
    }
    // emit_curve_stroke! (stroke.rs:134-168)
    Pt pb[4];
    stroke_power_basis(s, e, type, pb);
    const bool quadratic = type == ELEM_IQ || type == ELEM_RQ;
    float* params = s.line_inc + (cur.vertex_at >> 1); // the lane's own pair slots double as the parameter scratch
    uint32_t n;
    if (so.curve_approximation == CRH_CURVE_UNIFORMLY_SPACED_PARAMETERS) {
        n = so.steps;
        for (uint32_t i = 1; i < n + 1u; ++i) params[i - 1u] = (float)i / (float)n;
    } else {
        n = sample_parameters(type, pb, sg, so.angle_step, params);
    }
    Pt previous_point = sg.p0;
    for (uint32_t i = 0; i < n; ++i) {
        float t = params[i];
        Pl tangent = quadratic ? quadratic_tangent(pb, t) : cubic_tangent(pb, t);
        if (sqmag(tangent) == 0.0f) {
            if (t < 0.5f)
                t += kEpsilon;
            else
                t -= kEpsilon;
            tangent = quadratic ? quadratic_tangent(pb, t) : cubic_tangent(pb, t);
        }
        tangent = signum(tangent);
        Pt point = quadratic ? quadratic_point(pb, t) : cubic_point(pb, t);
        point = point * (1.0f / point.w);
        cur.pair(point, tangent, 0u, mag(join(previous_point, point)), PAIR_NORMAL, 0.0f);
        previous_point = point;
    }
}

// One lane per stroked path: replays `length_accumulator` (stroke.rs:215) over the path's vertex pairs in emission order
// and patches offset_along_path = length / width into the line vertices (texcoord.y) and joint vertices (texcoord.z).
__global__ __launch_bounds__(64) void k_stroke_lengths(SceneDev s) {
    const uint32_t path = blockIdx.x * 64u + threadIdx.x;
    if (path >= s.n_paths) return;
    const int32_t stroke = s.path_stroke[path];
    if (stroke < 0) return;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < NCH; ++c) ok = ok && s.totals[c] <= s.capacity[c];
    if (!ok) return;
    const float width = s.stroke_options[stroke].width;
    const uint32_t move = s.path_elem_begin[path], end = s.path_elem_begin[path + 1] - 1u;
    // the END element's own records come after its exclusive prefix: the path's last pair is found via the next element
    uint32_t pair_begin, pair_end, joint;
    if (s.n_runs) { // (k_tess_runs left them behind: there is no scan to ask)
        pair_begin = s.path_scan[3u * path], pair_end = s.path_scan[3u * path + 1u], joint = s.path_scan[3u * path + 2u];
    } else {
        pair_begin = gscan(s, move, CH_LINE_V) >> 1;
        pair_end = (end + 1u < s.n_elems ? gscan(s, end + 1u, CH_LINE_V) : s.totals[CH_LINE_V]) >> 1;
        joint = gscan(s, move, CH_JOINT);
    }
    float length = 0.0f;
    for (uint32_t p = pair_begin; p < pair_end; ++p) {
        const uint8_t mode = s.line_pair_mode[p];
        float y;
        if (mode == PAIR_FIXED) continue;
        if (mode == PAIR_AFTER_JOIN) {
            const float z = length / width; // stroke.rs:96
            for (int k = 0; k < 5; ++k) s.joint_v[5u * joint + k].w = z;
            joint += 1;
            length += s.line_inc[p]; // stroke.rs:111
            y = length / width;
        } else if (mode == PAIR_END_CAP) {
            y = (length + 0.5f * width) / width; // stroke.rs:458
        } else {
            length += s.line_inc[p]; // stroke.rs:156, :307, :414
            y = length / width;
        }
        s.line_v[2u * p].v = y;
        s.line_v[2u * p + 1u].v = y;
    }
}

} // namespace crh
