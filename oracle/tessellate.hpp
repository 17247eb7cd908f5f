// oracle/tessellate.hpp — TEST INFRASTRUCTURE (CPU oracle). Restatement of the reference's
// src/vertex.rs, src/convex_hull.rs, src/fill.rs, src/stroke.rs and the CPU part of
// Shape::from_paths (src/renderer.rs:20-60, :121-141, :177-215). Sequential, one Shape at a time,
// std::vector everywhere — the shape of the reference, not of the GPU implementation.
// PARITY UNPINNED against the real crate (no tests / golden vectors exist upstream):
// what pins this file are the hand-derived KATs in tests/test_oracle_kat.py.
#pragma once
#include <array>

#include "../include/contrast_hip.h"
#include "curve.hpp"

namespace oracle {

// ---- vertex.rs:1-26 (all members are 4-byte, so natural layout == #[repr(C, packed)]) ---------------
struct Vertex0 {
    float p[2];
};
struct Vertex2f {
    float p[2];
    float w[2];
};
struct Vertex2f1i {
    float p[2];
    float t[2];
    uint32_t u;
};
struct Vertex3f {
    float p[2];
    float w[3];
};
struct Vertex3f1i {
    float p[2];
    float t[3];
    uint32_t u;
};
struct Vertex4f {
    float p[2];
    float w[4];
};
static_assert(sizeof(Vertex0) == 8 && sizeof(Vertex2f) == 16 && sizeof(Vertex2f1i) == 20 && sizeof(Vertex3f) == 20 &&
                  sizeof(Vertex3f1i) == 24 && sizeof(Vertex4f) == 24,
              "vertex.rs sizes");

// vertex.rs:28-35
template <typename T>
std::vector<T> triangle_fan_to_strip(const std::vector<T>& vertices) {
    std::vector<T> result;
    result.reserve(vertices.size());
    for (size_t i = 0; i < vertices.size(); ++i) {
        const size_t src = (i & 1) == 0 ? (i >> 1) : vertices.size() - 1 - (i >> 1);
        result.push_back(vertices[src]);
    }
    return result;
}

struct ErrorSink {
    int status = CRH_OK;
    void raise(int code) {
        if (status == CRH_OK) status = code;
    }
};

// SafeFloat<f32, 2>::from (safe_float.rs:111-120): finite or panic, -0 -> +0.
struct Safe2 {
    float v[2];
};
inline Safe2 safe2(const float v[2], ErrorSink& err) {
    Safe2 s;
    for (int i = 0; i < 2; ++i) {
        float x = v[i];
        if (!std::isfinite(x)) err.raise(CRH_ERR_NON_FINITE);
        if (crh_f2u(x) == 0x80000000u) x = 0.0f;
        s.v[i] = x;
    }
    return s;
}
inline Safe2 safe2_of(Point p, ErrorSink& err) {
    float v[2];
    point_to_vec(p, v);
    return safe2(v, err);
}

// ---- convex_hull.rs:7-40 -----------------------------------------------------------------------------
inline std::vector<Vertex0> andrew(const std::vector<Safe2>& input) {
    std::vector<Safe2> pts = input;
    std::vector<Vertex0> hull;
    if (pts.size() < 3) {
        for (auto& p : pts) hull.push_back({{p.v[0], p.v[1]}});
        return hull;
    }
    // SafeFloat Ord: lexicographic partial_cmp (safe_float.rs:163-173)
    std::stable_sort(pts.begin(), pts.end(), [](const Safe2& a, const Safe2& b) {
        if (a.v[0] != b.v[0]) return a.v[0] < b.v[0];
        return a.v[1] < b.v[1];
    });
    auto turn = [&](const Vertex0& a, const Vertex0& b, const Safe2& c) {
        return regressive(regressive(vec_to_point(a.p), vec_to_point(b.p)), vec_to_point(c.v));
    };
    for (const Safe2& p : pts) {
        while (hull.size() > 1 && turn(hull[hull.size() - 2], hull[hull.size() - 1], p) <= ERROR_MARGIN) hull.pop_back();
        hull.push_back({{p.v[0], p.v[1]}});
    }
    hull.pop_back();
    const size_t t = hull.size() + 1;
    for (size_t k = pts.size(); k-- > 0;) {
        const Safe2& p = pts[k];
        while (hull.size() > t && turn(hull[hull.size() - 2], hull[hull.size() - 1], p) <= ERROR_MARGIN) hull.pop_back();
        hull.push_back({{p.v[0], p.v[1]}});
    }
    hull.pop_back();
    return hull;
}

// ---- a borrowed view of one Path (path.rs:213-230) -------------------------------------------------------
struct PathView {
    const float* start;
    const uint8_t* types;
    uint32_t n_segments;
    const float* control; // this path's records, laid out as in contrast_hip.h
    const crh_stroke_options* stroke; // nullptr = filled
};
constexpr int SEGMENT_FLOATS[5] = {2, 4, 6, 5, 10};

// ---- fill.rs -----------------------------------------------------------------------------------------------
struct FillBuilder { // fill.rs:252-260
    std::vector<uint16_t> solid_indices;
    std::vector<uint32_t> solid_restarts; // positions of the restart entries in solid_indices (not emitted; the rasterizer walks strips by position)
    std::vector<Vertex0> solid_vertices;
    std::vector<Vertex2f> integral_quadratic_vertices;
    std::vector<Vertex3f> integral_cubic_vertices;
    std::vector<Vertex3f> rational_quadratic_vertices;
    std::vector<Vertex4f> rational_cubic_vertices;
};

// fill.rs:14-32
inline bool find_double_point_issue(float discriminant, const Root roots[3], float& out) {
    if (discriminant < 0.0f) {
        float result = -1.0f;
        int inside = 0;
        for (int k = 0; k < 3; ++k) {
            if (roots[k].den != 0.0f) {
                const float parameter = roots[k].num_re / roots[k].den;
                if (0.0f < parameter && parameter < 1.0f) {
                    result = parameter;
                    inside += 1;
                }
            }
        }
        if (inside == 1) {
            out = result;
            return true;
        }
    }
    return false;
}

// fill.rs:34-49
inline void weight_derivatives(float weights[4][4], int column, Root r0, Root r1, Root r2) {
    const float power_basis[4] = {
        r0.num_re * r1.num_re * r2.num_re,
        -r0.den * r1.num_re * r2.num_re - r0.num_re * r1.den * r2.num_re - r0.num_re * r1.num_re * r2.den,
        r0.num_re * r1.den * r2.den + r0.den * r1.num_re * r2.den + r0.den * r1.den * r2.num_re,
        -r0.den * r1.den * r2.den,
    };
    weights[0][column] = power_basis[0];
    weights[1][column] = power_basis[0] + power_basis[1] * 1.0f / 3.0f;
    weights[2][column] = power_basis[0] + power_basis[1] * 2.0f / 3.0f + power_basis[2] * 1.0f / 3.0f;
    weights[3][column] = power_basis[0] + power_basis[1] + power_basis[2] + power_basis[3];
}

// fill.rs:51-68
inline void cubic_weights(float discriminant, const Root roots[3], float weights[4][4]) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) weights[i][j] = 0.0f;
    if (discriminant == 0.0f) {
        weight_derivatives(weights, 0, roots[0], roots[0], roots[2]);
        weight_derivatives(weights, 1, roots[0], roots[0], roots[0]);
        weight_derivatives(weights, 2, roots[0], roots[0], roots[0]);
    } else if (discriminant < 0.0f) {
        weight_derivatives(weights, 0, roots[0], roots[1], roots[2]);
        weight_derivatives(weights, 1, roots[0], roots[0], roots[1]);
        weight_derivatives(weights, 2, roots[1], roots[1], roots[0]);
    } else {
        weight_derivatives(weights, 0, roots[0], roots[1], roots[2]);
        weight_derivatives(weights, 1, roots[0], roots[0], roots[0]);
        weight_derivatives(weights, 2, roots[1], roots[1], roots[1]);
    }
    weight_derivatives(weights, 3, roots[2], roots[2], roots[2]);
}

// fill.rs:70-85
inline void weight_planes(const Point cp[4], const float weights[4][4], Plane planes[4]) {
    for (int i = 0; i < 4; ++i) {
        Point3 points[4];
        for (int j = 0; j < 4; ++j) points[j] = {{cp[j][0], cp[j][1], cp[j][2], weights[j][i]}};
        Plane3 plane_3d = regressive3(points[0], points[1], points[2]);
        if (squared_magnitude(plane_3d) < ERROR_MARGIN) plane_3d = regressive3(points[0], points[1], points[3]);
        plane_3d = plane_3d * (1.0f / -plane_3d[3]);
        planes[i] = {{plane_3d[0], plane_3d[1], plane_3d[2]}};
    }
}

// fill.rs:87-89
inline float implicit_curve_value(Point3 w) { return powi3(w[0]) - w[1] * w[2] * w[3]; }

// fill.rs:91-96
inline Plane implicit_curve_gradient(const Plane planes[4], const float w[4]) {
    return planes[0] * (3.0f * w[0] * w[0]) - planes[1] * (w[2] * w[3]) - planes[2] * (w[1] * w[3]) - planes[3] * (w[1] * w[2]);
}

// fill.rs:98-114 (the planes are not used after this point, so only the weights are flipped here)
inline void normalize_implicit_curve_side(float weights[4][4], const Point pb[4], Plane gradient) {
    const Plane tangent = rational_cubic_first_order_derivative(pb, 0.0f);
    if (inner(tangent, gradient) > 0.0f) {
        for (int r = 0; r < 4; ++r) {
            weights[r][0] *= -1.0f;
            weights[r][1] *= -1.0f;
        }
    }
}

// triangulate_cubic_curve_quadrilateral! + emit_cubic_curve_triangle! (fill.rs:116-204)
template <typename Emit>
inline void triangulate_cubic_curve_quadrilateral(std::vector<Vertex0>& fill_solid_vertices, const Point cp[4], Point3 weights[4], Emit emit,
                                                  ErrorSink& err) {
    for (int j = 0; j < 4; ++j) weights[j] = weights[j] * (1.0f / cp[j][0]);
    float signed_triangle_areas[4];
    for (int i = 0; i < 4; ++i) {
        Point pts[3];
        int n = 0;
        for (int j = 0; j < 4; ++j)
            if (i != j) pts[n++] = cp[j];
        signed_triangle_areas[i] = regressive(regressive(pts[0], pts[1]), pts[2]);
    }
    const float triangle_area_sum = std::fabs(signed_triangle_areas[0]) + std::fabs(signed_triangle_areas[1]) +
                                    std::fabs(signed_triangle_areas[2]) + std::fabs(signed_triangle_areas[3]);
    int enclosing_triangle = -1;
    for (int i = 0; i < 4; ++i) {
        const float equilibrium = 0.5f * triangle_area_sum;
        if (std::fabs(equilibrium - std::fabs(signed_triangle_areas[i])) <= ERROR_MARGIN) enclosing_triangle = enclosing_triangle < 0 ? i : -1;
    }
    auto emit_triangle = [&](int triangle_index) { // fill.rs:116-132
        int idx[3];
        int n = 0;
        for (int v = 0; v < 4; ++v)
            if (v != triangle_index) idx[n++] = v;
        const float area = signed_triangle_areas[triangle_index];
        if (std::fabs(area) > ERROR_MARGIN) {
            if (area < 0.0f) std::swap(idx[0], idx[2]);
            for (int k = 0; k < 3; ++k) {
                float v[2];
                point_to_vec(cp[idx[k]], v);
                emit(v, weights[idx[k]]);
            }
        }
    };
    if (enclosing_triangle >= 0) {
        emit_triangle(enclosing_triangle);
    } else {
        int opposite_triangle = 0;
        for (int j = 1; j < 4; ++j) {
            const float side_of_a = signed_triangle_areas[j];
            const float side_of_d = signed_triangle_areas[0] * (j == 2 ? -1.0f : 1.0f);
            if (side_of_a * side_of_d < 0.0f) {
                if (opposite_triangle != 0) err.raise(CRH_ERR_DEGENERATE_CUBIC); // assert_eq!, fill.rs:174
                opposite_triangle = j;
            }
        }
        if (opposite_triangle == 0) err.raise(CRH_ERR_DEGENERATE_CUBIC); // assert_ne!, fill.rs:178
        emit_triangle(0);
        emit_triangle(opposite_triangle);
    }
    int additional_vertices = 0;
    for (int i = 1; i < 3; ++i) {
        if (enclosing_triangle != i && implicit_curve_value(weights[i]) < 0.0f) {
            Vertex0 v;
            point_to_vec(cp[i], v.p);
            fill_solid_vertices.push_back(v);
            additional_vertices += 1;
        }
    }
    if (additional_vertices == 2 && signed_triangle_areas[0] * signed_triangle_areas[1] < 0.0f) {
        const size_t length = fill_solid_vertices.size();
        std::swap(fill_solid_vertices[length - 2], fill_solid_vertices[length - 1]);
    }
}

// emit_cubic_curve! + split_curve_at! (fill.rs:206-250)
template <typename Emit>
inline void emit_cubic_curve(std::vector<Safe2>& proto_hull, std::vector<Vertex0>& fill_solid_vertices, const Point cp[4], const Point pb[4],
                             float discriminant, const Root roots[3], Emit emit, ErrorSink& err) {
    float w[4][4];
    cubic_weights(discriminant, roots, w);
    Plane planes[4];
    weight_planes(cp, w, planes);
    const Plane gradient = implicit_curve_gradient(planes, w[0]);
    normalize_implicit_curve_side(w, pb, gradient);
    Point3 weights[4];
    for (int j = 0; j < 4; ++j) weights[j] = {{w[j][0], w[j][1], w[j][2], w[j][3]}};
    float param;
    if (find_double_point_issue(discriminant, roots, param)) {
        auto lerp2 = [&](Point a, Point b) { return a * (1.0f - param) + b * param; };
        auto lerp3 = [&](Point3 a, Point3 b) { return a * (1.0f - param) + b * param; };
        const Point p10 = lerp2(cp[0], cp[1]), p11 = lerp2(cp[1], cp[2]), p12 = lerp2(cp[2], cp[3]);
        const Point p20 = lerp2(p10, p11), p21 = lerp2(p11, p12);
        const Point p30 = lerp2(p20, p21);
        const Point cp_a[4] = {cp[0], p10, p20, p30};
        const Point cp_b[4] = {p30, p21, p12, cp[3]};
        const Point3 w10 = lerp3(weights[0], weights[1]), w11 = lerp3(weights[1], weights[2]), w12 = lerp3(weights[2], weights[3]);
        const Point3 w20 = lerp3(w10, w11), w21 = lerp3(w11, w12);
        const Point3 w30 = lerp3(w20, w21);
        Point3 weights_a[4] = {weights[0], w10, w20, w30};
        Point3 weights_b[4] = {w30, w21, w12, weights[3]};
        triangulate_cubic_curve_quadrilateral(fill_solid_vertices, cp_a, weights_a, emit, err);
        Vertex0 mid;
        point_to_vec(cp_b[0], mid.p);
        fill_solid_vertices.push_back(mid);
        for (auto& wb : weights_b) {
            wb[0] *= -1.0f;
            wb[1] *= -1.0f;
        }
        triangulate_cubic_curve_quadrilateral(fill_solid_vertices, cp_b, weights_b, emit, err);
    } else {
        triangulate_cubic_curve_quadrilateral(fill_solid_vertices, cp, weights, emit, err);
    }
    proto_hull.push_back(safe2_of(cp[1], err));
    proto_hull.push_back(safe2_of(cp[2], err));
    proto_hull.push_back(safe2_of(cp[3], err));
    Vertex0 last;
    point_to_vec(cp[3], last.p);
    fill_solid_vertices.push_back(last);
}

// FillBuilder::add_path (fill.rs:263-367)
inline void fill_add_path(FillBuilder& self, std::vector<Safe2>& proto_hull, const PathView& path, ErrorSink& err) {
    std::vector<Vertex0> path_solid_vertices;
    path_solid_vertices.push_back({{path.start[0], path.start[1]}});
    proto_hull.push_back(safe2(path.start, err));
    const float* seg = path.control;
    for (uint32_t s = 0; s < path.n_segments; ++s) {
        const int type = path.types[s];
        switch (type) {
            case CRH_SEGMENT_LINE: {
                proto_hull.push_back(safe2(seg, err));
                path_solid_vertices.push_back({{seg[0], seg[1]}});
                break;
            }
            case CRH_SEGMENT_INTEGRAL_QUADRATIC: {
                const Vertex0 prev = path_solid_vertices.back();
                self.integral_quadratic_vertices.push_back({{seg[2], seg[3]}, {1.0f, 1.0f}});
                self.integral_quadratic_vertices.push_back({{seg[0], seg[1]}, {0.5f, 0.0f}});
                self.integral_quadratic_vertices.push_back({{prev.p[0], prev.p[1]}, {0.0f, 0.0f}});
                proto_hull.push_back(safe2(seg, err));
                proto_hull.push_back(safe2(seg + 2, err));
                path_solid_vertices.push_back({{seg[2], seg[3]}});
                break;
            }
            case CRH_SEGMENT_INTEGRAL_CUBIC: {
                const Vertex0 prev = path_solid_vertices.back();
                const Point cp[4] = {vec_to_point(prev.p), vec_to_point(seg), vec_to_point(seg + 2), vec_to_point(seg + 4)};
                Point pb[4];
                rational_cubic_control_points_to_power_basis(cp, pb);
                float ippc[4];
                inflection_point_polynomial_coefficients(pb, true, ippc);
                Root roots[3];
                const float discriminant = integral_inflection_points(ippc, true, roots);
                emit_cubic_curve(
                    proto_hull, path_solid_vertices, cp, pb, discriminant, roots,
                    [&](const float v[2], Point3 w) { self.integral_cubic_vertices.push_back({{v[0], v[1]}, {w[0], w[1], w[2]}}); }, err);
                break;
            }
            case CRH_SEGMENT_RATIONAL_QUADRATIC: {
                const Vertex0 prev = path_solid_vertices.back();
                const float weight = 1.0f / seg[0];
                self.rational_quadratic_vertices.push_back({{seg[3], seg[4]}, {1.0f, 1.0f, 1.0f}});
                self.rational_quadratic_vertices.push_back({{seg[1], seg[2]}, {0.5f * weight, 0.0f, weight}});
                self.rational_quadratic_vertices.push_back({{prev.p[0], prev.p[1]}, {0.0f, 0.0f, 1.0f}});
                proto_hull.push_back(safe2(seg + 1, err));
                proto_hull.push_back(safe2(seg + 3, err));
                path_solid_vertices.push_back({{seg[3], seg[4]}});
                break;
            }
            case CRH_SEGMENT_RATIONAL_CUBIC: {
                const Vertex0 prev = path_solid_vertices.back();
                const Point cp[4] = {weighted_vec_to_point(seg[0], prev.p), weighted_vec_to_point(seg[1], seg + 4),
                                     weighted_vec_to_point(seg[2], seg + 6), weighted_vec_to_point(seg[3], seg + 8)};
                Point pb[4];
                rational_cubic_control_points_to_power_basis(cp, pb);
                float ippc[4];
                inflection_point_polynomial_coefficients(pb, false, ippc);
                Root roots[3];
                const float discriminant = rational_inflection_points(ippc, true, roots);
                emit_cubic_curve(
                    proto_hull, path_solid_vertices, cp, pb, discriminant, roots,
                    [&](const float v[2], Point3 w) { self.rational_cubic_vertices.push_back({{v[0], v[1]}, {w[0], w[1], w[2], w[3]}}); }, err);
                break;
            }
        }
        seg += SEGMENT_FLOATS[type];
    }
    const size_t start_index = self.solid_vertices.size();
    const std::vector<Vertex0> strip = triangle_fan_to_strip(path_solid_vertices);
    self.solid_vertices.insert(self.solid_vertices.end(), strip.begin(), strip.end());
    for (size_t i = start_index; i < self.solid_vertices.size() + 1; ++i) self.solid_indices.push_back((uint16_t)i);
    self.solid_indices.back() = 0xFFFF;
    self.solid_restarts.push_back((uint32_t)self.solid_indices.size() - 1u);
}

// ---- stroke.rs ----------------------------------------------------------------------------------------------
struct StrokeBuilder { // stroke.rs:170-177
    std::vector<uint16_t> line_indices;
    std::vector<uint16_t> joint_indices;
    std::vector<uint32_t> line_restarts, joint_restarts; // positions of the restart entries (not emitted)
    std::vector<Vertex2f1i> line_vertices;
    std::vector<Vertex3f1i> joint_vertices;
    std::vector<Vertex2f1i> path_line_vertices;
};

// stroke.rs:18-22
inline Point offset_control_point(Point control_point, Plane tangent, float offset) {
    Point direction = dual(tangent);
    direction[0] = 0.0f;
    return control_point + direction * offset;
}
// stroke.rs:24-26
inline void emit_stroke_vertex(std::vector<Vertex2f1i>& out, uint32_t path_index, float offset_along_path, Point vertex, float side) {
    Vertex2f1i v;
    point_to_vec(vertex, v.p);
    v.t[0] = side;
    v.t[1] = offset_along_path;
    v.u = path_index;
    out.push_back(v);
}
// stroke.rs:28-51
inline void emit_stroke_vertices(StrokeBuilder& builder, const crh_stroke_options& so, uint32_t path_index, float length_accumulator, Point point,
                                 Plane tangent) {
    const float offset_along_path = length_accumulator / so.width;
    emit_stroke_vertex(builder.path_line_vertices, path_index, offset_along_path, offset_control_point(point, tangent, (so.offset - 0.5f) * so.width),
                       -0.5f);
    emit_stroke_vertex(builder.path_line_vertices, path_index, offset_along_path, offset_control_point(point, tangent, (so.offset + 0.5f) * so.width),
                       0.5f);
}
// stroke.rs:123-132
inline void cut_stroke_polygon(StrokeBuilder& builder, std::vector<Safe2>& proto_hull, ErrorSink& err) {
    if (!builder.path_line_vertices.empty()) {
        for (auto& v : builder.path_line_vertices) proto_hull.push_back(safe2(v.p, err));
        const size_t start_index = builder.line_vertices.size();
        builder.line_vertices.insert(builder.line_vertices.end(), builder.path_line_vertices.begin(), builder.path_line_vertices.end());
        builder.path_line_vertices.clear();
        for (size_t i = start_index; i < builder.line_vertices.size() + 1; ++i) builder.line_indices.push_back((uint16_t)i);
        builder.line_indices.back() = 0xFFFF;
        builder.line_restarts.push_back((uint32_t)builder.line_indices.size() - 1u);
    }
}
// stroke.rs:53-121
inline void emit_stroke_join(StrokeBuilder& builder, std::vector<Safe2>& proto_hull, const crh_stroke_options& so, float& length_accumulator,
                             Point control_point, Plane previous_tangent, Plane next_tangent, ErrorSink& err) {
    const float tangets_dot_product = inner(previous_tangent, next_tangent);
    if (std::fabs(tangets_dot_product - 1.0f) <= ERROR_MARGIN) return;
    const float side_sign = f32_signum(outer(previous_tangent, next_tangent)[0]);
    const float miter_clip = so.width * so.miter_clip;
    const float side_offset = (so.offset - side_sign * 0.5f) * so.width;
    const Point previous_edge_vertex = offset_control_point(control_point, previous_tangent, side_offset);
    const Point next_edge_vertex = offset_control_point(control_point, next_tangent, side_offset);
    const Plane previous_edge_tangent = geometric_product_grade1(inner(previous_tangent, previous_edge_vertex), previous_edge_vertex);
    const Plane next_edge_tangent = geometric_product_grade1(inner(next_tangent, next_edge_vertex), next_edge_vertex);
    const Point intersection = line_line_intersection(previous_edge_tangent, next_edge_tangent);
    Point vertices[5] = {control_point, previous_edge_vertex, next_edge_vertex, intersection, intersection};
    const bool anti_parallel = std::fabs(tangets_dot_product + 1.0f) <= ERROR_MARGIN;
    if (anti_parallel || magnitude(regressive(control_point, intersection)) > miter_clip) {
        const Plane mid_tangent = anti_parallel ? -rotate_90_degree_clockwise(previous_tangent) : signum(previous_tangent + next_tangent);
        const Point clipping_vertex = offset_control_point(control_point, mid_tangent, -side_sign * miter_clip);
        const Plane clipping_plane = geometric_product_grade1(inner(mid_tangent, clipping_vertex), clipping_vertex);
        vertices[3] = line_line_intersection(previous_edge_tangent, clipping_plane);
        vertices[4] = line_line_intersection(clipping_plane, next_edge_tangent);
        proto_hull.push_back(safe2_of(vertices[3], err));
        proto_hull.push_back(safe2_of(vertices[4], err));
    } else {
        proto_hull.push_back(safe2_of(vertices[3], err));
    }
    const Plane scaled_tangent = previous_tangent * (1.0f / -so.width);
    const size_t start_index = builder.joint_vertices.size();
    const float offset_along_path = length_accumulator / so.width;
    for (const Point& vertex : vertices) {
        Vertex3f1i v;
        point_to_vec(vertex, v.p);
        v.t[0] = side_sign * regressive(vertex, scaled_tangent);
        v.t[1] = inner(regressive(vertex, control_point), scaled_tangent);
        v.t[2] = offset_along_path;
        v.u = so.dynamic_stroke_options_group;
        builder.joint_vertices.push_back(v);
    }
    for (size_t i = start_index; i < builder.joint_vertices.size() + 1; ++i) builder.joint_indices.push_back((uint16_t)i);
    builder.joint_indices.back() = 0xFFFF;
    builder.joint_restarts.push_back((uint32_t)builder.joint_indices.size() - 1u);
    length_accumulator += crh_acosf(tangets_dot_product) / (3.14159265358979323846f * 2.0f) * so.width;
    cut_stroke_polygon(builder, proto_hull, err);
    emit_stroke_vertices(builder, so, so.dynamic_stroke_options_group, length_accumulator, control_point, next_tangent);
}

// stroke.rs:179-187
inline void get_quadratic_tangents(const Point cp[3], Plane& start, Plane& end) {
    start = signum(regressive(cp[0], cp[1]));
    end = signum(regressive(cp[1], cp[2]));
    if (std::isnan(start[0]) || std::isnan(end[0])) {
        start = signum(regressive(cp[0], cp[2]));
        end = start;
    }
}
// stroke.rs:189-202
inline void get_cubic_tangents(const Point cp[4], Plane& start, Plane& end) {
    start = signum(regressive(cp[0], cp[1]));
    if (std::isnan(start[0])) start = signum(regressive(cp[0], cp[2]));
    end = signum(regressive(cp[2], cp[3]));
    if (std::isnan(end[0])) end = signum(regressive(cp[1], cp[3]));
    if (std::isnan(start[0]) || std::isnan(end[0])) end = signum(regressive(cp[0], cp[3]));
}

// emit_curve_stroke! (stroke.rs:134-168)
template <typename PointFn, typename TangentFn, typename Pb>
inline void emit_curve_stroke(StrokeBuilder& builder, const crh_stroke_options& so, float& length_accumulator, Point previous_control_point,
                              const Pb& power_basis, const std::vector<float>& parameters, PointFn point_fn, TangentFn tangent_fn) {
    Point previous_point = previous_control_point;
    for (float t : parameters) {
        Plane tangent = tangent_fn(power_basis, t);
        if (squared_magnitude(tangent) == 0.0f) {
            if (t < 0.5f)
                t += F32_EPSILON;
            else
                t -= F32_EPSILON;
            tangent = tangent_fn(power_basis, t);
        }
        tangent = signum(tangent);
        Point point = point_fn(power_basis, t);
        point = point * (1.0f / point[0]);
        length_accumulator += magnitude(regressive(previous_point, point));
        emit_stroke_vertices(builder, so, so.dynamic_stroke_options_group, length_accumulator, point, tangent);
        previous_point = point;
    }
}
inline std::vector<float> uniformly_spaced_parameters(uint32_t steps) { // stroke.rs:139
    std::vector<float> p;
    for (uint64_t i = 1; i < (uint64_t)steps + 1; ++i) p.push_back((float)i / (float)steps);
    return p;
}

// StrokeBuilder::add_path (stroke.rs:205-465)
inline void stroke_add_path(StrokeBuilder& self, std::vector<Safe2>& proto_hull, const PathView& path, ErrorSink& err) {
    const crh_stroke_options& so = *path.stroke;
    const bool uniform_angle = so.curve_approximation == CRH_CURVE_UNIFORM_TANGENT_ANGLE;
    Point previous_control_point = vec_to_point(path.start);
    Plane first_tangent = {{0, 0, 0}};
    Plane previous_tangent = {{0, 0, 0}};
    float length_accumulator = 0.0f;
    bool is_first_segment = true;
    // The reference walks five typed Vecs with five iterators (stroke.rs:210-214). Lines are taken with
    // `next()` (stroke.rs:223); curves are `peek()`ed (stroke.rs:229,238,248,257) and only `next()`ed after
    // the NaN-tangent `continue` (stroke.rs:267-269 vs :318,337,357,376) — so a skipped curve segment stays
    // at the head of its iterator and the next segment of that type reads the same record again.
    // Model: per-type record lists of this path + per-type cursors.
    std::vector<const float*> records[5];
    {
        const float* r = path.control;
        for (uint32_t s = 0; s < path.n_segments; ++s) {
            records[path.types[s]].push_back(r);
            r += SEGMENT_FLOATS[path.types[s]];
        }
    }
    size_t cursor[5] = {0, 0, 0, 0, 0};
    for (uint32_t s = 0; s < path.n_segments; ++s) {
        const int type = path.types[s];
        const float* seg = records[type][cursor[type]];
        if (type == CRH_SEGMENT_LINE) cursor[type] += 1;
        Point next_control_point;
        Plane segment_start_tangent, segment_end_tangent;
        switch (type) {
            case CRH_SEGMENT_LINE: {
                next_control_point = vec_to_point(seg);
                segment_start_tangent = signum(regressive(previous_control_point, next_control_point));
                segment_end_tangent = segment_start_tangent;
                break;
            }
            case CRH_SEGMENT_INTEGRAL_QUADRATIC: {
                next_control_point = vec_to_point(seg + 2);
                const Point cp[3] = {previous_control_point, vec_to_point(seg), next_control_point};
                get_quadratic_tangents(cp, segment_start_tangent, segment_end_tangent);
                break;
            }
            case CRH_SEGMENT_INTEGRAL_CUBIC: {
                next_control_point = vec_to_point(seg + 4);
                const Point cp[4] = {previous_control_point, vec_to_point(seg), vec_to_point(seg + 2), next_control_point};
                get_cubic_tangents(cp, segment_start_tangent, segment_end_tangent);
                break;
            }
            case CRH_SEGMENT_RATIONAL_QUADRATIC: {
                next_control_point = vec_to_point(seg + 3);
                const Point cp[3] = {previous_control_point, vec_to_point(seg + 1), next_control_point};
                get_quadratic_tangents(cp, segment_start_tangent, segment_end_tangent);
                break;
            }
            default: { // RationalCubicCurve
                next_control_point = vec_to_point(seg + 8);
                const Point cp[4] = {previous_control_point, vec_to_point(seg + 4), vec_to_point(seg + 6), next_control_point};
                get_cubic_tangents(cp, segment_start_tangent, segment_end_tangent);
                break;
            }
        }
        if (std::isnan(segment_start_tangent[0]) || std::isnan(segment_end_tangent[0])) continue;
        if (type != CRH_SEGMENT_LINE) cursor[type] += 1;
        if (is_first_segment) {
            is_first_segment = false;
            first_tangent = segment_start_tangent;
            if (!so.closed) {
                const Plane normal = rotate_90_degree_clockwise(segment_start_tangent);
                emit_stroke_vertices(self, so, so.dynamic_stroke_options_group, length_accumulator - 0.5f * so.width,
                                     offset_control_point(previous_control_point, normal, 0.5f * std::fabs(so.width)), segment_start_tangent);
            }
            if (so.closed || type != CRH_SEGMENT_LINE)
                emit_stroke_vertices(self, so, so.dynamic_stroke_options_group, length_accumulator, previous_control_point, segment_start_tangent);
        } else {
            emit_stroke_join(self, proto_hull, so, length_accumulator, previous_control_point, previous_tangent, segment_start_tangent, err);
        }
        switch (type) {
            case CRH_SEGMENT_LINE: {
                length_accumulator += magnitude(regressive(previous_control_point, next_control_point));
                emit_stroke_vertices(self, so, so.dynamic_stroke_options_group, length_accumulator, next_control_point, segment_end_tangent);
                break;
            }
            case CRH_SEGMENT_INTEGRAL_QUADRATIC: {
                const Point cp[3] = {previous_control_point, vec_to_point(seg), vec_to_point(seg + 2)};
                std::array<Point, 3> pb;
                rational_quadratic_control_points_to_power_basis(cp, pb.data());
                const std::vector<float> parameters =
                    uniform_angle ? integral_quadratic_uniform_tangent_angle(pb.data(), segment_start_tangent, segment_end_tangent, so.angle_step)
                                  : uniformly_spaced_parameters(so.steps);
                emit_curve_stroke(
                    self, so, length_accumulator, previous_control_point, pb, parameters,
                    [](const std::array<Point, 3>& b, float t) { return rational_quadratic_point(b.data(), t); },
                    [](const std::array<Point, 3>& b, float t) { return rational_quadratic_first_order_derivative(b.data(), t); });
                break;
            }
            case CRH_SEGMENT_INTEGRAL_CUBIC: {
                const Point cp[4] = {previous_control_point, vec_to_point(seg), vec_to_point(seg + 2), vec_to_point(seg + 4)};
                std::array<Point, 4> pb;
                rational_cubic_control_points_to_power_basis(cp, pb.data());
                const std::vector<float> parameters =
                    uniform_angle ? integral_cubic_uniform_tangent_angle(pb.data(), so.angle_step) : uniformly_spaced_parameters(so.steps);
                emit_curve_stroke(
                    self, so, length_accumulator, previous_control_point, pb, parameters,
                    [](const std::array<Point, 4>& b, float t) { return rational_cubic_point(b.data(), t); },
                    [](const std::array<Point, 4>& b, float t) { return rational_cubic_first_order_derivative(b.data(), t); });
                break;
            }
            case CRH_SEGMENT_RATIONAL_QUADRATIC: {
                const Point cp[3] = {previous_control_point, weighted_vec_to_point(seg[0], seg + 1), vec_to_point(seg + 3)};
                std::array<Point, 3> pb;
                rational_quadratic_control_points_to_power_basis(cp, pb.data());
                const std::vector<float> parameters =
                    uniform_angle ? rational_quadratic_uniform_tangent_angle(pb.data(), segment_start_tangent, segment_end_tangent, so.angle_step)
                                  : uniformly_spaced_parameters(so.steps);
                emit_curve_stroke(
                    self, so, length_accumulator, previous_control_point, pb, parameters,
                    [](const std::array<Point, 3>& b, float t) { return rational_quadratic_point(b.data(), t); },
                    [](const std::array<Point, 3>& b, float t) { return rational_quadratic_first_order_derivative(b.data(), t); });
                break;
            }
            default: {
                float prev_vec[2];
                point_to_vec(previous_control_point, prev_vec);
                const Point cp[4] = {weighted_vec_to_point(seg[0], prev_vec), weighted_vec_to_point(seg[1], seg + 4),
                                     weighted_vec_to_point(seg[2], seg + 6), weighted_vec_to_point(seg[3], seg + 8)};
                std::array<Point, 4> pb;
                rational_cubic_control_points_to_power_basis(cp, pb.data());
                const std::vector<float> parameters =
                    uniform_angle ? rational_cubic_uniform_tangent_angle(pb.data(), so.angle_step) : uniformly_spaced_parameters(so.steps);
                emit_curve_stroke(
                    self, so, length_accumulator, previous_control_point, pb, parameters,
                    [](const std::array<Point, 4>& b, float t) { return rational_cubic_point(b.data(), t); },
                    [](const std::array<Point, 4>& b, float t) { return rational_cubic_first_order_derivative(b.data(), t); });
                break;
            }
        }
        previous_control_point = next_control_point;
        previous_tangent = segment_end_tangent;
    }
    if (so.closed) {
        const Point start_point = vec_to_point(path.start);
        const Plane line_segment = regressive(previous_control_point, start_point);
        const float length = magnitude(line_segment);
        if (length > 0.0f) {
            const Plane segment_tangent = geometric_quotient(line_segment, length);
            emit_stroke_join(self, proto_hull, so, length_accumulator, previous_control_point, previous_tangent, segment_tangent, err);
            length_accumulator += length;
            emit_stroke_vertices(self, so, so.dynamic_stroke_options_group, length_accumulator, start_point, segment_tangent);
            emit_stroke_join(self, proto_hull, so, length_accumulator, start_point, segment_tangent, first_tangent, err);
        } else {
            emit_stroke_join(self, proto_hull, so, length_accumulator, start_point, previous_tangent, first_tangent, err);
        }
    } else {
        cut_stroke_polygon(self, proto_hull, err);
        emit_stroke_vertices(self, so, so.dynamic_stroke_options_group | 0x10000u, length_accumulator, previous_control_point, previous_tangent);
        const Plane normal = rotate_90_degree_clockwise(previous_tangent);
        emit_stroke_vertices(self, so, so.dynamic_stroke_options_group | 0x10000u, length_accumulator + 0.5f * so.width,
                             offset_control_point(previous_control_point, normal, -0.5f * std::fabs(so.width)), previous_tangent);
    }
    cut_stroke_polygon(self, proto_hull, err);
}

// ---- renderer.rs:20-60 -----------------------------------------------------------------------------------------
inline int convert_dynamic_stroke_options(const crh_dynamic_stroke_options& o, crh_dynamic_stroke_descriptor& result) {
    std::memset(&result, 0, sizeof(result));
    if (o.dashed) {
        if (o.pattern_len > CRH_MAX_DASH_INTERVALS) return CRH_ERR_TOO_MANY_DASH_INTERVALS;
        if (o.pattern_len == 0) return CRH_ERR_INVALID_ARGUMENT; // `pattern.len() as u32 - 1` underflows in the reference
        result.count_dashed_join = ((o.pattern_len - 1) << 3) | 4 | o.join;
        result.phase = o.phase;
        for (uint32_t i = 0; i < o.pattern_len; ++i) {
            result.gap_start[i] = o.pattern[i].gap_start;
            result.gap_end[i] = o.pattern[i].gap_end;
            result.caps |= o.pattern[i].dash_start << (((i + o.pattern_len - 1) % o.pattern_len) * 8);
            result.caps |= o.pattern[i].dash_end << (i * 8 + 4);
        }
    } else {
        result.caps = o.start | (o.end << 4);
        result.count_dashed_join = o.join;
        result.phase = 0.0f;
    }
    return CRH_OK;
}

// ---- the CPU half of Shape::from_paths (renderer.rs:177-215) -----------------------------------------------------
struct Shape {
    uint64_t vertex_offsets[8];
    uint64_t index_offsets[3];
    std::vector<uint8_t> vertex_buffer;
    std::vector<uint8_t> index_buffer;
    std::vector<crh_dynamic_stroke_descriptor> stroke_buffer;
    // kept un-concatenated for the software rasterizer
    StrokeBuilder stroke;
    FillBuilder fill;
    std::vector<Vertex0> convex_hull;
    std::vector<Safe2> hull_candidates; // what andrew() was given, in emission order (tools/proto_hull.cpp, tests/test_hull_formulations.py)
    int status = CRH_OK;
};

template <typename T>
inline void append_bytes(std::vector<uint8_t>& dst, const std::vector<T>& src) {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(src.data());
    dst.insert(dst.end(), p, p + src.size() * sizeof(T));
}

inline void shape_from_paths(Shape& shape, const crh_dynamic_stroke_options* dynamic_stroke_options, uint32_t n_dynamic, const PathView* paths,
                             uint32_t n_paths) {
    ErrorSink err;
    std::vector<Safe2> proto_hull;
    for (uint32_t p = 0; p < n_paths; ++p) {
        if (paths[p].stroke) {
            if (paths[p].stroke->dynamic_stroke_options_group >= n_dynamic) { // renderer.rs:189-191
                shape.status = CRH_ERR_DYNAMIC_STROKE_OPTIONS_INDEX_OUT_OF_BOUNDS;
                return;
            }
            stroke_add_path(shape.stroke, proto_hull, paths[p], err);
        } else {
            fill_add_path(shape.fill, proto_hull, paths[p], err);
        }
    }
    shape.convex_hull = triangle_fan_to_strip(andrew(proto_hull));
    shape.hull_candidates = std::move(proto_hull);
    // concat_buffers! (renderer.rs:121-141, :198-209)
    auto& vb = shape.vertex_buffer;
    append_bytes(vb, shape.stroke.line_vertices);
    shape.vertex_offsets[0] = vb.size();
    append_bytes(vb, shape.stroke.joint_vertices);
    shape.vertex_offsets[1] = vb.size();
    append_bytes(vb, shape.fill.solid_vertices);
    shape.vertex_offsets[2] = vb.size();
    append_bytes(vb, shape.fill.integral_quadratic_vertices);
    shape.vertex_offsets[3] = vb.size();
    append_bytes(vb, shape.fill.integral_cubic_vertices);
    shape.vertex_offsets[4] = vb.size();
    append_bytes(vb, shape.fill.rational_quadratic_vertices);
    shape.vertex_offsets[5] = vb.size();
    append_bytes(vb, shape.fill.rational_cubic_vertices);
    shape.vertex_offsets[6] = vb.size();
    append_bytes(vb, shape.convex_hull);
    shape.vertex_offsets[7] = vb.size();
    auto& ib = shape.index_buffer;
    append_bytes(ib, shape.stroke.line_indices);
    shape.index_offsets[0] = ib.size();
    append_bytes(ib, shape.stroke.joint_indices);
    shape.index_offsets[1] = ib.size();
    append_bytes(ib, shape.fill.solid_indices);
    shape.index_offsets[2] = ib.size();
    for (uint32_t i = 0; i < n_dynamic; ++i) { // renderer.rs:210-215
        crh_dynamic_stroke_descriptor d;
        const int rc = convert_dynamic_stroke_options(dynamic_stroke_options[i], d);
        if (rc != CRH_OK) {
            shape.status = rc;
            return;
        }
        shape.stroke_buffer.push_back(d);
    }
    shape.status = err.status;
}

} // namespace oracle
