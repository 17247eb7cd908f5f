// csrc/path.cpp — host-side path construction that needs real arithmetic: Path::push_elliptical_arc (path.rs:639-708), the SVG
// "arc to" command split into rational quadratic segments. One native implementation behind the C ABI so that the C++ and the Python
// mirror of the reference API produce the same floats. The reference computes with the un-vendored geometric_algebra crate (rotor
// sandwich products, complex powf / powi); this is the same construction written out on plain 2-vectors (the
// rotation convention is the one of csrc/text.cpp: rotate2d(a) turns counter-clockwise in y-up coordinates). DERIVED, not bit-pinned.
#include <stdint.h>

#include <cmath>

#include "../../include/contrast_hip.h"
#include "../../include/crh_fmath.h"

namespace {
struct V2 {
    float x, y;
};
struct Cx { // epga1d::ComplexNumber
    float re, im;
};
Cx mul(Cx a, Cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
Cx from_polar(float magnitude, float angle) {
    float s, c;
    crh_sincosf(angle, &s, &c);
    return {magnitude * c, magnitude * s};
}
Cx powf_unit(Cx z, float exponent) { // complex powf through polar form (|z| = 1 here up to rounding)
    const float magnitude = std::sqrt(z.re * z.re + z.im * z.im), argument = crh_atan2f(z.im, z.re);
    return from_polar(crh_powf(magnitude, exponent), argument * exponent);
}
Cx powi(Cx z, unsigned n) { // square and multiply
    Cx result = {1.0f, 0.0f};
    while (n) {
        if (n & 1u) result = mul(result, z);
        z = mul(z, z);
        n >>= 1;
    }
    return result;
}
Cx signum(Cx z) {
    const float inv = 1.0f / std::sqrt(z.re * z.re + z.im * z.im);
    return {z.re * inv, z.im * inv};
}
} // namespace

extern "C" crh_status crh_path_elliptical_arc(const float from[2], const float half_extent[2], float rotation, uint32_t large_arc, uint32_t sweep, const float to[2],
                                              float* records, uint32_t capacity, uint32_t* n_segments, uint32_t* is_line) {
    if (!from || !half_extent || !to || !n_segments || !is_line) return CRH_ERR_INVALID_ARGUMENT;
    const float inputs[7] = {from[0], from[1], half_extent[0], half_extent[1], rotation, to[0], to[1]};
    for (float v : inputs)
        if (!std::isfinite(v)) return CRH_ERR_NON_FINITE;
    *n_segments = 0;
    *is_line = 0;
    V2 radii = {std::fabs(half_extent[0]), std::fabs(half_extent[1])};
    if (radii.x == 0.0f || radii.y == 0.0f) { // path.rs:642-645: a plain line to `to`
        *is_line = 1;
        return CRH_OK;
    }
    float sn, cs;
    crh_sincosf(rotation, &sn, &cs);
    auto rotate = [&](V2 v) { return V2{v.x * cs - v.y * sn, v.x * sn + v.y * cs}; };
    auto rotate_back = [&](V2 v) { return V2{v.x * cs + v.y * sn, -v.x * sn + v.y * cs}; };
    const V2 vertex = rotate_back(V2{(to[0] - from[0]) * 0.5f, (to[1] - from[1]) * 0.5f});
    const V2 vertex_squared = {vertex.x * vertex.x, vertex.y * vertex.y};
    V2 radii_squared = {radii.x * radii.x, radii.y * radii.y};
    const float scale_factor_squared = vertex_squared.x / radii_squared.x + vertex_squared.y / radii_squared.y;
    if (scale_factor_squared > 1.0f) { // the radii cannot span from..to: scale them up
        const float scale = std::sqrt(scale_factor_squared);
        radii = {radii.x * scale, radii.y * scale};
        radii_squared = {radii.x * radii.x, radii.y * radii.y};
    }
    const V2 one_over_radii = {1.0f / radii.x, 1.0f / radii.y};
    const float radii_squared_vertex_squared = radii_squared.x * vertex_squared.y + radii_squared.y * vertex_squared.x;
    float offset = std::sqrt(std::fmax((radii_squared.x * radii_squared.y - radii_squared_vertex_squared) / radii_squared_vertex_squared, 0.0f));
    if ((large_arc != 0) == (sweep != 0)) offset = -offset;
    const V2 scaled = {vertex.x * one_over_radii.x, vertex.y * one_over_radii.y};
    const V2 center_offset = {radii.x * scaled.y * offset, radii.y * -scaled.x * offset}; // radii * rotate_90_degree_clockwise(vertex / radii) * offset
    const V2 turned = rotate(center_offset);
    const V2 center = {(to[0] + from[0]) * 0.5f + turned.x, (to[1] + from[1]) * 0.5f + turned.y};
    const V2 start_normal = {(-vertex.x - center_offset.x) * one_over_radii.x, (-vertex.y - center_offset.y) * one_over_radii.y};
    const V2 end_normal = {(vertex.x - center_offset.x) * one_over_radii.x, (vertex.y - center_offset.y) * one_over_radii.y};
    const Cx polar_start = signum(Cx{start_normal.x, start_normal.y}), polar_end = signum(Cx{end_normal.x, end_normal.y});
    Cx polar_range = mul(polar_end, Cx{polar_start.re, -polar_start.im}); // quotient of unit complex numbers
    {
        const float inv = 1.0f / (polar_start.re * polar_start.re + polar_start.im * polar_start.im);
        polar_range = {polar_range.re * inv, polar_range.im * inv};
    }
    float small_arc = crh_atan2f(polar_range.im, polar_range.re);
    if (small_arc < 0.0f) {
        polar_range.im = -polar_range.im; // reversal
        small_arc = -small_arc;
    }
    const float tau = crh_acosf(-1.0f) * 2.0f;
    float angle = small_arc;
    if (large_arc) angle -= tau;
    const float step_radians = crh_acosf(-1.0f) * 2.0f / 3.0f;
    const uint32_t steps = (uint32_t)std::ceil(std::fabs(angle) / step_radians);
    if ((large_arc != 0) != (sweep != 0)) angle = -angle;
    if (steps == 0) return CRH_OK; // from == to on the ellipse: nothing to draw
    if (steps > capacity || !records) {
        *n_segments = steps;
        return records ? CRH_ERR_INVALID_ARGUMENT : CRH_OK; // query
    }
    const Cx polar_step = powf_unit(polar_range, angle / (small_arc * (float)steps));
    const Cx half_polar_step_back = powf_unit(polar_step, -0.5f);
    float sin_unused, weight;
    crh_sincosf(std::fabs(angle) / (float)steps * 0.5f, &sin_unused, &weight);
    const V2 tangent_crossing_radii = {radii.x * (1.0f / weight), radii.y * (1.0f / weight)};
    for (uint32_t i = 1; i <= steps; ++i) {
        Cx interpolated = mul(polar_start, powi(polar_step, i));
        const V2 on_curve = rotate(V2{interpolated.re * radii.x, interpolated.im * radii.y});
        interpolated = mul(interpolated, half_polar_step_back);
        const V2 crossing = rotate(V2{interpolated.re * tangent_crossing_radii.x, interpolated.im * tangent_crossing_radii.y});
        float* r = records + 5 * (i - 1); // RationalQuadraticCurveSegment: weight, tangent crossing, vertex
        r[0] = weight;
        r[1] = center.x + crossing.x;
        r[2] = center.y + crossing.y;
        r[3] = center.x + on_curve.x;
        r[4] = center.y + on_curve.y;
        for (int k = 0; k < 5; ++k) {
            if (!std::isfinite(r[k])) return CRH_ERR_NON_FINITE;
            if (r[k] == 0.0f) r[k] = 0.0f; // SafeFloat: -0 -> +0
        }
    }
    *n_segments = steps;
    return CRH_OK;
}
