// contrast_renderer.hpp — header-only C++17 host mirror of the reference's public API for the hot path, over the C ABI of
// contrast_hip.h: the same type and function names, argument order, ownership and error behaviour as the Rust crate
// (Lichtso/contrast_renderer v0.1.4, citations relative to its tree), so that code written against
//     contrast_renderer::path::{Path, StrokeOptions, DynamicStrokeOptions, ...}          (src/path.rs)
//     contrast_renderer::renderer::{Configuration, Renderer, Shape, RenderOperation}     (src/renderer.rs)
//     contrast_renderer::text::{Font, Layout, paths_of_glyph, paths_of_text}             (src/text.rs)
// reads the same here. wgpu's Device / Queue / RenderPass become HIP-backed handles (a device ordinal, a Frame and a recorded
// RenderPass). Errors: Result<_, Error> becomes `throw Error` with the reference's variants (error.rs:5-16) in `Error::status`; the
// reference's panics (non-finite input, degenerate cubic) surface as status 6 / 7 instead of aborting the process.
//
// The Rust toolchain is absent from the build image, so this C++ mirror (and the ctypes one in contrast_renderer_amd/) is what the
// tests drive; INTEGRATION.md shows the equivalent Rust shim.
#pragma once
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "contrast_hip.h"

namespace contrast_renderer {

// ---------------------------------------------------------------------------------------------- error.rs
struct Error : std::runtime_error {
    crh_status status;
    explicit Error(crh_status s) : std::runtime_error(describe(s)), status(s) {}
    static std::string describe(crh_status s) {
        static const char* names[] = {"Ok", "NumberOfStencilBitsIsUnsupported", "ClipStackOverflow", "TooManyNestedOpacityGroups", "TooManyDashIntervals",
                                      "DynamicStrokeOptionsIndexOutOfBounds", "NonFinite (the reference panics: safe_float.rs:46)",
                                      "DegenerateCubic (the reference panics: fill.rs:174,178)", "Unsupported", "Hip", "InvalidArgument"};
        std::string text = s <= CRH_ERR_INVALID_ARGUMENT ? names[s] : "unknown";
        const char* detail = crh_last_error();
        if ((s == CRH_ERR_HIP || s == CRH_ERR_INVALID_ARGUMENT) && detail && *detail) text += std::string(": ") + detail;
        return text;
    }
};
inline void check(crh_status s) {
    if (s != CRH_OK) throw Error(s);
}

// ---------------------------------------------------------------------------------------------- safe_float.rs
// SafeFloat::from (safe_float.rs:44-52): finite or panic, -0.0 -> +0.0
inline float safe_float(float v) {
    if (!std::isfinite(v)) throw Error(CRH_ERR_NON_FINITE);
    return v == 0.0f ? 0.0f : v;
}
using Vec2 = std::pair<float, float>;

// ---------------------------------------------------------------------------------------------- path.rs
enum class SegmentType : uint8_t { Line = 0, IntegralQuadraticCurve = 1, IntegralCubicCurve = 2, RationalQuadraticCurve = 3, RationalCubicCurve = 4 }; // :56-67
enum class Join : uint32_t { Miter = 0, Bevel = 1, Round = 2 };                                                                                     // :71-82
enum class Cap : uint32_t { Square = 0, Round = 1, Out = 2, In = 3, Right = 4, Left = 5, Butt = 6 };                                               // :86-101

struct DashInterval { // path.rs:105-118
    float gap_start, gap_end;
    Cap dash_start, dash_end;
};

struct DynamicStrokeOptions { // path.rs:127-149: Dashed { join, pattern, phase } | Solid { join, start, end }
    bool dashed = false;
    Join join = Join::Miter;
    std::vector<DashInterval> pattern;
    float phase = 0.0f;
    Cap start = Cap::Butt, end = Cap::Butt;
    static DynamicStrokeOptions Dashed(Join join, std::vector<DashInterval> pattern, float phase) {
        DynamicStrokeOptions o;
        o.dashed = true;
        o.join = join;
        o.pattern = std::move(pattern);
        o.phase = safe_float(phase);
        return o;
    }
    static DynamicStrokeOptions Solid(Join join, Cap start, Cap end) {
        DynamicStrokeOptions o;
        o.join = join;
        o.start = start;
        o.end = end;
        return o;
    }
    crh_dynamic_stroke_options to_c() const {
        crh_dynamic_stroke_options c = {};
        c.dashed = dashed ? 1u : 0u;
        c.join = (uint32_t)join;
        c.pattern_len = (uint32_t)pattern.size(); // > 4 is reported by the library as TooManyDashIntervals, like renderer.rs:32-34
        for (size_t i = 0; i < pattern.size() && i < CRH_MAX_DASH_INTERVALS; ++i)
            c.pattern[i] = crh_dash_interval{safe_float(pattern[i].gap_start), safe_float(pattern[i].gap_end), (uint32_t)pattern[i].dash_start, (uint32_t)pattern[i].dash_end};
        c.phase = phase;
        c.start = (uint32_t)start;
        c.end = (uint32_t)end;
        return c;
    }
};

struct CurveApproximation { // path.rs:153-167
    bool uniform_tangent_angle = false;
    uint32_t steps = 1;
    float angle = 0.0f;
    static CurveApproximation UniformlySpacedParameters(uint32_t steps) { return CurveApproximation{false, steps, 0.0f}; }
    static CurveApproximation UniformTangentAngle(float angle) { return CurveApproximation{true, 0u, safe_float(angle)}; }
};

struct StrokeOptions { // path.rs:171-192
    float width = 1.0f, offset = 0.0f, miter_clip = 1.0f;
    bool closed = false;
    uint32_t dynamic_stroke_options_group = 0;
    CurveApproximation curve_approximation = CurveApproximation::UniformlySpacedParameters(1);
    void legalize() { // path.rs:196-200
        width = std::fabs(width);
        offset = std::fmin(std::fmax(offset, -0.5f), 0.5f);
        miter_clip = std::fabs(miter_clip);
    }
    crh_stroke_options to_c() const {
        crh_stroke_options c = {};
        c.width = safe_float(width);
        c.offset = safe_float(offset);
        c.miter_clip = safe_float(miter_clip);
        c.closed = closed ? 1u : 0u;
        c.dynamic_stroke_options_group = dynamic_stroke_options_group;
        c.curve_approximation = curve_approximation.uniform_tangent_angle ? CRH_CURVE_UNIFORM_TANGENT_ANGLE : CRH_CURVE_UNIFORMLY_SPACED_PARAMETERS;
        c.steps = curve_approximation.steps;
        c.angle_step = curve_approximation.angle;
        return c;
    }
};

// Path (path.rs:213-230). The five typed segment Vecs + segment_types of the reference are kept as one interleaved record list in
// segment order, which is what the SoA batch of the C ABI wants.
class Path {
  public:
    Vec2 start{0.0f, 0.0f};
    std::optional<StrokeOptions> stroke_options;
    std::vector<SegmentType> segment_types;
    std::vector<float> control; // per segment {2, 4, 6, 5, 10} floats, layouts of path.rs:15-52

    void push_line(Vec2 p) { // path.rs:234-237
        segment_types.push_back(SegmentType::Line);
        put(p);
    }
    void push_integral_quadratic_curve(Vec2 c0, Vec2 c1) { // path.rs:240-243
        segment_types.push_back(SegmentType::IntegralQuadraticCurve);
        put(c0), put(c1);
    }
    void push_integral_cubic_curve(Vec2 c0, Vec2 c1, Vec2 c2) { // path.rs:246-249
        segment_types.push_back(SegmentType::IntegralCubicCurve);
        put(c0), put(c1), put(c2);
    }
    void push_rational_quadratic_curve(float weight, Vec2 c0, Vec2 c1) { // path.rs:252-255
        segment_types.push_back(SegmentType::RationalQuadraticCurve);
        control.push_back(safe_float(weight));
        put(c0), put(c1);
    }
    void push_rational_cubic_curve(const float (&weights)[4], Vec2 c0, Vec2 c1, Vec2 c2) { // path.rs:258-261
        segment_types.push_back(SegmentType::RationalCubicCurve);
        for (float w : weights) control.push_back(safe_float(w));
        put(c0), put(c1), put(c2);
    }
    Vec2 get_end() const { return control.empty() ? start : Vec2{control[control.size() - 2], control[control.size() - 1]}; } // path.rs:266-290

    // ---- tangents (path.rs:296-372); Plane = (c, nx, ny)
    struct Plane {
        float c, nx, ny;
    };
    Plane get_start_tangent() const { // path.rs:296-322 — looks at segment_types.last() and the last segment of that type, as written there
        if (segment_types.empty()) return Plane{0.0f, 0.0f, 0.0f};
        const SegmentType last_type = segment_types.back();
        size_t at = 0, found = 0;
        for (size_t i = 0; i < segment_types.size(); ++i) {
            if (segment_types[i] == last_type) found = at;
            at += floats(segment_types[i]);
        }
        const size_t first_point = found + (last_type == SegmentType::RationalQuadraticCurve ? 1 : (last_type == SegmentType::RationalCubicCurve ? 4 : 0));
        return signum(tangent_from_points(start, Vec2{control[first_point], control[first_point + 1]}));
    }
    Plane get_end_tangent() const { // path.rs:326-372
        if (segment_types.empty()) return Plane{0.0f, 0.0f, 0.0f};
        const size_t n = control.size();
        if (segment_types.back() == SegmentType::Line) {
            const Vec2 previous = segment_types.size() > 1 ? Vec2{control[n - 4], control[n - 3]} : start;
            return signum(tangent_from_points(previous, Vec2{control[n - 2], control[n - 1]}));
        }
        return signum(tangent_from_points(Vec2{control[n - 4], control[n - 3]}, Vec2{control[n - 2], control[n - 1]}));
    }
    // path.rs:376-384 moves the typed segment Vecs of `other` but not its segment_types: nothing reachable is added (reproduced)
    void append(Path& other) {
        other.segment_types.clear();
        other.control.clear();
    }
    void reverse() { // path.rs:445-488
        std::vector<SegmentType> types;
        std::vector<std::vector<float>> records;
        Vec2 previous = start;
        size_t at = 0;
        for (SegmentType t : segment_types) {
            std::vector<float> rec(control.begin() + (long)at, control.begin() + (long)(at + floats(t)));
            at += floats(t);
            if (t == SegmentType::IntegralCubicCurve) {
                std::swap(rec[0], rec[2]);
                std::swap(rec[1], rec[3]);
            } else if (t == SegmentType::RationalCubicCurve) {
                std::swap(rec[0], rec[3]);
                std::swap(rec[1], rec[2]);
                std::swap(rec[4], rec[6]);
                std::swap(rec[5], rec[7]);
            }
            const Vec2 end{rec[rec.size() - 2], rec[rec.size() - 1]};
            rec[rec.size() - 2] = previous.first;
            rec[rec.size() - 1] = previous.second;
            previous = end;
            types.push_back(t);
            records.push_back(std::move(rec));
        }
        start = previous;
        segment_types.assign(types.rbegin(), types.rend());
        control.clear();
        for (auto it = records.rbegin(); it != records.rend(); ++it) control.insert(control.end(), it->begin(), it->end());
    }
    void convert_integral_curves_to_rational_curves() { // path.rs:492-534
        std::vector<float> out;
        size_t at = 0;
        for (SegmentType& t : segment_types) {
            const size_t n = floats(t);
            if (t == SegmentType::IntegralQuadraticCurve) {
                out.push_back(1.0f);
                t = SegmentType::RationalQuadraticCurve;
            } else if (t == SegmentType::IntegralCubicCurve) {
                out.insert(out.end(), 4, 1.0f);
                t = SegmentType::RationalCubicCurve;
            }
            out.insert(out.end(), control.begin() + (long)at, control.begin() + (long)(at + n));
            at += n;
        }
        control = std::move(out);
    }
    void convert_quadratic_curves_to_cubic_curves() { // path.rs:538-615
        std::vector<float> out;
        size_t at = 0;
        Vec2 previous = start;
        for (SegmentType& t : segment_types) {
            const size_t n = floats(t);
            const float* r = control.data() + at;
            at += n;
            if (t == SegmentType::IntegralQuadraticCurve) {
                const float c0x = previous.first + (r[0] - previous.first) * 2.0f / 3.0f, c0y = previous.second + (r[1] - previous.second) * 2.0f / 3.0f;
                const float c1x = r[2] + (r[0] - r[2]) * 2.0f / 3.0f, c1y = r[3] + (r[1] - r[3]) * 2.0f / 3.0f;
                for (float v : {c0x, c0y, c1x, c1y, r[2], r[3]}) out.push_back(safe_float(v));
                t = SegmentType::IntegralCubicCurve;
            } else if (t == SegmentType::RationalQuadraticCurve) {
                const float w = r[0];
                const float p0[3] = {1.0f, previous.first, previous.second}, p1[3] = {w, r[1] * w, r[2] * w}, p2[3] = {1.0f, r[3], r[4]};
                const float two_thirds = 2.0f / 3.0f;
                float n0[3], n1[3];
                for (int k = 0; k < 3; ++k) {
                    n0[k] = p0[k] + (p1[k] - p0[k]) * two_thirds;
                    n1[k] = p2[k] + (p1[k] - p2[k]) * two_thirds;
                }
                for (float v : {1.0f, n0[0], n1[0], 1.0f, n0[1] / n0[0], n0[2] / n0[0], n1[1] / n1[0], n1[2] / n1[0], r[3], r[4]}) out.push_back(safe_float(v));
                t = SegmentType::RationalCubicCurve;
            } else {
                out.insert(out.end(), r, r + n);
            }
            previous = Vec2{out[out.size() - 2], out[out.size() - 1]};
        }
        control = std::move(out);
    }
    void close() { // path.rs:621-628
        const Plane t = tangent_from_points(start, get_end());
        if (t.nx * t.nx + t.ny * t.ny <= 1e-4f) return;
        push_line(start);
    }
    void push_quarter_ellipse(Vec2 tangent_crossing, Vec2 to) { push_rational_quadratic_curve(0.70710678118654752440f, tangent_crossing, to); } // path.rs:631-636
    void push_elliptical_arc(Vec2 half_extent, float rotation, bool large_arc, bool sweep, Vec2 to) { // path.rs:639-708 (native: csrc/path.cpp)
        const Vec2 end = get_end();
        const float from[2] = {end.first, end.second}, half[2] = {half_extent.first, half_extent.second}, target[2] = {to.first, to.second};
        float records[20];
        uint32_t n = 0, is_line = 0;
        check(crh_path_elliptical_arc(from, half, rotation, large_arc, sweep, target, records, 4, &n, &is_line));
        if (is_line) {
            push_line(to);
            return;
        }
        for (uint32_t i = 0; i < n; ++i) push_rational_quadratic_curve(records[5 * i], {records[5 * i + 1], records[5 * i + 2]}, {records[5 * i + 3], records[5 * i + 4]});
    }

    static Path from_polygon(const std::vector<Vec2>& vertices) { // path.rs:711-724
        Path path;
        path.start = Vec2{safe_float(vertices.at(0).first), safe_float(vertices.at(0).second)};
        for (size_t i = 1; i < vertices.size(); ++i) path.push_line(vertices[i]);
        return path;
    }
    static Path from_regular_polygon(Vec2 center, float radius, float rotation, size_t vertex_count) { // path.rs:727-734
        std::vector<Vec2> vertices;
        for (size_t i = 0; i < vertex_count; ++i) {
            const float angle = rotation + (float)i / (float)vertex_count * 3.14159265358979323846f * 2.0f;
            vertices.push_back(Vec2{center.first + radius * std::cos(angle), center.second + radius * std::sin(angle)});
        }
        return from_polygon(vertices);
    }
    static Path from_rect(Vec2 center, Vec2 half_extent) { // path.rs:736-743
        const float cx = center.first, cy = center.second, hx = half_extent.first, hy = half_extent.second;
        return from_polygon({{cx - hx, cy - hy}, {cx - hx, cy + hy}, {cx + hx, cy + hy}, {cx + hx, cy - hy}});
    }

    static Path from_rounded_rect(Vec2 center, Vec2 half_extent, float radius) { // path.rs:746-780
        const float cx = center.first, cy = center.second, hx = half_extent.first, hy = half_extent.second, r = radius;
        const Vec2 v[4][3] = {{{cx - hx + r, cy - hy}, {cx - hx, cy - hy}, {cx - hx, cy - hy + r}},
                              {{cx - hx, cy + hy - r}, {cx - hx, cy + hy}, {cx - hx + r, cy + hy}},
                              {{cx + hx - r, cy + hy}, {cx + hx, cy + hy}, {cx + hx, cy + hy - r}},
                              {{cx + hx, cy - hy + r}, {cx + hx, cy - hy}, {cx + hx - r, cy - hy}}};
        Path path;
        path.start = Vec2{safe_float(v[3][2].first), safe_float(v[3][2].second)};
        for (const auto& corner : v) {
            path.push_line(corner[0]);
            path.push_quarter_ellipse(corner[1], corner[2]);
        }
        return path;
    }
    static Path from_ellipse(Vec2 center, Vec2 half_extent) { // path.rs:783-810
        const float cx = center.first, cy = center.second, hx = half_extent.first, hy = half_extent.second;
        const Vec2 v[4][2] = {{{cx - hx, cy - hy}, {cx - hx, cy}}, {{cx - hx, cy + hy}, {cx, cy + hy}}, {{cx + hx, cy + hy}, {cx + hx, cy}}, {{cx + hx, cy - hy}, {cx, cy - hy}}};
        Path path;
        path.start = Vec2{safe_float(v[3][1].first), safe_float(v[3][1].second)};
        for (const auto& corner : v) path.push_quarter_ellipse(corner[0], corner[1]);
        return path;
    }
    static Path from_circle(Vec2 center, float radius) { return from_ellipse(center, {radius, radius}); } // path.rs:813-815

  private:
    static size_t floats(SegmentType t) {
        static const size_t n[5] = {2, 4, 6, 5, 10};
        return n[(size_t)t];
    }
    static Plane tangent_from_points(Vec2 a, Vec2 b) { // path.rs:203-205: a v b
        return Plane{a.second * b.first - a.first * b.second, b.second - a.second, a.first - b.first};
    }
    static Plane signum(Plane p) {
        const float inv = 1.0f / std::sqrt(p.nx * p.nx + p.ny * p.ny);
        return Plane{p.c * inv, p.nx * inv, p.ny * inv};
    }
    void put(Vec2 p) {
        control.push_back(safe_float(p.first));
        control.push_back(safe_float(p.second));
    }
};

// Flattens the argument lists of Shape::from_paths (renderer.rs:177-183), one per Shape, into the SoA crh_path_batch.
class PathBatch {
  public:
    void add_shape(const std::vector<DynamicStrokeOptions>& dynamic_stroke_options, const std::vector<Path>& paths) {
        for (const Path& path : paths) {
            path_start_.push_back(path.start.first);
            path_start_.push_back(path.start.second);
            if (path.stroke_options) {
                path_stroke_.push_back((int32_t)stroke_options_.size());
                stroke_options_.push_back(path.stroke_options->to_c());
            } else {
                path_stroke_.push_back(-1);
            }
            for (SegmentType t : path.segment_types) types_.push_back((uint8_t)t);
            control_.insert(control_.end(), path.control.begin(), path.control.end());
            path_segment_begin_.push_back((uint32_t)types_.size());
        }
        shape_path_begin_.push_back((uint32_t)path_stroke_.size());
        for (const DynamicStrokeOptions& o : dynamic_stroke_options) dynamic_.push_back(o.to_c());
        shape_dynamic_begin_.push_back((uint32_t)dynamic_.size());
    }
    uint32_t n_shapes() const { return (uint32_t)shape_path_begin_.size() - 1u; }
    crh_path_batch view() const {
        crh_path_batch b = {};
        b.n_shapes = n_shapes();
        b.shape_path_begin = shape_path_begin_.data();
        b.n_paths = (uint32_t)path_stroke_.size();
        b.path_segment_begin = path_segment_begin_.data();
        b.path_start = path_start_.data();
        b.path_stroke_options = path_stroke_.data();
        b.n_segments = (uint32_t)types_.size();
        b.segment_types = types_.data();
        b.control_data = control_.data();
        b.n_control_floats = (uint32_t)control_.size();
        b.n_stroke_options = (uint32_t)stroke_options_.size();
        b.stroke_options = stroke_options_.data();
        b.shape_dynamic_begin = shape_dynamic_begin_.data();
        b.n_dynamic_stroke_options = (uint32_t)dynamic_.size();
        b.dynamic_stroke_options = dynamic_.data();
        return b;
    }

  private:
    std::vector<uint32_t> shape_path_begin_{0u}, path_segment_begin_{0u}, shape_dynamic_begin_{0u};
    std::vector<float> path_start_, control_;
    std::vector<int32_t> path_stroke_;
    std::vector<uint8_t> types_;
    std::vector<crh_stroke_options> stroke_options_;
    std::vector<crh_dynamic_stroke_options> dynamic_;
};

// ---------------------------------------------------------------------------------------------- renderer.rs
enum class RenderOperation : uint32_t { Stencil = 0, Clip = 1, UnClip = 2, Color = 3, SaveAlphaContext = 4, ScaleAlphaContext = 5, RestoreAlphaContext = 6 }; // :145-160

struct Configuration { // renderer.rs:380-405, the fields that change results on this path
    uint32_t msaa_sample_count = 1, clip_nesting_counter_bits = 4, winding_counter_bits = 4, alpha_layer_count = 0;
    // of the colour cover only, as in the reference (renderer.rs:743-745)
    crh_cull cull_mode = CRH_CULL_NONE;              // Option<wgpu::Face>
    crh_compare depth_compare = CRH_COMPARE_ALWAYS;  // wgpu::CompareFunction
    bool depth_write_enabled = false;
};

class Renderer { // renderer.rs:408-435
  public:
    Renderer(int device, const Configuration& config) { // Renderer::new(&device, config) -> Result<Renderer, Error>
        const crh_config c = {config.msaa_sample_count, config.clip_nesting_counter_bits, config.winding_counter_bits, config.alpha_layer_count,
                              (uint32_t)config.cull_mode,  (uint32_t)config.depth_compare,      config.depth_write_enabled ? 1u : 0u};
        check(crh_renderer_create(&c, device, &handle_));
        samples_ = config.msaa_sample_count;
    }
    ~Renderer() { crh_renderer_destroy(handle_); }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;
    Configuration get_config() const { // renderer.rs:887
        crh_config c;
        check(crh_renderer_get_config(handle_, &c));
        return Configuration{c.msaa_sample_count, c.clip_nesting_counter_bits, c.winding_counter_bits, c.alpha_layer_count,
                             (crh_cull)c.cull_mode, (crh_compare)c.depth_compare, c.depth_write_enabled != 0};
    }
    void synchronize() { check(crh_renderer_synchronize(handle_)); }
    crh_renderer* raw() const { return handle_; }
    uint32_t msaa_sample_count() const { return samples_; }

  private:
    crh_renderer* handle_ = nullptr;
    uint32_t samples_ = 1;
};

// The caller-owned colour + depth/stencil attachments of the render pass (examples/showcase/main.rs:217-230).
class Frame {
  public:
    // `format`: CRH_FORMAT_RGBA8 (f32 colours during a pass, one rounding at the end) or CRH_FORMAT_RGBA8_ATTACHMENT (every blend rounded to 8 bits, as
    // the wgpu Rgba8Unorm attachment of main.rs:205-215 would)
    Frame(Renderer& renderer, uint32_t width, uint32_t height, uint32_t format = CRH_FORMAT_RGBA8) : width_(width), height_(height), samples_(renderer.msaa_sample_count()) {
        check(crh_frame_create_format(renderer.raw(), width, height, format, &handle_));
    }
    ~Frame() { crh_frame_destroy(handle_); }
    Frame(const Frame&) = delete;
    Frame& operator=(const Frame&) = delete;
    void clear() { check(crh_frame_clear(handle_)); } // LoadOp::Clear(TRANSPARENT) + depth clear 1.0 + stencil clear
    void keep_pass_state() { check(crh_frame_keep_pass_state(handle_)); } // stencil, alpha layers and sample colours stay with the frame between passes, until clear()
    void set_tile_rows(uint32_t row_begin, uint32_t row_end) { check(crh_frame_set_tile_rows(handle_, row_begin, row_end)); } // the tile split of the multi-GPU path: draw these pixel rows only
    // the depth attachment (present when the Configuration tests or writes depth): LoadOp::Clear(value), the depth of the 3-D scene
    // the Shapes are decals in ([height][width], replicated to the samples), and read back of every sample
    void clear_depth(float value) { check(crh_frame_clear_depth(handle_, value)); }
    void upload_depth(const std::vector<float>& depth) {
        if (depth.size() != (size_t)width_ * height_) throw Error(CRH_ERR_INVALID_ARGUMENT);
        check(crh_frame_upload_depth(handle_, depth.data()));
    }
    std::vector<float> download_depth() {
        std::vector<float> out((size_t)width_ * height_ * samples_);
        check(crh_frame_download_depth(handle_, out.data()));
        return out;
    }
    std::vector<uint8_t> download() {                 // MSAA resolve + read back: premultiplied RGBA8, row 0 = top
        std::vector<uint8_t> out((size_t)width_ * height_ * 4);
        check(crh_frame_download(handle_, out.data()));
        return out;
    }
    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    crh_frame* raw() const { return handle_; }

  private:
    crh_frame* handle_ = nullptr;
    uint32_t width_, height_, samples_;
};

struct ShapeBuffers { // the byte image Shape::from_paths uploads (renderer.rs:198-209)
    uint64_t vertex_offsets[8], index_offsets[3];
    std::vector<uint8_t> vertex_bytes, index_bytes;
};

class RenderPass;

// A batch of Shapes built together (one launch tessellates all of them). Shape below is the n == 1 case with the reference's signature.
class Scene {
  public:
    Scene(Renderer& renderer, const PathBatch& batch, Scene* existing = nullptr) : n_shapes_(batch.n_shapes()) {
        const crh_path_batch view = batch.view();
        // `existing` is given up only once the upload has taken it over: a batch that fails validation leaves it with its owner
        check(crh_scene_upload(renderer.raw(), &view, existing ? existing->raw() : nullptr, &handle_));
        if (existing) existing->release();
        crh_status st = crh_scene_tessellate(handle_);
        if (st == CRH_OK) st = crh_scene_status(handle_); // surfaces the reference's panics at the call site, like the reference
        if (st != CRH_OK) { // a constructor that throws runs no destructor: free the handle here
            crh_scene_destroy(handle_);
            handle_ = nullptr;
            throw Error(st);
        }
    }
    ~Scene() {
        if (handle_) crh_scene_destroy(handle_);
    }
    Scene(Scene&& other) noexcept : handle_(other.release()), n_shapes_(other.n_shapes_) {}
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    uint32_t n_shapes() const { return n_shapes_; }
    ShapeBuffers buffers(uint32_t shape) const {
        ShapeBuffers b;
        check(crh_scene_shape_layout(handle_, shape, b.vertex_offsets, b.index_offsets));
        b.vertex_bytes.resize(b.vertex_offsets[7]);
        b.index_bytes.resize(b.index_offsets[2]);
        check(crh_scene_shape_download(handle_, shape, b.vertex_bytes.data(), b.index_bytes.data()));
        return b;
    }
    // Shape::set_dynamic_stroke_options (renderer.rs:360-376)
    void set_dynamic_stroke_options(uint32_t shape, size_t dynamic_stroke_options_index, const DynamicStrokeOptions& options) {
        const crh_dynamic_stroke_options c = options.to_c();
        check(crh_scene_set_dynamic_stroke_options(handle_, shape, (uint32_t)dynamic_stroke_options_index, &c));
    }
    // Stencil + Color of every Shape in index order, instance i = Shape i (the loop of examples/showcase/main.rs:236-250)
    void render(Frame& frame, const std::vector<float>& transforms, const std::vector<float>& colors) {
        if (transforms.size() != (size_t)n_shapes_ * 16 || colors.size() != (size_t)n_shapes_ * 4) throw Error(CRH_ERR_INVALID_ARGUMENT);
        check(crh_scene_render(handle_, frame.raw(), transforms.data(), colors.data()));
    }
    crh_scene* raw() const { return handle_; }
    crh_scene* release() {
        crh_scene* h = handle_;
        handle_ = nullptr;
        return h;
    }

  private:
    crh_scene* handle_ = nullptr;
    uint32_t n_shapes_ = 0;
};

// wgpu::RenderPass stand-in: records Shape::render calls with the pass state they see and submits them as one draw list.
class RenderPass {
  public:
    RenderPass(Renderer& renderer, Frame& frame) : config_(renderer.get_config()), frame_(frame) {}
    // instance data of the pass (the instance buffers bound at slot 0 / 2, renderer.rs:462-466): returns the instance index
    uint32_t push_instance(const float (&transform)[16], const float (&color)[4]) {
        transforms_.insert(transforms_.end(), transform, transform + 16);
        colors_.insert(colors_.end(), color, color + 4);
        return (uint32_t)(colors_.size() / 4 - 1);
    }
    // Renderer::set_clip_depth (renderer.rs:932-938)
    void set_clip_depth(size_t clip_depth) {
        if (clip_depth >= ((size_t)1 << config_.clip_nesting_counter_bits)) throw Error(CRH_ERR_CLIP_STACK_OVERFLOW);
        clip_depth_ = (uint32_t)clip_depth;
    }
    // Renderer::save_alpha_context / restore_alpha_context (renderer.rs:941-985) select the layer of the following alpha-context draws
    void set_alpha_layer(size_t alpha_layer) {
        if (alpha_layer >= config_.alpha_layer_count) throw Error(CRH_ERR_TOO_MANY_NESTED_OPACITY_GROUPS);
        alpha_layer_ = (uint32_t)alpha_layer;
    }
    // Shape::render(&renderer, &mut render_pass, instance_indices, render_operation) (renderer.rs:267-273) for Shape `shape` of `scene`
    // The Scenes / Shapes of a pass may be different objects: the frame keeps clip nesting counters, winding counters, saved alphas and sample
    // colours between the submissions, as the caller-owned stencil attachment and alpha layers of the reference do (renderer.rs:148-158, 257-266).
    void render(const Scene& scene, uint32_t shape, uint32_t first_instance, uint32_t end_instance, RenderOperation op) {
        for (uint32_t i = first_instance; i < end_instance; ++i) {
            draws_.push_back(crh_draw{shape, i, (uint32_t)op, clip_depth_, alpha_layer_});
            scenes_.push_back(&scene);
        }
    }
    // end of the pass: everything recorded executes in order, one crh_scene_render_draws per run of draws of the same object
    void submit() {
        for (const Scene* scene : scenes_)
            if (scene != scenes_.front()) {
                frame_.keep_pass_state(); // the pass spans objects: every sample's colour and stencil stay with the frame from its first draw on
                break;
            }
        for (size_t begin = 0; begin < draws_.size();) {
            size_t end = begin;
            while (end < draws_.size() && scenes_[end] == scenes_[begin]) ++end;
            check(crh_scene_render_draws(scenes_[begin]->raw(), frame_.raw(), transforms_.data(), colors_.data(), (uint32_t)(colors_.size() / 4), draws_.data() + begin, (uint32_t)(end - begin)));
            begin = end;
        }
        draws_.clear();
        scenes_.clear();
    }

  private:
    Configuration config_;
    Frame& frame_;
    std::vector<float> transforms_, colors_;
    std::vector<crh_draw> draws_;
    std::vector<const Scene*> scenes_; // the object every draw belongs to
    uint32_t clip_depth_ = 0, alpha_layer_ = 0;
};

// Shape (renderer.rs:163-171): `Shape::from_paths(&device, &renderer, &dynamic_stroke_options, &paths, existing_shape)` (renderer.rs:177-183)
class Shape : public Scene {
  public:
    static Shape from_paths(Renderer& renderer, const std::vector<DynamicStrokeOptions>& dynamic_stroke_options, const std::vector<Path>& paths,
                            Shape* existing_shape = nullptr) {
        PathBatch batch;
        batch.add_shape(dynamic_stroke_options, paths);
        return Shape(renderer, batch, existing_shape);
    }
    ShapeBuffers buffers() const { return Scene::buffers(0); }
    void set_dynamic_stroke_options(size_t dynamic_stroke_options_index, const DynamicStrokeOptions& options) { // renderer.rs:360-376
        Scene::set_dynamic_stroke_options(0, dynamic_stroke_options_index, options);
    }
    // Shape::render for one pass: see RenderPass::render(scene, 0, instances..., op)
    void render(RenderPass& pass, uint32_t first_instance, uint32_t end_instance, RenderOperation op) const { pass.render(*this, 0, first_instance, end_instance, op); }

  private:
    Shape(Renderer& renderer, const PathBatch& batch, Shape* existing) : Scene(renderer, batch, existing) {}
};

// ---------------------------------------------------------------------------------------------- text.rs
enum class Orientation : uint32_t { RightToLeft = 0, LeftToRight = 1, TopToBottom = 2, BottomToTop = 3 }; // :106-117
enum class Alignment : uint32_t { Begin = 0, Baseline = 1, Center = 2, End = 3 };                       // :119-131
struct Layout {                                                                                         // :133-143
    float size;
    Orientation orientation = Orientation::LeftToRight;
    Alignment major_alignment = Alignment::Begin, minor_alignment = Alignment::Baseline;
};

class Font { // text.rs:11-38; face() of the reference returns the parsed ttf_parser::Face — here the Font is the face
  public:
    Font(std::string name, const std::vector<uint8_t>& font_data) : name_(std::move(name)) { check(crh_font_create(font_data.data(), font_data.size(), &handle_)); }
    ~Font() { crh_font_destroy(handle_); }
    Font(const Font&) = delete;
    Font& operator=(const Font&) = delete;
    const std::string& name() const { return name_; }
    const Font& face() const { return *this; }
    crh_font_metrics metrics() const {
        crh_font_metrics m;
        check(crh_font_get_metrics(handle_, &m));
        return m;
    }
    std::optional<uint16_t> glyph_index(char32_t c) const {
        uint16_t glyph;
        uint32_t found;
        check(crh_font_glyph_index(handle_, (uint32_t)c, &glyph, &found));
        return found ? std::optional<uint16_t>(glyph) : std::nullopt;
    }
    crh_font* raw() const { return handle_; }

  private:
    std::string name_;
    crh_font* handle_ = nullptr;
};

// ---- utils.rs:168-203: column-major 4x4 matrices as the instance buffer holds them (shaders.wgsl:13-27), element [4 * column + row]
using Mat4 = std::array<float, 16>;
inline Mat4 perspective_projection(float field_of_view_y, float aspect_ratio, float near, float far) { // utils.rs:181-192
    const float height = 1.0f / std::tan(field_of_view_y * 0.5f), denominator = 1.0f / (near - far);
    Mat4 m{};
    m[0] = height / aspect_ratio;
    m[5] = height;
    m[10] = -far * denominator;
    m[11] = 1.0f;
    m[14] = near * far * denominator;
    return m;
}
inline Mat4 matrix_multiplication(const Mat4& a, const Mat4& b) { // utils.rs:194-203: a * b, accumulated left to right like the reference
    Mat4 out{};
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float acc = a[r] * b[4 * c];
            for (int k = 1; k < 4; ++k) acc = acc + a[4 * k + r] * b[4 * c + k];
            out[4 * c + r] = acc;
        }
    return out;
}
inline Mat4 translation_matrix(float x, float y, float z) { // what motor3d_to_mat4 (utils.rs:168-179) yields for a pure translator
    Mat4 m{};
    m[0] = m[5] = m[10] = m[15] = 1.0f;
    m[12] = x, m[13] = y, m[14] = z;
    return m;
}

namespace detail {
inline std::vector<Path> take_paths(crh_path_list* list) {
    crh_path_batch view;
    const crh_status st = crh_path_list_view(list, &view);
    std::vector<Path> out;
    if (st == CRH_OK) {
        static const int floats[5] = {2, 4, 6, 5, 10};
        size_t at = 0;
        for (uint32_t p = 0; p < view.n_paths; ++p) {
            Path path;
            path.start = Vec2{view.path_start[2 * p], view.path_start[2 * p + 1]};
            for (uint32_t s = view.path_segment_begin[p]; s < view.path_segment_begin[p + 1]; ++s) {
                path.segment_types.push_back((SegmentType)view.segment_types[s]);
                path.control.insert(path.control.end(), view.control_data + at, view.control_data + at + floats[view.segment_types[s]]);
                at += floats[view.segment_types[s]];
            }
            out.push_back(std::move(path));
        }
    }
    crh_path_list_destroy(list);
    check(st);
    return out;
}
} // namespace detail

inline std::vector<Path> paths_of_glyph(const Font& face, uint16_t glyph_id) { // text.rs:97-104
    crh_path_list* list = nullptr;
    check(crh_paths_of_glyph(face.raw(), glyph_id, &list));
    return detail::take_paths(list);
}
// text.rs:236-263; clipping_area = clockwise convex polygon of (x, y) pairs, or empty
inline std::vector<Path> paths_of_text(const Font& face, const Layout& layout, const std::u32string& text, const std::vector<Vec2>& clipping_area = {}) {
    const crh_text_layout c = {layout.size, (uint32_t)layout.orientation, (uint32_t)layout.major_alignment, (uint32_t)layout.minor_alignment};
    std::vector<float> clip;
    for (const Vec2& p : clipping_area) clip.push_back(p.first), clip.push_back(p.second);
    crh_path_list* list = nullptr;
    check(crh_paths_of_text(face.raw(), &c, reinterpret_cast<const uint32_t*>(text.data()), text.size(), clip.empty() ? nullptr : clip.data(), clip.size() / 2, &list));
    return detail::take_paths(list);
}

// text.rs:266-347: bounding box of the entire text and the glyph positions of its characters, line by line
struct TextGeometry {
    size_t major_axis;                                           // 0 horizontal, 1 vertical
    Vec2 half_extent;
    std::vector<std::pair<size_t, std::vector<Vec2>>> lines;     // (line_range_end, glyph positions incl. the line break)

    static TextGeometry make(const Font& face, const Layout& layout, const std::u32string& text) { // TextGeometry::new, text.rs:284-307
        const crh_text_layout c = {layout.size, (uint32_t)layout.orientation, (uint32_t)layout.major_alignment, (uint32_t)layout.minor_alignment};
        const uint32_t* chars = reinterpret_cast<const uint32_t*>(text.data());
        uint64_t n_lines = 0;
        check(crh_text_aligned_positions(face.raw(), &c, chars, text.size(), nullptr, nullptr, nullptr, nullptr, nullptr, &n_lines));
        int64_t extent[2], offset[2];
        std::vector<int64_t> positions((text.size() + 1) * 3);
        std::vector<uint64_t> ends(n_lines), lengths(n_lines);
        check(crh_text_aligned_positions(face.raw(), &c, chars, text.size(), extent, offset, positions.data(), ends.data(), lengths.data(), &n_lines));
        TextGeometry g;
        g.major_axis = (layout.orientation == Orientation::RightToLeft || layout.orientation == Orientation::LeftToRight) ? 0 : 1;
        const float scale = layout.size / (float)face.metrics().height;
        g.half_extent = {(float)extent[0] * scale * 0.5f, (float)extent[1] * scale * 0.5f};
        size_t at = 0;
        for (uint64_t l = 0; l < n_lines; ++l) {
            std::vector<Vec2> line;
            for (uint64_t i = 0; i < lengths[l]; ++i, ++at)
                line.push_back({(float)(positions[at * 3] - offset[0]) * scale, (float)(positions[at * 3 + 1] - offset[1]) * scale});
            g.lines.emplace_back((size_t)ends[l], std::move(line));
        }
        return g;
    }
    size_t line_index_from_char_index(size_t char_index) const { // text.rs:310-315 (the reference unwraps: out of range throws here)
        for (size_t i = 0; i < lines.size(); ++i)
            if (lines[i].first > char_index) return i;
        throw Error(CRH_ERR_INVALID_ARGUMENT);
    }
    size_t char_index_from_position(Vec2 cursor) const { // text.rs:318-331
        const auto axis = [](const Vec2& v, size_t a) { return a == 0 ? v.first : v.second; };
        const float minor_half_extent = axis(half_extent, 1 - major_axis);
        float v = (minor_half_extent - axis(cursor, 1 - major_axis)) * (float)lines.size() / (minor_half_extent * 2.0f);
        v = std::fmin(std::fmax(v, 0.0f), (float)(lines.size() - 1)); // f32::max / min return the other operand for NaN
        const size_t line_index = (size_t)v;
        const std::vector<Vec2>& glyph_positions = lines[line_index].second;
        size_t found = glyph_positions.size() - 1;
        for (size_t i = 0; i + 1 < glyph_positions.size(); ++i)
            if ((axis(glyph_positions[i], major_axis) + axis(glyph_positions[i + 1], major_axis)) * 0.5f > axis(cursor, major_axis)) {
                found = i;
                break;
            }
        return found + (line_index == 0 ? 0 : lines[line_index - 1].first);
    }
    size_t advance_char_index_by_line_index(size_t char_index, ptrdiff_t relative_line_index) const { // text.rs:334-346
        const size_t line_index = line_index_from_char_index(char_index);
        if (relative_line_index < 0 && line_index == 0) return 0;
        if (relative_line_index > 0 && line_index == lines.size() - 1) return lines.back().first - 1;
        const std::vector<Vec2>& glyph_positions = lines[line_index].second;
        Vec2 cursor = glyph_positions[char_index + glyph_positions.size() - lines[line_index].first];
        const float line_minor_extent = (major_axis == 0 ? half_extent.second : half_extent.first) * 2.0f / (float)lines.size();
        (major_axis == 0 ? cursor.second : cursor.first) -= line_minor_extent * (float)relative_line_index;
        return char_index_from_position(cursor);
    }
};
inline size_t byte_offset_of_char_index(const std::string& utf8, size_t char_index) { // text.rs:350-352
    size_t chars = 0;
    for (size_t i = 0; i < utf8.size(); ++i)
        if (((unsigned char)utf8[i] & 0xC0u) != 0x80u && chars++ == char_index) return i;
    return utf8.size();
}

} // namespace contrast_renderer
