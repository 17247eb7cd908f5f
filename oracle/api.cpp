// oracle/api.cpp — TEST INFRASTRUCTURE (CPU oracle): C entry points for ctypes (tests/, smoke(),
// bench.py's cpu_baseline leg). Never linked into or called by the product library.
#include <atomic>
#include <chrono>
#include <thread>

#include "raster.hpp"

using namespace oracle;

namespace {
struct Scene {
    std::vector<Shape> shapes;
};

void build_range(const crh_path_batch* b, const std::vector<uint64_t>& seg_off, Scene& scene, uint32_t begin, uint32_t end) {
    for (uint32_t s = begin; s < end; ++s) {
        std::vector<PathView> views;
        for (uint32_t p = b->shape_path_begin[s]; p < b->shape_path_begin[s + 1]; ++p) {
            PathView v;
            v.start = b->path_start + 2 * (size_t)p;
            v.types = b->segment_types + b->path_segment_begin[p];
            v.n_segments = b->path_segment_begin[p + 1] - b->path_segment_begin[p];
            v.control = b->control_data + seg_off[b->path_segment_begin[p]];
            const int32_t so = b->path_stroke_options[p];
            v.stroke = so < 0 ? nullptr : b->stroke_options + so;
            views.push_back(v);
        }
        const uint32_t d0 = b->shape_dynamic_begin ? b->shape_dynamic_begin[s] : 0;
        const uint32_t d1 = b->shape_dynamic_begin ? b->shape_dynamic_begin[s + 1] : 0;
        shape_from_paths(scene.shapes[s], b->dynamic_stroke_options + d0, d1 - d0, views.data(), (uint32_t)views.size());
    }
}

Scene* tessellate(const crh_path_batch* b, int n_threads) {
    Scene* scene = new Scene;
    scene->shapes.resize(b->n_shapes);
    std::vector<uint64_t> seg_off(b->n_segments + 1, 0);
    for (uint32_t i = 0; i < b->n_segments; ++i) seg_off[i + 1] = seg_off[i] + SEGMENT_FLOATS[b->segment_types[i]];
    if (n_threads <= 1) {
        build_range(b, seg_off, *scene, 0, b->n_shapes);
    } else {
        std::atomic<uint32_t> next{0};
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t)
            pool.emplace_back([&] {
                for (;;) {
                    const uint32_t begin = next.fetch_add(256);
                    if (begin >= b->n_shapes) break;
                    build_range(b, seg_off, *scene, begin, std::min(begin + 256, b->n_shapes));
                }
            });
        for (auto& th : pool) th.join();
    }
    return scene;
}
} // namespace

extern "C" {

void* oracle_tessellate(const crh_path_batch* batch, int n_threads) { return tessellate(batch, n_threads); }
void oracle_free(void* h) { delete static_cast<Scene*>(h); }
uint32_t oracle_n_shapes(void* h) { return (uint32_t) static_cast<Scene*>(h)->shapes.size(); }
int oracle_shape_status(void* h, uint32_t s) { return static_cast<Scene*>(h)->shapes[s].status; }
int oracle_status(void* h) {
    for (auto& s : static_cast<Scene*>(h)->shapes)
        if (s.status != CRH_OK) return s.status;
    return CRH_OK;
}
void oracle_shape_layout(void* h, uint32_t s, uint64_t vo[8], uint64_t io[3]) {
    const Shape& sh = static_cast<Scene*>(h)->shapes[s];
    std::memcpy(vo, sh.vertex_offsets, sizeof(sh.vertex_offsets));
    std::memcpy(io, sh.index_offsets, sizeof(sh.index_offsets));
}
void oracle_shape_download(void* h, uint32_t s, void* vb, void* ib) {
    const Shape& sh = static_cast<Scene*>(h)->shapes[s];
    if (vb && !sh.vertex_buffer.empty()) std::memcpy(vb, sh.vertex_buffer.data(), sh.vertex_buffer.size());
    if (ib && !sh.index_buffer.empty()) std::memcpy(ib, sh.index_buffer.data(), sh.index_buffer.size());
}
void oracle_layout_all(void* h, uint64_t* layout, uint64_t* total_v, uint64_t* total_i) {
    Scene* sc = static_cast<Scene*>(h);
    uint64_t tv = 0, ti = 0;
    for (size_t s = 0; s < sc->shapes.size(); ++s) {
        const Shape& sh = sc->shapes[s];
        for (int k = 0; k < 8; ++k) layout[s * 11 + k] = sh.vertex_offsets[k];
        for (int k = 0; k < 3; ++k) layout[s * 11 + 8 + k] = sh.index_offsets[k];
        tv += sh.vertex_buffer.size();
        ti += sh.index_buffer.size();
    }
    *total_v = tv;
    *total_i = ti;
}
void oracle_download_all(void* h, void* vb, void* ib) {
    Scene* sc = static_cast<Scene*>(h);
    uint8_t* v = static_cast<uint8_t*>(vb);
    uint8_t* i = static_cast<uint8_t*>(ib);
    for (const Shape& sh : sc->shapes) {
        if (!sh.vertex_buffer.empty()) std::memcpy(v, sh.vertex_buffer.data(), sh.vertex_buffer.size());
        if (!sh.index_buffer.empty()) std::memcpy(i, sh.index_buffer.data(), sh.index_buffer.size());
        v += sh.vertex_buffer.size();
        i += sh.index_buffer.size();
    }
}
uint32_t oracle_shape_descriptors(void* h, uint32_t s, crh_dynamic_stroke_descriptor* out, uint32_t capacity) {
    const Shape& sh = static_cast<Scene*>(h)->shapes[s];
    for (uint32_t k = 0; k < sh.stroke_buffer.size() && k < capacity; ++k) out[k] = sh.stroke_buffer[k];
    return (uint32_t)sh.stroke_buffer.size();
}
int oracle_convert_dynamic_stroke_options(const crh_dynamic_stroke_options* o, crh_dynamic_stroke_descriptor* out) {
    return convert_dynamic_stroke_options(*o, *out);
}

// The loop of examples/showcase/main.rs:236-250 over shapes [shape_begin, shape_end) into a cleared frame.
// frames created after this call behave as Rgba8Unorm attachments (per-blend rounding) or not: see oracle/raster.hpp Frame::attachment8
void oracle_set_attachment8(int on) { Frame::attachment8_default() = on != 0; }

int oracle_render(void* h, uint32_t width, uint32_t height, uint32_t msaa, uint32_t winding_bits, const float* transforms, const float* colors,
                  uint32_t shape_begin, uint32_t shape_end, uint8_t* rgba8) {
    Scene* sc = static_cast<Scene*>(h);
    if (!(msaa == 1 || msaa == 4) || winding_bits == 0 || winding_bits > 8) return CRH_ERR_INVALID_ARGUMENT;
    Frame f;
    f.create(width, height, msaa, winding_bits);
    for (uint32_t s = shape_begin; s < shape_end && s < sc->shapes.size(); ++s) {
        render_stencil(f, sc->shapes[s], transforms + 16 * (size_t)s);
        render_color(f, sc->shapes[s], transforms + 16 * (size_t)s, colors + 4 * (size_t)s);
    }
    resolve_rgba8(f, rgba8);
    return CRH_OK;
}

// A recorded render pass: draws[i] = {shape, instance, op (crh_render_op), clip_depth, alpha_layer}, executed in order into a cleared frame.
// `depth_state` (optional) = {cull_mode, depth_compare, depth_write_enabled}; `depth` (optional, with depth_state) = the depth attachment
// [height][width][msaa] the pass starts from and, on return, what it left there.
// `load_rgba8` (optional): LoadOp::Load — the pass starts from this resolved image (every sample of a pixel = its RGBA8 value / 255).
int oracle_render_pass_over(void* h, uint32_t width, uint32_t height, uint32_t msaa, uint32_t winding_bits, uint32_t clip_bits, uint32_t alpha_layers,
                            const uint32_t* depth_state, float* depth, const float* transforms, const float* colors, const uint32_t* draws, uint32_t n_draws,
                            const uint8_t* load_rgba8, uint8_t* rgba8);
int oracle_render_pass(void* h, uint32_t width, uint32_t height, uint32_t msaa, uint32_t winding_bits, uint32_t clip_bits, uint32_t alpha_layers,
                       const uint32_t* depth_state, float* depth, const float* transforms, const float* colors, const uint32_t* draws, uint32_t n_draws,
                       uint8_t* rgba8) {
    return oracle_render_pass_over(h, width, height, msaa, winding_bits, clip_bits, alpha_layers, depth_state, depth, transforms, colors, draws, n_draws, nullptr, rgba8);
}
int oracle_render_pass_over(void* h, uint32_t width, uint32_t height, uint32_t msaa, uint32_t winding_bits, uint32_t clip_bits, uint32_t alpha_layers,
                            const uint32_t* depth_state, float* depth, const float* transforms, const float* colors, const uint32_t* draws, uint32_t n_draws,
                            const uint8_t* load_rgba8, uint8_t* rgba8) {
    Scene* sc = static_cast<Scene*>(h);
    if (!(msaa == 1 || msaa == 4) || winding_bits == 0 || winding_bits + clip_bits > 8) return CRH_ERR_INVALID_ARGUMENT;
    Frame f;
    f.create(width, height, msaa, winding_bits, clip_bits, alpha_layers);
    if (load_rgba8)
        for (size_t p = 0; p < (size_t)width * height; ++p)
            for (uint32_t s = 0; s < msaa; ++s)
                for (int c = 0; c < 4; ++c) f.color[(p * msaa + s) * 4 + c] = (float)load_rgba8[p * 4 + c] * (1.0f / 255.0f);
    if (depth_state) {
        f.cull_mode = depth_state[0];
        f.depth_compare = depth_state[1];
        f.depth_write = depth_state[2];
        if (depth) f.depth.assign(depth, depth + (size_t)width * height * msaa);
    }
    for (uint32_t i = 0; i < n_draws; ++i) {
        const uint32_t shape = draws[5 * i], instance = draws[5 * i + 1], op = draws[5 * i + 2], clip_depth = draws[5 * i + 3], layer = draws[5 * i + 4];
        if (shape >= sc->shapes.size()) return CRH_ERR_INVALID_ARGUMENT;
        if (clip_depth >= (1u << clip_bits)) return CRH_ERR_CLIP_STACK_OVERFLOW; // renderer.rs:933-935
        if (op >= CRH_OP_SAVE_ALPHA_CONTEXT && layer >= alpha_layers) return CRH_ERR_TOO_MANY_NESTED_OPACITY_GROUPS; // renderer.rs:947,980
        f.reference = clip_depth << winding_bits;
        if (op == CRH_OP_STENCIL)
            render_stencil(f, sc->shapes[shape], transforms + 16 * (size_t)instance);
        else
            render_cover(f, sc->shapes[shape], transforms + 16 * (size_t)instance, colors + 4 * (size_t)instance, op, layer);
    }
    resolve_rgba8(f, rgba8);
    if (depth && !f.depth.empty()) std::copy(f.depth.begin(), f.depth.end(), depth);
    return CRH_OK;
}
int oracle_render_draws(void* h, uint32_t width, uint32_t height, uint32_t msaa, uint32_t winding_bits, uint32_t clip_bits, uint32_t alpha_layers,
                        const float* transforms, const float* colors, const uint32_t* draws, uint32_t n_draws, uint8_t* rgba8) {
    return oracle_render_pass(h, width, height, msaa, winding_bits, clip_bits, alpha_layers, nullptr, nullptr, transforms, colors, draws, n_draws, rgba8);
}

// cpu_baseline: wall seconds of `repeats` full tessellations (restatement of the CPU part of from_paths). The threads are created once,
// before the clock starts, and take blocks of 64 Shapes from one counter over all repeats (glibc malloc serves every thread from its own
// arena); the result scenes are destroyed after the clock stops. n_threads <= 1: the calling thread alone, like renderer.rs:187.
double oracle_time_tessellate(const crh_path_batch* batch, int n_threads, int repeats) {
    std::vector<uint64_t> seg_off(batch->n_segments + 1, 0);
    for (uint32_t i = 0; i < batch->n_segments; ++i) seg_off[i + 1] = seg_off[i] + SEGMENT_FLOATS[batch->segment_types[i]];
    std::vector<Scene*> scenes((size_t)repeats);
    for (Scene*& sc : scenes) {
        sc = new Scene;
        sc->shapes.resize(batch->n_shapes);
    }
    const uint64_t blocks_per_repeat = ((uint64_t)batch->n_shapes + 63) / 64, total_blocks = blocks_per_repeat * (uint64_t)repeats;
    std::atomic<uint64_t> next{0};
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    auto work = [&] {
        ready.fetch_add(1);
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        for (;;) {
            const uint64_t block = next.fetch_add(1);
            if (block >= total_blocks) break;
            const uint64_t r = block / blocks_per_repeat, first = (block % blocks_per_repeat) * 64;
            build_range(batch, seg_off, *scenes[r], (uint32_t)first, (uint32_t)std::min<uint64_t>(first + 64, batch->n_shapes));
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(work);
    while (ready.load() < (int)pool.size()) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    work(); // the calling thread is one of the workers
    for (auto& th : pool) th.join();
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (Scene* sc : scenes) delete sc;
    return seconds;
}

// the fragment helpers of shaders.wgsl:165-231 on their own, for the known-answer tests
int oracle_cap(float x, float y, uint32_t cap_type) { return cap(x, y, cap_type) ? 1 : 0; }
int oracle_stroke_dashed(const crh_dynamic_stroke_descriptor* d, float tx, float ty) { return stroke_dashed(*d, tx, ty) ? 1 : 0; }

// elementary functions, for the GPU bit-identity test: fn 0 atan2(a,b) 1 acos(a) 2 sin(a) 3 cos(a) 4 pow(a,b) 5 wgsl_mod(a,b)
void oracle_fmath_eval(int fn, const float* a, const float* b, float* out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        float s, c;
        switch (fn) {
            case 0: out[i] = crh_atan2f(a[i], b[i]); break;
            case 1: out[i] = crh_acosf(a[i]); break;
            case 2: crh_sincosf(a[i], &s, &c); out[i] = s; break;
            case 3: crh_sincosf(a[i], &s, &c); out[i] = c; break;
            case 4: out[i] = crh_powf(a[i], b[i]); break;
            default: out[i] = crh_wgsl_mod(a[i], b[i]); break;
        }
    }
}

// polynomial solvers, for unit tests: degree 1..4, ascending coefficients; returns root count, writes (re, im, den) triples
int oracle_solve(int degree, const float* coefficients, float* roots_out, float* discriminant_out) {
    Roots r;
    r.n = 0;
    float d = 0.0f;
    int real_root = 0;
    switch (degree) {
        case 1: d = solve_linear(coefficients, ERROR_MARGIN, r); break;
        case 2: d = solve_quadratic(coefficients, ERROR_MARGIN, r); break;
        case 3: d = solve_cubic(coefficients, ERROR_MARGIN, r, real_root); break;
        default: d = solve_quartic(coefficients, ERROR_MARGIN, r); break;
    }
    for (int k = 0; k < r.n; ++k) {
        roots_out[3 * k] = r.r[k].num_re;
        roots_out[3 * k + 1] = r.r[k].num_im;
        roots_out[3 * k + 2] = r.r[k].den;
    }
    *discriminant_out = d;
    return r.n;
}
}
