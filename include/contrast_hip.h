/* contrast_hip.h — C ABI of the MI355X-native tessellate + raster hot path of contrast_renderer.
 *
 * Every entry point names the reference interface it replaces (file:line relative to the
 * reference tree, contrast_renderer v0.1.4). The reference has no FFI; its seam for this path is
 * `Shape::from_paths` / `Shape::render` (renderer.rs:177, :267) over the private modules
 * fill / stroke / vertex (lib.rs:12,16,20). A Rust shim that keeps those signatures binds exactly
 * the functions below (see INTEGRATION.md for the `extern "C"` block).
 *
 * Conventions: plain pointers and sizes, host memory unless a name ends in `_dev`; all functions
 * return crh_status; handles are thread-compatible (one HIP stream per renderer), not thread-safe.
 * Batch first: the unit of work is a *scene* = many Shapes built and rendered together, because one
 * kernel launch per Shape would be launch-bound on a 256-CU part. A scene with n_shapes == 1 is the
 * reference's single `Shape`.
 */
#ifndef CONTRAST_HIP_H
#define CONTRAST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: 1..5 are error.rs:5-16 in declaration order ---------------------------------- */
typedef enum crh_status {
    CRH_OK = 0,
    CRH_ERR_NUMBER_OF_STENCIL_BITS_IS_UNSUPPORTED = 1,    /* renderer.rs:433-435 */
    CRH_ERR_CLIP_STACK_OVERFLOW = 2,                      /* renderer.rs:933-935 */
    CRH_ERR_TOO_MANY_NESTED_OPACITY_GROUPS = 3,           /* renderer.rs:947-949, :980-982 */
    CRH_ERR_TOO_MANY_DASH_INTERVALS = 4,                  /* renderer.rs:32-34 */
    CRH_ERR_DYNAMIC_STROKE_OPTIONS_INDEX_OUT_OF_BOUNDS = 5, /* renderer.rs:189-191, :366-368 */
    CRH_ERR_NON_FINITE = 6,        /* the reference panics: safe_float.rs:46,114 */
    CRH_ERR_DEGENERATE_CUBIC = 7,  /* the reference panics: fill.rs:174,178 */
    CRH_ERR_UNSUPPORTED = 8,       /* documented limit of this implementation */
    CRH_ERR_HIP = 9,               /* a HIP runtime call failed; crh_last_error() has the text */
    CRH_ERR_INVALID_ARGUMENT = 10
} crh_status;

/* ---- the Path data model (path.rs:15-230) -------------------------------------------------------- */

/* SegmentType discriminants, path.rs:56-67 */
enum {
    CRH_SEGMENT_LINE = 0,               /* control data: x y                          (path.rs:15-18)  */
    CRH_SEGMENT_INTEGRAL_QUADRATIC = 1, /* control data: c0x c0y c1x c1y              (path.rs:22-25)  */
    CRH_SEGMENT_INTEGRAL_CUBIC = 2,     /* control data: c0x c0y c1x c1y c2x c2y      (path.rs:29-32)  */
    CRH_SEGMENT_RATIONAL_QUADRATIC = 3, /* control data: weight c0x c0y c1x c1y       (path.rs:36-43)  */
    CRH_SEGMENT_RATIONAL_CUBIC = 4      /* control data: w0 w1 w2 w3 c0x c0y .. c2y   (path.rs:47-52)  */
};
/* Join, path.rs:71-82 ; Cap, path.rs:86-101 */
enum { CRH_JOIN_MITER = 0, CRH_JOIN_BEVEL = 1, CRH_JOIN_ROUND = 2 };
enum { CRH_CAP_SQUARE = 0, CRH_CAP_ROUND = 1, CRH_CAP_OUT = 2, CRH_CAP_IN = 3, CRH_CAP_RIGHT = 4, CRH_CAP_LEFT = 5, CRH_CAP_BUTT = 6 };
/* CurveApproximation, path.rs:153-167 */
enum { CRH_CURVE_UNIFORMLY_SPACED_PARAMETERS = 0, CRH_CURVE_UNIFORM_TANGENT_ANGLE = 1 };
#define CRH_MAX_DASH_INTERVALS 4 /* path.rs:121 */

/* StrokeOptions, path.rs:171-192 (already legalize()d by the caller, path.rs:196-200) */
typedef struct crh_stroke_options {
    float width;
    float offset;
    float miter_clip;
    uint32_t closed;
    uint32_t dynamic_stroke_options_group;
    uint32_t curve_approximation; /* CRH_CURVE_* */
    uint32_t steps;               /* UniformlySpacedParameters(steps) */
    float angle_step;             /* UniformTangentAngle(angle_step) */
} crh_stroke_options;

/* DashInterval, path.rs:105-118 */
typedef struct crh_dash_interval {
    float gap_start;
    float gap_end;
    uint32_t dash_start; /* CRH_CAP_* */
    uint32_t dash_end;   /* CRH_CAP_* */
} crh_dash_interval;

/* DynamicStrokeOptions, path.rs:127-149 */
typedef struct crh_dynamic_stroke_options {
    uint32_t dashed;      /* 1 = Dashed{join, pattern, phase}, 0 = Solid{join, start, end} */
    uint32_t join;        /* CRH_JOIN_* */
    uint32_t pattern_len; /* > CRH_MAX_DASH_INTERVALS -> CRH_ERR_TOO_MANY_DASH_INTERVALS; 0 is invalid as in the reference */
    crh_dash_interval pattern[CRH_MAX_DASH_INTERVALS];
    float phase;
    uint32_t start; /* CRH_CAP_* (Solid) */
    uint32_t end;   /* CRH_CAP_* (Solid) */
} crh_dynamic_stroke_options;

/* DynamicStrokeDescriptor, renderer.rs:20-27: the 48-byte record the raster reads */
typedef struct crh_dynamic_stroke_descriptor {
    float gap_start[CRH_MAX_DASH_INTERVALS];
    float gap_end[CRH_MAX_DASH_INTERVALS];
    uint32_t caps;
    uint32_t count_dashed_join;
    float phase;
    uint32_t _padding;
} crh_dynamic_stroke_descriptor;

/* A batch of Shapes, each a contiguous run of Paths (the argument `paths: &[Path]` of
 * renderer.rs:181, for many shapes at once). Struct-of-arrays flattening of path.rs:213-230:
 * `control_data` holds the segments' floats in path order, one record per segment with the layout
 * given at CRH_SEGMENT_* above; `path_start` is Path::start. All floats must be finite; -0.0 is
 * canonicalised to +0.0 on upload (SafeFloat::from, safe_float.rs:44-52). */
typedef struct crh_path_batch {
    uint32_t n_shapes;
    const uint32_t* shape_path_begin; /* [n_shapes + 1] */
    uint32_t n_paths;
    const uint32_t* path_segment_begin;  /* [n_paths + 1] into segment_types */
    const float* path_start;             /* [n_paths][2] */
    const int32_t* path_stroke_options;  /* [n_paths]: index into stroke_options, or -1 = filled (Path::stroke_options == None) */
    uint32_t n_segments;
    const uint8_t* segment_types; /* [n_segments] CRH_SEGMENT_* */
    const float* control_data;    /* sum over segments of {2,4,6,5,10}[type] floats */
    uint32_t n_control_floats;
    uint32_t n_stroke_options;
    const crh_stroke_options* stroke_options;
    const uint32_t* shape_dynamic_begin; /* [n_shapes + 1] into dynamic_stroke_options (the per-shape `&[DynamicStrokeOptions]`, renderer.rs:180) */
    uint32_t n_dynamic_stroke_options;
    const crh_dynamic_stroke_options* dynamic_stroke_options;
} crh_path_batch;

/* ---- Renderer (renderer.rs:380-435) --------------------------------------------------------------- */

/* Configuration, renderer.rs:380-405. The fields that change results on this path:
 * blending is fixed to premultiplied "over" (One, OneMinusSrcAlpha — examples/showcase/main.rs:32-43), the depth attachment is
 * f32 per sample (depth_stencil_format is a wgpu detail), color_attachment_in_stencil_pass has no meaning in a compute rasterizer.
 * The three depth / cull fields act on the colour cover only, as in the reference (renderer.rs:743-745: every other pipeline is
 * built with cull None, CompareFunction::Always and no depth write). Zero-initialised = no culling, Always, no write. */
typedef enum crh_cull { CRH_CULL_NONE = 0, CRH_CULL_FRONT = 1, CRH_CULL_BACK = 2 } crh_cull; /* Option<wgpu::Face>; front = counter-clockwise on screen (renderer.rs:477) */
typedef enum crh_compare { /* wgpu::CompareFunction of `fragment depth  OP  stored depth` */
    CRH_COMPARE_ALWAYS = 0,
    CRH_COMPARE_NEVER = 1,
    CRH_COMPARE_LESS = 2,
    CRH_COMPARE_EQUAL = 3,
    CRH_COMPARE_LESS_EQUAL = 4,
    CRH_COMPARE_GREATER = 5,
    CRH_COMPARE_NOT_EQUAL = 6,
    CRH_COMPARE_GREATER_EQUAL = 7
} crh_compare;
typedef struct crh_config {
    uint32_t msaa_sample_count;         /* 1 or 4 */
    uint32_t clip_nesting_counter_bits; /* validated as in renderer.rs:433 */
    uint32_t winding_counter_bits;      /* >= 1, sum <= 8 */
    uint32_t alpha_layer_count;         /* <= 4; layers of the alpha-context operations (renderer.rs:403-404) */
    uint32_t cull_mode;                 /* crh_cull, renderer.rs:383-384 */
    uint32_t depth_compare;             /* crh_compare, renderer.rs:387-388 */
    uint32_t depth_write_enabled;       /* 0 / 1, renderer.rs:389-390 */
} crh_config;

typedef struct crh_renderer crh_renderer; /* Renderer, renderer.rs:408 */
typedef struct crh_scene crh_scene;       /* n Shapes (renderer.rs:163-171) built together */
typedef struct crh_frame crh_frame;       /* the caller-owned colour + stencil attachments of the render pass */

/* RenderOperation, renderer.rs:145-160, same order */
typedef enum crh_render_op {
    CRH_OP_STENCIL = 0,
    CRH_OP_CLIP = 1,
    CRH_OP_UNCLIP = 2,
    CRH_OP_COLOR = 3,
    CRH_OP_SAVE_ALPHA_CONTEXT = 4,
    CRH_OP_SCALE_ALPHA_CONTEXT = 5,
    CRH_OP_RESTORE_ALPHA_CONTEXT = 6
} crh_render_op;

/* Renderer::new, renderer.rs:432-435. device_ordinal = HIP device (cuda:N). */
crh_status crh_renderer_create(const crh_config* config, int device_ordinal, crh_renderer** out);
void crh_renderer_destroy(crh_renderer* renderer);
/* Renderer::get_config, renderer.rs:887 */
crh_status crh_renderer_get_config(const crh_renderer* renderer, crh_config* out);

/* convert_dynamic_stroke_options, renderer.rs:29-60 (host-side, pure) */
crh_status crh_convert_dynamic_stroke_options(const crh_dynamic_stroke_options* options, crh_dynamic_stroke_descriptor* out);

/* ---- Shape::from_paths (renderer.rs:177-249) ------------------------------------------------------ */

/* Host -> HBM: validates (finite, stroke-group bounds renderer.rs:188-191, dash count renderer.rs:32-34),
 * canonicalises -0, lays the batch out for the kernels and copies it to the device. This is the part of
 * from_paths that is not arithmetic; it is outside bench.py's timed region. `existing` may be NULL; when
 * given, its device allocations are reused if large enough (Buffer::update, renderer.rs:89-95). Paths of the
 * STRUCTURE `existing` holds (as many Shapes, paths and segments, stroked or not — an animation of control
 * points) also keep the capacities of its vertex streams: the next crh_scene_tessellate does not wait for
 * the totals of the new paths; a tessellation that does not fit after all is noticed when a frame drawn from
 * it is settled (crh_frame_synchronize, crh_frame_download, ...) or by crh_scene_status, sized and repeated,
 * the frame drawn again. The host arrays of `batch` are copied before the call returns. */
crh_status crh_scene_upload(crh_renderer* renderer, const crh_path_batch* batch, crh_scene* existing, crh_scene** out);
/* The arithmetic of from_paths for every shape of the scene, on the GPU:
 * StrokeBuilder::add_path (stroke.rs:205-465), FillBuilder::add_path (fill.rs:263-367),
 * convex_hull::andrew + triangle_fan_to_strip (convex_hull.rs:7-40, vertex.rs:28-35) and the
 * concat_buffers! offsets (renderer.rs:198-209). Asynchronous on the renderer's stream. */
crh_status crh_scene_tessellate(crh_scene* scene);
/* Waits for the stream and returns the first per-path error raised by the kernels (codes 6, 7, 8). */
crh_status crh_scene_status(crh_scene* scene);
void crh_scene_destroy(crh_scene* scene);

/* The reference's single-shape entry point: upload + tessellate of a batch with n_shapes == 1. */
crh_status crh_shape_from_paths(crh_renderer* renderer, const crh_path_batch* one_shape, crh_scene* existing, crh_scene** out);

/* Parity tap: exactly the byte image renderer.rs:198-209 uploads for shape `shape_index`.
 * vertex_offsets[8] / index_offsets[3] are the cumulative byte END offsets of
 * [line, joint, solid, integral_quadratic, integral_cubic, rational_quadratic, rational_cubic, hull]
 * and [line_indices, joint_indices, solid_indices]. Call with NULL byte pointers to query sizes. */
crh_status crh_scene_shape_layout(crh_scene* scene, uint32_t shape_index, uint64_t vertex_offsets[8], uint64_t index_offsets[3]);
crh_status crh_scene_shape_download(crh_scene* scene, uint32_t shape_index, void* vertex_bytes, void* index_bytes);
/* All shapes at once: layout[shape][11] u64 (8 vertex + 3 index END offsets); sizes in bytes of the
 * concatenation over shapes are returned so the caller can allocate, then download. */
crh_status crh_scene_layout_all(crh_scene* scene, uint64_t* layout /* [n_shapes][11] */, uint64_t* total_vertex_bytes, uint64_t* total_index_bytes);
crh_status crh_scene_download_all(crh_scene* scene, void* vertex_bytes, void* index_bytes);
/* Bytes the tessellation kernels read and wrote (the algorithmic traffic of SURVEY.md §8(d)). */
crh_status crh_scene_traffic(crh_scene* scene, uint64_t* bytes_read, uint64_t* bytes_written);

/* Shape::set_dynamic_stroke_options, renderer.rs:360-376 */
crh_status crh_scene_set_dynamic_stroke_options(crh_scene* scene, uint32_t shape_index, uint32_t group_index, const crh_dynamic_stroke_options* options);

/* ---- render pass (renderer.rs:267-355 + shaders.wgsl) --------------------------------------------- */

/* The colour target (RGBA8 unorm, premultiplied) plus the per-sample winding ("stencil") state. */
crh_status crh_frame_create(crh_renderer* renderer, uint32_t width, uint32_t height, crh_frame** out);
/* The same with the storage format of the resolved image named. CRH_FORMAT_RGBA8 is the reference's target (main.rs:205-215 renders to
 * the surface format, 8 bits per channel). CRH_FORMAT_RGBA16F keeps four binary16 values per pixel — the per-rank LAYERS of the multi-GPU
 * exchange (layers exchanged as RGBA16F keep the composite within 1/255 of a single-GPU render, RGBA8 layers within 2/255). */
enum { CRH_FORMAT_RGBA8 = 0, CRH_FORMAT_RGBA16F = 1, CRH_FORMAT_RGBA8_ATTACHMENT = 2 };
/* CRH_FORMAT_RGBA8_ATTACHMENT: RGBA8 storage like CRH_FORMAT_RGBA8, and the frame behaves like the wgpu::TextureFormat::Rgba8Unorm colour
 * ATTACHMENT the reference blends into (renderer.rs:736-754, examples/showcase/main.rs:32-43,205-215): every colour write of a cover — the
 * premultiplied "over" of Color, the alpha writes of Scale / RestoreAlphaContext — is rounded to 8 bits per channel where it happens, as a
 * hardware blender reads and writes the texture. CRH_FORMAT_RGBA8 keeps f32 colours for the whole pass and rounds once at the end: up to a
 * few 1/255 closer to the exact composite where many translucent Shapes overlap. Same download / upload entry points as CRH_FORMAT_RGBA8. */
crh_status crh_frame_create_format(crh_renderer* renderer, uint32_t width, uint32_t height, uint32_t format, crh_frame** out);
crh_status crh_frame_format(const crh_frame* frame, uint32_t* format);
void crh_frame_destroy(crh_frame* frame);
/* LoadOp::Clear(TRANSPARENT) + depth clear 1.0 + stencil clear 0 (examples/showcase/main.rs:217-230) */
crh_status crh_frame_clear(crh_frame* frame);
/* The stencil attachment and the alpha layers of the reference are caller-owned textures: what one Shape::render call leaves in them
 * — an open Clip, the winding of a Stencil without its cover, a saved alpha context — is seen by the next call, whatever Shape (object) it
 * belongs to (renderer.rs:148-158, 257-266, 932-985). A frame keeps that state across crh_scene_render* calls, in HBM: the stencil byte
 * (clip nesting counter << winding bits | winding counter, renderer.rs:565-566, 936), the saved alphas and the f32 colour of every sample.
 * It starts to do so by itself at the first recorded pass that ENDS with state left over; this call starts it now — a pass that will
 * span several Shape / Scene objects calls it in front of its first crh_scene_render_draws, so that colours drawn before the state
 * appears are kept per sample and unrounded too (CRH_FORMAT_RGBA8 rounds once, at the resolve — as within one crh_scene_render_draws).
 * In force until crh_frame_clear. Passes into such a frame are drawn by the general (triangle) formulation. */
crh_status crh_frame_keep_pass_state(crh_frame* frame);
/* The depth attachment (f32 per sample, [height][width][msaa_sample_count]); it exists when the renderer's configuration tests or
 * writes depth. clear_depth = LoadOp::Clear(value) (main.rs:223-226); upload_depth places the depth of a 3-D scene the Shapes are
 * decals in (README.md:8-12): `depth` = [height][width] host floats, replicated to every sample; download_depth copies all samples out. */
crh_status crh_frame_clear_depth(crh_frame* frame, float value);
crh_status crh_frame_upload_depth(crh_frame* frame, const float* depth);
crh_status crh_frame_download_depth(crh_frame* frame, float* depth_samples);

/* For every shape i of the scene, in index order: Shape::render(Stencil) then Shape::render(Color)
 * with instance transform `transforms[i]` (column-major mat4, 64 B: shaders.wgsl:13-27) and colour
 * `colors[i]` (straight RGBA, 16 B: shaders.wgsl:304-309) — the loop of examples/showcase/main.rs:236-250.
 * Any 4x4 matrix is accepted: perspective instances (utils.rs:181-203, main.rs:162-202) are rasterised with homogeneous edge functions,
 * a per-sample near / far test and perspective-correct attributes.
 * Pointers are host memory; they are copied to the device when they change. Asynchronous. */
crh_status crh_scene_render(crh_scene* scene, crh_frame* frame, const float* transforms, const float* colors);
/* Same, with per-shape data already in HBM (used by bench.py so that PCIe is outside the step). */
crh_status crh_scene_set_instances(crh_scene* scene, const float* transforms, const float* colors);
crh_status crh_scene_render_resident(crh_scene* scene, crh_frame* frame);

/* A recorded render pass. One crh_draw = one call of Shape::render(renderer, pass, instance_indices, op) (renderer.rs:267-273) together
 * with the pass state it sees: the stencil reference set by Renderer::set_clip_depth (renderer.rs:932-938) and, for the alpha-context
 * operations, the layer bound by Renderer::save_alpha_context / restore_alpha_context (renderer.rs:941-985). `instance` indexes
 * `transforms` / `colors` (the instance buffers, shaders.wgsl:13-27): a Shape may be drawn any number of times.
 *   Stencil                     the seven stencil pipelines (renderer.rs:275-337)
 *   Clip / UnClip               increment / decrement_clip_nesting_counter pipelines (renderer.rs:692-729)
 *   Color                       color_cover (renderer.rs:736-754, shaders.wgsl:304-309)
 *   Save / Scale / Restore      the alpha-context covers (renderer.rs:761-861, shaders.wgsl:311-355); the saved layers live with the frame
 * Draws execute in order. Errors as the reference: clip_depth >= 2^clip_nesting_counter_bits -> CRH_ERR_CLIP_STACK_OVERFLOW,
 * alpha_layer >= alpha_layer_count -> CRH_ERR_TOO_MANY_NESTED_OPACITY_GROUPS. At most 4 alpha layers are supported. */
typedef struct crh_draw {
    uint32_t shape;       /* index of the Shape in the scene */
    uint32_t instance;    /* index into transforms / colors */
    uint32_t op;          /* crh_render_op */
    uint32_t clip_depth;  /* Renderer::set_clip_depth value in effect */
    uint32_t alpha_layer; /* alpha_layer of save_alpha_context / restore_alpha_context in effect (alpha-context operations only) */
} crh_draw;
crh_status crh_scene_render_draws(crh_scene* scene, crh_frame* frame, const float* transforms, const float* colors, uint32_t n_instances,
                                  const crh_draw* draws, uint32_t n_draws);

/* MSAA resolve (box average, examples/showcase/main.rs:215) + copy to host, `rgba8` = width*height*4 bytes, row 0 = top. */
crh_status crh_frame_download(crh_frame* frame, void* rgba8);
/* The same for a CRH_FORMAT_RGBA16F frame: width*height*8 bytes (four IEEE binary16 per pixel). Each entry point refuses the other format. */
crh_status crh_frame_download_f16(crh_frame* frame, void* rgba16f);
/* Device pointer of the resolved RGBA8 image (for the RCCL tile exchange); valid until the frame is destroyed. */
crh_status crh_frame_device_pointer(crh_frame* frame, void** rgba8_dev);
/* Ordered premultiplied "over" of n_layers RGBA8 images that live in HBM: dst = layers[0] under layers[1] ....
 * Runs on a stream of its own and returns when dst is complete, without waiting for renders in flight. */
crh_status crh_composite_over(crh_renderer* renderer, const void* const* layers_dev, uint32_t n_layers, uint64_t n_pixels, void* dst_dev);

/* ---- stream plumbing ------------------------------------------------------------------------------- */
crh_status crh_renderer_synchronize(crh_renderer* renderer);
/* Blocks the host until the last render INTO THIS FRAME has finished; work queued afterwards (the next frame of a double-buffered
 * loop) keeps running. The resolved image behind crh_frame_device_pointer is then complete (a frame whose tile lists turned out too
 * small is rendered again here, which waits for everything in flight — once, while the capacities are being learned). */
crh_status crh_frame_synchronize(crh_frame* frame);
/* hipStream_t of the renderer, as void* (for HIP events in bench.py). */
void* crh_renderer_stream(crh_renderer* renderer);
/* Milliseconds spent by the last crh_scene_tessellate / crh_scene_render* on the GPU, per kernel,
 * measured with HIP events on the renderer's streams when timing is enabled. enabled = 1: every kernel of a step (a dozen events per
 * step: they cost a pipelined loop about 4 % of its rate); 2: the kernels of the raster lane only (two events per step); 0: off. */
crh_status crh_renderer_enable_timing(crh_renderer* renderer, int enabled);
typedef struct crh_kernel_time {
    char name[48];
    float ms;
    uint64_t algorithmic_bytes;
} crh_kernel_time;
crh_status crh_renderer_kernel_times(crh_renderer* renderer, crh_kernel_time* out, uint32_t capacity, uint32_t* count);
/* Self-test tap: evaluates include/crh_fmath.h ON THE GPU (fn 0 atan2(a,b), 1 acos(a), 2 sin(a), 3 cos(a), 4 pow(a,b), 5 wgsl_mod(a,b))
 * so that tests can check device results bit for bit against the host evaluation of the same header. Host pointers. */
crh_status crh_selftest_fmath(crh_renderer* renderer, int fn, const float* a, const float* b, float* out, uint64_t n);
/* ---- glyph producer: text.rs (config 3 input) ----------------------------------------------------
 * Host-side code (the reference's is host-side too). The TrueType reading is done by the crate
 * ttf-parser 0.14.0 in the reference (Cargo.toml:20, not vendored); src/csrc/text.cpp restates the
 * published TrueType `glyf` outline walk and the tables text.rs queries. */
typedef struct crh_font crh_font;           /* Font, text.rs:11-38 (ttf_parser::Face over owned bytes) */
typedef struct crh_path_list crh_path_list; /* Vec<Path>, as returned by text.rs:97 and :236 */

/* Font::new, text.rs:19-27. The bytes are copied. CRH_ERR_INVALID_ARGUMENT when the face cannot be parsed (the reference unwrap()s). */
crh_status crh_font_create(const void* ttf_bytes, size_t n_bytes, crh_font** out);
void crh_font_destroy(crh_font* font);

/* The Face getters text.rs calls (text.rs:156-158, :209-211, :238); font units. */
typedef struct crh_font_metrics {
    uint32_t units_per_em;
    uint32_t number_of_glyphs;
    int32_t ascender, descender, line_gap, height; /* height = ascender - descender (Face::height) */
    int32_t has_x_height, x_height;                /* Face::x_height() -> Option */
    int32_t has_vertical_metrics, vertical_height, vertical_line_gap; /* Face::vertical_height() / vertical_line_gap() -> Option */
    int32_t has_kerning;                           /* first subtable of `kern` usable (text.rs:148) */
} crh_font_metrics;
crh_status crh_font_get_metrics(const crh_font* font, crh_font_metrics* out);
/* Face::glyph_index (text.rs:147,181): *found = 0 when no Unicode cmap subtable maps the code point. */
crh_status crh_font_glyph_index(const crh_font* font, uint32_t code_point, uint16_t* glyph_id, uint32_t* found);
/* Face::glyph_hor_advance / glyph_ver_advance (text.rs:189-191) */
crh_status crh_font_glyph_advance(const crh_font* font, uint16_t glyph_id, uint32_t vertical, uint16_t* advance, uint32_t* found);
/* Face::glyph_bounding_box (text.rs:244): x_min y_min x_max y_max */
crh_status crh_font_glyph_bounding_box(const crh_font* font, uint16_t glyph_id, int16_t box[4], uint32_t* found);
/* kerning_table.glyphs_kerning(left, right) (text.rs:183) */
crh_status crh_font_glyphs_kerning(const crh_font* font, uint16_t left, uint16_t right, int16_t* kerning, uint32_t* found);

/* Orientation text.rs:106-117, Alignment :119-131 (declaration order) */
enum { CRH_ORIENTATION_RIGHT_TO_LEFT = 0, CRH_ORIENTATION_LEFT_TO_RIGHT = 1, CRH_ORIENTATION_TOP_TO_BOTTOM = 2, CRH_ORIENTATION_BOTTOM_TO_TOP = 3 };
enum { CRH_ALIGNMENT_BEGIN = 0, CRH_ALIGNMENT_BASELINE = 1, CRH_ALIGNMENT_CENTER = 2, CRH_ALIGNMENT_END = 3 };
/* Layout, text.rs:133-143 */
typedef struct crh_text_layout {
    float size;
    uint32_t orientation;     /* CRH_ORIENTATION_* */
    uint32_t major_alignment; /* CRH_ALIGNMENT_* */
    uint32_t minor_alignment; /* CRH_ALIGNMENT_* */
} crh_text_layout;

/* paths_of_glyph, text.rs:97-104: one Path per contour, no stroke options; an empty list for glyphs without outline. */
crh_status crh_paths_of_glyph(const crh_font* font, uint16_t glyph_id, crh_path_list** out);
/* paths_of_text, text.rs:236-263. `text` = Unicode scalar values (Rust chars); `clipping_area` = n_clip (x, y) pairs of a convex
 * polygon in clockwise order (utils.rs:83-98) or NULL. */
crh_status crh_paths_of_text(const crh_font* font, const crh_text_layout* layout, const uint32_t* text, size_t n_chars, const float* clipping_area,
                             size_t n_clip, crh_path_list** out);
/* calculate_aligned_positions!, text.rs:145-230 (integer font units): `positions` receives, line by line, one (x, y, glyph_id) triple of
 * int64 per character plus one terminating entry per line (the '\n' or the end of the text, glyph id 0), i.e. n_chars + 1 triples in
 * total; line_ends[l] = the reference's `line_range_end` (text.rs:169,199). Pass NULL pointers to query *n_lines only. */
crh_status crh_text_aligned_positions(const crh_font* font, const crh_text_layout* layout, const uint32_t* text, size_t n_chars, int64_t extent[2],
                                      int64_t offset[2], int64_t* positions /* [n_chars + 1][3] */, uint64_t* line_ends, uint64_t* line_lengths,
                                      uint64_t* n_lines);
/* Path::push_elliptical_arc, path.rs:639-708 (the SVG "arc to" command): the rational quadratic segments that continue a path whose
 * current end point is `from`. `records` receives n_segments x {weight, tangent_crossing.xy, vertex.xy} (the record layout of
 * CRH_SEGMENT_RATIONAL_QUADRATIC); *is_line = 1 when a radius is zero and the reference pushes a plain line to `to` instead.
 * Call with records == NULL to query *n_segments (at most 3). Host code. */
crh_status crh_path_elliptical_arc(const float from[2], const float half_extent[2], float rotation, uint32_t large_arc, uint32_t sweep, const float to[2],
                                   float* records, uint32_t capacity, uint32_t* n_segments, uint32_t* is_line);
/* Path::transform(scale, &motor), path.rs:387-439, on every path of the list. motor = ppga2d::Motor [scalar, e12, e01, e02]
 * (utils.rs:122-129: rotate2d, translate2d). */
crh_status crh_path_list_transform(crh_path_list* list, float scale, const float motor[4]);
/* Views the list as one Shape, all paths filled, in crh_path_batch form; the pointers stay valid until the list is changed or destroyed. */
crh_status crh_path_list_view(const crh_path_list* list, crh_path_batch* out);
void crh_path_list_destroy(crh_path_list* list);

/* ---- multi-GPU: path-index sharding + the framebuffer exchange ---------------------------------------
 * The reference is single-GPU (no such component upstream); this group is the exchange step of SURVEY.md §8(e) behind
 * the C ABI, so that a host in any language can shard: one process per GPU, rank g renders Shapes crh_comm_shard(n, g, world) into a
 * private full-size layer (a crh_frame), crh_frame_exchange composites the layers in rank order — premultiplied "over", lower rank
 * underneath — and leaves the image in rank 0's `result` frame. Only 16x16 tiles that hold something travel (occupancy bitmaps are
 * all-gathered first); transfers are grouped ncclSend / ncclRecv of row slabs over RCCL (the librccl the process has already mapped —
 * e.g. the one bundled with torch — or librccl.so, opened on first use).
 * Layers may be RGBA8 frames (<= 2/255 per channel against a single-GPU render of the whole scene) or CRH_FORMAT_RGBA16F frames
 * (<= 1/255: one RGBA8 quantisation, in the composite); all ranks use the same format, the result frame is RGBA8.
 * A rank whose layer cannot be read (a failed pass) still takes part in the collective and every rank returns an error together. */
typedef struct crh_comm crh_comm;
#define CRH_COMM_ID_BYTES 128 /* ncclUniqueId */
/* contiguous, order-preserving split of [0, n_items): sizes differ by at most one */
crh_status crh_comm_shard(uint32_t n_items, uint32_t rank, uint32_t world, uint32_t* begin, uint32_t* end);
/* the pixel rows of rank `rank`'s slab of a frame `height` pixels high (whole 16-pixel tile rows) */
crh_status crh_comm_slab_rows(uint32_t height, uint32_t rank, uint32_t world, uint32_t* row_begin, uint32_t* row_end);
/* rank 0 calls this and hands the 128 bytes to the other ranks by any means (ncclGetUniqueId) */
crh_status crh_comm_unique_id(void* id128);
/* collective over all ranks (ncclCommInitRank); the renderer names the device. CRH_ERR_UNSUPPORTED when RCCL cannot be loaded. */
crh_status crh_comm_create(crh_renderer* renderer, uint32_t rank, uint32_t world, const void* id128, crh_comm** out);
void crh_comm_destroy(crh_comm* comm);
/* collective: `layer` = this rank's frame; `result` = the frame that receives the image on rank 0, NULL on every other rank.
 * Waits for the last pass into `layer` only; runs on a stream of its own, so the renderer may already be drawing the next step. */
crh_status crh_frame_exchange(crh_comm* comm, crh_frame* layer, crh_frame* result);
/* The other split of SURVEY.md §8(e) — shard by TILE instead of by path index (no such component upstream either; it mirrors what a wgpu
 * scissor rectangle on renderer.rs:267-355's passes would do): the passes into `frame` draw the tile rows that cover the pixel rows
 * [row_begin, row_end) only (multiples of 16, or the frame's height; crh_comm_slab_rows gives rank g's), the rest of the frame is and stays
 * transparent. Every rank uploads, tessellates and bins ALL paths and draws 1 / world of the tiles; crh_frame_exchange of such layers
 * moves nothing in its all-to-all, composites nothing, and gathers an image that is bit-equal to a single GPU's (path sharding with RGBA8
 * layers: <= 2/255). (0, height) gives the whole frame back. Waits for the frame's last pass. */
crh_status crh_frame_set_tile_rows(crh_frame* frame, uint32_t row_begin, uint32_t row_end);
/* collective, the tile split's own exchange: every rank's `layer` holds its slab of rows (crh_frame_set_tile_rows with crh_comm_slab_rows' rows)
 * and the slabs travel straight from the layers' pixel rows into rank 0's `result` frame (NULL elsewhere): one grouped ncclSend / ncclRecv
 * per rank, no bitmaps, packing, plan, composite or unpacking. RGBA8 storage on both sides, frames of one size on every rank (checked).
 * crh_comm_last_timing then reports the transfer under [4], crh_comm_last_traffic the slab's bytes. */
crh_status crh_frame_gather_slabs(crh_comm* comm, crh_frame* layer, crh_frame* result);
/* bytes this rank sent in the last exchange, and what dense slabs (no empty-tile suppression) would have been */
crh_status crh_comm_last_traffic(const crh_comm* comm, uint64_t* bytes_sent, uint64_t* bytes_dense);
/* GPU time of the phases of this rank's last exchange, in milliseconds (HIP events on the communicator's stream; waits for the exchange):
 * [0] occupancy bitmap + packing of the non-empty tiles, [1] all-gather of the bitmaps + their prefix sums, [2] all-to-all of the slab
 * tiles, [3] ordered composite of the slab, [4] gather of the composited tiles on rank 0, [5] unpacking into the result frame (rank 0). */
#define CRH_COMM_PHASES 6
crh_status crh_comm_last_timing(crh_comm* comm, float ms[CRH_COMM_PHASES]);
/* bytes this rank sent to every peer in the all-to-all of the last exchange: per_peer[world] (its own entry is 0) */
crh_status crh_comm_last_peer_bytes(const crh_comm* comm, uint64_t* per_peer);
/* What the transport says about this communicator: *nranks = ncclCommCount of an RCCL communicator (the size of the loopback group for a
 * local one), *rccl_version = ncclGetVersion's code (major * 10000 + minor * 100 + patch; 0 for a local communicator). A line of a
 * multi-GPU measurement carries both, so that "did RCCL see N ranks" is answered by RCCL. */
crh_status crh_comm_info(const crh_comm* comm, uint32_t* nranks, int32_t* rccl_version);
/* The same exchange without RCCL, for several communicators on ONE device driven by one thread (tests, single-GPU validation):
 * rank 0's communicator founds the group (rank0 = NULL), ranks 1.. join it; crh_comm_local_exchange(rank 0's comm, layers[world],
 * result) then runs every rank's part with device-to-device copies in place of the transfers. */
crh_status crh_comm_create_local(crh_renderer* renderer, uint32_t rank, uint32_t world, crh_comm* rank0, crh_comm** out);
crh_status crh_comm_local_exchange(crh_comm* rank0, crh_frame* const* layers, crh_frame* result);
crh_status crh_comm_local_gather_slabs(crh_comm* rank0, crh_frame* const* layers, crh_frame* result); /* crh_frame_gather_slabs over the loopback group */

const char* crh_last_error(void);
const char* crh_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CONTRAST_HIP_H */
