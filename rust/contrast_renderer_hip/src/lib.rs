//! contrast_renderer's `Renderer` / `Shape` / `Path` API (Lichtso/contrast_renderer v0.1.4) over `libcontrast_hip.so`.
//!
//! The reference tessellates on the CPU (`stroke.rs`, `fill.rs`, `convex_hull.rs`) and rasterises through wgpu (`shaders.wgsl`); this
//! crate keeps its public types and call order and sends the work to hand-written HIP kernels for MI355X through the C ABI of
//! `include/contrast_hip.h` (raw bindings: [`ffi`], generated from that header). What changes for a caller:
//!
//! * `&wgpu::Device` becomes the HIP device ordinal given to [`Renderer::new`]; `&wgpu::Queue` disappears (uploads are ordered on
//!   the renderer's own HIP streams);
//! * the caller-owned `wgpu::RenderPass` becomes [`RenderPass`], which records `Shape::render` calls in order and submits them as one
//!   pass over a [`Frame`] (the colour + depth / stencil attachments);
//! * `SafeFloat<f32, N>` fields are plain `[f32; N]` / `f32` here: the library validates finiteness and canonicalises `-0.0` on
//!   upload, where the reference does it in `SafeFloat::from` (`safe_float.rs:44-52`).
//!
//! Statuses 1..=5 are the reference's [`Error`] variants in declaration order (`error.rs:5-16`). What the reference *panics* on
//! (non-finite input `safe_float.rs:46`, degenerate cubics `fill.rs:174,178`) panics here too, with the library's message.
//!
//! NOTE: the image this crate was written in has no Rust toolchain; the crate ships as source. Its raw bindings are generated from the
//! header and checked against it by `tests/test_rust_shim.py`; the tested callers of the same ABI are the C++ and Python mirrors.
pub mod ffi;

use std::ffi::CStr;
use std::ops::Range;
use std::ptr;

/// error.rs:5-16, same order
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum Error {
    NumberOfStencilBitsIsUnsupported,
    ClipStackOverflow,
    TooManyNestedOpacityGroups,
    TooManyDashIntervals,
    DynamicStrokeOptionsIndexOutOfBounds,
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::crh_last_error()) }.to_string_lossy().into_owned()
}

fn status(code: ffi::crh_status) -> Result<(), Error> {
    match code {
        ffi::CRH_OK => Ok(()),
        ffi::CRH_ERR_NUMBER_OF_STENCIL_BITS_IS_UNSUPPORTED => Err(Error::NumberOfStencilBitsIsUnsupported),
        ffi::CRH_ERR_CLIP_STACK_OVERFLOW => Err(Error::ClipStackOverflow),
        ffi::CRH_ERR_TOO_MANY_NESTED_OPACITY_GROUPS => Err(Error::TooManyNestedOpacityGroups),
        ffi::CRH_ERR_TOO_MANY_DASH_INTERVALS => Err(Error::TooManyDashIntervals),
        ffi::CRH_ERR_DYNAMIC_STROKE_OPTIONS_INDEX_OUT_OF_BOUNDS => Err(Error::DynamicStrokeOptionsIndexOutOfBounds),
        // the reference panics where the library reports 6 (safe_float.rs:46,114) and 7 (fill.rs:174,178); 8.. are not reference states
        other => panic!("contrast_hip status {}: {}", other, last_error()),
    }
}

// ------------------------------------------------------------------------------------------------ path.rs

/// path.rs:15-18
#[derive(Debug, Clone, Copy, PartialEq)]
pub struct LineSegment {
    pub control_points: [[f32; 2]; 1],
}
/// path.rs:22-25
#[derive(Debug, Clone, Copy, PartialEq)]
pub struct IntegralQuadraticCurveSegment {
    pub control_points: [[f32; 2]; 2],
}
/// path.rs:29-32
#[derive(Debug, Clone, Copy, PartialEq)]
pub struct IntegralCubicCurveSegment {
    pub control_points: [[f32; 2]; 3],
}
/// path.rs:36-43
#[derive(Debug, Clone, Copy, PartialEq)]
pub struct RationalQuadraticCurveSegment {
    pub weight: f32,
    pub control_points: [[f32; 2]; 2],
}
/// path.rs:47-52
#[derive(Debug, Clone, Copy, PartialEq)]
pub struct RationalCubicCurveSegment {
    pub weights: [f32; 4],
    pub control_points: [[f32; 2]; 3],
}
/// path.rs:56-67 (discriminants = CRH_SEGMENT_*)
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum SegmentType {
    Line = 0,
    IntegralQuadraticCurve = 1,
    IntegralCubicCurve = 2,
    RationalQuadraticCurve = 3,
    RationalCubicCurve = 4,
}
/// path.rs:71-82
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum Join {
    Miter = 0,
    Bevel = 1,
    Round = 2,
}
/// path.rs:86-101
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum Cap {
    Square = 0,
    Round = 1,
    Out = 2,
    In = 3,
    Right = 4,
    Left = 5,
    Butt = 6,
}
/// path.rs:105-118
#[derive(Debug, Clone, Copy, PartialEq)]
pub struct DashInterval {
    pub gap_start: f32,
    pub gap_end: f32,
    pub dash_start: Cap,
    pub dash_end: Cap,
}
/// path.rs:121
pub const MAX_DASH_INTERVALS: usize = ffi::CRH_MAX_DASH_INTERVALS;
/// path.rs:127-149
#[derive(Debug, Clone, PartialEq)]
pub enum DynamicStrokeOptions {
    Dashed { join: Join, pattern: Vec<DashInterval>, phase: f32 },
    Solid { join: Join, start: Cap, end: Cap },
}
/// path.rs:153-167
#[derive(Debug, Clone, Copy, PartialEq)]
pub enum CurveApproximation {
    UniformlySpacedParameters(usize),
    UniformTangentAngle(f32),
}
/// path.rs:171-192
#[derive(Debug, Clone, PartialEq)]
pub struct StrokeOptions {
    pub width: f32,
    pub offset: f32,
    pub miter_clip: f32,
    pub closed: bool,
    pub dynamic_stroke_options_group: usize,
    pub curve_approximation: CurveApproximation,
}
impl StrokeOptions {
    /// path.rs:196-200
    pub fn legalize(&mut self) {
        self.width = self.width.abs();
        self.offset = self.offset.clamp(-0.5, 0.5);
        self.miter_clip = self.miter_clip.abs();
    }
}
/// path.rs:213-230
#[derive(Default, Debug, Clone, PartialEq)]
pub struct Path {
    pub stroke_options: Option<StrokeOptions>,
    pub start: [f32; 2],
    pub line_segments: Vec<LineSegment>,
    pub integral_quadratic_curve_segments: Vec<IntegralQuadraticCurveSegment>,
    pub integral_cubic_curve_segments: Vec<IntegralCubicCurveSegment>,
    pub rational_quadratic_curve_segments: Vec<RationalQuadraticCurveSegment>,
    pub rational_cubic_curve_segments: Vec<RationalCubicCurveSegment>,
    pub segment_types: Vec<SegmentType>,
}
impl Path {
    /// path.rs:232-236 and the push_* family :240-318
    pub fn push_line(&mut self, segment: LineSegment) {
        self.line_segments.push(segment);
        self.segment_types.push(SegmentType::Line);
    }
    pub fn push_integral_quadratic_curve(&mut self, segment: IntegralQuadraticCurveSegment) {
        self.integral_quadratic_curve_segments.push(segment);
        self.segment_types.push(SegmentType::IntegralQuadraticCurve);
    }
    pub fn push_integral_cubic_curve(&mut self, segment: IntegralCubicCurveSegment) {
        self.integral_cubic_curve_segments.push(segment);
        self.segment_types.push(SegmentType::IntegralCubicCurve);
    }
    pub fn push_rational_quadratic_curve(&mut self, segment: RationalQuadraticCurveSegment) {
        self.rational_quadratic_curve_segments.push(segment);
        self.segment_types.push(SegmentType::RationalQuadraticCurve);
    }
    pub fn push_rational_cubic_curve(&mut self, segment: RationalCubicCurveSegment) {
        self.rational_cubic_curve_segments.push(segment);
        self.segment_types.push(SegmentType::RationalCubicCurve);
    }
}

fn convert_dynamic(options: &DynamicStrokeOptions) -> ffi::crh_dynamic_stroke_options {
    let empty = ffi::crh_dash_interval { gap_start: 0.0, gap_end: 0.0, dash_start: 0, dash_end: 0 };
    let mut out = ffi::crh_dynamic_stroke_options { dashed: 0, join: 0, pattern_len: 0, pattern: [empty; ffi::CRH_MAX_DASH_INTERVALS], phase: 0.0, start: 0, end: 0 };
    match options {
        DynamicStrokeOptions::Dashed { join, pattern, phase } => {
            out.dashed = 1;
            out.join = *join as u32;
            out.pattern_len = pattern.len() as u32; // > MAX_DASH_INTERVALS comes back as TooManyDashIntervals (renderer.rs:32-34)
            for (slot, interval) in out.pattern.iter_mut().zip(pattern.iter()) {
                *slot = ffi::crh_dash_interval { gap_start: interval.gap_start, gap_end: interval.gap_end, dash_start: interval.dash_start as u32, dash_end: interval.dash_end as u32 };
            }
            out.phase = *phase;
        }
        DynamicStrokeOptions::Solid { join, start, end } => {
            out.join = *join as u32;
            out.start = *start as u32;
            out.end = *end as u32;
        }
    }
    out
}

fn convert_stroke(options: &StrokeOptions) -> ffi::crh_stroke_options {
    let (curve_approximation, steps, angle_step) = match options.curve_approximation {
        CurveApproximation::UniformlySpacedParameters(steps) => (ffi::CRH_CURVE_UNIFORMLY_SPACED_PARAMETERS, steps as u32, 0.0),
        CurveApproximation::UniformTangentAngle(angle_step) => (ffi::CRH_CURVE_UNIFORM_TANGENT_ANGLE, 0, angle_step),
    };
    ffi::crh_stroke_options {
        width: options.width,
        offset: options.offset,
        miter_clip: options.miter_clip,
        closed: options.closed as u32,
        dynamic_stroke_options_group: options.dynamic_stroke_options_group as u32,
        curve_approximation,
        steps,
        angle_step,
    }
}

/// `&[Path]` of one or more Shapes flattened into the struct-of-arrays batch of the C ABI (segment order = `Path::segment_types`, record
/// layouts = path.rs:15-52). Owns the vectors the `crh_path_batch` view points into.
#[derive(Default)]
pub struct PathBatch {
    shape_path_begin: Vec<u32>,
    path_segment_begin: Vec<u32>,
    path_start: Vec<f32>,
    path_stroke_options: Vec<i32>,
    segment_types: Vec<u8>,
    control_data: Vec<f32>,
    stroke_options: Vec<ffi::crh_stroke_options>,
    shape_dynamic_begin: Vec<u32>,
    dynamic_stroke_options: Vec<ffi::crh_dynamic_stroke_options>,
}
impl PathBatch {
    pub fn new() -> Self {
        Self { shape_path_begin: vec![0], path_segment_begin: vec![0], shape_dynamic_begin: vec![0], ..Default::default() }
    }
    /// Appends one Shape (the arguments of `Shape::from_paths`, renderer.rs:180-181).
    pub fn push_shape(&mut self, dynamic_stroke_options: &[DynamicStrokeOptions], paths: &[Path]) {
        for path in paths {
            self.path_start.extend_from_slice(&path.start);
            let mut line = path.line_segments.iter();
            let mut integral_quadratic = path.integral_quadratic_curve_segments.iter();
            let mut integral_cubic = path.integral_cubic_curve_segments.iter();
            let mut rational_quadratic = path.rational_quadratic_curve_segments.iter();
            let mut rational_cubic = path.rational_cubic_curve_segments.iter();
            for segment_type in &path.segment_types {
                self.segment_types.push(*segment_type as u8);
                match segment_type {
                    SegmentType::Line => {
                        for point in &line.next().unwrap().control_points {
                            self.control_data.extend_from_slice(point);
                        }
                    }
                    SegmentType::IntegralQuadraticCurve => {
                        for point in &integral_quadratic.next().unwrap().control_points {
                            self.control_data.extend_from_slice(point);
                        }
                    }
                    SegmentType::IntegralCubicCurve => {
                        for point in &integral_cubic.next().unwrap().control_points {
                            self.control_data.extend_from_slice(point);
                        }
                    }
                    SegmentType::RationalQuadraticCurve => {
                        let segment = rational_quadratic.next().unwrap();
                        self.control_data.push(segment.weight);
                        for point in &segment.control_points {
                            self.control_data.extend_from_slice(point);
                        }
                    }
                    SegmentType::RationalCubicCurve => {
                        let segment = rational_cubic.next().unwrap();
                        self.control_data.extend_from_slice(&segment.weights);
                        for point in &segment.control_points {
                            self.control_data.extend_from_slice(point);
                        }
                    }
                }
            }
            self.path_segment_begin.push(self.segment_types.len() as u32);
            self.path_stroke_options.push(match &path.stroke_options {
                None => -1,
                Some(options) => {
                    self.stroke_options.push(convert_stroke(options));
                    self.stroke_options.len() as i32 - 1
                }
            });
        }
        self.shape_path_begin.push(self.path_stroke_options.len() as u32);
        self.dynamic_stroke_options.extend(dynamic_stroke_options.iter().map(convert_dynamic));
        self.shape_dynamic_begin.push(self.dynamic_stroke_options.len() as u32);
    }
    pub fn shape_count(&self) -> usize {
        self.shape_path_begin.len() - 1
    }
    fn view(&self) -> ffi::crh_path_batch {
        ffi::crh_path_batch {
            n_shapes: self.shape_count() as u32,
            shape_path_begin: self.shape_path_begin.as_ptr(),
            n_paths: self.path_stroke_options.len() as u32,
            path_segment_begin: self.path_segment_begin.as_ptr(),
            path_start: self.path_start.as_ptr(),
            path_stroke_options: self.path_stroke_options.as_ptr(),
            n_segments: self.segment_types.len() as u32,
            segment_types: self.segment_types.as_ptr(),
            control_data: self.control_data.as_ptr(),
            n_control_floats: self.control_data.len() as u32,
            n_stroke_options: self.stroke_options.len() as u32,
            stroke_options: self.stroke_options.as_ptr(),
            shape_dynamic_begin: self.shape_dynamic_begin.as_ptr(),
            n_dynamic_stroke_options: self.dynamic_stroke_options.len() as u32,
            dynamic_stroke_options: self.dynamic_stroke_options.as_ptr(),
        }
    }
}

// ------------------------------------------------------------------------------------------------ renderer.rs

/// renderer.rs:145-160, same order (= crh_render_op)
#[derive(Clone, Copy, PartialOrd, Ord, PartialEq, Eq, Debug)]
pub enum RenderOperation {
    Stencil = 0,
    Clip = 1,
    UnClip = 2,
    Color = 3,
    SaveAlphaContext = 4,
    ScaleAlphaContext = 5,
    RestoreAlphaContext = 6,
}

/// `Option<wgpu::Face>` of `Configuration::cull_mode`; front = counter-clockwise on screen (renderer.rs:477)
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub enum Face {
    Front = 1,
    Back = 2,
}
/// `wgpu::CompareFunction` of `Configuration::depth_compare` (fragment depth OP stored depth)
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub enum CompareFunction {
    Always = 0,
    Never = 1,
    Less = 2,
    Equal = 3,
    LessEqual = 4,
    Greater = 5,
    NotEqual = 6,
    GreaterEqual = 7,
}

/// renderer.rs:380-405. `blending` is fixed to premultiplied "over" (examples/showcase/main.rs:32-43), `depth_stencil_format` and
/// `color_attachment_in_stencil_pass` are wgpu details without meaning for a compute rasterizer.
#[derive(Clone, Copy, Debug)]
pub struct Configuration {
    pub cull_mode: Option<Face>,
    pub depth_compare: CompareFunction,
    pub depth_write_enabled: bool,
    pub msaa_sample_count: u32,
    pub clip_nesting_counter_bits: usize,
    pub winding_counter_bits: usize,
    pub alpha_layer_count: usize,
}
impl Default for Configuration {
    /// examples/showcase/main.rs:45-56
    fn default() -> Self {
        Self { cull_mode: None, depth_compare: CompareFunction::Always, depth_write_enabled: false, msaa_sample_count: 4, clip_nesting_counter_bits: 4, winding_counter_bits: 4, alpha_layer_count: 0 }
    }
}

/// renderer.rs:408
pub struct Renderer {
    raw: *mut ffi::crh_renderer,
    config: Configuration,
}
impl Renderer {
    /// renderer.rs:432 — `device` is the HIP device ordinal where the reference takes `&wgpu::Device`
    pub fn new(device: i32, config: Configuration) -> Result<Self, Error> {
        let c = ffi::crh_config {
            msaa_sample_count: config.msaa_sample_count,
            clip_nesting_counter_bits: config.clip_nesting_counter_bits as u32,
            winding_counter_bits: config.winding_counter_bits as u32,
            alpha_layer_count: config.alpha_layer_count as u32,
            cull_mode: config.cull_mode.map_or(ffi::CRH_CULL_NONE, |face| face as u32),
            depth_compare: config.depth_compare as u32,
            depth_write_enabled: config.depth_write_enabled as u32,
        };
        let mut raw = ptr::null_mut();
        status(unsafe { ffi::crh_renderer_create(&c, device, &mut raw) })?;
        Ok(Self { raw, config })
    }
    /// renderer.rs:887
    pub fn get_config(&self) -> &Configuration {
        &self.config
    }
    pub fn synchronize(&self) {
        status(unsafe { ffi::crh_renderer_synchronize(self.raw) }).unwrap()
    }
}
impl Drop for Renderer {
    fn drop(&mut self) {
        unsafe { ffi::crh_renderer_destroy(self.raw) }
    }
}

/// The colour attachment (RGBA8, premultiplied) plus the depth / stencil attachments of a render pass: caller-owned in the reference
/// (`wgpu::TextureView`s handed to `begin_render_pass`), an object of the library here.
pub struct Frame {
    raw: *mut ffi::crh_frame,
    width: u32,
    height: u32,
}
impl Frame {
    pub fn new(renderer: &Renderer, width: u32, height: u32) -> Result<Self, Error> {
        let mut raw = ptr::null_mut();
        status(unsafe { ffi::crh_frame_create(renderer.raw, width, height, &mut raw) })?;
        Ok(Self { raw, width, height })
    }
    /// `LoadOp::Clear` of colour, depth (1.0) and stencil (examples/showcase/main.rs:217-230)
    pub fn clear(&mut self) {
        status(unsafe { ffi::crh_frame_clear(self.raw) }).unwrap()
    }
    /// The reference's stencil attachment and alpha layers are caller-owned and outlive a `Shape::render` call (renderer.rs:148-158, 892-985): from this
    /// call until `clear` the frame keeps clip / winding counters, saved alphas and the colour of every sample between passes. (It also starts by itself at
    /// the first pass that ends with state left over; `RenderPass::submit` calls this when its draws span several objects.)
    pub fn keep_pass_state(&mut self) {
        status(unsafe { ffi::crh_frame_keep_pass_state(self.raw) }).unwrap()
    }
    /// The depth of the 3-D scene the Shapes are decals in: `[height][width]`
    pub fn upload_depth(&mut self, depth: &[f32]) {
        assert_eq!(depth.len(), (self.width * self.height) as usize);
        status(unsafe { ffi::crh_frame_upload_depth(self.raw, depth.as_ptr()) }).unwrap()
    }
    /// Premultiplied RGBA8, row 0 = top
    pub fn download(&mut self) -> Vec<u8> {
        let mut pixels = vec![0u8; (self.width * self.height * 4) as usize];
        status(unsafe { ffi::crh_frame_download(self.raw, pixels.as_mut_ptr() as *mut _) }).unwrap();
        pixels
    }
    /// Waits for the last pass into this frame only
    pub fn synchronize(&self) {
        status(unsafe { ffi::crh_frame_synchronize(self.raw) }).unwrap()
    }
    /// The tile split of the multi-GPU path: passes into this frame draw the pixel rows `rows` only (whole 16-pixel tile rows)
    pub fn set_tile_rows(&self, rows: std::ops::Range<u32>) {
        status(unsafe { ffi::crh_frame_set_tile_rows(self.raw, rows.start, rows.end) }).unwrap()
    }
}
impl Drop for Frame {
    fn drop(&mut self) {
        unsafe { ffi::crh_frame_destroy(self.raw) }
    }
}

/// renderer.rs:163-171 — a set of Paths which is always rendered together
pub struct Shape {
    raw: *mut ffi::crh_scene,
    dynamic_stroke_options_count: usize,
}
impl Shape {
    /// renderer.rs:177-183. `existing_shape` is consumed like the reference's `Option<(Shape, &wgpu::Queue)>`: its device allocations
    /// are reused when large enough (renderer.rs:216-221).
    pub fn from_paths(renderer: &Renderer, dynamic_stroke_options: &[DynamicStrokeOptions], paths: &[Path], existing_shape: Option<Shape>) -> Result<Self, Error> {
        let mut batch = PathBatch::new();
        batch.push_shape(dynamic_stroke_options, paths);
        let view = batch.view();
        // `existing_shape` is moved in (renderer.rs:182): from here on the library owns its handle
        let existing = match existing_shape {
            Some(shape) => {
                let raw = shape.raw;
                std::mem::forget(shape);
                raw
            }
            None => ptr::null_mut(),
        };
        let mut raw = ptr::null_mut();
        match status(unsafe { ffi::crh_shape_from_paths(renderer.raw, &view, existing, &mut raw) }) {
            Ok(()) => Ok(Self { raw, dynamic_stroke_options_count: dynamic_stroke_options.len() }),
            Err(error) => {
                // like the reference, an Err drops the Shape that was moved in: `raw` is it once the upload took it over, `existing` before
                let victim = if raw.is_null() { existing } else { raw };
                if !victim.is_null() {
                    unsafe { ffi::crh_scene_destroy(victim) }
                }
                Err(error)
            }
        }
    }
    /// renderer.rs:267-273 — records one draw into the pass; the instances are indices into the pass' instance data
    pub fn render(&self, _renderer: &Renderer, render_pass: &mut RenderPass, instance_indices: Range<u32>, render_operation: RenderOperation) {
        for instance in instance_indices {
            render_pass.draws.push((self.raw, ffi::crh_draw { shape: 0, instance, op: render_operation as u32, clip_depth: render_pass.clip_depth, alpha_layer: render_pass.alpha_layer }));
        }
    }
    /// renderer.rs:360-376
    pub fn set_dynamic_stroke_options(&self, dynamic_stroke_options_group_index: usize, dynamic_stroke_options_group: &DynamicStrokeOptions) -> Result<(), Error> {
        if dynamic_stroke_options_group_index >= self.dynamic_stroke_options_count {
            return Err(Error::DynamicStrokeOptionsIndexOutOfBounds);
        }
        let options = convert_dynamic(dynamic_stroke_options_group);
        status(unsafe { ffi::crh_scene_set_dynamic_stroke_options(self.raw, 0, dynamic_stroke_options_group_index as u32, &options) })
    }
    /// Parity tap: the byte image `renderer.rs:198-209` uploads, with its cumulative END offsets
    pub fn buffers(&self) -> ([u64; 8], [u64; 3], Vec<u8>, Vec<u8>) {
        let (mut vertex_offsets, mut index_offsets) = ([0u64; 8], [0u64; 3]);
        status(unsafe { ffi::crh_scene_shape_layout(self.raw, 0, vertex_offsets.as_mut_ptr(), index_offsets.as_mut_ptr()) }).unwrap();
        let (mut vertices, mut indices) = (vec![0u8; vertex_offsets[7] as usize], vec![0u8; index_offsets[2] as usize]);
        status(unsafe { ffi::crh_scene_shape_download(self.raw, 0, vertices.as_mut_ptr() as *mut _, indices.as_mut_ptr() as *mut _) }).unwrap();
        (vertex_offsets, index_offsets, vertices, indices)
    }
}
impl Drop for Shape {
    fn drop(&mut self) {
        unsafe { ffi::crh_scene_destroy(self.raw) }
    }
}

/// Many Shapes built together: one launch tessellates all of them (the reference's one-call-per-Shape loop, renderer.rs:187, is
/// launch-latency bound on a GPU). No counterpart upstream; `Shape` is the `n == 1` case.
pub struct Scene {
    raw: *mut ffi::crh_scene,
    shape_count: usize,
}
impl Scene {
    pub fn new(renderer: &Renderer, batch: &PathBatch) -> Result<Self, Error> {
        let view = batch.view();
        let mut raw = ptr::null_mut();
        status(unsafe { ffi::crh_scene_upload(renderer.raw, &view, ptr::null_mut(), &mut raw) })?;
        let scene = Self { raw, shape_count: batch.shape_count() };
        scene.tessellate()?;
        status(unsafe { ffi::crh_scene_status(scene.raw) })?;
        Ok(scene)
    }
    /// The arithmetic of `from_paths` for every Shape, asynchronously on the renderer's tessellation stream
    pub fn tessellate(&self) -> Result<(), Error> {
        status(unsafe { ffi::crh_scene_tessellate(self.raw) })
    }
    /// Stencil + Color of every Shape in index order, instance i = Shape i (the loop of examples/showcase/main.rs:236-250)
    pub fn render(&self, frame: &mut Frame, transforms: &[[f32; 16]], colors: &[[f32; 4]]) -> Result<(), Error> {
        assert!(transforms.len() == self.shape_count && colors.len() == self.shape_count);
        status(unsafe { ffi::crh_scene_render(self.raw, frame.raw, transforms.as_ptr() as *const f32, colors.as_ptr() as *const f32) })
    }
    /// A recorded pass over the Shapes of this Scene: `draws[i]` = (index of the Shape, index into `transforms` / `colors`, operation, clip
    /// depth and alpha layer in effect) — what a sequence of `Shape::render` calls between `Renderer::set_clip_depth` /
    /// `save_alpha_context` calls records in the reference (renderer.rs:267-355, :932-985). Clip nesting and alpha contexts may span Shapes
    /// here: one call is one pass.
    pub fn render_draws(&self, frame: &mut Frame, transforms: &[[f32; 16]], colors: &[[f32; 4]], draws: &[(u32, u32, RenderOperation, u32, u32)]) -> Result<(), Error> {
        assert!(transforms.len() == colors.len());
        let raw_draws: Vec<ffi::crh_draw> = draws
            .iter()
            .map(|&(shape, instance, op, clip_depth, alpha_layer)| {
                assert!((shape as usize) < self.shape_count && (instance as usize) < transforms.len());
                ffi::crh_draw { shape, instance, op: op as u32, clip_depth, alpha_layer }
            })
            .collect();
        status(unsafe {
            ffi::crh_scene_render_draws(self.raw, frame.raw, transforms.as_ptr() as *const f32, colors.as_ptr() as *const f32, transforms.len() as u32, raw_draws.as_ptr(), raw_draws.len() as u32)
        })
    }
}
impl Drop for Scene {
    fn drop(&mut self) {
        unsafe { ffi::crh_scene_destroy(self.raw) }
    }
}

/// What the reference records into a caller-owned `wgpu::RenderPass` (renderer.rs:267-355): `Shape::render` calls in order, the clip
/// depth (`Renderer::set_clip_depth`, renderer.rs:932-938) and the alpha layer in effect. `submit` runs the pass over the frame.
pub struct RenderPass<'a> {
    frame: &'a mut Frame,
    transforms: Vec<[f32; 16]>,
    colors: Vec<[f32; 4]>,
    draws: Vec<(*mut ffi::crh_scene, ffi::crh_draw)>,
    clip_depth: u32,
    alpha_layer: u32,
}
impl<'a> RenderPass<'a> {
    pub fn new(frame: &'a mut Frame) -> Self {
        Self { frame, transforms: Vec::new(), colors: Vec::new(), draws: Vec::new(), clip_depth: 0, alpha_layer: 0 }
    }
    /// One entry of the instance buffers the reference binds at vertex slots 0 and 1 (column-major mat4 + straight-alpha colour,
    /// shaders.wgsl:13-27); returns its index for `Shape::render`'s `instance_indices`
    pub fn push_instance(&mut self, transform: [f32; 16], color: [f32; 4]) -> u32 {
        self.transforms.push(transform);
        self.colors.push(color);
        self.transforms.len() as u32 - 1
    }
    /// renderer.rs:932-938
    pub fn set_clip_depth(&mut self, renderer: &Renderer, clip_depth: usize) -> Result<(), Error> {
        if clip_depth >= (1 << renderer.config.clip_nesting_counter_bits) {
            return Err(Error::ClipStackOverflow);
        }
        self.clip_depth = clip_depth as u32;
        Ok(())
    }
    /// renderer.rs:940-977 / :979-985: the alpha layer of the following Save / Scale / RestoreAlphaContext draws
    pub fn set_alpha_layer(&mut self, renderer: &Renderer, alpha_layer: usize) -> Result<(), Error> {
        if alpha_layer >= renderer.config.alpha_layer_count {
            return Err(Error::TooManyNestedOpacityGroups);
        }
        self.alpha_layer = alpha_layer as u32;
        Ok(())
    }
    /// Runs the recorded draws, in order, as one `crh_scene_render_draws` per run of draws of the same Shape object. Clip nesting counters,
    /// winding counters, saved alpha contexts and the colour of every sample stay with the [`Frame`] between those calls (the library keeps
    /// them in HBM from the first pass that ends with state left over until `Frame::clear`), so the reference's pattern — `a.render(Stencil)`,
    /// `set_clip_depth(1)`, `a.render(Clip)`, other Shapes, `set_clip_depth(0)`, `a.render(UnClip)` (renderer.rs:257-266) — works with `a`
    /// and the clipped Shapes as separate objects, as does an opacity group around other Shapes (renderer.rs:941-985).
    pub fn submit(self) -> Result<(), Error> {
        if self.draws.windows(2).any(|pair| pair[0].0 != pair[1].0) {
            // the pass spans objects: every sample's colour and stencil stay with the frame from its first draw on
            status(unsafe { ffi::crh_frame_keep_pass_state(self.frame.raw) })?;
        }
        let mut begin = 0;
        while begin < self.draws.len() {
            let scene = self.draws[begin].0;
            let mut end = begin;
            while end < self.draws.len() && self.draws[end].0 == scene {
                end += 1;
            }
            let draws: Vec<ffi::crh_draw> = self.draws[begin..end].iter().map(|(_, draw)| *draw).collect();
            status(unsafe {
                ffi::crh_scene_render_draws(scene, self.frame.raw, self.transforms.as_ptr() as *const f32, self.colors.as_ptr() as *const f32, self.transforms.len() as u32, draws.as_ptr(), draws.len() as u32)
            })?;
            begin = end;
        }
        Ok(())
    }
}

// ------------------------------------------------------------------------------------------------ multi-GPU (no counterpart upstream)

/// One rank of the framebuffer exchange (include/contrast_hip.h `crh_comm_*`): one process per GPU, rank g renders the Shape range
/// [`Comm::shard`] gives it into a private [`Frame`]; [`Comm::exchange`] composites the layers in rank order into rank 0's result.
pub struct Comm {
    raw: *mut ffi::crh_comm,
    rank: u32,
}
impl Comm {
    /// Rank 0 creates the id and hands it to the other ranks by any means
    pub fn unique_id() -> [u8; ffi::CRH_COMM_ID_BYTES] {
        let mut id = [0u8; ffi::CRH_COMM_ID_BYTES];
        status(unsafe { ffi::crh_comm_unique_id(id.as_mut_ptr() as *mut _) }).unwrap();
        id
    }
    pub fn new(renderer: &Renderer, rank: u32, world: u32, unique_id: &[u8; ffi::CRH_COMM_ID_BYTES]) -> Result<Self, Error> {
        let mut raw = ptr::null_mut();
        status(unsafe { ffi::crh_comm_create(renderer.raw, rank, world, unique_id.as_ptr() as *const _, &mut raw) })?;
        Ok(Self { raw, rank })
    }
    /// The contiguous, order-preserving Shape range of a rank
    pub fn shard(n_items: u32, rank: u32, world: u32) -> Range<u32> {
        let (mut begin, mut end) = (0u32, 0u32);
        status(unsafe { ffi::crh_comm_shard(n_items, rank, world, &mut begin, &mut end) }).unwrap();
        begin..end
    }
    /// The pixel rows of rank `rank`'s slab of a frame `height` pixels high (whole 16-pixel tile rows)
    pub fn slab_rows(height: u32, rank: u32, world: u32) -> Range<u32> {
        let (mut begin, mut end) = (0u32, 0u32);
        status(unsafe { ffi::crh_comm_slab_rows(height, rank, world, &mut begin, &mut end) }).unwrap();
        begin..end
    }
    /// Collective. `result` must be `Some` on rank 0 and `None` elsewhere.
    pub fn exchange(&self, layer: &mut Frame, result: Option<&mut Frame>) -> Result<(), Error> {
        assert_eq!(self.rank == 0, result.is_some());
        status(unsafe { ffi::crh_frame_exchange(self.raw, layer.raw, result.map_or(ptr::null_mut(), |frame| frame.raw)) })
    }
    /// Collective, the tile split: every rank's slab of rows (`Frame::set_tile_rows`) straight into rank 0's `result`.
    pub fn gather_slabs(&self, layer: &mut Frame, result: Option<&mut Frame>) -> Result<(), Error> {
        assert_eq!(self.rank == 0, result.is_some());
        status(unsafe { ffi::crh_frame_gather_slabs(self.raw, layer.raw, result.map_or(ptr::null_mut(), |frame| frame.raw)) })
    }
}
impl Drop for Comm {
    fn drop(&mut self) {
        unsafe { ffi::crh_comm_destroy(self.raw) }
    }
}
