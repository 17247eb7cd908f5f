"""Host-side mirror of the reference's renderer.rs API over the C ABI (ctypes): Renderer, Shape, RenderOperation — plus the
batch forms (Scene = many Shapes built and rendered together, Frame = the render pass attachments).

All arithmetic runs in libcontrast_hip.so on the GPU; this module only marshals arguments.
"""
import ctypes as C
from dataclasses import dataclass
from enum import IntEnum

import numpy as np

from . import _ffi
from ._ffi import ContrastError, check
from .path import batch_from_shapes


class RenderOperation(IntEnum):  # renderer.rs:145-160
    Stencil = 0
    Clip = 1
    UnClip = 2
    Color = 3
    SaveAlphaContext = 4
    ScaleAlphaContext = 5
    RestoreAlphaContext = 6


class Cull(IntEnum):  # Option<wgpu::Face> of Configuration::cull_mode (renderer.rs:383-384); front = counter-clockwise on screen
    Disabled = 0
    Front = 1
    Back = 2


class Compare(IntEnum):  # wgpu::CompareFunction of Configuration::depth_compare (renderer.rs:387-388): fragment depth OP stored depth
    Always = 0
    Never = 1
    Less = 2
    Equal = 3
    LessEqual = 4
    Greater = 5
    NotEqual = 6
    GreaterEqual = 7


@dataclass
class Configuration:  # renderer.rs:380-405 (fields that change results on this path)
    msaa_sample_count: int = 1
    clip_nesting_counter_bits: int = 4
    winding_counter_bits: int = 4
    alpha_layer_count: int = 0
    cull_mode: int = Cull.Disabled          # the three depth / cull fields act on the colour cover only (renderer.rs:743-745)
    depth_compare: int = Compare.Always
    depth_write_enabled: bool = False


class Renderer:
    """Renderer::new (renderer.rs:432): validates the stencil bit budget, owns one HIP stream on `device`."""

    def __init__(self, config: Configuration = None, device: int = 0):
        self.lib = _ffi.load_library()
        config = config or Configuration()
        c = _ffi.ConfigC(config.msaa_sample_count, config.clip_nesting_counter_bits, config.winding_counter_bits, config.alpha_layer_count,
                         int(config.cull_mode), int(config.depth_compare), 1 if config.depth_write_enabled else 0)
        handle = C.c_void_p()
        check(self.lib.crh_renderer_create(C.byref(c), device, C.byref(handle)))
        self.handle = handle
        self.config = config
        self.device = device

    def get_config(self):
        return self.config

    def synchronize(self):
        check(self.lib.crh_renderer_synchronize(self.handle))

    def enable_timing(self, enabled=True):
        """True / 1: HIP events around every kernel; 2: around the raster lane's kernels only (cheap enough for a timed loop); False / 0: off"""
        check(self.lib.crh_renderer_enable_timing(self.handle, int(enabled)))

    def kernel_times(self):
        """[(kernel name, milliseconds, algorithmic bytes)] of the last tessellate / render call (HIP events on the renderer's stream)."""
        n = C.c_uint32()
        check(self.lib.crh_renderer_kernel_times(self.handle, None, 0, C.byref(n)))
        out = (_ffi.KernelTimeC * max(1, n.value))()
        check(self.lib.crh_renderer_kernel_times(self.handle, out, n.value, C.byref(n)))
        return [(out[i].name.decode(), out[i].ms, out[i].algorithmic_bytes) for i in range(n.value)]

    def selftest_fmath(self, fn, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(a if b is None else b, dtype=np.float32)
        out = np.zeros_like(a)
        fp = C.POINTER(C.c_float)
        check(self.lib.crh_selftest_fmath(self.handle, fn, a.ctypes.data_as(fp), b.ctypes.data_as(fp), out.ctypes.data_as(fp), a.size))
        return out

    def __del__(self):
        if getattr(self, "handle", None) and self.lib is not None:
            self.lib.crh_renderer_destroy(self.handle)
            self.handle = None


FORMAT_RGBA8, FORMAT_RGBA16F, FORMAT_RGBA8_ATTACHMENT = 0, 1, 2  # (2: RGBA8 storage, every blend rounded to 8 bits like an Rgba8Unorm attachment)


class Frame:
    """The colour attachment (premultiplied; RGBA8, or RGBA16F for the layers of the multi-GPU exchange) and per-sample winding state of one render pass."""

    def __init__(self, renderer: Renderer, width: int, height: int, format: int = FORMAT_RGBA8):
        self.renderer = renderer
        self.lib = renderer.lib
        self.width, self.height, self.format = width, height, format
        handle = C.c_void_p()
        check(self.lib.crh_frame_create_format(renderer.handle, width, height, format, C.byref(handle)))
        self.handle = handle

    def clear(self):
        check(self.lib.crh_frame_clear(self.handle))

    def keep_pass_state(self):
        """From now until clear(): clip / winding counters, saved alphas and the f32 colour of every sample stay with the frame between
        passes (the reference's caller-owned stencil attachment and alpha layers, renderer.rs:148-158, 257-266)."""
        check(self.lib.crh_frame_keep_pass_state(self.handle))

    def synchronize(self):
        """Waits for the last render into this frame only (the next frame of a double-buffered loop keeps running)."""
        check(self.lib.crh_frame_synchronize(self.handle))

    def set_tile_rows(self, row_begin, row_end):
        """The tile split of the multi-GPU path: passes into this frame draw the pixel rows [row_begin, row_end) only (whole 16-pixel tile rows;
        slab_rows() gives a rank's), the rest stays transparent; (0, height) gives the whole frame back."""
        check(self.lib.crh_frame_set_tile_rows(self.handle, int(row_begin), int(row_end)))

    def clear_depth(self, value=1.0):
        """LoadOp::Clear(value) of the depth attachment (main.rs:223-226); it exists when the configuration tests or writes depth."""
        check(self.lib.crh_frame_clear_depth(self.handle, value))

    def upload_depth(self, depth):
        """The depth of the 3-D scene the Shapes are decals in: [height, width] floats, replicated to every sample."""
        d = np.ascontiguousarray(depth, dtype=np.float32).reshape(self.height, self.width)
        check(self.lib.crh_frame_upload_depth(self.handle, d.ctypes.data_as(C.POINTER(C.c_float))))

    def download_depth(self):
        """-> [height, width, msaa_sample_count] float32."""
        out = np.zeros((self.height, self.width, self.renderer.config.msaa_sample_count), dtype=np.float32)
        check(self.lib.crh_frame_download_depth(self.handle, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def download(self):
        """-> [height, width, 4] uint8 (an RGBA8 frame) or float16 (an RGBA16F frame)."""
        if self.format == FORMAT_RGBA16F:
            out = np.zeros((self.height, self.width, 4), dtype=np.float16)
            check(self.lib.crh_frame_download_f16(self.handle, out.ctypes.data))
            return out
        out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        check(self.lib.crh_frame_download(self.handle, out.ctypes.data))
        return out

    def device_pointer(self):
        p = C.c_void_p()
        check(self.lib.crh_frame_device_pointer(self.handle, C.byref(p)))
        return p.value

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.crh_frame_destroy(self.handle)
            self.handle = None


COMM_ID_BYTES = 128


def comm_unique_id(lib=None):
    """Rank 0: the 128-byte id (ncclGetUniqueId) every rank passes to Comm(...); distribute it by any means."""
    lib = lib or _ffi.load_library()
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    check(lib.crh_comm_unique_id(buf))
    return bytes(buf)


def shard_range(n_items, rank, world, lib=None):
    """crh_comm_shard: the contiguous Shape range of a rank (host arithmetic, no GPU)."""
    lib = lib or _ffi.load_library()
    b, e = C.c_uint32(), C.c_uint32()
    check(lib.crh_comm_shard(n_items, rank, world, C.byref(b), C.byref(e)))
    return b.value, e.value


def slab_rows(height, rank, world, lib=None):
    lib = lib or _ffi.load_library()
    b, e = C.c_uint32(), C.c_uint32()
    check(lib.crh_comm_slab_rows(height, rank, world, C.byref(b), C.byref(e)))
    return b.value, e.value


class Comm:
    """One rank of the framebuffer exchange (include/contrast_hip.h, crh_comm_*): RCCL when `unique_id` is given, otherwise a member of an
    in-process loopback group on one device (`rank0` = the group's founder for ranks > 0)."""

    def __init__(self, renderer: Renderer, rank: int, world: int, unique_id: bytes = None, rank0: "Comm" = None):
        self.renderer, self.lib, self.rank, self.world = renderer, renderer.lib, rank, world
        handle = C.c_void_p()
        if unique_id is not None:
            ident = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
            check(self.lib.crh_comm_create(renderer.handle, rank, world, ident, C.byref(handle)))
        else:
            check(self.lib.crh_comm_create_local(renderer.handle, rank, world, rank0.handle if rank0 else None, C.byref(handle)))
        self.handle = handle

    def exchange(self, layer: Frame, result: Frame = None):
        """Collective (RCCL): composites every rank's layer in rank order into rank 0's `result`."""
        check(self.lib.crh_frame_exchange(self.handle, layer.handle, result.handle if result is not None else None))

    def local_exchange(self, layers, result: Frame):
        """Loopback group, called on rank 0's communicator: layers[k] = rank k's frame."""
        arr = (C.c_void_p * len(layers))(*[f.handle for f in layers])
        check(self.lib.crh_comm_local_exchange(self.handle, arr, result.handle))

    def gather_slabs(self, layer: Frame, result: Frame = None):
        """Collective (RCCL), the tile split's exchange: every rank's slab of rows (Frame.set_tile_rows) straight into rank 0's `result`."""
        check(self.lib.crh_frame_gather_slabs(self.handle, layer.handle, result.handle if result is not None else None))

    def local_gather_slabs(self, layers, result: Frame):
        """... over the loopback group, called on rank 0's communicator."""
        arr = (C.c_void_p * len(layers))(*[f.handle for f in layers])
        check(self.lib.crh_comm_local_gather_slabs(self.handle, arr, result.handle))

    def last_traffic(self):
        sent, dense = C.c_uint64(), C.c_uint64()
        check(self.lib.crh_comm_last_traffic(self.handle, C.byref(sent), C.byref(dense)))
        return sent.value, dense.value

    PHASES = ("pack", "allgather_plan", "alltoall", "composite", "gather", "unpack")

    def last_timing(self):
        """GPU milliseconds of the phases of this rank's last exchange (waits for it): dict by phase name."""
        ms = (C.c_float * len(self.PHASES))()
        check(self.lib.crh_comm_last_timing(self.handle, ms))
        return dict(zip(self.PHASES, [float(v) for v in ms]))

    def last_peer_bytes(self):
        """Bytes this rank sent to every peer in the all-to-all of the last exchange."""
        out = (C.c_uint64 * self.world)()
        check(self.lib.crh_comm_last_peer_bytes(self.handle, out))
        return [int(v) for v in out]

    def info(self):
        """{"nranks": ncclCommCount (or the loopback group's size), "rccl_version": ncclGetVersion's code, 0 for a loopback communicator}"""
        n, v = C.c_uint32(), C.c_int32()
        check(self.lib.crh_comm_info(self.handle, C.byref(n), C.byref(v)))
        return {"nranks": int(n.value), "rccl_version": int(v.value)}

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.crh_comm_destroy(self.handle)
            self.handle = None


class Scene:
    """A batch of Shapes in HBM: upload + tessellate once, render many times."""

    def __init__(self, renderer: Renderer, batch: _ffi.PathBatch, tessellate=True, existing: "Scene" = None):
        self.renderer = renderer
        self.lib = renderer.lib
        self.batch = batch
        handle = C.c_void_p()
        check(self.lib.crh_scene_upload(renderer.handle, C.byref(batch.c), existing.handle if existing else None, C.byref(handle)))
        if existing is not None:
            existing.handle = None  # moved in, as `existing_shape` is in renderer.rs:182
        self.handle = handle
        self.n_shapes = batch.n_shapes
        if tessellate:
            self.tessellate()

    def tessellate(self):
        check(self.lib.crh_scene_tessellate(self.handle))

    def status(self):
        return self.lib.crh_scene_status(self.handle)

    def check(self):
        check(self.status())

    def shape(self, index):
        """-> (vertex_offsets[8], index_offsets[3], vertex bytes, index bytes): the byte image of renderer.rs:198-209."""
        vo = (C.c_uint64 * 8)()
        io = (C.c_uint64 * 3)()
        check(self.lib.crh_scene_shape_layout(self.handle, index, vo, io))
        vb = np.zeros(vo[7], dtype=np.uint8)
        ib = np.zeros(io[2], dtype=np.uint8)
        check(self.lib.crh_scene_shape_download(self.handle, index, vb.ctypes.data, ib.ctypes.data))
        return np.array(vo[:], dtype=np.uint64), np.array(io[:], dtype=np.uint64), vb, ib

    def all_shapes(self):
        layout = np.zeros((self.n_shapes, 11), dtype=np.uint64)
        tv, ti = C.c_uint64(), C.c_uint64()
        check(self.lib.crh_scene_layout_all(self.handle, layout.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(tv), C.byref(ti)))
        vb = np.zeros(tv.value, dtype=np.uint8)
        ib = np.zeros(ti.value, dtype=np.uint8)
        check(self.lib.crh_scene_download_all(self.handle, vb.ctypes.data, ib.ctypes.data))
        return layout, vb, ib

    def traffic(self):
        r, w = C.c_uint64(), C.c_uint64()
        check(self.lib.crh_scene_traffic(self.handle, C.byref(r), C.byref(w)))
        return r.value, w.value

    def set_instances(self, transforms, colors):
        t = np.ascontiguousarray(transforms, dtype=np.float32).reshape(self.n_shapes, 16)
        c = np.ascontiguousarray(colors, dtype=np.float32).reshape(self.n_shapes, 4)
        fp = C.POINTER(C.c_float)
        check(self.lib.crh_scene_set_instances(self.handle, t.ctypes.data_as(fp), c.ctypes.data_as(fp)))

    def render(self, frame: Frame, transforms=None, colors=None):
        """Stencil + Color of every shape in index order (the loop of examples/showcase/main.rs:236-250)."""
        if transforms is not None:
            self.set_instances(transforms, colors)
        check(self.lib.crh_scene_render_resident(self.handle, frame.handle))

    def render_draws(self, frame: Frame, transforms, colors, draws):
        """A recorded render pass: draws = [(shape, instance, RenderOperation, clip_depth, alpha_layer), ...] — one tuple per
        Shape::render call (renderer.rs:267-273) with the clip depth (Renderer::set_clip_depth, renderer.rs:932-938) and alpha layer
        (save/restore_alpha_context, renderer.rs:941-985) in effect. `instance` indexes transforms / colors (instancing)."""
        t = np.ascontiguousarray(transforms, dtype=np.float32).reshape(-1, 16)
        c = np.ascontiguousarray(colors, dtype=np.float32).reshape(-1, 4)
        assert len(t) == len(c)
        if isinstance(draws, np.ndarray):  # [n, 5] uint32 = crh_draw records: passed as they are (an animation re-submits the same array)
            table = np.ascontiguousarray(draws, dtype=np.uint32).reshape(-1, 5)
        else:
            table = np.zeros((len(draws), 5), dtype=np.uint32)
            for i, d in enumerate(draws):
                table[i, :len(d)] = [int(v) for v in d]
        fp = C.POINTER(C.c_float)
        check(self.lib.crh_scene_render_draws(self.handle, frame.handle, t.ctypes.data_as(fp), c.ctypes.data_as(fp), len(t),
                                              table.ctypes.data_as(C.POINTER(_ffi.DrawC)), len(table)))

    def set_dynamic_stroke_options(self, shape_index, group_index, options):
        c = options.to_c()
        check(self.lib.crh_scene_set_dynamic_stroke_options(self.handle, shape_index, group_index, C.byref(c)))

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.crh_scene_destroy(self.handle)
            self.handle = None


class Shape(Scene):
    """contrast_renderer::renderer::Shape — `Shape.from_paths` mirrors renderer.rs:177-249 (synchronous, raises ContrastError)."""

    @staticmethod
    def from_paths(renderer: Renderer, dynamic_stroke_options, paths, existing_shape: "Shape" = None):
        batch = batch_from_shapes([(list(dynamic_stroke_options), list(paths))])
        shape = Shape.__new__(Shape)
        Scene.__init__(shape, renderer, batch, tessellate=True, existing=existing_shape)
        shape.check()
        return shape

    def buffers(self):
        return self.shape(0)

    def render_in(self, render_pass: "RenderPass", instance_indices, render_operation):
        """Shape::render(&self, &renderer, &mut render_pass, instance_indices, render_operation), renderer.rs:267-273."""
        render_pass.render(self, instance_indices, render_operation, 0)

    def render_instance(self, frame: Frame, transform, color):
        """render(Stencil) followed by render(Color) for one instance."""
        self.render(frame, np.asarray(transform, dtype=np.float32).reshape(1, 16), np.asarray(color, dtype=np.float32).reshape(1, 4))


class RenderPass:
    """wgpu::RenderPass stand-in: records `Shape::render(&renderer, &mut render_pass, instance_indices, operation)` calls (renderer.rs:267-273)
    with the pass state they see — the stencil reference of Renderer::set_clip_depth (renderer.rs:932-938), the layer of save_ / restore_alpha_context
    (renderer.rs:941-985) — and submits them in order. The Shapes may be different objects (a Shape, a Scene): the frame keeps clip nesting counters,
    winding counters, saved alphas and sample colours between the submissions (crh_frame: `carry`), so the documented pattern works as it does in
    the reference (renderer.rs:257-266):

        a.render(pass, range(0, 1), Op.Stencil); pass.set_clip_depth(1); a.render(pass, range(0, 1), Op.Clip)
        b.render(pass, ...Stencil); b.render(pass, ...Color)        # another Shape object, clipped by a
        pass.set_clip_depth(0); a.render(pass, range(0, 1), Op.UnClip)
    """

    def __init__(self, renderer: Renderer, frame: Frame):
        self.renderer, self.frame = renderer, frame
        self.transforms, self.colors, self.draws = [], [], []  # draws: (scene, shape, instance, op, clip_depth, alpha_layer)
        self.clip_depth = self.alpha_layer = 0

    def push_instance(self, transform, color):
        """Instance data of the pass (the instance buffers bound at slots 0 / 2, renderer.rs:462-466): returns the instance index."""
        self.transforms.append(np.asarray(transform, dtype=np.float32).reshape(16))
        self.colors.append(np.asarray(color, dtype=np.float32).reshape(4))
        return len(self.colors) - 1

    def set_clip_depth(self, clip_depth):  # Renderer::set_clip_depth, renderer.rs:932-938
        if clip_depth >= (1 << self.renderer.config.clip_nesting_counter_bits):
            raise ContrastError(_ffi.ERR_CLIP_STACK_OVERFLOW, "ClipStackOverflow")
        self.clip_depth = int(clip_depth)

    def set_alpha_layer(self, alpha_layer):  # the layer save_alpha_context / restore_alpha_context bind, renderer.rs:941-985
        if alpha_layer >= self.renderer.config.alpha_layer_count:
            raise ContrastError(_ffi.ERR_TOO_MANY_NESTED_OPACITY_GROUPS, "TooManyNestedOpacityGroups")
        self.alpha_layer = int(alpha_layer)

    def render(self, scene: "Scene", instance_indices, operation, shape_index=0):
        for i in instance_indices:
            self.draws.append((scene, int(shape_index), int(i), int(operation), self.clip_depth, self.alpha_layer))

    def submit(self):
        """End of the pass: everything recorded executes in order, one crh_scene_render_draws per run of draws of the same Scene object."""
        if any(d[0] is not self.draws[0][0] for d in self.draws):
            self.frame.keep_pass_state()  # the pass spans objects: every sample's colour and stencil stay with the frame from its first draw on
        begin = 0
        while begin < len(self.draws):
            end = begin
            while end < len(self.draws) and self.draws[end][0] is self.draws[begin][0]:
                end += 1
            self.draws[begin][0].render_draws(self.frame, np.stack(self.transforms), np.stack(self.colors), [d[1:] for d in self.draws[begin:end]])
            begin = end
        self.draws = []

