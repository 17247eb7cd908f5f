"""ctypes mirror of include/contrast_hip.h (structs, enums) and the loader of libcontrast_hip.so.

Plumbing only: the product is the HIP library behind the C ABI. Nothing here computes geometry, and
nothing here may import `oracle` (the CPU oracle is test infrastructure).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcontrast_hip.so")

# ---- status codes (error.rs:5-16 + panics of the reference surfaced as codes) -------------------------------
OK = 0
ERR_NUMBER_OF_STENCIL_BITS_IS_UNSUPPORTED = 1
ERR_CLIP_STACK_OVERFLOW = 2
ERR_TOO_MANY_NESTED_OPACITY_GROUPS = 3
ERR_TOO_MANY_DASH_INTERVALS = 4
ERR_DYNAMIC_STROKE_OPTIONS_INDEX_OUT_OF_BOUNDS = 5
ERR_NON_FINITE = 6
ERR_DEGENERATE_CUBIC = 7
ERR_UNSUPPORTED = 8
ERR_HIP = 9
ERR_INVALID_ARGUMENT = 10

STATUS_NAMES = {
    0: "Ok",
    1: "NumberOfStencilBitsIsUnsupported",
    2: "ClipStackOverflow",
    3: "TooManyNestedOpacityGroups",
    4: "TooManyDashIntervals",
    5: "DynamicStrokeOptionsIndexOutOfBounds",
    6: "NonFinite (the reference panics: safe_float.rs:46,114)",
    7: "DegenerateCubic (the reference panics: fill.rs:174,178)",
    8: "Unsupported",
    9: "HipError",
    10: "InvalidArgument",
}

SEGMENT_LINE, SEGMENT_INTEGRAL_QUADRATIC, SEGMENT_INTEGRAL_CUBIC, SEGMENT_RATIONAL_QUADRATIC, SEGMENT_RATIONAL_CUBIC = range(5)
SEGMENT_FLOATS = (2, 4, 6, 5, 10)
MAX_DASH_INTERVALS = 4


class StrokeOptionsC(C.Structure):
    _fields_ = [
        ("width", C.c_float),
        ("offset", C.c_float),
        ("miter_clip", C.c_float),
        ("closed", C.c_uint32),
        ("dynamic_stroke_options_group", C.c_uint32),
        ("curve_approximation", C.c_uint32),
        ("steps", C.c_uint32),
        ("angle_step", C.c_float),
    ]


class DashIntervalC(C.Structure):
    _fields_ = [("gap_start", C.c_float), ("gap_end", C.c_float), ("dash_start", C.c_uint32), ("dash_end", C.c_uint32)]


class DynamicStrokeOptionsC(C.Structure):
    _fields_ = [
        ("dashed", C.c_uint32),
        ("join", C.c_uint32),
        ("pattern_len", C.c_uint32),
        ("pattern", DashIntervalC * MAX_DASH_INTERVALS),
        ("phase", C.c_float),
        ("start", C.c_uint32),
        ("end", C.c_uint32),
    ]


class DynamicStrokeDescriptorC(C.Structure):
    _fields_ = [
        ("gap_start", C.c_float * MAX_DASH_INTERVALS),
        ("gap_end", C.c_float * MAX_DASH_INTERVALS),
        ("caps", C.c_uint32),
        ("count_dashed_join", C.c_uint32),
        ("phase", C.c_float),
        ("_padding", C.c_uint32),
    ]


class PathBatchC(C.Structure):
    _fields_ = [
        ("n_shapes", C.c_uint32),
        ("shape_path_begin", C.POINTER(C.c_uint32)),
        ("n_paths", C.c_uint32),
        ("path_segment_begin", C.POINTER(C.c_uint32)),
        ("path_start", C.POINTER(C.c_float)),
        ("path_stroke_options", C.POINTER(C.c_int32)),
        ("n_segments", C.c_uint32),
        ("segment_types", C.POINTER(C.c_uint8)),
        ("control_data", C.POINTER(C.c_float)),
        ("n_control_floats", C.c_uint32),
        ("n_stroke_options", C.c_uint32),
        ("stroke_options", C.POINTER(StrokeOptionsC)),
        ("shape_dynamic_begin", C.POINTER(C.c_uint32)),
        ("n_dynamic_stroke_options", C.c_uint32),
        ("dynamic_stroke_options", C.POINTER(DynamicStrokeOptionsC)),
    ]


class ConfigC(C.Structure):
    _fields_ = [
        ("msaa_sample_count", C.c_uint32),
        ("clip_nesting_counter_bits", C.c_uint32),
        ("winding_counter_bits", C.c_uint32),
        ("alpha_layer_count", C.c_uint32),
        ("cull_mode", C.c_uint32),
        ("depth_compare", C.c_uint32),
        ("depth_write_enabled", C.c_uint32),
    ]


class KernelTimeC(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_float), ("algorithmic_bytes", C.c_uint64)]


def _ptr(array, ctype):
    return array.ctypes.data_as(C.POINTER(ctype))


class PathBatch:
    """Struct-of-arrays batch of Shapes (see crh_path_batch). Owns numpy arrays; `.c` is the C view."""

    def __init__(self, shape_path_begin, path_segment_begin, path_start, path_stroke_options, segment_types, control_data, stroke_options=(),
                 shape_dynamic_begin=None, dynamic_stroke_options=()):
        self.shape_path_begin = np.ascontiguousarray(shape_path_begin, dtype=np.uint32)
        self.path_segment_begin = np.ascontiguousarray(path_segment_begin, dtype=np.uint32)
        self.path_start = np.ascontiguousarray(path_start, dtype=np.float32).reshape(-1, 2)
        self.path_stroke_options = np.ascontiguousarray(path_stroke_options, dtype=np.int32)
        self.segment_types = np.ascontiguousarray(segment_types, dtype=np.uint8)
        self.control_data = np.ascontiguousarray(control_data, dtype=np.float32).reshape(-1)
        self.n_shapes = len(self.shape_path_begin) - 1
        self.n_paths = len(self.path_segment_begin) - 1
        self.n_segments = len(self.segment_types)
        assert self.path_start.shape[0] == self.n_paths and len(self.path_stroke_options) == self.n_paths
        assert int(self.shape_path_begin[-1]) == self.n_paths and int(self.path_segment_begin[-1]) == self.n_segments
        sizes = np.asarray(SEGMENT_FLOATS, dtype=np.int64)[self.segment_types]
        assert int(sizes.sum()) == self.control_data.size, "control_data does not match segment_types"
        self.stroke_options = (StrokeOptionsC * max(1, len(stroke_options)))(*stroke_options)
        self.n_stroke_options = len(stroke_options)
        if shape_dynamic_begin is None:
            shape_dynamic_begin = np.zeros(self.n_shapes + 1, dtype=np.uint32)
        self.shape_dynamic_begin = np.ascontiguousarray(shape_dynamic_begin, dtype=np.uint32)
        self.dynamic_stroke_options = (DynamicStrokeOptionsC * max(1, len(dynamic_stroke_options)))(*dynamic_stroke_options)
        self.n_dynamic_stroke_options = len(dynamic_stroke_options)
        c = PathBatchC()
        c.n_shapes = self.n_shapes
        c.shape_path_begin = _ptr(self.shape_path_begin, C.c_uint32)
        c.n_paths = self.n_paths
        c.path_segment_begin = _ptr(self.path_segment_begin, C.c_uint32)
        c.path_start = _ptr(self.path_start, C.c_float)
        c.path_stroke_options = _ptr(self.path_stroke_options, C.c_int32)
        c.n_segments = self.n_segments
        c.segment_types = _ptr(self.segment_types, C.c_uint8)
        c.control_data = _ptr(self.control_data, C.c_float)
        c.n_control_floats = self.control_data.size
        c.n_stroke_options = self.n_stroke_options
        c.stroke_options = C.cast(self.stroke_options, C.POINTER(StrokeOptionsC))
        c.shape_dynamic_begin = _ptr(self.shape_dynamic_begin, C.c_uint32)
        c.n_dynamic_stroke_options = self.n_dynamic_stroke_options
        c.dynamic_stroke_options = C.cast(self.dynamic_stroke_options, C.POINTER(DynamicStrokeOptionsC))
        self.c = c

    def input_bytes(self):
        """Algorithmic input bytes of SURVEY.md §8(d): control bytes + 1 type byte per segment, 8 B start (+ 32 B options) per path."""
        stroked = int((self.path_stroke_options >= 0).sum())
        return self.control_data.nbytes + self.n_segments + 8 * self.n_paths + 32 * stroked

    def slice_shapes(self, begin, end):
        """The contiguous shape range [begin, end) as its own batch (path-index sharding)."""
        p0, p1 = int(self.shape_path_begin[begin]), int(self.shape_path_begin[end])
        s0, s1 = int(self.path_segment_begin[p0]), int(self.path_segment_begin[p1])
        sizes = np.asarray(SEGMENT_FLOATS, dtype=np.int64)[self.segment_types]
        offsets = np.concatenate([[0], np.cumsum(sizes)])
        d0, d1 = int(self.shape_dynamic_begin[begin]), int(self.shape_dynamic_begin[end])
        return PathBatch(
            self.shape_path_begin[begin:end + 1] - p0,
            self.path_segment_begin[p0:p1 + 1] - s0,
            self.path_start[p0:p1],
            self.path_stroke_options[p0:p1],
            self.segment_types[s0:s1],
            self.control_data[int(offsets[s0]):int(offsets[s1])],
            [self.stroke_options[i] for i in range(self.n_stroke_options)],
            self.shape_dynamic_begin[begin:end + 1] - d0,
            [self.dynamic_stroke_options[i] for i in range(d0, d1)],
        )


class DrawC(C.Structure):
    _fields_ = [("shape", C.c_uint32), ("instance", C.c_uint32), ("op", C.c_uint32), ("clip_depth", C.c_uint32), ("alpha_layer", C.c_uint32)]


class FontMetricsC(C.Structure):
    _fields_ = [("units_per_em", C.c_uint32), ("number_of_glyphs", C.c_uint32), ("ascender", C.c_int32), ("descender", C.c_int32),
                ("line_gap", C.c_int32), ("height", C.c_int32), ("has_x_height", C.c_int32), ("x_height", C.c_int32),
                ("has_vertical_metrics", C.c_int32), ("vertical_height", C.c_int32), ("vertical_line_gap", C.c_int32), ("has_kerning", C.c_int32)]


class TextLayoutC(C.Structure):
    _fields_ = [("size", C.c_float), ("orientation", C.c_uint32), ("major_alignment", C.c_uint32), ("minor_alignment", C.c_uint32)]


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so. If this library (linked against the system ROCm) is loaded first and torch
    afterwards, the process holds two HIP runtimes and the second one sees no device. Importing torch first (when it is installed) makes
    the dynamic loader bind libcontrast_hip.so to the runtime torch uses — which bench.py needs anyway to hand frames to RCCL."""
    import importlib.util
    import sys
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def load_library():
    """Loads the HIP library. Fails loudly when it is missing: there is no CPU fallback in the product."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). contrast_renderer_amd has no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    V = C.c_void_p
    sig = {
        "crh_renderer_create": (C.c_int, [C.POINTER(ConfigC), C.c_int, C.POINTER(V)]),
        "crh_renderer_destroy": (None, [V]),
        "crh_renderer_get_config": (C.c_int, [V, C.POINTER(ConfigC)]),
        "crh_convert_dynamic_stroke_options": (C.c_int, [C.POINTER(DynamicStrokeOptionsC), C.POINTER(DynamicStrokeDescriptorC)]),
        "crh_scene_upload": (C.c_int, [V, C.POINTER(PathBatchC), V, C.POINTER(V)]),
        "crh_scene_tessellate": (C.c_int, [V]),
        "crh_scene_status": (C.c_int, [V]),
        "crh_scene_destroy": (None, [V]),
        "crh_shape_from_paths": (C.c_int, [V, C.POINTER(PathBatchC), V, C.POINTER(V)]),
        "crh_scene_shape_layout": (C.c_int, [V, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "crh_scene_shape_download": (C.c_int, [V, C.c_uint32, V, V]),
        "crh_scene_layout_all": (C.c_int, [V, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "crh_scene_download_all": (C.c_int, [V, V, V]),
        "crh_scene_traffic": (C.c_int, [V, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "crh_scene_set_dynamic_stroke_options": (C.c_int, [V, C.c_uint32, C.c_uint32, C.POINTER(DynamicStrokeOptionsC)]),
        "crh_frame_create": (C.c_int, [V, C.c_uint32, C.c_uint32, C.POINTER(V)]),
        "crh_frame_create_format": (C.c_int, [V, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(V)]),
        "crh_frame_format": (C.c_int, [V, C.POINTER(C.c_uint32)]),
        "crh_frame_destroy": (None, [V]),
        "crh_frame_clear": (C.c_int, [V]),
        "crh_frame_keep_pass_state": (C.c_int, [V]),
        "crh_frame_synchronize": (C.c_int, [V]),
        "crh_frame_set_tile_rows": (C.c_int, [V, C.c_uint32, C.c_uint32]),
        "crh_frame_clear_depth": (C.c_int, [V, C.c_float]),
        "crh_frame_upload_depth": (C.c_int, [V, C.POINTER(C.c_float)]),
        "crh_frame_download_depth": (C.c_int, [V, C.POINTER(C.c_float)]),
        "crh_scene_render": (C.c_int, [V, V, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "crh_scene_set_instances": (C.c_int, [V, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "crh_scene_render_resident": (C.c_int, [V, V]),
        "crh_scene_render_draws": (C.c_int, [V, V, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.POINTER(DrawC), C.c_uint32]),
        "crh_frame_download": (C.c_int, [V, V]),
        "crh_frame_download_f16": (C.c_int, [V, V]),
        "crh_frame_device_pointer": (C.c_int, [V, C.POINTER(V)]),
        "crh_composite_over": (C.c_int, [V, C.POINTER(V), C.c_uint32, C.c_uint64, V]),
        "crh_comm_shard": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "crh_comm_slab_rows": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "crh_comm_unique_id": (C.c_int, [V]),
        "crh_comm_create": (C.c_int, [V, C.c_uint32, C.c_uint32, V, C.POINTER(V)]),
        "crh_comm_create_local": (C.c_int, [V, C.c_uint32, C.c_uint32, V, C.POINTER(V)]),
        "crh_comm_destroy": (None, [V]),
        "crh_frame_exchange": (C.c_int, [V, V, V]),
        "crh_frame_gather_slabs": (C.c_int, [V, V, V]),
        "crh_comm_local_exchange": (C.c_int, [V, C.POINTER(V), V]),
        "crh_comm_local_gather_slabs": (C.c_int, [V, C.POINTER(V), V]),
        "crh_comm_last_traffic": (C.c_int, [V, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "crh_comm_last_timing": (C.c_int, [V, C.POINTER(C.c_float)]),
        "crh_comm_last_peer_bytes": (C.c_int, [V, C.POINTER(C.c_uint64)]),
        "crh_comm_info": (C.c_int, [V, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]),
        "crh_renderer_synchronize": (C.c_int, [V]),
        "crh_renderer_stream": (V, [V]),
        "crh_renderer_enable_timing": (C.c_int, [V, C.c_int]),
        "crh_renderer_kernel_times": (C.c_int, [V, C.POINTER(KernelTimeC), C.c_uint32, C.POINTER(C.c_uint32)]),
        "crh_selftest_fmath": (C.c_int, [V, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint64]),
        "crh_font_create": (C.c_int, [V, C.c_size_t, C.POINTER(V)]),
        "crh_font_destroy": (None, [V]),
        "crh_font_get_metrics": (C.c_int, [V, C.POINTER(FontMetricsC)]),
        "crh_font_glyph_index": (C.c_int, [V, C.c_uint32, C.POINTER(C.c_uint16), C.POINTER(C.c_uint32)]),
        "crh_font_glyph_advance": (C.c_int, [V, C.c_uint16, C.c_uint32, C.POINTER(C.c_uint16), C.POINTER(C.c_uint32)]),
        "crh_font_glyph_bounding_box": (C.c_int, [V, C.c_uint16, C.POINTER(C.c_int16), C.POINTER(C.c_uint32)]),
        "crh_font_glyphs_kerning": (C.c_int, [V, C.c_uint16, C.c_uint16, C.POINTER(C.c_int16), C.POINTER(C.c_uint32)]),
        "crh_paths_of_glyph": (C.c_int, [V, C.c_uint16, C.POINTER(V)]),
        "crh_paths_of_text": (C.c_int, [V, C.POINTER(TextLayoutC), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_float), C.c_size_t, C.POINTER(V)]),
        "crh_text_aligned_positions": (C.c_int, [V, C.POINTER(TextLayoutC), C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                                 C.POINTER(C.c_int64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "crh_path_elliptical_arc": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_uint32, C.c_uint32, C.POINTER(C.c_float),
                                              C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "crh_path_list_transform": (C.c_int, [V, C.c_float, C.POINTER(C.c_float)]),
        "crh_path_list_view": (C.c_int, [V, C.POINTER(PathBatchC)]),
        "crh_path_list_destroy": (None, [V]),
        "crh_last_error": (C.c_char_p, []),
        "crh_version": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here == a symbol of include/contrast_hip.h is not exported
        fn.restype = res
        fn.argtypes = args
    lib._crh_signatures = sig
    _lib = lib
    return lib


class ContrastError(RuntimeError):
    """Mirrors contrast_renderer::error::Error (error.rs:5-16) plus the panics surfaced as codes."""

    def __init__(self, status, detail=""):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}{(': ' + detail) if detail else ''}")


def check(status):
    if status != OK:
        detail = ""
        if status == ERR_HIP and _lib is not None:
            detail = (_lib.crh_last_error() or b"").decode()
        raise ContrastError(status, detail)
