"""Host-side mirror of contrast_renderer::text (text.rs): Font, Orientation, Alignment, Layout, paths_of_glyph, paths_of_text,
TextGeometry.new — thin ctypes calls into the native glyph producer of libcontrast_hip.so (csrc/text.cpp). No font parsing or
layout arithmetic happens in Python."""
import ctypes as C
import enum
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from .path import Path, SegmentType

_FLOATS = (2, 4, 6, 5, 10)


class Orientation(enum.IntEnum):  # text.rs:106-117
    RightToLeft = 0
    LeftToRight = 1
    TopToBottom = 2
    BottomToTop = 3


class Alignment(enum.IntEnum):  # text.rs:119-131
    Begin = 0
    Baseline = 1
    Center = 2
    End = 3


@dataclass
class Layout:  # text.rs:133-143
    size: float
    orientation: Orientation = Orientation.LeftToRight
    major_alignment: Alignment = Alignment.Begin
    minor_alignment: Alignment = Alignment.Baseline

    def to_c(self):
        return _ffi.TextLayoutC(float(self.size), int(self.orientation), int(self.major_alignment), int(self.minor_alignment))


def _check(status):
    if status != _ffi.OK:
        detail = (_ffi.load_library().crh_last_error() or b"").decode()
        raise _ffi.ContrastError(status, detail)


def _chars(text) -> np.ndarray:
    if isinstance(text, str):
        return np.asarray([ord(c) for c in text], dtype=np.uint32)
    return np.ascontiguousarray(text, dtype=np.uint32)


class PathList:
    """Vec<Path> owned by the native side."""

    def __init__(self, handle):
        self.lib = _ffi.load_library()
        self.handle = handle

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.crh_path_list_destroy(self.handle)
            self.handle = None

    def transform(self, scale: float, motor: Sequence[float]):  # Path::transform on every path, path.rs:387-439
        m = (C.c_float * 4)(*[float(v) for v in motor])
        _check(self.lib.crh_path_list_transform(self.handle, float(scale), m))
        return self

    def arrays(self):
        """-> (path_segment_begin[n_paths + 1], path_start[n_paths, 2], segment_types[n_segments], control_data) as numpy copies."""
        view = _ffi.PathBatchC()
        _check(self.lib.crh_path_list_view(self.handle, C.byref(view)))
        n_paths, n_seg, n_ctl = view.n_paths, view.n_segments, view.n_control_floats
        seg_begin = np.ctypeslib.as_array(view.path_segment_begin, (n_paths + 1,)).copy()
        start = np.ctypeslib.as_array(view.path_start, (n_paths, 2)).copy() if n_paths else np.zeros((0, 2), np.float32)
        types = np.ctypeslib.as_array(view.segment_types, (n_seg,)).copy() if n_seg else np.zeros(0, np.uint8)
        control = np.ctypeslib.as_array(view.control_data, (n_ctl,)).copy() if n_ctl else np.zeros(0, np.float32)
        return seg_begin, start, types, control

    def to_paths(self) -> List[Path]:
        seg_begin, start, types, control = self.arrays()
        offsets = np.concatenate([[0], np.cumsum(np.asarray(_FLOATS)[types])]).astype(np.int64)
        out = []
        for p in range(len(start)):
            path = Path(start=(float(start[p, 0]), float(start[p, 1])))
            for s in range(seg_begin[p], seg_begin[p + 1]):
                path.segment_types.append(SegmentType(int(types[s])))
                path.records.append(tuple(float(v) for v in control[offsets[s]:offsets[s + 1]]))
            out.append(path)
        return out


class Font:
    """text.rs:11-38. `face()` of the reference returns the ttf_parser::Face; here the Font is the face."""

    def __init__(self, name: str, font_data: bytes):
        self.lib = _ffi.load_library()
        self._name = name
        handle = C.c_void_p()
        buf = (C.c_uint8 * len(font_data)).from_buffer_copy(font_data)
        _check(self.lib.crh_font_create(buf, len(font_data), C.byref(handle)))
        self.handle = handle
        m = _ffi.FontMetricsC()
        _check(self.lib.crh_font_get_metrics(self.handle, C.byref(m)))
        self.metrics = m

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.crh_font_destroy(self.handle)
            self.handle = None

    def name(self):
        return self._name

    def face(self):
        return self

    # ---- the ttf_parser::Face calls text.rs makes
    def units_per_em(self): return int(self.metrics.units_per_em)
    def number_of_glyphs(self): return int(self.metrics.number_of_glyphs)
    def ascender(self): return int(self.metrics.ascender)
    def descender(self): return int(self.metrics.descender)
    def line_gap(self): return int(self.metrics.line_gap)
    def height(self): return int(self.metrics.height)
    def x_height(self): return int(self.metrics.x_height) if self.metrics.has_x_height else None
    def vertical_height(self): return int(self.metrics.vertical_height) if self.metrics.has_vertical_metrics else None
    def vertical_line_gap(self): return int(self.metrics.vertical_line_gap) if self.metrics.has_vertical_metrics else None

    def glyph_index(self, char) -> Optional[int]:
        gid, found = C.c_uint16(), C.c_uint32()
        _check(self.lib.crh_font_glyph_index(self.handle, ord(char) if isinstance(char, str) else int(char), C.byref(gid), C.byref(found)))
        return int(gid.value) if found.value else None

    def glyph_hor_advance(self, glyph_id) -> Optional[int]:
        adv, found = C.c_uint16(), C.c_uint32()
        _check(self.lib.crh_font_glyph_advance(self.handle, int(glyph_id), 0, C.byref(adv), C.byref(found)))
        return int(adv.value) if found.value else None

    def glyph_ver_advance(self, glyph_id) -> Optional[int]:
        adv, found = C.c_uint16(), C.c_uint32()
        _check(self.lib.crh_font_glyph_advance(self.handle, int(glyph_id), 1, C.byref(adv), C.byref(found)))
        return int(adv.value) if found.value else None

    def glyph_bounding_box(self, glyph_id) -> Optional[Tuple[int, int, int, int]]:
        box, found = (C.c_int16 * 4)(), C.c_uint32()
        _check(self.lib.crh_font_glyph_bounding_box(self.handle, int(glyph_id), box, C.byref(found)))
        return tuple(int(v) for v in box) if found.value else None

    def glyphs_kerning(self, left, right) -> Optional[int]:
        k, found = C.c_int16(), C.c_uint32()
        _check(self.lib.crh_font_glyphs_kerning(self.handle, int(left), int(right), C.byref(k), C.byref(found)))
        return int(k.value) if found.value else None


def glyph_path_list(face: Font, glyph_id: int) -> PathList:
    handle = C.c_void_p()
    _check(face.lib.crh_paths_of_glyph(face.handle, int(glyph_id), C.byref(handle)))
    return PathList(handle)


def paths_of_glyph(face: Font, glyph_id: int) -> List[Path]:  # text.rs:97-104
    return glyph_path_list(face, glyph_id).to_paths()


def text_path_list(face: Font, layout: Layout, text, clipping_area=None) -> PathList:
    chars = _chars(text)
    handle = C.c_void_p()
    lay = layout.to_c()
    clip_ptr, n_clip = None, 0
    if clipping_area is not None:
        clip = np.ascontiguousarray(clipping_area, dtype=np.float32).reshape(-1, 2)
        clip_ptr, n_clip = clip.ctypes.data_as(C.POINTER(C.c_float)), len(clip)
    _check(face.lib.crh_paths_of_text(face.handle, C.byref(lay), chars.ctypes.data_as(C.POINTER(C.c_uint32)), len(chars), clip_ptr, n_clip, C.byref(handle)))
    return PathList(handle)


def paths_of_text(face: Font, layout: Layout, text, clipping_area=None) -> List[Path]:  # text.rs:236-263
    return text_path_list(face, layout, text, clipping_area).to_paths()


def calculate_aligned_positions(face: Font, layout: Layout, text):
    """calculate_aligned_positions!, text.rs:145-230 -> (extent[2], offset[2], lines) with lines = [(line_range_end, [((x, y), glyph_id), ...])],
    integer font units."""
    chars = _chars(text)
    lay = layout.to_c()
    n_lines = C.c_uint64()
    ptr = chars.ctypes.data_as(C.POINTER(C.c_uint32))
    _check(face.lib.crh_text_aligned_positions(face.handle, C.byref(lay), ptr, len(chars), None, None, None, None, None, C.byref(n_lines)))
    extent, offset = (C.c_int64 * 2)(), (C.c_int64 * 2)()
    positions = np.zeros((len(chars) + 1, 3), dtype=np.int64)
    ends = np.zeros(n_lines.value, dtype=np.uint64)
    lengths = np.zeros(n_lines.value, dtype=np.uint64)
    _check(face.lib.crh_text_aligned_positions(face.handle, C.byref(lay), ptr, len(chars), extent, offset, positions.ctypes.data_as(C.POINTER(C.c_int64)),
                                               ends.ctypes.data_as(C.POINTER(C.c_uint64)), lengths.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(n_lines)))
    lines, at = [], 0
    for l in range(n_lines.value):
        n = int(lengths[l])
        lines.append((int(ends[l]), [((int(positions[i, 0]), int(positions[i, 1])), int(positions[i, 2])) for i in range(at, at + n)]))
        at += n
    return [int(extent[0]), int(extent[1])], [int(offset[0]), int(offset[1])], lines


@dataclass
class TextGeometry:  # text.rs:266-347
    major_axis: int
    half_extent: Tuple[float, float]
    lines: list

    @staticmethod
    def new(face: Font, layout: Layout, text):
        f = np.float32
        major_axis = 0 if layout.orientation in (Orientation.RightToLeft, Orientation.LeftToRight) else 1
        scale = f(layout.size) / f(face.height())
        extent, offset, lines = calculate_aligned_positions(face, layout, text)
        half = (float(f(extent[0]) * scale * f(0.5)), float(f(extent[1]) * scale * f(0.5)))
        out = [(end, [(float(f(p[0] - offset[0]) * scale), float(f(p[1] - offset[1]) * scale)) for p, _ in glyphs]) for end, glyphs in lines]
        return TextGeometry(major_axis, half, out)

    def line_index_from_char_index(self, char_index: int) -> int:  # text.rs:310-315 (panics when there is no such line: IndexError here)
        for index, (line_range_end, _) in enumerate(self.lines):
            if line_range_end > char_index:
                return index
        raise IndexError("char_index is beyond the last line")

    def char_index_from_position(self, cursor) -> int:  # text.rs:318-331
        f = np.float32
        minor = 1 - self.major_axis
        minor_half_extent = f(self.half_extent[minor])
        with np.errstate(invalid="ignore", divide="ignore"):
            v = (minor_half_extent - f(cursor[minor])) * f(len(self.lines)) / (minor_half_extent * f(2.0))
        v = min(max(float(v), 0.0), float(len(self.lines) - 1)) if v == v else 0.0  # f32::max / min ignore NaN; `as usize` saturates
        line_index = int(v)
        glyph_positions = self.lines[line_index][1]
        found = len(glyph_positions) - 1
        for i in range(len(glyph_positions) - 1):
            if (f(glyph_positions[i][self.major_axis]) + f(glyph_positions[i + 1][self.major_axis])) * f(0.5) > f(cursor[self.major_axis]):
                found = i
                break
        return found + (0 if line_index == 0 else self.lines[line_index - 1][0])

    def advance_char_index_by_line_index(self, char_index: int, relative_line_index: int) -> int:  # text.rs:334-346
        f = np.float32
        line_index = self.line_index_from_char_index(char_index)
        if relative_line_index < 0 and line_index == 0:
            return 0
        if relative_line_index > 0 and line_index == len(self.lines) - 1:
            return self.lines[-1][0] - 1
        line_range_end, glyph_positions = self.lines[line_index]
        cursor = list(glyph_positions[char_index + len(glyph_positions) - line_range_end])
        minor = 1 - self.major_axis
        line_minor_extent = f(self.half_extent[minor]) * f(2.0) / f(len(self.lines))
        cursor[minor] = float(f(cursor[minor]) - line_minor_extent * f(relative_line_index))
        return self.char_index_from_position(cursor)


def byte_offset_of_char_index(string: str, char_index: int) -> int:  # text.rs:350-352
    return len(string[:char_index].encode("utf-8")) if char_index < len(string) else len(string.encode("utf-8"))
