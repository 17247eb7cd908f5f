"""Host-side mirror of the reference's Path data model (src/path.rs:15-230) — same names, same meaning.

The data model, the constructors (polygon, rect, rounded rect, ellipse, circle, quarter ellipse, elliptical arc) and the conversions
(reverse, integral -> rational, quadratic -> cubic, close, append, end tangents) of path.rs. f32 arithmetic in the reference's operation
order; the elliptical arc is computed natively (csrc/path.cpp) so that this mirror and the C++ one agree to the bit.
"""
import math
from dataclasses import dataclass, field
from enum import IntEnum
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _ffi


class SegmentType(IntEnum):  # path.rs:56-67
    Line = 0
    IntegralQuadraticCurve = 1
    IntegralCubicCurve = 2
    RationalQuadraticCurve = 3
    RationalCubicCurve = 4


class Join(IntEnum):  # path.rs:71-82
    Miter = 0
    Bevel = 1
    Round = 2


class Cap(IntEnum):  # path.rs:86-101
    Square = 0
    Round = 1
    Out = 2
    In = 3
    Right = 4
    Left = 5
    Butt = 6


def safe_float(x) -> float:
    """SafeFloat::from (safe_float.rs:44-52): finite or panic; -0.0 becomes +0.0."""
    x = float(np.float32(x))
    if not math.isfinite(x):
        raise ValueError("SafeFloat: value is not finite (safe_float.rs:46)")
    return 0.0 if x == 0.0 else x


def _vec2(v):
    return (safe_float(v[0]), safe_float(v[1]))


@dataclass
class DashInterval:  # path.rs:105-118
    gap_start: float
    gap_end: float
    dash_start: Cap
    dash_end: Cap


@dataclass
class DynamicStrokeOptions:  # path.rs:127-149
    """Either Dashed{join, pattern, phase} or Solid{join, start, end}."""
    join: Join = Join.Miter
    pattern: Optional[List[DashInterval]] = None  # not None -> Dashed
    phase: float = 0.0
    start: Cap = Cap.Butt
    end: Cap = Cap.Butt

    @staticmethod
    def Dashed(join, pattern, phase):
        return DynamicStrokeOptions(join=join, pattern=list(pattern), phase=phase)

    @staticmethod
    def Solid(join, start, end):
        return DynamicStrokeOptions(join=join, start=start, end=end)

    def to_c(self) -> _ffi.DynamicStrokeOptionsC:
        c = _ffi.DynamicStrokeOptionsC()
        c.join = int(self.join)
        if self.pattern is not None:
            c.dashed = 1
            c.pattern_len = len(self.pattern)
            for i, interval in enumerate(self.pattern[:_ffi.MAX_DASH_INTERVALS]):
                c.pattern[i].gap_start = safe_float(interval.gap_start)
                c.pattern[i].gap_end = safe_float(interval.gap_end)
                c.pattern[i].dash_start = int(interval.dash_start)
                c.pattern[i].dash_end = int(interval.dash_end)
            c.phase = safe_float(self.phase)
        else:
            c.dashed = 0
            c.start = int(self.start)
            c.end = int(self.end)
        return c


@dataclass
class CurveApproximation:  # path.rs:153-167
    uniform_tangent_angle: bool
    value: Union[int, float]

    @staticmethod
    def UniformlySpacedParameters(steps: int):
        return CurveApproximation(False, int(steps))

    @staticmethod
    def UniformTangentAngle(angle: float):
        return CurveApproximation(True, safe_float(angle))


@dataclass
class StrokeOptions:  # path.rs:171-192
    width: float
    offset: float
    miter_clip: float
    closed: bool
    dynamic_stroke_options_group: int
    curve_approximation: CurveApproximation

    def legalize(self):  # path.rs:196-200
        self.width = abs(self.width)
        self.offset = min(max(self.offset, -0.5), 0.5)
        self.miter_clip = abs(self.miter_clip)

    def to_c(self) -> _ffi.StrokeOptionsC:
        c = _ffi.StrokeOptionsC()
        c.width = safe_float(self.width)
        c.offset = safe_float(self.offset)
        c.miter_clip = safe_float(self.miter_clip)
        c.closed = 1 if self.closed else 0
        c.dynamic_stroke_options_group = int(self.dynamic_stroke_options_group)
        if self.curve_approximation.uniform_tangent_angle:
            c.curve_approximation = 1
            c.angle_step = float(self.curve_approximation.value)
        else:
            c.curve_approximation = 0
            c.steps = int(self.curve_approximation.value)
        return c


@dataclass
class Path:  # path.rs:213-230 (the five typed Vecs + segment_types are kept as one interleaved record list)
    start: Tuple[float, float] = (0.0, 0.0)
    stroke_options: Optional[StrokeOptions] = None
    segment_types: List[int] = field(default_factory=list)
    records: List[Tuple[float, ...]] = field(default_factory=list)

    def __post_init__(self):
        self.start = _vec2(self.start)  # SafeFloat<f32, 2>: finite, f32, -0 -> +0

    def push_line(self, control_point):  # path.rs:234-237
        self.segment_types.append(SegmentType.Line)
        self.records.append(_vec2(control_point))

    def push_integral_quadratic_curve(self, c0, c1):  # path.rs:240-243
        self.segment_types.append(SegmentType.IntegralQuadraticCurve)
        self.records.append(_vec2(c0) + _vec2(c1))

    def push_integral_cubic_curve(self, c0, c1, c2):  # path.rs:246-249
        self.segment_types.append(SegmentType.IntegralCubicCurve)
        self.records.append(_vec2(c0) + _vec2(c1) + _vec2(c2))

    def push_rational_quadratic_curve(self, weight, c0, c1):  # path.rs:252-255
        self.segment_types.append(SegmentType.RationalQuadraticCurve)
        self.records.append((safe_float(weight),) + _vec2(c0) + _vec2(c1))

    def push_rational_cubic_curve(self, weights, c0, c1, c2):  # path.rs:258-261
        self.segment_types.append(SegmentType.RationalCubicCurve)
        self.records.append(tuple(safe_float(w) for w in weights) + _vec2(c0) + _vec2(c1) + _vec2(c2))

    def get_end(self):  # path.rs:266-290
        if not self.records:
            return self.start
        return self.records[-1][-2:]

    # ---- tangents (path.rs:296-372); Plane = (c, nx, ny), signum = multiply by 1 / sqrt(nx^2 + ny^2)
    def get_start_tangent(self):
        """path.rs:296-322. The reference looks at `segment_types.last()` and the last segment of that type — reproduced as written."""
        if not self.segment_types:
            return (0.0, 0.0, 0.0)
        last_type = self.segment_types[-1]
        record = [r for t, r in zip(self.segment_types, self.records) if t == last_type][-1]
        first_point = record[1:3] if last_type == SegmentType.RationalQuadraticCurve else (record[4:6] if last_type == SegmentType.RationalCubicCurve else record[0:2])
        return _signum(_tangent_from_points(self.start, first_point))

    def get_end_tangent(self):
        """path.rs:326-372."""
        if not self.segment_types:
            return (0.0, 0.0, 0.0)
        last_type, record = self.segment_types[-1], self.records[-1]
        if last_type == SegmentType.Line:
            previous = self.records[-2][-2:] if len(self.records) > 1 else self.start
            return _signum(_tangent_from_points(previous, record[0:2]))
        return _signum(_tangent_from_points(record[-4:-2], record[-2:]))

    def append(self, other: "Path"):
        """path.rs:376-384 moves the five typed Vecs of `other` but NOT its segment_types, so the appended segments are unreachable
        through segment_types. Reproduced: `other` is emptied, nothing reachable is added."""
        other.records = []
        other.segment_types = []

    def reverse(self):
        """path.rs:445-488: swaps start and end, reverses the segment order and every segment's direction."""
        f = np.float32
        previous = self.start
        new_records = []
        for t, rec in zip(self.segment_types, self.records):
            rec = list(rec)
            if t == SegmentType.IntegralCubicCurve:
                rec[0:4] = rec[2:4] + rec[0:2]
            elif t == SegmentType.RationalCubicCurve:
                rec[0:4] = rec[0:4][::-1]
                rec[4:8] = rec[6:8] + rec[4:6]
            end = tuple(rec[-2:])
            rec[-2:] = list(previous)
            previous = end
            new_records.append(tuple(float(f(v)) for v in rec))
        self.start = previous
        self.segment_types.reverse()
        new_records.reverse()
        self.records = new_records

    def convert_integral_curves_to_rational_curves(self):
        """path.rs:492-534: weight 1 for quadratics, weights [1, 1, 1, 1] for cubics."""
        for i, t in enumerate(self.segment_types):
            if t == SegmentType.IntegralQuadraticCurve:
                self.segment_types[i] = SegmentType.RationalQuadraticCurve
                self.records[i] = (1.0,) + tuple(self.records[i])
            elif t == SegmentType.IntegralCubicCurve:
                self.segment_types[i] = SegmentType.RationalCubicCurve
                self.records[i] = (1.0, 1.0, 1.0, 1.0) + tuple(self.records[i])

    def convert_quadratic_curves_to_cubic_curves(self):
        """path.rs:538-615: degree elevation; `(a - p) * 2.0 / 3.0` is ((a - p) * 2) / 3 in f32, the rational case works on
        homogeneous points with the f32 constant 2.0 / 3.0."""
        f = np.float32
        previous = (f(self.start[0]), f(self.start[1]))
        for i, (t, rec) in enumerate(zip(self.segment_types, self.records)):
            rec = [f(v) for v in rec]
            if t == SegmentType.IntegralQuadraticCurve:
                a, b = (rec[0], rec[1]), (rec[2], rec[3])
                c0 = tuple(previous[k] + (a[k] - previous[k]) * f(2.0) / f(3.0) for k in range(2))
                c1 = tuple(b[k] + (a[k] - b[k]) * f(2.0) / f(3.0) for k in range(2))
                self.segment_types[i] = SegmentType.IntegralCubicCurve
                self.records[i] = _vec2(c0) + _vec2(c1) + _vec2(b)
            elif t == SegmentType.RationalQuadraticCurve:
                w, a, b = rec[0], (rec[1], rec[2]), (rec[3], rec[4])
                p0, p1, p2 = (f(1.0), previous[0], previous[1]), (w, a[0] * w, a[1] * w), (f(1.0), b[0], b[1])
                two_thirds = f(2.0) / f(3.0)
                n0 = tuple(p0[k] + (p1[k] - p0[k]) * two_thirds for k in range(3))
                n1 = tuple(p2[k] + (p1[k] - p2[k]) * two_thirds for k in range(3))
                self.segment_types[i] = SegmentType.RationalCubicCurve
                self.records[i] = ((1.0, safe_float(n0[0]), safe_float(n1[0]), 1.0) + _vec2((n0[1] / n0[0], n0[2] / n0[0])) + _vec2((n1[1] / n1[0], n1[2] / n1[0])) + _vec2(b))
            previous = (f(self.records[i][-2]), f(self.records[i][-1]))

    def close(self):
        """path.rs:621-628: an explicit closing line unless the end is (within ERROR_MARGIN) the start."""
        t = _tangent_from_points(self.start, self.get_end())
        if np.float32(t[1]) * np.float32(t[1]) + np.float32(t[2]) * np.float32(t[2]) <= np.float32(1e-4):
            return
        self.push_line(self.start)

    def push_quarter_ellipse(self, tangent_crossing, to):  # path.rs:631-636
        self.push_rational_quadratic_curve(np.float32(0.70710678118654752440), tangent_crossing, to)

    def push_elliptical_arc(self, half_extent, rotation, large_arc, sweep, to):
        """path.rs:639-708, the SVG "arc to" command (native: crh_path_elliptical_arc)."""
        import ctypes as C
        lib = _ffi.load_library()
        fp = C.POINTER(C.c_float)
        start = (C.c_float * 2)(*self.get_end())
        half = (C.c_float * 2)(float(half_extent[0]), float(half_extent[1]))
        end = (C.c_float * 2)(float(to[0]), float(to[1]))
        records = (C.c_float * 20)()
        n, is_line = C.c_uint32(), C.c_uint32()
        _ffi.check(lib.crh_path_elliptical_arc(start, half, float(rotation), int(bool(large_arc)), int(bool(sweep)), end, records, 4, C.byref(n), C.byref(is_line)))
        if is_line.value:
            self.push_line(to)
            return
        for i in range(n.value):
            r = records[5 * i:5 * i + 5]
            self.push_rational_quadratic_curve(r[0], (r[1], r[2]), (r[3], r[4]))

    @staticmethod
    def from_polygon(vertices: Sequence[Sequence[float]]):  # path.rs:711-724
        path = Path(start=_vec2(vertices[0]))
        for v in vertices[1:]:
            path.push_line(v)
        return path

    @staticmethod
    def from_regular_polygon(center, radius, rotation, vertex_count):  # path.rs:727-734 (f32 arithmetic as in the reference)
        f = np.float32
        vertices = []
        for i in range(vertex_count):
            angle = f(rotation) + f(i) / f(vertex_count) * f(math.pi) * f(2.0)
            vertices.append((f(center[0]) + f(radius) * f(math.cos(angle)), f(center[1]) + f(radius) * f(math.sin(angle))))
        return Path.from_polygon(vertices)

    @staticmethod
    def from_rect(center, half_extent):  # path.rs:736-743
        f = np.float32
        cx, cy, hx, hy = f(center[0]), f(center[1]), f(half_extent[0]), f(half_extent[1])
        return Path.from_polygon([(cx - hx, cy - hy), (cx - hx, cy + hy), (cx + hx, cy + hy), (cx + hx, cy - hy)])


def _rounded_corner_path(start, corners):
    path = Path(start=_vec2(start))
    for corner in corners:
        if len(corner) == 3:
            path.push_line(corner[0])
            path.push_quarter_ellipse(corner[1], corner[2])
        else:
            path.push_quarter_ellipse(corner[0], corner[1])
    return path


def _from_rounded_rect(center, half_extent, radius):  # path.rs:746-780
    f = np.float32
    cx, cy, hx, hy, r = f(center[0]), f(center[1]), f(half_extent[0]), f(half_extent[1]), f(radius)
    vertices = [((cx - hx + r, cy - hy), (cx - hx, cy - hy), (cx - hx, cy - hy + r)),
                ((cx - hx, cy + hy - r), (cx - hx, cy + hy), (cx - hx + r, cy + hy)),
                ((cx + hx - r, cy + hy), (cx + hx, cy + hy), (cx + hx, cy + hy - r)),
                ((cx + hx, cy - hy + r), (cx + hx, cy - hy), (cx + hx - r, cy - hy))]
    return _rounded_corner_path(vertices[3][2], vertices)


def _from_ellipse(center, half_extent):  # path.rs:783-810
    f = np.float32
    cx, cy, hx, hy = f(center[0]), f(center[1]), f(half_extent[0]), f(half_extent[1])
    vertices = [((cx - hx, cy - hy), (cx - hx, cy)), ((cx - hx, cy + hy), (cx, cy + hy)), ((cx + hx, cy + hy), (cx + hx, cy)), ((cx + hx, cy - hy), (cx, cy - hy))]
    return _rounded_corner_path(vertices[3][1], vertices)


Path.from_rounded_rect = staticmethod(_from_rounded_rect)
Path.from_ellipse = staticmethod(_from_ellipse)
Path.from_circle = staticmethod(lambda center, radius: _from_ellipse(center, (radius, radius)))  # path.rs:813-815


def _tangent_from_points(a, b):  # path.rs:203-205: a v b = [ay bx - ax by, by - ay, ax - bx]
    f = np.float32
    ax, ay, bx, by = f(a[0]), f(a[1]), f(b[0]), f(b[1])
    return (ay * bx - ax * by, by - ay, ax - bx)


def _signum(plane):
    f = np.float32
    inv = f(1.0) / np.sqrt(plane[1] * plane[1] + plane[2] * plane[2], dtype=f)
    return tuple(float(f(v) * inv) for v in plane)


def batch_from_shapes(shapes: Sequence[Tuple[Sequence[DynamicStrokeOptions], Sequence[Path]]]) -> _ffi.PathBatch:
    """Flattens [(dynamic_stroke_options, paths), ...] — the arguments of Shape::from_paths (renderer.rs:177-183),
    one tuple per Shape — into the struct-of-arrays crh_path_batch."""
    shape_path_begin, shape_dynamic_begin = [0], [0]
    path_segment_begin, path_start, path_stroke = [0], [], []
    segment_types, control, stroke_options, dynamic = [], [], [], []
    for dynamic_stroke_options, paths in shapes:
        for path in paths:
            path_start.append(_vec2(path.start))
            if path.stroke_options is None:
                path_stroke.append(-1)
            else:
                path_stroke.append(len(stroke_options))
                stroke_options.append(path.stroke_options.to_c())
            segment_types.extend(int(t) for t in path.segment_types)
            for record in path.records:
                control.extend(record)
            path_segment_begin.append(len(segment_types))
        shape_path_begin.append(len(path_start))
        dynamic.extend(o.to_c() for o in dynamic_stroke_options)
        shape_dynamic_begin.append(len(dynamic))
    return _ffi.PathBatch(
        shape_path_begin, path_segment_begin,
        np.asarray(path_start, dtype=np.float32).reshape(-1, 2), path_stroke, segment_types, np.asarray(control, dtype=np.float32),
        stroke_options, shape_dynamic_begin, dynamic)
