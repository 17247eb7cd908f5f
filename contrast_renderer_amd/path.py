"""Host-side mirror of the reference's Path data model (src/path.rs:15-230) — same names, same meaning.

Only the data model and the trivial polygon constructors live here (they are the *input layout* of the
hot path); arcs / ellipses / reverse / convert are scene-building helpers that SURVEY.md §8(f) ranks "next".
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
