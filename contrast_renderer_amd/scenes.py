"""Synthetic scenes of SURVEY.md §8(d) / BASELINE.json `configs` (there are no datasets: the reference ships none).

PRNG = PCG32 (XSH-RR 64/32), seed 0xC0FFEE ^ config_index, one stream per path, vectorised over paths with
numpy uint64 arithmetic. One Shape per path.

Coordinates: every path is modelled in LOCAL units around the origin (radius ~1) and placed by its Shape's instance
transform (scale = radius in pixels, translation = centre) — the way the reference is driven (one Shape, many
instances with their own mat4: examples/showcase/main.rs:162-202,236-250). This is not cosmetic: the reference's
ERROR_MARGIN = 1e-4 tests (fill.rs:125,153; convex_hull.rs:19; curve.rs:153) are absolute, and f32 triple products of
absolute pixel coordinates (~4096) carry ~0.1 of cancellation error, which makes fill.rs:174/178 panic on ordinary curves.
"""
import math

import numpy as np

from . import _ffi
from .path import Cap, CurveApproximation, DashInterval, DynamicStrokeOptions, Join, StrokeOptions

_MULT = np.uint64(6364136223846793005)


class PCG32:
    """n independent PCG32 streams advanced in lock step."""

    def __init__(self, seed: int, n_streams: int, first_stream: int = 0):
        with np.errstate(over="ignore"):
            self.inc = ((np.arange(first_stream, first_stream + n_streams, dtype=np.uint64) << np.uint64(1)) | np.uint64(1))
            self.state = np.zeros(n_streams, dtype=np.uint64)
            self._step()
            self.state = self.state + np.uint64(seed)
            self._step()

    def _step(self):
        with np.errstate(over="ignore"):
            self.state = self.state * _MULT + self.inc

    def next_u32(self):
        old = self.state.copy()
        self._step()
        xorshifted = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
        rot = (old >> np.uint64(59)).astype(np.uint32)
        return (xorshifted >> rot) | (xorshifted << ((np.uint32(32) - rot) & np.uint32(31)))

    def uniform(self, lo=0.0, hi=1.0):
        u = (self.next_u32() >> np.uint32(8)).astype(np.float64) * (1.0 / 16777216.0)
        return lo + (hi - lo) * u


def ortho_pixels(width, height):
    """Column-major mat4 mapping y-up pixel units to NDC (the one instance transform the benchmark uses)."""
    m = np.zeros(16, dtype=np.float32)
    m[0] = 2.0 / width
    m[5] = 2.0 / height
    m[10] = 1.0
    m[12] = -1.0
    m[13] = -1.0
    m[15] = 1.0
    return m


def place(width, height, cx, cy, scale):
    """Per-shape instance transforms: ortho_pixels(width, height) * translate(cx, cy) * scale(scale), column-major mat4."""
    n = len(cx)
    m = np.zeros((n, 16), dtype=np.float64)
    m[:, 0] = 2.0 * scale / width
    m[:, 5] = 2.0 * scale / height
    m[:, 10] = 1.0
    m[:, 12] = 2.0 * cx / width - 1.0
    m[:, 13] = 2.0 * cy / height - 1.0
    m[:, 15] = 1.0
    return m.astype(np.float32)


def _colors(rng, n):
    rgb = np.stack([rng.uniform(), rng.uniform(), rng.uniform()], axis=1)
    opaque = rng.uniform() < 0.5
    alpha = np.where(opaque, 1.0, rng.uniform(0.25, 1.0))
    return np.concatenate([rgb, alpha[:, None]], axis=1).astype(np.float32)


def _blob_points(rng, n, size, r_lo, r_hi, log_radius, n_seg=8):
    """On-curve points of n closed blobs: centres U(0,size)^2, radius (log-)U(r_lo, r_hi), angles 2 pi k/n_seg + U(+-0.2)."""
    cx, cy = rng.uniform(0.0, size[0]), rng.uniform(0.0, size[1])
    if log_radius:
        radius = np.exp(rng.uniform(math.log(r_lo), math.log(r_hi)))
    else:
        radius = rng.uniform(r_lo, r_hi)
    angles = np.stack([2.0 * math.pi * k / n_seg + rng.uniform(-0.2, 0.2) for k in range(n_seg)], axis=1)
    rr = np.stack([rng.uniform(0.8, 1.2) for _ in range(n_seg)], axis=1)
    # All paths run CLOCKWISE (y up): that is the reference's convention for solid paths — from_rect / from_rounded_rect / from_ellipse
    # (path.rs:736-810) wind clockwise and holes are made with reverse(). The cubic fill (fill.rs:116-250) is only geometrically exact for
    # that orientation: a stand-alone counter-clockwise cubic path is filled out to its control polygon (tests/test_oracle_ground_truth.py).
    angles = -angles
    px = rr * np.cos(angles)  # local units: centre 0, radius ~1
    py = rr * np.sin(angles)
    return cx, cy, radius, angles, px, py


def _f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32)


def scene_quadratic(n_paths=100, size=(1024, 1024), config_index=1, stroke_steps=8):
    """S100q (config 1): n closed paths of 8 integral-quadratic segments; shape 2i = path i filled, shape 2i+1 = path i stroked
    (width U(1,4), offset 0, miter_clip 4, closed, UniformlySpacedParameters(8), Solid/Miter/Butt)."""
    rng = PCG32(0xC0FFEE ^ config_index, n_paths)
    cx = rng.uniform(64.0, size[0] - 64.0)
    cy = rng.uniform(64.0, size[1] - 64.0)
    radius = rng.uniform(16.0, 48.0)
    n_seg = 8
    angles = np.stack([2.0 * math.pi * k / n_seg + rng.uniform(-0.2, 0.2) for k in range(n_seg)], axis=1)
    px = _f32(np.cos(angles))  # local units: centre 0, radius 1
    py = _f32(np.sin(angles))
    nxt = np.roll(angles, -1, axis=1)
    nxt[:, -1] += 2.0 * math.pi
    mid = 0.5 * (angles + nxt)
    cr = np.stack([rng.uniform(0.9, 1.4) for _ in range(n_seg)], axis=1)
    qx = _f32(cr * np.cos(mid))
    qy = _f32(cr * np.sin(mid))
    width = _f32(rng.uniform(1.0, 4.0) / radius)  # U(1,4) pixels, expressed in local units
    colors_fill = _colors(rng, n_paths)
    colors_stroke = _colors(rng, n_paths)
    # records per path: 8 x (ctrl, end); end of segment k = point k+1 (closing on point 0)
    ex, ey = np.roll(px, -1, axis=1), np.roll(py, -1, axis=1)
    rec = np.stack([qx, qy, ex, ey], axis=2).reshape(n_paths, -1)  # [n, 32]
    rec2 = np.repeat(rec, 2, axis=0)  # each path twice
    starts = np.repeat(np.stack([px[:, 0], py[:, 0]], axis=1), 2, axis=0)
    n_total = 2 * n_paths
    stroke_idx = np.full(n_total, -1, dtype=np.int32)
    stroke_idx[1::2] = np.arange(n_paths)
    stroke_options = []
    for i in range(n_paths):
        so = StrokeOptions(float(width[i]), 0.0, 4.0, True, 0, CurveApproximation.UniformlySpacedParameters(stroke_steps))
        stroke_options.append(so.to_c())
    dyn = [DynamicStrokeOptions.Solid(Join.Miter, Cap.Butt, Cap.Butt).to_c() for _ in range(n_paths)]
    shape_dynamic_begin = np.zeros(n_total + 1, dtype=np.uint32)
    shape_dynamic_begin[1:] = np.cumsum(np.arange(n_total) % 2)
    batch = _ffi.PathBatch(
        np.arange(n_total + 1), np.arange(n_total + 1) * n_seg, starts, stroke_idx,
        np.full(n_total * n_seg, _ffi.SEGMENT_INTEGRAL_QUADRATIC, dtype=np.uint8), rec2.reshape(-1), stroke_options, shape_dynamic_begin, dyn)
    colors = np.empty((n_total, 4), dtype=np.float32)
    colors[0::2] = colors_fill
    colors[1::2] = colors_stroke
    transforms = np.repeat(place(size[0], size[1], cx, cy, radius), 2, axis=0)
    return dict(batch=batch, transforms=transforms, colors=colors, width=size[0], height=size[1], msaa=1, winding_bits=4, name="S100q")


def _cubic_records(rng, n, px, py, rational):
    """8 cubic segments per path through the on-curve points: handles = +-(U(0.2,0.45) x chord) rotated by U(+-0.3) rad
    (SURVEY.md §8(d) says U(0.2,0.6): beyond ~0.45 neighbouring handles cross often enough that the reference's own
    degenerate-cubic assertions, fill.rs:174,178, reject a visible share of the scene — the generator stays inside what it accepts).
    Even segments integral, odd segments rational (weights U(0.5,2)) when `rational` == 'mixed'; all rational when 'all'."""
    n_seg = px.shape[1]
    ex, ey = np.roll(px, -1, axis=1), np.roll(py, -1, axis=1)
    dx, dy = ex - px, ey - py
    recs, types = [], []
    for k in range(n_seg):
        l1, l2 = rng.uniform(0.2, 0.45), rng.uniform(0.2, 0.45)
        a1, a2 = rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3)
        # bulge outwards (clockwise travel: the outside is on the LEFT): rotate the chord direction by ~+0.4 rad plus the perturbation
        h1x = px[:, k] + l1 * (np.cos(a1 + 0.4) * dx[:, k] - np.sin(a1 + 0.4) * dy[:, k])
        h1y = py[:, k] + l1 * (np.sin(a1 + 0.4) * dx[:, k] + np.cos(a1 + 0.4) * dy[:, k])
        h2x = ex[:, k] - l2 * (np.cos(a2 - 0.4) * dx[:, k] - np.sin(a2 - 0.4) * dy[:, k])
        h2y = ey[:, k] - l2 * (np.sin(a2 - 0.4) * dx[:, k] + np.cos(a2 - 0.4) * dy[:, k])
        w = [rng.uniform(0.5, 2.0) for _ in range(4)]
        is_rational = rational == "all" or (rational == "mixed" and k % 2 == 1)
        if is_rational:
            # Rational segments get a CONVEX control polygon: the reference's enclosing-triangle test compares weighted
            # areas (fill.rs:141-156), so with non-uniform weights a control point inside the triangle of the other three
            # trips assert_eq!/assert_ne! (fill.rs:174,178). P1, P2 sit on the legs of a triangle over the chord.
            depth, along = rng.uniform(0.25, 0.6), rng.uniform(0.4, 0.6)
            apex_x = px[:, k] + along * dx[:, k] - depth * dy[:, k]   # left of travel (outside of a clockwise path): (-dy, dx)
            apex_y = py[:, k] + along * dy[:, k] + depth * dx[:, k]
            t1, t2 = rng.uniform(0.3, 0.55), rng.uniform(0.3, 0.55)
            h1x = px[:, k] + t1 * (apex_x - px[:, k])
            h1y = py[:, k] + t1 * (apex_y - py[:, k])
            h2x = ex[:, k] + t2 * (apex_x - ex[:, k])
            h2y = ey[:, k] + t2 * (apex_y - ey[:, k])
        pts = [h1x, h1y, h2x, h2y, ex[:, k], ey[:, k]]
        if is_rational:
            recs.append(np.stack(w + pts, axis=1))
            types.append(_ffi.SEGMENT_RATIONAL_CUBIC)
        else:
            recs.append(np.stack(pts, axis=1))
            types.append(_ffi.SEGMENT_INTEGRAL_CUBIC)
    control = _f32(np.concatenate(recs, axis=1))  # [n, floats per path]
    return control, np.asarray(types, dtype=np.uint8)


def _loop_double_point_in_range(start, control, types, margin=0.04):
    """True for paths holding a cubic whose self-intersection (double point) parameter lies within `margin` of 0 or 1.

    Such segments are legal input, but the reference splits the curve AT the double point (fill.rs:232-241) and when
    that parameter is within ~1e-2 of 0 or 1 one half degenerates and trips assert_ne! (fill.rs:178). The generators
    re-draw these paths. Oracle-free: d0..d3 = inflection polynomial coefficients (curve.rs:133-144), double point
    parameters = roots of its Hessian (curve.rs:206-213), in float64."""
    n = start.shape[0]
    bad = np.zeros(n, dtype=bool)
    prev = start.astype(np.float64)
    off = 0
    control = control.astype(np.float64)
    for t in types:
        if t == _ffi.SEGMENT_INTEGRAL_CUBIC:
            w = np.ones((n, 4))
            pts = np.concatenate([prev, control[:, off:off + 6]], axis=1).reshape(n, 4, 2)
            off += 6
        else:
            w = control[:, off:off + 4]
            pts = np.concatenate([prev, control[:, off + 4:off + 10]], axis=1).reshape(n, 4, 2)
            off += 10
        h = np.concatenate([w[:, :, None], pts * w[:, :, None]], axis=2)  # [n, 4, (w, xw, yw)]
        c = np.stack([h[:, 0], 3 * (h[:, 1] - h[:, 0]), 3 * (h[:, 0] - 2 * h[:, 1] + h[:, 2]), h[:, 3] - 3 * h[:, 2] + 3 * h[:, 1] - h[:, 0]], axis=1)
        det = lambda a, b, cc: np.linalg.det(np.stack([c[:, a], c[:, b], c[:, cc]], axis=1))
        d = np.stack([-det(1, 2, 3), det(0, 2, 3), -det(0, 1, 3), det(0, 1, 2)], axis=1)
        d = d / np.linalg.norm(d, axis=1, keepdims=True)
        c0 = d[:, 1] * d[:, 3] - d[:, 2] ** 2
        c1 = d[:, 1] * d[:, 2] - d[:, 0] * d[:, 3]
        c2 = d[:, 0] * d[:, 2] - d[:, 1] ** 2
        disc = c1 * c1 - 4 * c2 * c0
        with np.errstate(invalid="ignore", divide="ignore"):
            sq = np.sqrt(np.maximum(disc, 0.0))
            r1, r2 = (-c1 + sq) / (2 * c2), (-c1 - sq) / (2 * c2)
        loop = disc > -1e-6
        near = lambda r: (np.abs(r) < margin) | (np.abs(r - 1.0) < margin)
        inside = near(r1) | near(r2)
        bad |= loop & inside & np.isfinite(r1)
        prev = pts[:, 3]
    return bad


def scene_cubic_fill(n_paths=10000, size=(4096, 4096), config_index=2, r_lo=8.0, r_hi=128.0, first_path=0):
    """S10k / S100k (configs 2, 4): n filled closed paths, 8 cubic segments each alternating integral / rational,
    centres U(0,size)^2, radius log-U(r_lo, r_hi). `first_path` selects the PCG streams (path-index sharding)."""
    rng = PCG32(0xC0FFEE ^ config_index, n_paths, first_path)
    cx, cy, radius, angles, px, py = _blob_points(rng, n_paths, size, r_lo, r_hi, True)
    px, py = _f32(px), _f32(py)
    control = None
    for attempt in range(16):  # re-draw (from the same per-path streams) the paths the reference cannot tessellate
        fresh, types = _cubic_records(rng, n_paths, px.astype(np.float64), py.astype(np.float64), "mixed")
        # close exactly: the last record ends on the start point bit for bit
        fresh[:, -2] = px[:, 0]
        fresh[:, -1] = py[:, 0]
        if control is None:
            control = fresh
        else:
            control[redo] = fresh[redo]
        redo = _loop_double_point_in_range(np.stack([px[:, 0], py[:, 0]], axis=1), control, types)
        if not redo.any():
            break
    colors = _colors(rng, n_paths)
    n_seg = 8
    batch = _ffi.PathBatch(
        np.arange(n_paths + 1), np.arange(n_paths + 1) * n_seg, np.stack([px[:, 0], py[:, 0]], axis=1), np.full(n_paths, -1, dtype=np.int32),
        np.tile(types, n_paths), control.reshape(-1))
    transforms = place(size[0], size[1], cx, cy, radius)
    return dict(batch=batch, transforms=transforms, colors=colors, width=size[0], height=size[1], msaa=1, winding_bits=4,
                name=f"S{n_paths}c")


def scene_dashed_strokes(n_paths=2000, size=(4096, 4096), config_index=5, angle_step=0.1, msaa=4):
    """Sdash (config 5): open + closed rational-cubic paths stroked with UniformTangentAngle(angle_step), width U(4,24),
    joins cycling Miter/Round, dashed with two intervals (gaps [2,3] and [5,6], caps cycling all 7)."""
    rng = PCG32(0xC0FFEE ^ config_index, n_paths)
    cx, cy, radius, angles, px, py = _blob_points(rng, n_paths, size, 24.0, 160.0, True)
    px, py = _f32(px), _f32(py)
    control, types = _cubic_records(rng, n_paths, px.astype(np.float64), py.astype(np.float64), "all")
    control[:, -2] = px[:, 0]
    control[:, -1] = py[:, 0]
    width = _f32(rng.uniform(4.0, 24.0) / radius)  # U(4,24) pixels in local units
    colors = _colors(rng, n_paths)
    n_seg = 8
    closed = (np.arange(n_paths) % 2) == 0
    # open paths drop their last segment
    seg_counts = np.where(closed, n_seg, n_seg - 1)
    path_segment_begin = np.concatenate([[0], np.cumsum(seg_counts)])
    seg_types, ctrl = [], []
    floats_per_seg = 10
    for i in range(n_paths):
        k = int(seg_counts[i])
        seg_types.append(types[:k])
        ctrl.append(control[i, :k * floats_per_seg])
    stroke_options, dyn = [], []
    for i in range(n_paths):
        so = StrokeOptions(float(width[i]), 0.0, 4.0, bool(closed[i]), 0, CurveApproximation.UniformTangentAngle(angle_step))
        stroke_options.append(so.to_c())
        c0, c1 = Cap(i % 7), Cap((i + 3) % 7)
        pattern = [DashInterval(2.0, 3.0, c0, c1), DashInterval(5.0, 6.0, c1, c0)]
        dyn.append(DynamicStrokeOptions.Dashed(Join.Miter if (i // 2) % 2 == 0 else Join.Round, pattern, 0.0).to_c())
    batch = _ffi.PathBatch(
        np.arange(n_paths + 1), path_segment_begin, np.stack([px[:, 0], py[:, 0]], axis=1), np.arange(n_paths, dtype=np.int32),
        np.concatenate(seg_types), np.concatenate(ctrl), stroke_options, np.arange(n_paths + 1), dyn)
    transforms = place(size[0], size[1], cx, cy, radius)
    return dict(batch=batch, transforms=transforms, colors=colors, width=size[0], height=size[1], msaa=msaa, winding_bits=4, name="Sdash")


def scene_mixed(n_shapes=64, size=(512, 512), seed=7):
    """A small everything-scene for parity tests: every segment type, filled and stroked paths in the same Shape,
    open / closed strokes, both curve approximations, solid and dashed groups, all joins and caps."""
    rng = np.random.RandomState(seed)
    from .path import Path, batch_from_shapes
    shapes, colors, centres, radii = [], [], [], []
    for s in range(n_shapes):
        centres.append((rng.uniform(40, size[0] - 40), rng.uniform(40, size[1] - 40)))
        radii.append(rng.uniform(10, 60))
        cx, cy, r = 0.0, 0.0, 1.0  # local units; placement goes through the instance transform
        paths, dyn = [], []
        n_paths = 1 + s % 3
        for p in range(n_paths):
            n_seg = rng.randint(2, 7)
            ang = np.sort(rng.uniform(0, 2 * math.pi, n_seg + 1))
            if (s + p) % 2:
                ang = ang[::-1]
            pts = [(cx + r * math.cos(a) * rng.uniform(0.7, 1.2), cy + r * math.sin(a) * rng.uniform(0.7, 1.2)) for a in ang]
            path = Path(start=(np.float32(pts[0][0]), np.float32(pts[0][1])))
            for k in range(1, n_seg + 1):
                a, b = pts[k - 1], pts[k]
                kind = (s + p + k) % 5
                jitter = lambda t: (a[0] + (b[0] - a[0]) * t + rng.uniform(-0.4, 0.4) * r, a[1] + (b[1] - a[1]) * t + rng.uniform(-0.4, 0.4) * r)
                if kind == 0:
                    path.push_line(b)
                elif kind == 1:
                    path.push_integral_quadratic_curve(jitter(0.5), b)
                elif kind == 2:
                    path.push_integral_cubic_curve(jitter(0.33), jitter(0.66), b)
                elif kind == 3:
                    path.push_rational_quadratic_curve(rng.uniform(0.5, 2.0), jitter(0.5), b)
                else:
                    nx, ny = (b[1] - a[1]), -(b[0] - a[0])
                    depth, along = rng.uniform(0.25, 0.7), rng.uniform(0.35, 0.65)
                    apex = (a[0] + along * (b[0] - a[0]) + depth * nx, a[1] + along * (b[1] - a[1]) + depth * ny)
                    t1, t2 = rng.uniform(0.35, 0.8), rng.uniform(0.35, 0.8)
                    c0 = (a[0] + t1 * (apex[0] - a[0]), a[1] + t1 * (apex[1] - a[1]))
                    c1 = (b[0] + t2 * (apex[0] - b[0]), b[1] + t2 * (apex[1] - b[1]))
                    path.push_rational_cubic_curve(rng.uniform(0.5, 2.0, 4), c0, c1, b)
            if (s + p) % 3 == 0:
                group = len(dyn)
                if s % 4 == 0:
                    caps = [Cap(int(c)) for c in rng.randint(0, 7, 4)]
                    n_int = 1 + s % 3
                    pattern, pos = [], 0.0
                    for i in range(n_int):
                        g0 = pos + rng.uniform(0.5, 2.0)
                        g1 = g0 + rng.uniform(0.5, 1.5)
                        pattern.append(DashInterval(g0, g1, caps[i], caps[(i + 1) % 4]))
                        pos = g1
                    dyn.append(DynamicStrokeOptions.Dashed(Join(s % 3), pattern, rng.uniform(-1, 1)))
                else:
                    dyn.append(DynamicStrokeOptions.Solid(Join(s % 3), Cap(int(rng.randint(0, 7))), Cap(int(rng.randint(0, 7)))))
                approx = CurveApproximation.UniformTangentAngle(rng.uniform(0.08, 0.3)) if s % 2 else CurveApproximation.UniformlySpacedParameters(int(rng.randint(1, 9)))
                path.stroke_options = StrokeOptions(rng.uniform(1.0, 9.0) / radii[-1], rng.uniform(-0.5, 0.5) if s % 5 == 0 else 0.0, rng.uniform(0.6, 4.0), bool((s + p) % 2), group, approx)
            paths.append(path)
        shapes.append((dyn, paths))
        a = 1.0 if s % 2 else rng.uniform(0.3, 1.0)
        colors.append((rng.uniform(), rng.uniform(), rng.uniform(), a))
    batch = batch_from_shapes(shapes)
    centres = np.asarray(centres)
    transforms = place(size[0], size[1], centres[:, 0], centres[:, 1], np.asarray(radii))
    return dict(batch=batch, transforms=transforms, colors=np.asarray(colors, dtype=np.float32), width=size[0], height=size[1], msaa=4,
                winding_bits=4, name="Smixed")


def default_font_path():
    """The bundled OpenSans-Regular.ttf data fixture (Apache-2.0; the reference ships the same file under examples/fonts/)."""
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "fonts", "OpenSans-Regular.ttf")


def scene_glyphs(n_glyphs=50000, size=(2048, 2048), config_index=3, font_path=None, sizes=(12.0, 16.0, 24.0, 32.0, 48.0)):
    """S50kg (BASELINE configs[2]): `n_glyphs` glyph instances produced by text::paths_of_text (native glyph producer, text.rs:236-263) —
    many tiny line / quadratic paths, one filled Shape per glyph instance. The text cycles through every character of the font's cmap
    whose glyph has an outline; it is set in bands of lines, Layout{size cycling 12..48 px, LeftToRight, Begin, Baseline}, stacked down
    the frame (wrapping, so later bands overlay earlier ones). Glyph coordinates are layout coordinates (origin = frame centre, y up),
    placed by Path::transform exactly as paths_of_text does; all Shapes share one instance transform."""
    from . import text as T
    width, height = size
    data = open(font_path or default_font_path(), "rb").read()
    font = T.Font("OpenSans", data)
    chars, contours, advances = [], {}, {}
    for code in range(0x21, 0x3000):
        gid = font.glyph_index(code)
        if gid is None or font.glyph_bounding_box(gid) is None:
            continue
        if gid not in contours:
            contours[gid] = len(T.glyph_path_list(font, gid).arrays()[1])
            advances[gid] = font.glyph_hor_advance(gid) or 0
        if contours[gid] > 0:
            chars.append((code, gid))
    assert chars, "the font maps no outline glyph"
    seg_begin_all, start_all, types_all, control_all, shape_paths = [np.zeros(1, np.int64)], [], [], [], []
    produced, cursor, band, y_top = 0, 0, 0, 0.0
    while produced < n_glyphs:
        px = float(sizes[band % len(sizes)])
        scale = px / font.height()
        line_height = px
        lines_in_band = max(1, int(96.0 // line_height))
        text, counts = [], []
        for _ in range(lines_in_band):
            x = 0.0
            while produced + len(counts) < n_glyphs:
                code, gid = chars[cursor % len(chars)]
                if x + advances[gid] * scale > width and x > 0.0:
                    break
                cursor += 1
                text.append(code)
                counts.append(contours[gid])
                x += advances[gid] * scale
            text.append(10)
            if produced + len(counts) >= n_glyphs:
                break
        text = text[:-1] if text and text[-1] == 10 else text
        plist = T.text_path_list(font, T.Layout(px, T.Orientation.LeftToRight, T.Alignment.Begin, T.Alignment.Baseline), np.asarray(text, dtype=np.uint32))
        band_height = lines_in_band * line_height
        # the block is centred on the origin by the layout; move its centre to the band's centre (frame centre = origin, y up)
        centre_y = height / 2.0 - ((y_top + band_height / 2.0) % height)
        plist.transform(1.0, (1.0, 0.0, -0.5 * float(np.float32(centre_y)), 0.0))  # translate2d((0, centre_y)), utils.rs:127-129
        seg_begin, start, types, control = plist.arrays()
        assert len(start) == sum(counts)
        seg_begin_all.append(seg_begin[1:].astype(np.int64) + seg_begin_all[-1][-1])
        start_all.append(start)
        types_all.append(types)
        control_all.append(control)
        shape_paths.extend(counts)
        produced += len(counts)
        y_top += band_height
        band += 1
    n_shapes = len(shape_paths)
    shape_path_begin = np.concatenate([[0], np.cumsum(shape_paths)]).astype(np.uint32)
    n_paths = int(shape_path_begin[-1])
    batch = _ffi.PathBatch(shape_path_begin, np.concatenate(seg_begin_all).astype(np.uint32), np.concatenate(start_all).astype(np.float32),
                           np.full(n_paths, -1, dtype=np.int32), np.concatenate(types_all).astype(np.uint8), np.concatenate(control_all).astype(np.float32),
                           [], np.zeros(n_shapes + 1, dtype=np.uint32), [])
    m = np.zeros(16, dtype=np.float32)
    m[0], m[5], m[10], m[15] = 2.0 / width, 2.0 / height, 1.0, 1.0
    transforms = np.tile(m, (n_shapes, 1))
    rng = PCG32(0xC0FFEE ^ config_index, n_shapes)
    colors = _colors(rng, n_shapes)
    return dict(batch=batch, transforms=transforms, colors=colors, width=width, height=height, msaa=1, winding_bits=4, name="S50kg",
                n_glyphs=n_shapes, n_paths=n_paths)
