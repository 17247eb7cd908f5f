"""The oracle against mathematics instead of against itself. Its parity with the reference is unpinned (no reference tests, no Rust
toolchain), so this file checks what can be checked independently: the pixels the oracle's stencil-then-cover pipeline produces for a
filled path must be the pixels whose centres have a non-zero winding number with respect to the exact curve, and a round-joined,
round-capped stroke must cover exactly the pixels closer than width / 2 to the curve. The ground truth is brute force: every Bezier
(integral and rational) is flattened into thousands of chords by direct evaluation of its Bernstein form — no Loop-Blinn, no tessellation
code shared with the oracle. Disagreements are only tolerated where the pixel centre lies within a hair of the boundary."""
import numpy as np
import pytest

from contrast_renderer_amd import Cap, CurveApproximation, DynamicStrokeOptions, Join, Path, SegmentType, StrokeOptions, batch_from_shapes, scenes

SIZE = 96


def flatten(path, samples=400, closed=True):
    """-> [n, 2] points along the exact curve in path coordinates."""
    t = np.linspace(0.0, 1.0, samples, endpoint=False)[:, None]
    out = []
    p0 = np.asarray(path.start, dtype=np.float64)
    for kind, rec in zip(path.segment_types, path.records):
        r = np.asarray(rec, dtype=np.float64)
        if kind == SegmentType.Line:
            pts, w = [p0, r[0:2]], [1.0, 1.0]
        elif kind == SegmentType.IntegralQuadraticCurve:
            pts, w = [p0, r[0:2], r[2:4]], [1.0, 1.0, 1.0]
        elif kind == SegmentType.IntegralCubicCurve:
            pts, w = [p0, r[0:2], r[2:4], r[4:6]], [1.0] * 4
        elif kind == SegmentType.RationalQuadraticCurve:
            pts, w = [p0, r[1:3], r[3:5]], [1.0, r[0], 1.0]
        else:
            pts, w = [p0, r[4:6], r[6:8], r[8:10]], list(r[0:4])
        n = len(pts) - 1
        binom = [1, n, n * (n - 1) // 2, 1][:n] + [1] if n == 3 else ([1, 2, 1] if n == 2 else [1, 1])
        basis = [b * t ** k * (1 - t) ** (n - k) * wk for k, (b, wk) in enumerate(zip(binom, w))]
        den = sum(basis)
        out.append(sum(bk * np.asarray(pk)[None, :] for bk, pk in zip(basis, pts)) / den)
        p0 = np.asarray(pts[-1], dtype=np.float64)
    out.append(p0[None, :] if not closed else np.asarray(path.start, dtype=np.float64)[None, :])
    return np.concatenate(out)


def to_pixels(points, transform, size):
    m = np.asarray(transform, dtype=np.float64)
    x = (m[0] * points[:, 0] + m[4] * points[:, 1] + m[12]) * 0.5 + 0.5
    y = 0.5 - (m[1] * points[:, 0] + m[5] * points[:, 1] + m[13]) * 0.5
    return np.stack([x * size, y * size], axis=1)


def pixel_centres(size):
    c = np.arange(size) + 0.5
    return np.stack(np.meshgrid(c, c), axis=-1).reshape(-1, 2)  # row-major: y outer, x inner


def winding_numbers(polygon, centres):
    a, b = polygon[:-1], polygon[1:]
    px, py = centres[:, 0:1], centres[:, 1:2]
    upward = (a[None, :, 1] <= py) & (b[None, :, 1] > py)
    downward = (a[None, :, 1] > py) & (b[None, :, 1] <= py)
    cross = (b[None, :, 0] - a[None, :, 0]) * (py - a[None, :, 1]) - (px - a[None, :, 0]) * (b[None, :, 1] - a[None, :, 1])
    return (upward & (cross > 0)).sum(axis=1) - (downward & (cross < 0)).sum(axis=1)


def distance_to_polyline(polyline, centres):
    a, b = polyline[:-1], polyline[1:]
    d = b - a
    length2 = np.maximum((d * d).sum(axis=1), 1e-30)
    best = np.full(len(centres), np.inf)
    for chunk in range(0, len(a), 512):
        aa, dd, ll = a[chunk:chunk + 512], d[chunk:chunk + 512], length2[chunk:chunk + 512]
        rel = centres[:, None, :] - aa[None, :, :]
        t = np.clip((rel * dd[None]).sum(axis=2) / ll[None], 0.0, 1.0)
        diff = rel - t[..., None] * dd[None]
        best = np.minimum(best, np.sqrt((diff * diff).sum(axis=2)).min(axis=1))
    return best


def test_filled_cubic_paths_cover_exactly_the_pixels_with_nonzero_winding(oracle_lib):
    sc = scenes.scene_cubic_fill(16, (SIZE, SIZE), r_lo=10.0, r_hi=40.0)
    batch = sc["batch"]
    oracle = oracle_lib.Oracle(batch)
    assert oracle.status() == 0
    centres = pixel_centres(SIZE)
    floats = np.asarray((2, 4, 6, 5, 10))
    offsets = np.concatenate([[0], np.cumsum(floats[batch.segment_types])])
    checked = 0
    for shape in range(batch.n_shapes):
        p = int(batch.shape_path_begin[shape])
        path = Path(start=tuple(batch.path_start[p]))
        for s in range(int(batch.path_segment_begin[p]), int(batch.path_segment_begin[p + 1])):
            path.segment_types.append(SegmentType(int(batch.segment_types[s])))
            path.records.append(tuple(float(v) for v in batch.control_data[offsets[s]:offsets[s + 1]]))
        colors = np.zeros_like(sc["colors"])
        colors[shape] = (1.0, 1.0, 1.0, 1.0)  # only this shape is visible; every other one blends nothing
        image = oracle.render(SIZE, SIZE, 1, 8, sc["transforms"], colors)
        polygon = to_pixels(flatten(path, 600), sc["transforms"][shape], SIZE)
        inside = winding_numbers(polygon, centres) != 0
        covered = image[..., 3].reshape(-1) > 0
        wrong = np.flatnonzero(inside != covered)
        if len(wrong):
            assert distance_to_polyline(polygon, centres[wrong]).max() < 2e-3, f"shape {shape}: {len(wrong)} pixels differ away from the boundary"
        checked += int(inside.sum())
    assert checked > 5000


@pytest.mark.parametrize("approximation", [CurveApproximation.UniformTangentAngle(0.02), CurveApproximation.UniformlySpacedParameters(64)])
def test_round_strokes_cover_exactly_the_pixels_within_half_the_width(oracle_lib, approximation):
    width = 0.12
    paths = []
    p = Path(start=(-0.7, -0.5))
    p.push_integral_cubic_curve((-0.2, 0.9), (0.3, -0.9), (0.7, 0.4))
    p.push_rational_quadratic_curve(0.6, (0.8, 0.8), (0.1, 0.7))
    p.push_line((-0.5, 0.6))
    paths.append((p, False))
    q = Path(start=(0.5, 0.0))
    q.push_integral_quadratic_curve((0.5, 0.5), (0.0, 0.5))
    q.push_rational_cubic_curve((1.0, 0.8, 1.4, 1.0), (-0.6, 0.5), (-0.6, -0.5), (0.0, -0.5))
    q.push_integral_quadratic_curve((0.5, -0.5), (0.5, 0.0))
    paths.append((q, True))
    centres = pixel_centres(SIZE)
    for path, closed in paths:
        path.stroke_options = StrokeOptions(width, 0.0, 4.0, closed, 0, approximation)
        batch = batch_from_shapes([([DynamicStrokeOptions.Solid(Join.Round, Cap.Round, Cap.Round)], [path])])
        oracle = oracle_lib.Oracle(batch)
        assert oracle.status() == 0
        transform = scenes.place(SIZE, SIZE, np.array([SIZE / 2.0]), np.array([SIZE / 2.0]), np.array([SIZE * 0.45]))
        image = oracle.render(SIZE, SIZE, 1, 4, transform, np.array([[1.0, 1.0, 1.0, 1.0]], dtype=np.float32))
        covered = image[..., 3].reshape(-1) > 0
        polyline = to_pixels(flatten(path, 800, closed), transform[0], SIZE)
        distance = distance_to_polyline(polyline, centres)
        half = width * 0.5 * SIZE * 0.45
        # the stroke is a polygonal approximation of the offset curve: allow a thin band around the exact boundary
        sure_inside, sure_outside = distance < half - 0.12, distance > half + 0.12
        assert covered[sure_inside].all() and not covered[sure_outside].any()
        assert sure_inside.sum() > 300
