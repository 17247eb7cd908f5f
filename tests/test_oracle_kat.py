"""Known-answer tests that pin the CPU oracle (KAT-A..F, hand-derived from the reference's source text,
because the reference has no tests or golden vectors of its own), plus geometric property tests of the stroker."""
import math

import numpy as np
import pytest

from contrast_renderer_amd import (Cap, CurveApproximation, DashInterval, DynamicStrokeOptions, Join, Path, StrokeOptions, batch_from_shapes)

pytestmark = pytest.mark.usefixtures("oracle_lib")


def tessellate(paths, dynamic=()):
    from oracle import Oracle
    from oracle.binding import split_shape
    o = Oracle(batch_from_shapes([(list(dynamic), list(paths))]))
    vo, io, vb, ib = o.shape(0)
    return o, vo, io, split_shape(vo, io, vb, ib)


def f32(buf, width):
    return buf.view(np.float32).reshape(-1, width)


def test_kat_a_rect_fill():
    """fill.rs:271-284,361-365 + vertex.rs:28-35 + path.rs:736-743."""
    o, vo, io, d = tessellate([Path.from_rect((3.0, 4.0), (1.0, 2.0))])
    assert o.status() == 0
    v = [(2.0, 2.0), (2.0, 6.0), (4.0, 6.0), (4.0, 2.0)]
    assert f32(d["solid"], 2).tolist() == [list(v[0]), list(v[3]), list(v[1]), list(v[2])]
    assert d["solid_indices"].tolist() == [0, 1, 2, 3, 0xFFFF]
    # convex_hull.rs:7-40 with the sign convention of SURVEY.md A.1: clockwise from the lexicographic minimum, then fan -> strip
    assert f32(d["hull"], 2).tolist() == [[2.0, 2.0], [4.0, 2.0], [2.0, 6.0], [4.0, 6.0]]
    assert vo.tolist() == [0, 0, 32, 32, 32, 32, 32, 64] and io.tolist() == [0, 0, 10]


def test_kat_b_integral_quadratic():
    """fill.rs:285-296: pure copies with constant texcoords."""
    p = Path(start=(1.0, 1.0))
    p.push_integral_quadratic_curve((2.0, 3.0), (4.0, 1.0))
    o, vo, io, d = tessellate([p])
    assert f32(d["integral_quadratic"], 4).tolist() == [[4, 1, 1, 1], [2, 3, 0.5, 0], [1, 1, 0, 0]]
    assert f32(d["solid"], 2).tolist() == [[1, 1], [4, 1]]
    assert d["solid_indices"].tolist() == [0, 1, 0xFFFF]
    # proto_hull = [s, c, e]; andrew of a clockwise-turning triple keeps all three; hull strip = fan_to_strip
    assert sorted(map(tuple, f32(d["hull"], 2).tolist())) == [(1, 1), (2, 3), (4, 1)]


def test_kat_c_rational_quadratic():
    """fill.rs:321-333: [(e,[1,1,1]), (c,[0.5*(1/w), 0, 1/w]), (s,[0,0,1])]."""
    w = np.float32(1.7)
    p = Path(start=(0.0, 0.0))
    p.push_rational_quadratic_curve(w, (1.0, 2.0), (2.0, 0.0))
    o, vo, io, d = tessellate([p])
    inv = np.float32(1.0) / w
    expect = np.array([[2, 0, 1, 1, 1], [1, 2, np.float32(0.5) * inv, 0, inv], [0, 0, 0, 0, 1]], dtype=np.float32)
    assert np.array_equal(f32(d["rational_quadratic"], 5), expect)


@pytest.mark.parametrize("n", range(1, 10))
def test_kat_d_fan_to_strip(n):
    """vertex.rs:28-35: gather [0, n-1, 1, n-2, ...]."""
    pts = [(float(i), float(i * i)) for i in range(n)]
    o, vo, io, d = tessellate([Path.from_polygon(pts)])
    gather = [(i >> 1) if i % 2 == 0 else n - 1 - (i >> 1) for i in range(n)]
    assert f32(d["solid"], 2).tolist() == [list(pts[g]) for g in gather]
    assert d["solid_indices"].tolist() == list(range(n)) + [0xFFFF]


def test_kat_e_descriptor_packing():
    """renderer.rs:29-60."""
    from oracle.binding import _load
    from contrast_renderer_amd import _ffi
    import ctypes as C
    lib = _load()
    out = _ffi.DynamicStrokeDescriptorC()
    solid = DynamicStrokeOptions.Solid(Join.Round, Cap.Out, Cap.Left).to_c()
    assert lib.oracle_convert_dynamic_stroke_options(C.byref(solid), C.byref(out)) == 0
    assert (out.caps, out.count_dashed_join, out.phase) == (2 | (5 << 4), 2, 0.0)
    pattern = [DashInterval(1.0, 2.0, Cap.Round, Cap.In), DashInterval(3.0, 4.5, Cap.Butt, Cap.Square), DashInterval(5.0, 7.0, Cap.Left, Cap.Right)]
    dashed = DynamicStrokeOptions.Dashed(Join.Bevel, pattern, 0.25).to_c()
    assert lib.oracle_convert_dynamic_stroke_options(C.byref(dashed), C.byref(out)) == 0
    assert out.count_dashed_join == ((3 - 1) << 3) | 4 | 1
    caps = 0
    for i, iv in enumerate(pattern):
        caps |= int(iv.dash_start) << (((i + 3 - 1) % 3) * 8)
        caps |= int(iv.dash_end) << (i * 8 + 4)
    assert out.caps == caps and out.phase == 0.25
    assert list(out.gap_start)[:3] == [1.0, 3.0, 5.0] and list(out.gap_end)[:3] == [2.0, 4.5, 7.0]
    too_many = DynamicStrokeOptions.Dashed(Join.Miter, pattern + pattern, 0.0).to_c()
    assert lib.oracle_convert_dynamic_stroke_options(C.byref(too_many), C.byref(out)) == 4  # TooManyDashIntervals
    assert C.sizeof(_ffi.DynamicStrokeDescriptorC) == 48


def stroke_options(width=0.5, closed=True, steps=4, offset=0.0, miter_clip=4.0, angle=None):
    approx = CurveApproximation.UniformTangentAngle(angle) if angle else CurveApproximation.UniformlySpacedParameters(steps)
    return StrokeOptions(width, offset, miter_clip, closed, 0, approx)


SOLID = [DynamicStrokeOptions.Solid(Join.Miter, Cap.Butt, Cap.Butt)]


@pytest.mark.parametrize("n", [3, 4, 7])
def test_kat_f_stroke_counts_closed_polygon(n):
    """SURVEY.md Appendix B.2 for a closed n-gon (m = n-1 Line segments + the implicit closing line, stroke.rs:400-431):
    line vertices 2 + 2m + 2(m-1) + 6 = 4m+6, n joins (5 vertices / 6 indices each), n+1 strips."""
    path = Path.from_regular_polygon((0.0, 0.0), 2.0, 0.3, n)
    path.stroke_options = stroke_options()
    o, vo, io, d = tessellate([path], SOLID)
    assert o.status() == 0
    m = n - 1
    assert len(d["line"]) == 20 * (4 * m + 6)
    assert len(d["joint"]) == 24 * 5 * n and len(d["joint_indices"]) == 6 * n
    assert len(d["line_indices"]) == (4 * m + 6) + (n + 1)
    assert int((d["line_indices"] == 0xFFFF).sum()) == n + 1
    ji = d["joint_indices"].reshape(n, 6)
    assert np.array_equal(ji[:, :5], np.arange(5 * n).reshape(n, 5)) and (ji[:, 5] == 0xFFFF).all()
    assert len(d["solid"]) == 0 and len(d["solid_indices"]) == 0


def test_stroke_counts_open_polyline_and_caps():
    """Open path starting with a Line: start cap pair only (stroke.rs:273-293), end: cut + 4 flagged vertices (stroke.rs:444-462)."""
    path = Path.from_polygon([(0.0, 0.0), (2.0, 0.0), (2.0, 2.0)])
    path.stroke_options = stroke_options(width=0.5, closed=False)
    o, vo, io, d = tessellate([path], SOLID)
    line = d["line"].view(np.dtype([("p", "<f4", 2), ("t", "<f4", 2), ("u", "<u4")]))
    # 2 (cap) + 2 (line 1) | join: cut, 2 + 2 (line 2) | cut, 4 end-cap vertices
    assert len(line) == 12
    assert d["line_indices"].tolist() == [0, 1, 2, 3, 0xFFFF, 4, 5, 6, 7, 0xFFFF, 8, 9, 10, 11, 0xFFFF]
    assert line["u"].tolist() == [0] * 8 + [0x10000] * 4
    # the start cap extends BACKWARDS by half a width and its texcoord.y is -0.5 (stroke.rs:274-282)
    assert np.allclose(line["p"][0], (-0.25, 0.25)) and np.allclose(line["p"][1], (-0.25, -0.25))
    assert line["t"][0].tolist() == [-0.5, -0.5] and line["t"][1].tolist() == [0.5, -0.5]
    # side -0.5 is the LEFT side of travel (path.rs:179, stroke.rs:41-42): travelling +x, left is +y
    assert line["p"][2].tolist() == [2.0, 0.25] and line["t"][2][0] == -0.5
    # the end cap extends forwards: (2, 2) + 0.25 along +y, texcoord.y = (length + 0.5 w) / w
    assert np.allclose(line["p"][10], (1.75, 2.25)) and np.allclose(line["p"][11], (2.25, 2.25))
    length = 4.0 + math.acos(0.0) / (2 * math.pi) * 0.5
    assert np.allclose(line["t"][8][1], length / 0.5, rtol=1e-6) and np.allclose(line["t"][10][1], (length + 0.25) / 0.5, rtol=1e-6)


def test_stroke_offsets_are_half_width_from_the_curve():
    """Property: every line vertex sits at |offset -+ 0.5| * width from the sampled curve point."""
    p = Path(start=(0.0, 0.0))
    p.push_integral_cubic_curve((1.0, 2.0), (3.0, 2.0), (4.0, 0.0))
    p.push_rational_quadratic_curve(1.3, (5.0, -1.0), (6.0, 1.0))
    for offset in (0.0, 0.3, -0.5):
        p.stroke_options = stroke_options(width=0.4, closed=False, offset=offset, angle=0.2)
        o, vo, io, d = tessellate([p], SOLID)
        assert o.status() == 0
        line = d["line"].view(np.dtype([("p", "<f4", 2), ("t", "<f4", 2), ("u", "<u4")]))
        pairs = line["p"].reshape(-1, 2, 2)
        assert np.allclose(np.linalg.norm(pairs[:, 0] - pairs[:, 1], axis=1), 0.4, atol=1e-5)
        # texcoord.y never decreases within a strip, apart from the -0.5 of the start cap
        ty = line["t"][:, 1]
        assert (np.diff(ty[2:-4:2]) >= -1e-6).all()


def test_join_vertices_lie_on_the_offset_lines():
    """emit_stroke_join (stroke.rs:53-121): vertex 0 = control point, 1/2 = offset points, 3/4 = miter tip (or clip points)."""
    path = Path.from_polygon([(0.0, 0.0), (4.0, 0.0), (4.0, 3.0)])
    path.stroke_options = stroke_options(width=1.0, closed=False, miter_clip=4.0)
    o, vo, io, d = tessellate([path], SOLID)
    joint = d["joint"].view(np.dtype([("p", "<f4", 2), ("t", "<f4", 3), ("u", "<u4")]))
    assert len(joint) == 5
    assert joint["p"][0].tolist() == [4.0, 0.0]
    # left turn: the join polygon is on the outer (right) side; previous edge point (4, -0.5), next edge point (4.5, 0), miter tip (4.5, -0.5)
    assert np.allclose(joint["p"][1], (4.0, -0.5)) and np.allclose(joint["p"][2], (4.5, 0.0))
    assert np.allclose(joint["p"][3], (4.5, -0.5)) and np.allclose(joint["p"][4], (4.5, -0.5))
    # texcoord.z = length / width at the join (4.0), texcoord.xy of the centre = 0
    assert np.allclose(joint["t"][:, 2], 4.0) and np.allclose(joint["t"][0][:2], 0.0, atol=1e-6)
    # a sharper limit clips the miter: two distinct clip points (stroke.rs:79-90)
    path.stroke_options = stroke_options(width=1.0, closed=False, miter_clip=0.6)
    o, vo, io, d = tessellate([path], SOLID)
    joint = d["joint"].view(np.dtype([("p", "<f4", 2), ("t", "<f4", 3), ("u", "<u4")]))
    assert not np.allclose(joint["p"][3], joint["p"][4])


def test_parallel_join_is_skipped_and_degenerate_line_is_skipped():
    """stroke.rs:62-65 (|dot - 1| <= 1e-4 returns early) and stroke.rs:267-269 (NaN tangent -> continue)."""
    path = Path.from_polygon([(0.0, 0.0), (1.0, 0.0), (1.0, 0.0), (3.0, 0.0)])
    path.stroke_options = stroke_options(width=0.5, closed=False)
    o, vo, io, d = tessellate([path], SOLID)
    assert o.status() == 0
    assert len(d["joint"]) == 0
    # cap pair + line 1 + (zero-length line skipped) + (collinear join skipped) line 3 + 4 end-cap vertices
    assert len(d["line"]) == 20 * (2 + 2 + 2 + 4)


def test_cubic_fill_implicit_function_vanishes_on_the_curve():
    """Property: k^3 - l m n interpolated over the emitted triangles is ~0 on the curve and its sign differs
    on the two sides (fill.rs:34-114). Evaluated through the software rasterizer's attribute planes: compare coverage with a
    flattened-polygon point-in-polygon test."""
    from oracle import Oracle
    from contrast_renderer_amd import scenes
    p = Path(start=(-0.8, -0.5))
    p.push_integral_cubic_curve((-0.6, 0.9), (0.5, 1.0), (0.8, -0.4))
    p.push_rational_cubic_curve((1.0, 0.7, 1.6, 1.2), (0.6, -0.9), (-0.3, -1.0), (-0.8, -0.5))
    batch = batch_from_shapes([([], [p])])
    o = Oracle(batch)
    assert o.status() == 0
    size = 256
    img = o.render(size, size, 1, 4, scenes.place(size, size, np.array([128.0]), np.array([128.0]), np.array([100.0])), np.array([[1, 1, 1, 1]], dtype=np.float32))
    # flatten both curves densely
    def bez(P, W, n=400):
        t = np.linspace(0, 1, n)[:, None]
        B = np.concatenate([(1 - t) ** 3, 3 * t * (1 - t) ** 2, 3 * t * t * (1 - t), t ** 3], axis=1) * W
        return (B @ P) / B.sum(axis=1, keepdims=True)
    c1 = bez(np.array([(-0.8, -0.5), (-0.6, 0.9), (0.5, 1.0), (0.8, -0.4)]), np.ones(4))
    c2 = bez(np.array([(0.8, -0.4), (0.6, -0.9), (-0.3, -1.0), (-0.8, -0.5)]), np.array([1.0, 0.7, 1.6, 1.2]))
    poly = np.concatenate([c1, c2]) * 100.0 + 128.0
    ys, xs = np.mgrid[0:size, 0:size]
    px, py = xs + 0.5, size - (ys + 0.5)  # pixel centres in y-up scene units
    inside = np.zeros((size, size), dtype=bool)
    x0, y0 = poly[:-1, 0], poly[:-1, 1]
    x1, y1 = poly[1:, 0], poly[1:, 1]
    for a, b, c, e in zip(x0, y0, x1, y1):
        cond = ((b > py) != (e > py)) & (px < (c - a) * (py - b) / (e - b + 1e-30) + a)
        inside ^= cond
    covered = img[..., 3] > 127
    mismatch = covered != inside
    # only pixels within ~1 px of the outline may differ
    assert mismatch.sum() < 0.02 * inside.sum(), (mismatch.sum(), inside.sum())
    from scipy.ndimage import binary_erosion, binary_dilation
    core = binary_erosion(inside, iterations=2)
    outer = ~binary_dilation(inside, iterations=2)
    assert covered[core].all() and not covered[outer].any()


def test_reference_panics_are_surfaced_as_status_codes():
    """Non-finite coordinates (safe_float.rs:46) cannot enter through the Path mirror; a collinear cubic trips fill.rs:178."""
    with pytest.raises(ValueError):
        Path(start=(0.0, 0.0)).push_line((float("nan"), 0.0))
    p = Path(start=(0.0, 0.0))
    p.push_integral_cubic_curve((1.0, 0.0), (2.0, 0.0), (3.0, 0.0))
    o, vo, io, d = tessellate([p])
    assert o.status() == 7
    # stroke group out of range: renderer.rs:189-191
    q = Path.from_rect((0.0, 0.0), (1.0, 1.0))
    q.stroke_options = stroke_options()
    q.stroke_options.dynamic_stroke_options_group = 3
    o, vo, io, d = tessellate([q], SOLID)
    assert o.status() == 5


def test_stale_iterator_quirk_of_skipped_curves():
    """stroke.rs peeks curve segments before the NaN-tangent `continue` and advances the typed iterator only afterwards
    (stroke.rs:229 vs :318), so a fully degenerate curve segment is re-read by the next segment of its type. The oracle
    restates that: the second quadratic is stroked with the FIRST record (all three points equal) and is skipped too."""
    p = Path(start=(1.0, 1.0))
    p.push_integral_quadratic_curve((1.0, 1.0), (1.0, 1.0))  # degenerate: NaN tangents -> skipped, iterator not advanced
    p.push_integral_quadratic_curve((2.0, 2.0), (3.0, 1.0))  # reads the stale record again -> also skipped
    p.push_line((4.0, 1.0))
    p.stroke_options = stroke_options(width=0.2, closed=False, steps=3)
    o, vo, io, d = tessellate([p], SOLID)
    assert o.status() == 0
    # only the Line is stroked, from the un-advanced previous_control_point (1,1): cap pair + line end + 4 cap vertices
    line = d["line"].view(np.dtype([("p", "<f4", 2), ("t", "<f4", 2), ("u", "<u4")]))
    assert len(line) == 2 + 2 + 4 and len(d["joint"]) == 0
    assert np.allclose(line["p"][2], (4.0, 1.1))


def test_canonical_scenes_are_tessellable():
    """The benchmark generators must not emit input on which the reference panics (SURVEY.md §7 hard part 6)."""
    from oracle import Oracle
    from contrast_renderer_amd import scenes
    for sc in (scenes.scene_quadratic(100), scenes.scene_cubic_fill(10000), scenes.scene_dashed_strokes(300, (1024, 1024)), scenes.scene_mixed()):
        o = Oracle(sc["batch"], 4)
        assert o.status() == 0, sc["name"]


# ---- KAT-G / H / I: fragment helpers and descriptor packing, worked out by hand from shaders.wgsl:165-231 and renderer.rs:29-60 ----------
# (no third-party semantics involved: WGSL comparisons, `%` on positive operands and integer shifts)

def test_kat_g_cap_all_seven_types():
    """shaders.wgsl:165-189 at hand-picked texcoords; the numbers in the comments are the two sides of each comparison."""
    from oracle.binding import cap
    SQUARE, ROUND, OUT, IN, RIGHT, LEFT, BUTT = range(7)
    table = [
        (SQUARE, 0.3, 0.6, True), (SQUARE, 0.3, 0.5, False),                    # y > 0.5
        (ROUND, 0.3, 0.39, True), (ROUND, 0.3, 0.41, False),                    # 0.09 + 0.1521 = 0.2421 < 0.25 ; 0.09 + 0.1681 = 0.2581
        (OUT, 0.2, 0.25, True), (OUT, -0.2, 0.25, True), (OUT, 0.3, 0.25, False),   # 0.5 - 0.25 = 0.25 > |x|
        (IN, 0.4, 0.3, True), (IN, -0.4, 0.3, True), (IN, 0.2, 0.3, False),      # y < |x|
        (RIGHT, 0.2, 0.25, True), (RIGHT, 0.3, 0.25, False), (RIGHT, -0.4, 0.25, True),  # 0.5 - y > x (signed x)
        (LEFT, -0.2, 0.25, True), (LEFT, -0.3, 0.25, False), (LEFT, 0.4, 0.25, True),    # y - 0.5 = -0.25 < x
        (BUTT, 0.0, -0.01, True), (BUTT, 0.0, 0.0, False),                      # y < 0
        (BUTT | 0x30, 0.0, -1.0, True), (SQUARE | 0x70, 0.0, 0.75, True),       # only the low nibble selects (cap_type & 15)
        (9, 0.0, -1.0, True), (9, 0.0, 1.0, False),                             # any other value: the default arm = Butt
    ]
    for cap_type, x, y, expect in table:
        assert cap(x, y, cap_type) is expect, (cap_type, x, y)


def _descriptor(pattern, phase, join=0):
    from contrast_renderer_amd import Cap, DashInterval, DynamicStrokeOptions, Join
    from oracle.binding import _load
    import ctypes as C
    from contrast_renderer_amd import _ffi
    o = DynamicStrokeOptions.Dashed(Join(join), [DashInterval(a, b, Cap(s), Cap(e)) for a, b, s, e in pattern], phase)
    out = _ffi.DynamicStrokeDescriptorC()
    rc = _load().oracle_convert_dynamic_stroke_options(C.byref(o.to_c()), C.byref(out))
    assert rc == 0
    return out


def test_kat_h_descriptor_packing_for_two_three_and_four_intervals():
    """renderer.rs:29-60: count_dashed_join = (len - 1) << 3 | 4 | join; dash_start of interval i goes to byte (i + len - 1) % len, low
    nibble... of the PREVIOUS interval's byte; dash_end of interval i to the high nibble of byte i."""
    BUTT, ROUND, OUT, IN, RIGHT = 6, 1, 2, 3, 4
    d = _descriptor([(2.0, 3.0, ROUND, OUT), (5.0, 6.0, IN, RIGHT)], 0.25, join=2)
    assert d.count_dashed_join == (1 << 3) | 4 | 2
    assert list(d.gap_start) == [2.0, 5.0, 0.0, 0.0] and list(d.gap_end) == [3.0, 6.0, 0.0, 0.0] and d.phase == 0.25
    # interval 0: start ROUND -> byte (0 + 1) % 2 = 1, end OUT -> byte 0 high nibble; interval 1: start IN -> byte 0, end RIGHT -> byte 1 high nibble
    assert d.caps == (IN | (OUT << 4)) | ((ROUND | (RIGHT << 4)) << 8)
    d = _descriptor([(1.0, 2.0, 0, 1), (3.0, 4.0, 2, 3), (5.0, 7.0, 4, 5)], 0.0, join=1)
    assert d.count_dashed_join == (2 << 3) | 4 | 1
    # starts: i=0 -> byte 2, i=1 -> byte 0, i=2 -> byte 1 ; ends: byte i high nibble
    assert d.caps == (2 | (1 << 4)) | ((4 | (3 << 4)) << 8) | ((0 | (5 << 4)) << 16)
    d = _descriptor([(1.0, 2.0, 6, 6), (3.0, 4.0, 1, 1), (5.0, 6.0, 2, 2), (7.0, 9.0, 3, 3)], 1.5)
    assert d.count_dashed_join == (3 << 3) | 4 | 0 and list(d.gap_end) == [2.0, 4.0, 6.0, 9.0]
    # starts: 0 -> byte 3, 1 -> byte 0, 2 -> byte 1, 3 -> byte 2
    assert d.caps == (1 | (6 << 4)) | ((2 | (1 << 4)) << 8) | ((3 | (2 << 4)) << 16) | ((6 | (3 << 4)) << 24)


def test_kat_i_stroke_dashed_interval_selection():
    """shaders.wgsl:205-231 with Butt caps everywhere (cap = y < 0: never true for the positive gap distances, so a sample is filled
    exactly when it is NOT inside a gap): pattern length = gap_end[last]; position = (y - phase) mod length; the first interval whose
    gap_end >= position is selected (or the last one); inside its gap <=> position > gap_start."""
    from oracle.binding import stroke_dashed
    BUTT = 6
    d = _descriptor([(2.0, 3.0, BUTT, BUTT), (5.0, 6.0, BUTT, BUTT)], 0.0)  # dashes [0,2] and [3,5], gaps (2,3] and (5,6], period 6
    for y, filled in [(0.0, True), (1.99, True), (2.0, True), (2.01, False), (2.99, False), (3.0, False), (3.01, True), (4.5, True), (5.0, True),
                      (5.5, False), (6.0 + 1.0, True), (6.0 + 2.5, False), (12.0 + 5.75, False), (-0.5, False), (-1.5, True), (-6.0 + 2.5, False)]:
        assert stroke_dashed(d, 0.0, y) is filled, y   # negative positions wrap: -0.5 -> 5.5 (gap), -1.5 -> 4.5 (dash)
    d = _descriptor([(2.0, 3.0, BUTT, BUTT), (5.0, 6.0, BUTT, BUTT)], 1.0)  # phase shifts the pattern by +1
    for y, filled in [(1.0, True), (3.0, True), (3.5, False), (4.0, False), (4.01, True), (6.5, False), (6.99, False), (7.0, True), (7.5, True)]:  # position = (y - 1) mod 6: 7.0 -> 0 (a new period starts)
        assert stroke_dashed(d, 0.0, y) is filled, y
    # with Round caps the dash ends grow half-discs into the gap: a sample at distance 0.3 behind the dash end, on the centre line, is inside
    ROUND = 1
    d = _descriptor([(2.0, 3.0, ROUND, ROUND), (5.0, 6.0, ROUND, ROUND)], 0.0)
    assert stroke_dashed(d, 0.0, 2.3) is True       # gap_start distance 0.3: 0.09 < 0.25
    assert stroke_dashed(d, 0.0, 2.5) is False      # 0.5 from either end: 0.25 < 0.25 is false
    assert stroke_dashed(d, 0.45, 2.3) is False     # off the centre line: 0.2025 + 0.09 = 0.2925 (start cap), 0.2025 + 0.49 (end cap)


def test_kat_j_orientation_sign_of_the_winding_contribution():
    """path.rs:210-211: "Filled Paths increment the winding counter when they are counterclockwise and decrement it when they are clockwise."
    What pixels can pin of that sentence is the RELATIVE sign (the stencil value itself is only ever tested against zero, renderer.rs:736-754):
    two overlapping rectangles in one Shape — same direction of travel: the overlap carries winding +-2 and is covered under the non-zero
    rule (winding_counter_bits = 8), not under even-odd (bits = 1); opposite directions: the contributions cancel in the overlap, which
    stays uncovered under either rule. The absolute sign follows from the code, not from the sentence: from_rect's vertex
    order (path.rs:736-743) is clockwise in y-up user coordinates, vertex.rs:28-35 reverses the facing of a polygon when it turns the fan
    into a strip, front = counter-clockwise ON SCREEN (renderer.rs:477) increments (renderer.rs:577-582) — so a path that is clockwise in
    y-up coordinates, i.e. counter-clockwise by the signed area of its y-DOWN (screen, SVG-style) coordinates, increments. Read in y-down
    coordinates the sentence and the code agree; read in y-up coordinates the sentence has the sign backwards."""
    from oracle import Oracle
    W = H = 64
    a = Path.from_rect((24.0, 32.0), (16.0, 16.0))   # x 8..40, y 16..48
    b = Path.from_rect((40.0, 32.0), (16.0, 16.0))   # x 24..56: overlap x 24..40
    b_reversed = Path.from_rect((40.0, 32.0), (16.0, 16.0))
    b_reversed.reverse()
    transform = np.zeros((1, 16), dtype=np.float32)  # user (x, y-up) in pixels -> clip space
    transform[0, 0], transform[0, 5], transform[0, 10], transform[0, 15], transform[0, 12], transform[0, 13] = 2.0 / W, 2.0 / H, 1.0, 1.0, -1.0, -1.0
    color = np.array([[1.0, 1.0, 1.0, 1.0]], dtype=np.float32)

    def covered(paths, bits):
        o = Oracle(batch_from_shapes([([], paths)]))
        assert o.status() == 0
        return o.render(W, H, 1, bits, transform, color)[..., 3] > 0

    inside_a, overlap, inside_b = (slice(20, 44), slice(10, 22)), (slice(20, 44), slice(26, 38)), (slice(20, 44), slice(42, 54))  # (rows, columns), away from the edges
    same8, same1 = covered([a, b], 8), covered([a, b], 1)
    assert same8[inside_a].all() and same8[overlap].all() and same8[inside_b].all()          # +-1, +-2, +-1: all non-zero
    assert same1[inside_a].all() and not same1[overlap].any() and same1[inside_b].all()      # even-odd: 2 = 0 (mod 2)
    for bits in (8, 4, 1):
        opposite = covered([a, b_reversed], bits)
        assert opposite[inside_a].all() and not opposite[overlap].any() and opposite[inside_b].all()  # +1 - 1 = 0 whatever the counter width
    # the direction of from_rect (path.rs:736-743) in y-up user coordinates: clockwise = negative signed area
    pts = np.array([a.start] + [rec[-2:] for rec in a.records], dtype=np.float64)
    x, y = pts[:, 0], pts[:, 1]
    assert 0.5 * np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y) < 0.0


def _stroke_over_fill_case():
    """One Shape: a stroked horizontal line (width 8) crossed by a filled rectangle — clockwise in y-up user coordinates as from_rect makes
    it (path.rs:736-743), or reverse()d. Returns (paths builder, transform, colour, regions) for a 64 x 64 frame."""
    W = H = 64
    line = Path(start=(8.0, 32.0))
    line.push_line((56.0, 32.0))
    line.stroke_options = StrokeOptions(8.0, 0.0, 4.0, False, 0, CurveApproximation.UniformlySpacedParameters(1))
    dynamic = [DynamicStrokeOptions.Solid(Join.Miter, Cap.Butt, Cap.Butt)]
    transform = np.zeros((1, 16), dtype=np.float32)  # user (x, y-up) in pixels -> clip space
    transform[0, 0], transform[0, 5], transform[0, 10], transform[0, 15], transform[0, 12], transform[0, 13] = 2.0 / W, 2.0 / H, 1.0, 1.0, -1.0, -1.0
    color = np.array([[1.0, 1.0, 1.0, 1.0]], dtype=np.float32)

    def shape(reverse_fill):
        rect = Path.from_rect((32.0, 32.0), (8.0, 16.0))  # x 24..40, y 16..48: crosses the stroke band y 28..36
        if reverse_fill:
            rect.reverse()
        return batch_from_shapes([(dynamic, [line, rect])])
    # (rows, columns) well inside: the stroke alone, the fill alone (above the band), stroke and fill together; rows count from the top, y is up
    regions = dict(stroke_only=(slice(29, 35), slice(10, 22)), fill_only=(slice(18, 26), slice(26, 38)), both=(slice(29, 35), slice(26, 38)))
    return shape, transform, color, regions, W, H


def test_kat_k_a_stroke_and_a_fill_overlapping_in_one_shape_show_the_absolute_sign():
    """The ONE place where the absolute sign of a filled path's winding contribution reaches a pixel. Shape::render(Stencil) draws the
    strokes first — stencil Equal(0) -> IncrementWrap on both faces, i.e. "set to 1 once" (renderer.rs:275-303, 571-576) — and the fills
    behind them with IncrementWrap / DecrementWrap by facing (renderer.rs:304-336, 577-582); the cover keeps a sample iff the counter is not
    zero (renderer.rs:736-754). Where a stroke and a fill of the same Shape overlap the counter is 1 + 1 = 2 for a fill that increments
    (covered under the non-zero rule; 0 under even-odd, bits = 1) and 1 - 1 = 0 for a fill that decrements (a HOLE under every rule).
    By the chain of facts in KAT-J / DESIGN.md §2 a rectangle as from_rect emits it — clockwise in y-up user coordinates — increments when
    the instance transform preserves orientation, so the restatement covers the overlap at bits 4 and 8 and punches it out at bits 1,
    and the reverse()d rectangle punches it out at every width. If the real crate behaved by the y-up reading of path.rs:210-211
    ("increment when counterclockwise") these two rows would swap — which is what showcase/main.rs:82-84 (every glyph path reverse()d
    in a Shape that also strokes a rounded rectangle) suggests its author expects; DESIGN.md §2 says which un-vendored conventions decide."""
    from oracle import Oracle
    shape, transform, color, regions, W, H = _stroke_over_fill_case()
    for reverse_fill, expect_both in ((False, {8: True, 4: True, 1: False}), (True, {8: False, 4: False, 1: False})):
        for bits, both_covered in expect_both.items():
            o = Oracle(shape(reverse_fill))
            assert o.status() == 0
            covered = o.render(W, H, 1, bits, transform, color)[..., 3] > 0
            assert covered[regions["stroke_only"]].all() and covered[regions["fill_only"]].all(), (reverse_fill, bits)
            assert covered[regions["both"]].all() == both_covered and covered[regions["both"]].any() == both_covered, (reverse_fill, bits)
