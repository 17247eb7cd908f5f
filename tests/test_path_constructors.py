"""Path constructors and conversions of path.rs in the two host mirrors: geometric known answers and properties
for the Python mirror, bit-for-bit agreement of the C++ mirror (include/contrast_renderer.hpp, compiled with g++) with it. Host code only."""
import math
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from contrast_renderer_amd import Path, SegmentType

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = SegmentType


def mixed_path():
    p = Path(start=(0.1, 0.2))
    p.push_line((1.0, 0.3))
    p.push_integral_quadratic_curve((1.5, 1.0), (0.7, 1.4))
    p.push_rational_quadratic_curve(0.6, (0.2, 1.9), (-0.4, 1.1))
    p.push_integral_cubic_curve((-0.9, 0.8), (-0.8, 0.1), (-0.3, -0.2))
    p.push_rational_cubic_curve((1.0, 1.3, 0.7, 1.0), (0.0, -0.6), (0.3, -0.5), (0.4, -0.1))
    return p


def clone(p):
    q = Path(start=p.start)
    q.segment_types, q.records = list(p.segment_types), list(p.records)
    return q


def rational_quadratic_point(p0, w, p1, p2, t):
    b0, b1, b2 = (1 - t) ** 2, 2 * t * (1 - t) * w, t * t
    d = b0 + b1 + b2
    return ((b0 * p0[0] + b1 * p1[0] + b2 * p2[0]) / d, (b0 * p0[1] + b1 * p1[1] + b2 * p2[1]) / d)


def test_rounded_rect_ellipse_circle_known_answers():
    p = Path.from_rounded_rect((0.0, 0.0), (1.0, 0.5), 0.25)
    assert p.start == (0.75, -0.5) and [int(t) for t in p.segment_types] == [0, 3] * 4
    assert p.records[0] == (-0.75, -0.5) and p.records[1][1:] == (-1.0, -0.5, -1.0, -0.25)
    assert p.get_end() == p.start  # closed
    c = Path.from_circle((1.0, 2.0), 0.5)
    assert c.start == (1.0, 1.5) and [int(t) for t in c.segment_types] == [3] * 4
    # every quarter is an exact circular arc: weight cos(45 deg), points at distance r from the centre
    previous = c.start
    for w, cx, cy, x, y in c.records:
        assert abs(w - math.sqrt(0.5)) < 1e-7
        for t in (0.0, 0.3, 0.5, 0.8, 1.0):
            px, py = rational_quadratic_point(previous, w, (cx, cy), (x, y), t)
            assert abs(math.hypot(px - 1.0, py - 2.0) - 0.5) < 1e-6
        previous = (x, y)
    e = Path.from_ellipse((0.0, 0.0), (2.0, 1.0))
    previous = e.start
    for w, cx, cy, x, y in e.records:
        px, py = rational_quadratic_point(previous, w, (cx, cy), (x, y), 0.37)
        assert abs((px / 2.0) ** 2 + py ** 2 - 1.0) < 1e-6
        previous = (x, y)


def test_reverse_convert_close():
    p = mixed_path()
    r = clone(p)
    r.reverse()
    assert r.start == p.get_end() and r.get_end() == p.start
    assert [int(t) for t in r.segment_types] == [int(t) for t in reversed(p.segment_types)]
    assert r.records[0][:4] == (0.7, 1.3, 1.3, 1.0) or np.allclose(r.records[0][:4], (1.0, 0.7, 1.3, 1.0))  # weights reversed
    assert np.allclose(r.records[0][:4], (1.0, 0.7, 1.3, 1.0))
    r.reverse()
    assert r.start == p.start and r.records == p.records and r.segment_types == p.segment_types
    c = clone(p)
    c.convert_integral_curves_to_rational_curves()
    assert [int(t) for t in c.segment_types] == [0, 3, 3, 4, 4]
    assert c.records[1] == (1.0,) + p.records[1] and c.records[3] == (1.0, 1.0, 1.0, 1.0) + p.records[3]
    d = clone(p)
    d.convert_quadratic_curves_to_cubic_curves()
    assert [int(t) for t in d.segment_types] == [0, 2, 4, 2, 4]
    # degree elevation keeps the curve: compare points of the quadratic and of the cubic at a few parameters
    p0, (ax, ay, bx, by) = p.records[0], p.records[1]
    c0x, c0y, c1x, c1y, ex, ey = d.records[1]
    for t in (0.25, 0.5, 0.75):
        q = ((1 - t) ** 2 * p0[0] + 2 * t * (1 - t) * ax + t * t * bx, (1 - t) ** 2 * p0[1] + 2 * t * (1 - t) * ay + t * t * by)
        k = ((1 - t) ** 3 * p0[0] + 3 * t * (1 - t) ** 2 * c0x + 3 * t * t * (1 - t) * c1x + t ** 3 * ex,
             (1 - t) ** 3 * p0[1] + 3 * t * (1 - t) ** 2 * c0y + 3 * t * t * (1 - t) * c1y + t ** 3 * ey)
        assert np.allclose(q, k, atol=1e-6)
    w, qx, qy, zx, zy = p.records[2]
    w0, w1, w2, w3, a0x, a0y, a1x, a1y, fx, fy = d.records[2]
    start = p.records[1][-2:]
    for t in (0.25, 0.5, 0.75):
        q = rational_quadratic_point(start, w, (qx, qy), (zx, zy), t)
        b = [(1 - t) ** 3 * w0, 3 * t * (1 - t) ** 2 * w1, 3 * t * t * (1 - t) * w2, t ** 3 * w3]
        pts = [start, (a0x, a0y), (a1x, a1y), (fx, fy)]
        k = (sum(bi * pt[0] for bi, pt in zip(b, pts)) / sum(b), sum(bi * pt[1] for bi, pt in zip(b, pts)) / sum(b))
        assert np.allclose(q, k, atol=1e-6)
    e = clone(p)
    e.close()
    assert int(e.segment_types[-1]) == 0 and e.records[-1] == p.start
    n = len(e.records)
    e.close()
    assert len(e.records) == n  # already closed
    other = mixed_path()
    p.append(other)
    assert other.records == [] and len(p.records) == 5  # path.rs:376-384 as written


def test_elliptical_arc_properties():
    rx, ry, phi = 1.5, 0.75, 0.4
    start, to = (1.0, 0.25), (-0.5, 1.0)
    seen = set()
    for large in (False, True):
        for sweep in (False, True):
            a = Path(start=start)
            a.push_elliptical_arc((rx, ry), phi, large, sweep, to)
            assert 1 <= len(a.records) <= 3 and all(int(t) == 3 for t in a.segment_types)
            assert np.allclose(a.get_end(), to, atol=1e-5)
            # all four arcs lie on an ellipse with these radii and this rotation: recover the centre from two points, check the rest
            previous, pts, total = start, [], 0.0
            for w, cx, cy, x, y in a.records:
                assert 0.0 < w <= 1.0
                for t in (0.0, 0.25, 0.5, 0.75, 1.0):
                    pts.append(rational_quadratic_point(previous, w, (cx, cy), (x, y), t))
                previous = (x, y)
            c, s_ = math.cos(phi), math.sin(phi)
            def residual(center):
                out = []
                for px, py in pts:
                    dx, dy = px - center[0], py - center[1]
                    u, v = dx * c + dy * s_, -dx * s_ + dy * c
                    out.append((u / rx) ** 2 + (v / ry) ** 2 - 1.0)
                return np.abs(out).max()
            # the centre is one of the two SVG solutions; find it by a coarse-to-fine search around the chord midpoint
            best = min(((residual((mx, my)), (mx, my)) for mx in np.linspace(-2, 2.5, 91) for my in np.linspace(-1.5, 2.5, 81)), key=lambda r: r[0])[1]
            for _ in range(5):
                span = 0.06
                best = min(((residual((mx, my)), (mx, my)) for mx in np.linspace(best[0] - span, best[0] + span, 25) for my in np.linspace(best[1] - span, best[1] + span, 25)),
                           key=lambda r: r[0])[1]
            assert residual(best) < 5e-3
            # the swept angle is > pi exactly for the large arcs
            angles = [math.atan2((-(px - best[0]) * s_ + (py - best[1]) * c) / ry, ((px - best[0]) * c + (py - best[1]) * s_) / rx) for px, py in pts]
            swept = sum(abs((b - a_ + math.pi) % (2 * math.pi) - math.pi) for a_, b in zip(angles, angles[1:]))
            assert (swept > math.pi) == large
            seen.add(tuple(round(v, 3) for v in a.records[0]))
    assert len(seen) == 4  # four different arcs
    z = Path(start=(0.0, 0.0))
    z.push_elliptical_arc((0.0, 1.0), 0.0, False, True, (1.0, 1.0))
    assert [int(t) for t in z.segment_types] == [0] and z.records == [(1.0, 1.0)]  # zero radius: a line (path.rs:642-645)


def test_tangents():
    p = mixed_path()
    c, nx, ny = p.get_end_tangent()
    assert abs(math.hypot(nx, ny) - 1.0) < 1e-6
    # end tangent of the last (rational cubic) segment: the line through its last two control points, normal = direction rotated clockwise
    dx, dy = 0.4 - 0.3, -0.1 - (-0.5)
    assert np.allclose((nx, ny), (dy / math.hypot(dx, dy), -dx / math.hypot(dx, dy)), atol=1e-6)
    assert Path(start=(0.0, 0.0)).get_start_tangent() == (0.0, 0.0, 0.0)


def python_dump():
    def bits(v):
        return "%08x" % struct.unpack("<I", struct.pack("<f", v))[0]

    def line(name, p):
        return " ".join([name, str(len(p.segment_types)), bits(p.start[0]), bits(p.start[1])] + ["t%d" % int(t) for t in p.segment_types] +
                        [bits(v) for r in p.records for v in r])
    out = [line("rounded_rect", Path.from_rounded_rect((0.25, -0.5), (1.0, 0.5), 0.2)), line("ellipse", Path.from_ellipse((1.0, 2.0), (0.75, 0.5))),
           line("circle", Path.from_circle((-1.0, 0.5), 0.3))]
    p = mixed_path()
    out.append(line("mixed", p))
    r = clone(p)
    r.reverse()
    out.append(line("reversed", r))
    r.reverse()
    out.append(line("reversed_twice", r))
    c = clone(p)
    c.convert_integral_curves_to_rational_curves()
    out.append(line("rational", c))
    d = clone(p)
    d.convert_quadratic_curves_to_cubic_curves()
    out.append(line("cubic", d))
    e = clone(p)
    e.close()
    out.append(line("closed", e))
    e.close()
    out.append(line("closed_again", e))
    for large in (0, 1):
        for sweep in (0, 1):
            a = Path(start=(1.0, 0.25))
            a.push_elliptical_arc((1.5, 0.75), 0.4, large, sweep, (-0.5, 1.0))
            out.append(line(f"arc_{large}{sweep}", a))
    z = Path(start=(0.0, 0.0))
    z.push_elliptical_arc((0.0, 1.0), 0.0, False, True, (1.0, 1.0))
    out.append(line("arc_zero_radius", z))
    out.append("tangents 0 " + " ".join(bits(v) for v in p.get_start_tangent() + p.get_end_tangent()))
    return out


def test_cpp_mirror_agrees_bit_for_bit():
    import __graft_entry__ as entry
    entry.build()
    lib_dir = os.path.join(ROOT, "contrast_renderer_amd")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "path_harness")
        cmd = ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
               os.path.join(ROOT, "tests", "cpp", "path_harness.cpp"), "-o", exe, "-L", lib_dir, "-lcontrast_hip", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib",
               "-Wl,-rpath-link,/opt/rocm/lib"]
        built = subprocess.run(cmd, capture_output=True, text=True)
        assert built.returncode == 0, built.stderr
        run = subprocess.run([exe], capture_output=True, text=True)
        assert run.returncode == 0, run.stderr
    assert run.stdout.strip().splitlines() == python_dump()
