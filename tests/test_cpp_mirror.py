"""The C++ host mirror of the reference API (include/contrast_renderer.hpp): it must compile as plain C++17 against the C ABI (CPU
test), and an application written against it must produce exactly what the oracle says (GPU test: tessellation bytes, a plain pass, a
recorded pass with clipping + instancing, and the reference's error variants)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "cpp", "mirror_harness.cpp")
FONT = os.path.join(ROOT, "contrast_renderer_amd", "data", "fonts", "OpenSans-Regular.ttf")


def build_harness(out_dir):
    import __graft_entry__ as entry
    entry.build()
    lib_dir = os.path.join(ROOT, "contrast_renderer_amd")
    exe = os.path.join(out_dir, "mirror_harness")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), HARNESS, "-o", exe,
           "-L", lib_dir, "-lcontrast_hip", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_cpp_mirror_compiles_and_links_against_the_c_abi():
    with tempfile.TemporaryDirectory() as tmp:
        try:
            exe = build_harness(tmp)
        except subprocess.CalledProcessError as e:
            raise AssertionError(e.stderr)
        assert os.path.getsize(exe) > 0


def python_scene():
    from contrast_renderer_amd import (Cap, CurveApproximation, DashInterval, DynamicStrokeOptions, Join, Path, StrokeOptions, batch_from_shapes)
    from contrast_renderer_amd import text as T
    font = T.Font("OpenSans", open(FONT, "rb").read())
    shapes = [([], [Path.from_rect((0.0, 0.0), (0.75, 0.5))])]
    p = Path(start=(0.5, 0.0))
    p.push_integral_quadratic_curve((0.5, 0.5), (0.0, 0.5))
    p.push_integral_quadratic_curve((-0.5, 0.5), (-0.5, 0.0))
    p.push_rational_quadratic_curve(0.75, (-0.5, -0.5), (0.0, -0.5))
    p.push_line((0.5, -0.25))
    p.stroke_options = StrokeOptions(0.125, 0.25, 2.0, True, 0, CurveApproximation.UniformTangentAngle(0.25))
    shapes.append(([DynamicStrokeOptions.Solid(Join.Round, Cap.Butt, Cap.Butt)], [p]))
    p = Path(start=(-0.5, -0.25))
    p.push_integral_cubic_curve((-0.25, 0.75), (0.25, 0.75), (0.5, -0.25))
    p.push_rational_cubic_curve((1.0, 1.5, 1.5, 1.0), (0.25, -0.75), (-0.25, -0.75), (-0.5, -0.25))
    shapes.append(([], [p]))
    shapes.append(([], T.paths_of_text(font, T.Layout(1.0, T.Orientation.LeftToRight, T.Alignment.Center, T.Alignment.Center), "g8")))
    p = Path.from_polygon([(-0.75, -0.5), (-0.25, 0.5), (0.25, -0.5), (0.75, 0.5)])
    p.stroke_options = StrokeOptions(0.0625, 0.0, 4.0, False, 0, CurveApproximation.UniformlySpacedParameters(4))
    shapes.append(([DynamicStrokeOptions.Dashed(Join.Miter, [DashInterval(0.5, 1.0, Cap.Round, Cap.Out), DashInterval(2.0, 2.5, Cap.Butt, Cap.Square)], 0.25)], [p]))
    batch = batch_from_shapes(shapes)
    place = [(-0.5, 0.5, 0.4), (0.5, 0.5, 0.4), (-0.5, -0.5, 0.4), (0.5, -0.5, 0.6), (0.0, 0.0, 0.9)]
    transforms = np.zeros((6, 16), dtype=np.float32)
    for i, (cx, cy, s) in enumerate(place):
        transforms[i, [0, 5, 10, 12, 13, 15]] = (s, s, 1.0, cx, cy, 1.0)
    colors = np.array([[1, 0, 0, 1], [0, 0.5, 1, 0.75], [0, 1, 0, 0.5], [0.25, 0.25, 0.25, 1], [1, 0.5, 0, 1], [0, 0, 1, 1]], dtype=np.float32)
    return batch, transforms, colors


@pytest.mark.gpu
def test_cpp_application_matches_the_oracle(oracle_lib):
    from contrast_renderer_amd.renderer import RenderOperation as Op
    from oracle.binding import Oracle, render_draws
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_harness(tmp)
        out = os.path.join(tmp, "out.bin")
        run = subprocess.run([exe, FONT, out], capture_output=True, text=True)
        assert run.returncode == 0, run.stderr
        assert "3 reference errors reproduced" in run.stdout
        assert "geometry 3 1 4 1 7 8 3 1.500000" in run.stdout  # tests/test_text_cpu.py::test_text_geometry_and_its_cursor_helpers
        blob = np.fromfile(out, dtype=np.uint8)
    batch, transforms, colors = python_scene()
    oracle = Oracle(batch)
    assert oracle.status() == 0
    at = 0

    def take(n):
        nonlocal at
        part = blob[at:at + n]
        at += n
        return part
    assert int(take(8).view(np.uint64)[0]) == 5
    for s in range(5):
        vo, io, vb, ib = oracle.shape(s)
        assert np.array_equal(take(64).view(np.uint64), vo) and np.array_equal(take(24).view(np.uint64), io), f"shape {s} offsets"
        assert np.array_equal(take(len(vb)), vb), f"shape {s} vertex bytes"
        assert np.array_equal(take(len(ib)), ib), f"shape {s} index bytes"
    width, height = 160, 128
    plain = take(width * height * 4).reshape(height, width, 4)
    assert np.array_equal(plain, oracle.render(width, height, 4, 4, transforms[:5], colors[:5]))
    t2 = transforms.copy()
    t2[0, [0, 5, 12, 13]] = (1.2, 1.0, 0.0, 0.0)
    t2[5, [0, 5, 10, 12, 13, 15]] = (0.3, 0.3, 1.0, 0.6, -0.6, 1.0)
    draws = [(0, 0, Op.Stencil, 0, 0), (0, 0, Op.Clip, 1, 0), (3, 3, Op.Stencil, 1, 0), (3, 3, Op.Color, 1, 0), (4, 4, Op.Stencil, 1, 0), (4, 4, Op.Color, 1, 0),
             (0, 0, Op.UnClip, 0, 0), (1, 1, Op.Stencil, 0, 0), (1, 1, Op.Color, 0, 0), (2, 2, Op.Stencil, 0, 0), (2, 2, Op.Color, 0, 0),
             (2, 5, Op.Stencil, 0, 0), (2, 5, Op.Color, 0, 0)]
    recorded = take(width * height * 4).reshape(height, width, 4)
    expect = render_draws(oracle, width, height, 4, 4, 4, 1, t2, colors, [tuple(int(v) for v in d) for d in draws])
    assert np.array_equal(recorded, expect)
    assert not np.array_equal(recorded, plain) and (recorded[..., 3] > 0).mean() > 0.05
    # perspective instances + depth attachment + cull mode through the C++ mirror
    from oracle.binding import render_pass
    t3d = take(5 * 64).view(np.float32).reshape(5, 16)
    assert (t3d[:, 11] == 1.0).all() and (t3d[:, 15] > 1.0).all()  # clip.w = view z: perspective
    image3d = take(width * height * 4).reshape(height, width, 4)
    depth3d = take(width * height * 4).view(np.float32).reshape(height, width, 1)
    wall = np.ones((height, width, 1), dtype=np.float32)
    wall[:, :40] = 0.25
    plain_draws = [d for i in range(5) for d in ((i, i, int(Op.Stencil), 0, 0), (i, i, int(Op.Color), 0, 0))]
    o3 = Oracle(batch)
    expect3d, expect_depth = render_pass(o3, width, height, 1, 4, 2, 0, t3d, colors[:5], plain_draws, cull_mode=2, depth_compare=4, depth_write=1, depth=wall)
    assert np.array_equal(image3d, expect3d) and np.array_equal(depth3d, expect_depth)
    assert (image3d[..., 3] > 0).mean() > 0.03 and not (image3d[:, :40, 3] > 0).any() and (depth3d < 1.0).mean() > 0.25
    assert int(take(8).view(np.uint64)[0]) == 3
    n = int(take(8).view(np.uint64)[0])
    assert take(n).view(np.float32).reshape(-1, 2).tolist() == [[2, 2], [4, 2], [2, 6], [4, 6], [2, 2], [4, 2], [2, 6], [4, 6]]  # KAT-A through Shape::from_paths
    assert at == len(blob)
