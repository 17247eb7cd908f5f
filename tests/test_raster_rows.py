"""k_raster_rows (csrc/raster_edges.hip): the plain Stencil + Color pass with the winding numbers accumulated in LDS — lanes spread over
(entry, sample row), switch columns by bisection of the exact edge predicate, row prefix sums; renderer.rs:304-318,340-354,565-582,
vertex.rs:28-35, shaders.wgsl:233-266,304-309. The library picks it per Scene by measurement (and never for fewer than 256 Shapes), so
these tests pin it with CRH_ROWS=1 and compare with the oracle bit for bit: structured and unstructured filled paths, glyph runs, tile
lists of many chunks with and without the late start, deep stacks of opaque covers, every winding rule, frames whose size is no multiple
of 16 or of 4, RGBA16F targets, passes over existing content, and the per-sample kernel on the same lists."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from contrast_renderer_amd import renderer
    return renderer


def _renderer(gpu, bits=4):
    return gpu.Renderer(gpu.Configuration(msaa_sample_count=1, winding_counter_bits=bits), device=0)


def random_filled_paths(n_shapes, seed):
    """Unstructured filled outlines (self-intersecting, cusps, coincident points, weights 0.2 .. 5), one to three paths per Shape"""
    from contrast_renderer_amd import Path
    rng = np.random.RandomState(seed)
    shapes = []
    for _ in range(n_shapes):
        paths = []
        for _ in range(rng.randint(1, 4)):
            scale = float(np.exp(rng.uniform(math.log(0.05), math.log(3.0))))
            pt = lambda: (float(np.float32(rng.normal(0, scale))), float(np.float32(rng.normal(0, scale))))
            pen = pt()
            p = Path(start=pen)
            for _ in range(rng.randint(2, 10)):
                kind = rng.randint(0, 5)
                end = pen if rng.uniform() < 0.05 else pt()
                if kind == 0:
                    p.push_line(end)
                elif kind == 1:
                    p.push_integral_quadratic_curve(pt(), end)
                elif kind == 2:
                    p.push_integral_cubic_curve(pt(), pt(), end)
                elif kind == 3:
                    p.push_rational_quadratic_curve(float(np.exp(rng.uniform(-1.6, 1.6))), pt(), end)
                else:
                    p.push_rational_cubic_curve(np.exp(rng.uniform(-1.6, 1.6, 4)), pt(), pt(), end)
                pen = end
            paths.append(p)
        shapes.append(([], paths))
    return shapes


def _check(gpu, oracle_lib, sc, what, bits=None, colors=None, passes=2, fmt=None):
    bits = sc["winding_bits"] if bits is None else bits
    colors = sc["colors"] if colors is None else colors
    r = _renderer(gpu, bits)
    scene = gpu.Scene(r, sc["batch"])
    oracle = oracle_lib.Oracle(sc["batch"], 4)
    assert scene.status() == oracle.status() == 0
    expect = oracle.render(sc["width"], sc["height"], 1, bits, sc["transforms"], colors)
    frame = gpu.Frame(r, sc["width"], sc["height"]) if fmt is None else gpu.Frame(r, sc["width"], sc["height"], fmt)
    for k in range(passes):  # (the later passes run without the read-back of the list sizes, the third with the lists in place)
        frame.clear()
        scene.render(frame, sc["transforms"], colors)
        got = frame.download()
        assert np.array_equal(got, expect), f"{what}, pass {k}: {(got != expect).any(axis=2).sum()} pixels differ"
    return expect


@pytest.mark.parametrize("bits", [1, 2, 4])
def test_structured_scenes_every_winding_rule(gpu, oracle_lib, bits, monkeypatch):
    from contrast_renderer_amd import scenes
    monkeypatch.setenv("CRH_ROWS", "1")
    for sc in (scenes.scene_cubic_fill(400, (512, 512), r_lo=6.0, r_hi=64.0), scenes.scene_glyphs(300, (320, 256)), scenes.scene_cubic_fill(60, (203, 177), r_lo=3.0, r_hi=40.0),
               scenes.scene_glyphs(150, (250, 131), sizes=(64.0, 96.0))):
        _check(gpu, oracle_lib, sc, f"{sc['name']} bits {bits}", bits=bits, passes=3)


@pytest.mark.parametrize("seed", range(3))
def test_unstructured_filled_paths(gpu, oracle_lib, seed, monkeypatch):
    """Random outlines at random placements, slivers and huge coordinates included: fills that leave their hulls, hull strips that fold."""
    from contrast_renderer_amd import batch_from_shapes, scenes
    from oracle.binding import Oracle
    monkeypatch.setenv("CRH_ROWS", "1")
    shapes = random_filled_paths(500, 300 + seed)
    probe = Oracle(batch_from_shapes(shapes), 8)
    good = [s for s in range(len(shapes)) if probe.shape_status(s) == 0]
    assert len(good) > 200
    batch = batch_from_shapes([shapes[s] for s in good])
    n = batch.n_shapes
    rng = np.random.RandomState(seed)
    colors = np.concatenate([rng.uniform(0, 1, (n, 3)), np.where(rng.uniform(size=(n, 1)) < 0.5, 1.0, rng.uniform(0.2, 1, (n, 1)))], axis=1).astype(np.float32)
    for w, h, radius in ((256, 256, rng.uniform(5, 60, n)), (301, 199, np.exp(rng.uniform(math.log(0.05), math.log(4000.0), n)))):
        t = scenes.place(w, h, rng.uniform(-0.2 * w, 1.2 * w, n), rng.uniform(-0.2 * h, 1.2 * h, n), radius)
        sc = dict(batch=batch, transforms=t, colors=colors, width=w, height=h, winding_bits=4, name=f"random fills {seed}")
        _check(gpu, oracle_lib, sc, f"seed {seed} {w}x{h}", bits=[4, 1, 2][seed % 3])


@pytest.mark.parametrize("mode", ["0", "1", "restart", "debug_off"])
def test_long_lists_and_opaque_stacks(gpu, oracle_lib, mode, monkeypatch):
    """Hundreds of entries per tile in several chunks (groups and grids carried across chunk boundaries), the late start behind the last
    whole-tile reset in front of an opaque whole-tile cover (found inside one chunk, or across chunks by the LONG variant) and the walk from
    the top when that cover does not overwrite every sample (debug bit 25: always), or with the shortcut off in the binning (bit 15)."""
    from contrast_renderer_amd import scenes
    monkeypatch.setenv("CRH_ROWS", "1")
    if mode == "restart":
        monkeypatch.setenv("CRH_LONG_LISTS", "1")
        monkeypatch.setenv("CRH_RASTER_DEBUG", str(1 << 25))
    elif mode == "debug_off":
        monkeypatch.setenv("CRH_RASTER_DEBUG", "32768")
    else:
        monkeypatch.setenv("CRH_LONG_LISTS", mode)
    rng = np.random.RandomState(12)
    for sc in (scenes.scene_cubic_fill(3000, (384, 384), r_lo=20.0, r_hi=120.0), scenes.scene_glyphs(1500, (192, 192)), scenes.scene_cubic_fill(2500, (512, 512), r_lo=6.0, r_hi=90.0, config_index=7)):
        colors = np.asarray(sc["colors"], np.float32).copy()
        colors[rng.uniform(size=len(colors)) < 0.75, 3] = 1.0
        _check(gpu, oracle_lib, sc, f"{sc['name']} mode {mode}", colors=colors)


def test_rgba16f_target_and_a_pass_over_existing_content(gpu, oracle_lib, monkeypatch):
    """The same kernel stores binary16 layers (the multi-GPU exchange) and composites over what a frame already shows (LoadOp::Load)."""
    from contrast_renderer_amd import scenes
    monkeypatch.setenv("CRH_ROWS", "1")
    sc = scenes.scene_cubic_fill(300, (330, 270), r_lo=6.0, r_hi=64.0)
    r = _renderer(gpu)
    scene = gpu.Scene(r, sc["batch"])
    f8, f16 = gpu.Frame(r, 330, 270), gpu.Frame(r, 330, 270, gpu.FORMAT_RGBA16F)
    for f in (f8, f16):
        f.clear()
        scene.render(f, sc["transforms"], sc["colors"])
    monkeypatch.delenv("CRH_ROWS")
    monkeypatch.setenv("CRH_EDGE_PASS", "1")
    g8, g16 = gpu.Frame(r, 330, 270), gpu.Frame(r, 330, 270, gpu.FORMAT_RGBA16F)
    for f in (g8, g16):
        f.clear()
        scene.render(f, sc["transforms"], sc["colors"])
    assert np.array_equal(f8.download(), g8.download())
    assert np.array_equal(f16.download(), g16.download())  # (the RGBA8 view of the binary16 layer)
    # a second Scene over the first one's pixels, without a clear in between: both kernels, the same bytes
    other = scenes.scene_cubic_fill(200, (330, 270), r_lo=10.0, r_hi=50.0, config_index=9)
    scene2 = gpu.Scene(r, other["batch"])
    scene2.render(g8, other["transforms"], other["colors"])
    monkeypatch.setenv("CRH_ROWS", "1")
    scene2.render(f8, other["transforms"], other["colors"])
    assert np.array_equal(f8.download(), g8.download())


def test_the_trial_picks_a_kernel_and_the_frames_stay_the_same(gpu, oracle_lib, monkeypatch):
    """Left alone, the library draws a Scene's first frames with each formulation in turn (edges per sample, strip triangles, edges as row
    spans), times a group of three frames of each and keeps the fastest: twenty-four frames of a glyph scene, every one the oracle's."""
    from contrast_renderer_amd import scenes
    for name in ("CRH_ROWS", "CRH_EDGE_PASS", "CRH_TRIANGLE_PASS", "CRH_NO_ROWS"):
        monkeypatch.delenv(name, raising=False)
    sc = scenes.scene_glyphs(1200, (512, 384))
    r = _renderer(gpu)
    scene = gpu.Scene(r, sc["batch"])
    oracle = oracle_lib.Oracle(sc["batch"], 4)
    expect = oracle.render(512, 384, 1, 4, sc["transforms"], sc["colors"])
    frame = gpu.Frame(r, 512, 384)
    scene.set_instances(sc["transforms"], sc["colors"])
    for k in range(24):
        scene.tessellate()
        frame.clear()
        scene.render(frame)
        if k in (2, 8, 14, 19, 23):
            assert np.array_equal(frame.download(), expect), f"frame {k}"
