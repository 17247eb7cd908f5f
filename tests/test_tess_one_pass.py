"""The one-pass tessellation (k_tess_fused, csrc/tessellate.hip) against the oracle and against the two-pass path it replaces.

Shape-aligned workgroups reserve their ranges of the scene-wide streams with one atomic per channel, so the ORDER of the Shapes inside the
streams differs from launch to launch; the parity surface — every Shape's byte image, renderer.rs:198-209 — and the pixels must not.
CRH_TESS_TWO_PASS=1 (read per upload) keeps a Scene on k_count / k_scan_* / k_emit; a Shape with more elements than a workgroup has lanes
takes that path by itself. Which path ran is read off the kernel marks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from contrast_renderer_amd import renderer
    return renderer


def _renderer(gpu, sc):
    return gpu.Renderer(gpu.Configuration(msaa_sample_count=sc["msaa"], winding_counter_bits=sc["winding_bits"]), device=0)


def _marks(r, scene):
    r.enable_timing(True)
    scene.tessellate()
    r.synchronize()
    names = [name for name, _, _ in r.kernel_times()]
    r.enable_timing(False)
    return names


def _assert_equal(scene, oracle, what):
    layout, vb, ib = scene.all_shapes()
    olayout, ovb, oib = oracle.all_shapes()
    assert np.array_equal(layout, olayout), f"{what}: offsets differ in {(layout != olayout).any(axis=1).sum()} shapes"
    assert np.array_equal(ib, oib), f"{what}: index bytes differ"
    assert np.array_equal(vb, ovb), f"{what}: vertex bytes differ"


def _cases():
    from contrast_renderer_amd import scenes
    return {
        "cubic_fill_3000": lambda: scenes.scene_cubic_fill(3000, (1024, 1024), r_lo=8.0, r_hi=64.0),
        "dashed_strokes_400_msaa4": lambda: scenes.scene_dashed_strokes(400, (1024, 1024)),
        "mixed_96": lambda: scenes.scene_mixed(96, (512, 512), seed=11),
        "quadratic_100": lambda: scenes.scene_quadratic(100),
        "glyphs_600": lambda: scenes.scene_glyphs(600, (512, 512)),
    }


@pytest.mark.parametrize("two_pass", [False, True])
@pytest.mark.parametrize("case", list(_cases()))
def test_both_tessellation_paths_match_the_oracle(gpu, oracle_lib, case, two_pass, monkeypatch):
    sc = _cases()[case]()
    if two_pass:
        monkeypatch.setenv("CRH_TESS_TWO_PASS", "1")
    r = _renderer(gpu, sc)
    scene = gpu.Scene(r, sc["batch"])
    oracle = oracle_lib.Oracle(sc["batch"], 4)
    assert scene.status() == oracle.status() == 0
    _assert_equal(scene, oracle, case)
    frame = gpu.Frame(r, sc["width"], sc["height"])
    expect = oracle.render(sc["width"], sc["height"], sc["msaa"], sc["winding_bits"], sc["transforms"], sc["colors"])
    for step in range(3):  # again on the other set of buffers, and on the first one once more: another order of the runs each time
        if step:
            scene.tessellate()
        frame.clear()
        scene.render(frame, sc["transforms"], sc["colors"])
        assert np.array_equal(frame.download(), expect), f"{case}: pixels differ at step {step}"
        _assert_equal(scene, oracle, f"{case}, run {step + 1}")
    names = _marks(r, scene)
    assert ("tess_fused" in names) == (not two_pass) and ("tess_emit" in names) == two_pass, names
    assert scene.status() == 0


def _long_path_scene(n_segments, with_small_shapes):
    """One closed polygon of n_segments line segments (n_segments + 1 elements: more than a workgroup has lanes from 256 segments on), optionally between
    small Shapes and empty ones."""
    from contrast_renderer_amd.path import Path, batch_from_shapes
    rng = np.random.RandomState(5)
    shapes = []

    def small(cx, cy):
        p = Path(start=(cx + 10.0, cy))
        for k in range(1, 6):
            a = 2.0 * np.pi * k / 6.0
            p.push_line((cx + 10.0 * np.cos(a), cy + 10.0 * np.sin(a)))
        return p

    if with_small_shapes:
        shapes += [[small(40.0, 40.0)], [], [small(80.0, 40.0), small(120.0, 40.0)]]
    big = Path(start=(128.0 + 90.0, 128.0))
    for k in range(1, n_segments):
        a = 2.0 * np.pi * k / n_segments
        rad = 90.0 + 8.0 * rng.uniform(-1, 1)
        big.push_line((128.0 + rad * np.cos(a), 128.0 + rad * np.sin(a)))
    shapes.append([big])
    if with_small_shapes:
        shapes += [[], [small(200.0, 220.0)], []]
    return batch_from_shapes([([], paths) for paths in shapes])


@pytest.mark.parametrize("n_segments,with_small_shapes,fused", [(200, True, True), (255, False, True), (256, False, False), (700, True, False)])
def test_a_shape_beyond_one_workgroup_takes_the_two_pass_path(gpu, oracle_lib, n_segments, with_small_shapes, fused):
    from contrast_renderer_amd import scenes
    batch = _long_path_scene(n_segments, with_small_shapes)
    n = batch.n_shapes
    r = gpu.Renderer(gpu.Configuration(msaa_sample_count=1, winding_counter_bits=4), device=0)
    scene = gpu.Scene(r, batch)
    oracle = oracle_lib.Oracle(batch, 2)
    assert scene.status() == oracle.status() == 0
    _assert_equal(scene, oracle, f"{n_segments} segments")
    transforms = np.tile(scenes.ortho_pixels(256, 256), (n, 1, 1))
    colors = np.tile(np.asarray([0.2, 0.5, 0.9, 0.75], np.float32), (n, 1))
    frame = gpu.Frame(r, 256, 256)
    frame.clear()
    scene.render(frame, transforms, colors)
    assert np.array_equal(frame.download(), oracle.render(256, 256, 1, 4, transforms, colors))
    names = _marks(r, scene)
    assert ("tess_fused" in names) == fused, names


def test_stale_capacities_are_caught_and_sized_again(gpu, oracle_lib):
    """New paths uploaded into an existing Scene start from unknown capacities (a counting pass, then the whole kernel); the same Scene
    then takes larger and smaller geometry in turn."""
    from contrast_renderer_amd import scenes
    a = scenes.scene_mixed(40, (320, 320), seed=9)
    r = _renderer(gpu, a)
    scene = None
    for k, (n, seed) in enumerate([(40, 9), (96, 11), (12, 4), (64, 8)]):
        sc = scenes.scene_mixed(n, (320, 320), seed=seed)
        scene = gpu.Scene(r, sc["batch"], existing=scene)
        oracle = oracle_lib.Oracle(sc["batch"], 4)
        assert scene.status() == oracle.status() == 0
        for run in range(2):
            if run:
                scene.tessellate()
            _assert_equal(scene, oracle, f"upload {k}, run {run}")
        frame = gpu.Frame(r, 320, 320)
        frame.clear()
        scene.render(frame, sc["transforms"], sc["colors"])
        assert np.array_equal(frame.download(), oracle.render(320, 320, sc["msaa"], sc["winding_bits"], sc["transforms"], sc["colors"]))
