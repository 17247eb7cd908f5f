"""The one-pass tessellation (k_tess_runs, csrc/tessellate.hip) against the oracle and against the two-pass path it replaces, and what goes
with it: new paths of the structure a Scene holds uploaded without a wait for their totals (crh_scene::optimistic).

What an element emits is counted once per upload (k_tess_count_runs, k_scan_runs); every tessellation is then ONE kernel of Shape-aligned
workgroups that scans in LDS and analyses each element once. The parity surface — every Shape's byte image, renderer.rs:198-209 — and the
pixels must be the oracle's on either path. CRH_TESS_TWO_PASS=1 (read per upload) keeps a Scene on k_count / k_scan_* / k_emit; a Shape with
more elements than a workgroup has lanes takes that path by itself. Which path ran is read off the kernel marks."""
import os

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


@pytest.mark.parametrize("two_pass", [False, True, "runs of 128"])
@pytest.mark.parametrize("case", list(_cases()))
def test_both_tessellation_paths_match_the_oracle(gpu, oracle_lib, case, two_pass, monkeypatch):
    sc = _cases()[case]()
    if two_pass == "runs of 128":  # the one-pass kernel's 128-lane build (taken by itself beyond 4 096 runs; a Shape beyond 128 elements keeps 256)
        monkeypatch.setenv("CRH_TESS_RUN_BLOCK", "128")
        two_pass = False
    if two_pass:
        monkeypatch.setenv("CRH_TESS_TWO_PASS", "1")
    two_pass = two_pass or bool(os.environ.get("CRH_TESS_TWO_PASS"))  # (the suite run under the pin, tools/r06_pins.sh: every case takes that path)
    r = _renderer(gpu, sc)
    scene = gpu.Scene(r, sc["batch"])
    oracle = oracle_lib.Oracle(sc["batch"], 4)
    assert scene.status() == oracle.status() == 0
    _assert_equal(scene, oracle, case)
    frame = gpu.Frame(r, sc["width"], sc["height"])
    expect = oracle.render(sc["width"], sc["height"], sc["msaa"], sc["winding_bits"], sc["transforms"], sc["colors"])
    for step in range(3):  # again on the other set of buffers, and on the first one once more
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
    assert ("tess_fused" in names) == (fused and not os.environ.get("CRH_TESS_TWO_PASS")), names  # (the suite under the pin: never)


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


def _wobbly_scene(n_shapes, seed, kind, size=(512, 512)):
    """n_shapes Shapes of one closed path of eight segments each — one structure (as many Shapes, paths and elements) whatever `kind` is:
    kind 0 octagons (few records per segment), kind 1 the benchmark's mixed integral / rational cubic blobs drawn from the PCG streams
    behind path `seed * n_shapes` (several times the records). Returns (batch, transforms, colors)."""
    from contrast_renderer_amd import scenes
    from contrast_renderer_amd.path import Path, batch_from_shapes
    if kind == 1:
        sc = scenes.scene_cubic_fill(n_shapes, size, r_lo=8.0, r_hi=40.0, first_path=seed * n_shapes)
        return sc["batch"], sc["transforms"], sc["colors"]
    rng = np.random.RandomState(seed)
    shapes = []
    for _ in range(n_shapes):
        cx, cy = rng.uniform(40, size[0] - 40), rng.uniform(40, size[1] - 40)
        rad = rng.uniform(10, 36)
        pts = [(cx + rad * np.cos(-2 * np.pi * i / 8 + 0.1 * rng.uniform(-1, 1)), cy + rad * np.sin(-2 * np.pi * i / 8 + 0.1 * rng.uniform(-1, 1))) for i in range(8)]
        p = Path(start=pts[0])
        for i in range(1, 9):
            p.push_line(pts[i % 8])
        shapes.append(([], [p]))
    batch = batch_from_shapes(shapes)
    transforms = np.tile(scenes.ortho_pixels(*size), (n_shapes, 1, 1))
    colors = np.concatenate([rng.uniform(0, 1, (n_shapes, 3)), rng.uniform(0.4, 1.0, (n_shapes, 1))], axis=1).astype(np.float32)
    return batch, transforms, colors


@pytest.mark.parametrize("optimistic", [True, False, "two passes"])
def test_new_paths_of_the_same_structure_keep_the_capacities(gpu, oracle_lib, optimistic, monkeypatch, capfd):
    """crh_scene_upload into an existing Scene with paths of the structure it holds: no wait for the totals (api.hip: crh_scene::optimistic).
    Control points that move (the counts change a little: the headroom), then polygons replaced by cubics of the same structure (several
    times the records: the run does not fit, the frame finds the overflow code among its flags, everything is sized and drawn again), then
    back. Every frame and every Shape's bytes against the oracle; CRH_NO_OPTIMISTIC_UPLOAD is the old way."""
    if optimistic == "two passes":  # (the two-pass tessellation under optimistic uploads: k_shape_rows must leave the rows of a run that does not fit alone)
        monkeypatch.setenv("CRH_TESS_TWO_PASS", "1")
        optimistic = True
    if not optimistic:
        monkeypatch.setenv("CRH_NO_OPTIMISTIC_UPLOAD", "1")
    monkeypatch.setenv("CRH_PASS_VERBOSE", "1")  # (the library says on stderr when a frame finds its tessellation's overflow code)
    r = gpu.Renderer(gpu.Configuration(msaa_sample_count=1, winding_counter_bits=4), device=0)
    frames = [gpu.Frame(r, 512, 512), gpu.Frame(r, 512, 512)]
    scene = None
    n = 300
    sequence = [(0, 4), (0, 5), (1, 1), (1, 2), (1, 3), (0, 8), (1, 6), (1, 7), (0, 9), (1, 10)]  # (the streams never shrink: the octagons come first)
    for step, (kind, seed) in enumerate(sequence):
        batch, transforms, colors = _wobbly_scene(n, seed, kind)
        oracle = oracle_lib.Oracle(batch, 4)
        assert oracle.status() == 0
        scene = gpu.Scene(r, batch, tessellate=False, existing=scene)
        scene.set_instances(transforms, colors)
        scene.tessellate()
        f = frames[step % 2]
        f.clear()
        scene.render(f)
        if step % 3 == 2:  # some steps: the Shape bytes as well (a synchronising call between upload and download)
            _assert_equal(scene, oracle, f"step {step} (kind {kind})")
        image = f.download()
        expect = oracle.render(512, 512, 1, 4, transforms, colors)
        assert np.array_equal(image, expect), f"step {step} (kind {kind}): {(image != expect).any(axis=2).sum()} pixels differ"
        assert scene.status() == 0
        _assert_equal(scene, oracle, f"step {step} (kind {kind}), after the frame")
    redrawn = capfd.readouterr().err.count("outgrew the streams")
    assert (redrawn >= 1) if optimistic else (redrawn == 0), redrawn  # octagons -> cubics of the same structure does not fit: found by the frame


def test_two_scenes_in_turn_with_frames_in_flight(gpu, oracle_lib):
    """The loop of bench.py --reupload: two Scenes and two frames in turn, new paths of the same structure every step, nothing waited for
    until the end; then every frame's pixels."""
    r = gpu.Renderer(gpu.Configuration(msaa_sample_count=1, winding_counter_bits=4), device=0)
    n = 400
    frames = [gpu.Frame(r, 512, 512), gpu.Frame(r, 512, 512)]
    first = _wobbly_scene(n, 100, 0)
    scenes_ = [gpu.Scene(r, first[0]), gpu.Scene(r, first[0])]
    for s in scenes_:
        assert s.status() == 0
    last = [None, None]
    for step in range(12):
        batch, transforms, colors = _wobbly_scene(n, 101 + step, 0 if step % 5 == 3 else 1)
        k = step % 2
        if last[k] is not None:  # the frame about to be reused is consumed first
            image = frames[k].download()
            o, tr, co = last[k]
            assert np.array_equal(image, o.render(512, 512, 1, 4, tr, co)), f"frame of step {step - 2}"
        scenes_[k] = gpu.Scene(r, batch, tessellate=False, existing=scenes_[k])
        scenes_[k].set_instances(transforms, colors)
        scenes_[k].tessellate()
        frames[k].clear()
        scenes_[k].render(frames[k])
        oracle = oracle_lib.Oracle(batch, 4)
        assert oracle.status() == 0
        last[k] = (oracle, transforms, colors)
    for k in range(2):
        o, tr, co = last[k]
        assert np.array_equal(frames[k].download(), o.render(512, 512, 1, 4, tr, co))
        assert scenes_[k].status() == 0


def test_slabs_of_re_uploaded_paths(gpu, oracle_lib):
    """The tile split's passes (a slab of tile rows per frame; the items whose Shape misses the slab are left out before they are set up:
    k_shape_bounds, k_slab_items) over paths that are uploaded again and again into the same Scene — the Shapes' boxes belong to an upload,
    and to a tessellation that fitted: the octagons-to-cubics step does not, and is drawn again."""
    r = gpu.Renderer(gpu.Configuration(msaa_sample_count=1, winding_counter_bits=4), device=0)
    n = 300
    slabs = [(0, 176), (176, 336), (336, 512)]
    frames = []
    for a, b in slabs:
        f = gpu.Frame(r, 512, 512)
        f.set_tile_rows(a, b)
        frames.append(f)
    scene = None
    for step, (kind, seed) in enumerate([(0, 4), (0, 5), (1, 1), (1, 2), (0, 8), (1, 3)]):
        batch, transforms, colors = _wobbly_scene(n, seed, kind)
        oracle = oracle_lib.Oracle(batch, 4)
        assert oracle.status() == 0
        scene = gpu.Scene(r, batch, tessellate=False, existing=scene)
        scene.set_instances(transforms, colors)
        scene.tessellate()
        for f in frames:
            f.clear()
            scene.render(f)
        expect = oracle.render(512, 512, 1, 4, transforms, colors)
        for (a, b), f in zip(slabs, frames):
            image = f.download()
            assert np.array_equal(image[a:b], expect[a:b]), f"step {step} (kind {kind}), rows {a}..{b}: {(image[a:b] != expect[a:b]).any(axis=2).sum()} pixels differ"
        assert scene.status() == 0
