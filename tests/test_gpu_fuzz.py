"""Randomised parity sweep of the HIP path against the oracle: scenes with every segment type, fills and strokes, random frame sizes
(not multiples of the tile), sample counts, winding rules, plain or perspective instance transforms, depth / cull states and recorded
passes with clip nesting — each case small enough for the oracle to render in a fraction of a second. Bit for bit."""
import math

import numpy as np
import pytest

from contrast_renderer_amd import scenes, utils
from contrast_renderer_amd.renderer import Compare, Cull
from contrast_renderer_amd.renderer import RenderOperation as Op


def random_case(seed):
    rng = np.random.RandomState(1000 + seed)
    width, height = int(rng.randint(40, 300)), int(rng.randint(40, 300))
    msaa = int(rng.choice([1, 4]))
    winding_bits = int(rng.choice([1, 2, 4]))
    n_shapes = int(rng.randint(3, 40))
    base = scenes.scene_mixed(n_shapes, (max(width, 96), max(height, 96)), seed=seed)
    batch, colors = base["batch"], base["colors"]
    n = batch.n_shapes
    kind = seed % 6
    if kind == 0 or kind == 4:  # plain instances, placed over the (smaller) frame so that shapes hang over every border
        t = scenes.place(width, height, rng.uniform(-10, width + 10, n), rng.uniform(-10, height + 10, n), rng.uniform(8, 0.5 * min(width, height), n))
    elif kind == 5:  # extreme placements: tiny and enormous scales, centres far outside the frame (slivers, clamped boxes, huge coordinates)
        t = scenes.place(width, height, rng.uniform(-3 * width, 4 * width, n), rng.uniform(-3 * height, 4 * height, n),
                         np.exp(rng.uniform(math.log(0.05), math.log(40.0 * max(width, height)), n)))
    else:  # perspective: tilted decals at random depths, some reaching through the near plane
        projection = utils.perspective_projection(math.pi * 0.5, width / height, 1.0, 100.0)
        t = np.stack([utils.matrix_multiplication(projection, utils.matrix_multiplication(
            utils.translation_matrix(rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0.8, 6.0)),
            utils.rotation_matrix(rng.uniform(-1.4, 1.4), (math.cos(i * 1.7), math.sin(i * 1.7), 0.0)))) for i in range(n)])
    state = {}
    if kind == 2:
        state = dict(depth_compare=int(rng.choice([Compare.Less, Compare.LessEqual, Compare.Greater, Compare.NotEqual])), depth_write=int(rng.randint(0, 2)),
                     cull_mode=int(rng.choice([Cull.Disabled, Cull.Back])))
    draws = []
    if kind == 3:  # a recorded pass: the first shape clips the second half of the shapes; instancing reuses shape 0
        draws += [(0, 0, Op.Stencil, 0, 0), (0, 0, Op.Clip, 1, 0)]
        for i in range(n // 2, n):
            draws += [(i, i, Op.Stencil, 1, 0), (i, i, Op.Color, 1, 0)]
        draws += [(0, 0, Op.UnClip, 0, 0)]
        for i in range(1, n // 2):
            draws += [(i, i, Op.Stencil, 0, 0), (i, i, Op.Color, 0, 0)]
        draws += [(0, n - 1, Op.Stencil, 0, 0), (0, n - 1, Op.Color, 0, 0)]
    elif kind == 4:  # an opacity group (Save / Scale / content / Restore) over a background, twice nested
        draws += [(0, 0, Op.Stencil, 0, 0), (0, 0, Op.Color, 0, 0), (1, 1, Op.SaveAlphaContext, 0, 0), (1, 1, Op.ScaleAlphaContext, 0, 0)]
        for i in range(2, n // 2):
            draws += [(i, i, Op.Stencil, 0, 0), (i, i, Op.Color, 0, 0)]
        draws += [(2, 2, Op.SaveAlphaContext, 0, 1), (2, 2, Op.ScaleAlphaContext, 0, 1)]
        for i in range(n // 2, n):
            draws += [(i, i, Op.Stencil, 0, 0), (i, i, Op.Color, 0, 0)]
        draws += [(2, 2, Op.RestoreAlphaContext, 0, 1), (1, 1, Op.RestoreAlphaContext, 0, 0)]
    else:
        for i in range(n):
            draws += [(i, i, Op.Stencil, 0, 0), (i, i, Op.Color, 0, 0)]
    depth = None
    if state.get("depth_compare", 0) or state.get("depth_write", 0):
        depth = rng.choice([0.2, 0.6, 0.95, 1.0], size=(height, width)).astype(np.float32)
    return dict(width=width, height=height, msaa=msaa, winding_bits=winding_bits, batch=batch, transforms=t.astype(np.float32), colors=colors, draws=draws,
                state=state, depth=depth)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("CRH_FUZZ_SEEDS", "48"))))
def test_random_scene_matches_the_oracle(seed, oracle_lib):
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    from oracle.binding import Oracle, render_pass
    c = random_case(seed)
    o = Oracle(c["batch"])
    config = R.Configuration(msaa_sample_count=c["msaa"], clip_nesting_counter_bits=2, winding_counter_bits=c["winding_bits"], alpha_layer_count=2,
                             cull_mode=c["state"].get("cull_mode", 0), depth_compare=c["state"].get("depth_compare", 0),
                             depth_write_enabled=bool(c["state"].get("depth_write", 0)))
    r = R.Renderer(config, device=0)
    scene = R.Scene(r, c["batch"])
    assert scene.status() == o.status()  # a scene the reference cannot tessellate (fill.rs:174,178 panics) fails the same way
    if o.status() != 0:
        return
    expect, expect_depth = render_pass_rect(o, c, render_pass)
    frame = R.Frame(r, c["width"], c["height"])
    frame.clear()
    if c["depth"] is not None:
        frame.upload_depth(c["depth"])
    scene.render_draws(frame, c["transforms"], c["colors"], c["draws"])
    image = frame.download()
    assert np.array_equal(image, expect), f"seed {seed}: {(image != expect).any(axis=2).sum()} pixels differ"
    if c["depth"] is not None:
        assert np.array_equal(frame.download_depth(), expect_depth)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("CRH_FUZZ_SPLIT_SEEDS", "24"))))
def test_a_random_pass_cut_into_pieces_at_random_draws_gives_the_same_frame(seed, oracle_lib):
    """The recorded pass of random_case(seed) submitted as two to five passes cut at random draws (between a Stencil and its cover, inside an
    open clip, inside an opacity group): with the pass state kept on the frame (crh_frame_keep_pass_state: the reference's caller-owned stencil
    attachment, alpha layers and f32 colour, renderer.rs:148-158, 257-266) the frame is the single pass's, bit for bit."""
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    from oracle.binding import Oracle, render_pass
    c = random_case(seed)
    o = Oracle(c["batch"])
    if o.status() != 0:
        pytest.skip("the reference cannot tessellate this scene (covered by test_random_scene_matches_the_oracle)")
    rng = np.random.RandomState(7000 + seed)
    config = R.Configuration(msaa_sample_count=c["msaa"], clip_nesting_counter_bits=2, winding_counter_bits=c["winding_bits"], alpha_layer_count=2,
                             cull_mode=c["state"].get("cull_mode", 0), depth_compare=c["state"].get("depth_compare", 0),
                             depth_write_enabled=bool(c["state"].get("depth_write", 0)))
    r = R.Renderer(config, device=0)
    scene = R.Scene(r, c["batch"])
    expect, expect_depth = render_pass_rect(o, c, render_pass)
    draws = c["draws"]
    pieces = min(len(draws), int(rng.randint(2, 6)))
    cuts = sorted(rng.choice(np.arange(1, len(draws)), size=pieces - 1, replace=False).tolist()) if len(draws) > 1 else []
    bounds = [0] + cuts + [len(draws)]
    frame = R.Frame(r, c["width"], c["height"])
    for round_ in range(2):  # the second round after a clear: the state of the first must be gone
        frame.clear()
        if c["depth"] is not None:
            frame.upload_depth(c["depth"])
        frame.keep_pass_state()
        for a, b in zip(bounds[:-1], bounds[1:]):
            scene.render_draws(frame, c["transforms"], c["colors"], draws[a:b])
        image = frame.download()
        assert np.array_equal(image, expect), f"seed {seed} cuts {cuts} round {round_}: {(image != expect).any(axis=2).sum()} pixels differ"
        if c["depth"] is not None:
            assert np.array_equal(frame.download_depth(), expect_depth)


def render_pass_rect(o, c, render_pass):
    """oracle.binding.render_pass takes (width, height); kept in one place so that the argument order cannot drift from the GPU call."""
    return render_pass(o, c["width"], c["height"], c["msaa"], c["winding_bits"], 2, 2, c["transforms"], c["colors"], [tuple(int(v) for v in d) for d in c["draws"]],
                       depth=c["depth"], **c["state"])


def random_paths(n_shapes, seed):
    """Unstructured geometry: random control points (self-intersecting outlines, loops, cusps, coincident and collinear points, weights
    from 0.2 to 5), one to three paths per Shape, filled or stroked with random widths / offsets / joins / caps / approximations."""
    from contrast_renderer_amd import Cap, CurveApproximation, DashInterval, DynamicStrokeOptions, Join, Path, StrokeOptions
    rng = np.random.RandomState(seed)
    caps = list(Cap)
    shapes = []
    for s in range(n_shapes):
        paths, dynamic = [], []
        for _ in range(rng.randint(1, 4)):
            scale = float(np.exp(rng.uniform(math.log(0.05), math.log(3.0))))

            def pt():
                if rng.uniform() < 0.08 and paths:  # reuse an earlier point: coincident vertices
                    return paths[-1].start
                return (float(np.float32(rng.normal(0, scale))), float(np.float32(rng.normal(0, scale))))
            pen = pt()
            p = Path(start=pen)
            for _ in range(rng.randint(1, 10)):
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
            if rng.uniform() < 0.5:
                approx = CurveApproximation.UniformTangentAngle(float(rng.uniform(0.05, 0.8))) if rng.uniform() < 0.5 else CurveApproximation.UniformlySpacedParameters(int(rng.randint(1, 12)))
                p.stroke_options = StrokeOptions(float(rng.uniform(0.001, 0.5)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(1.0, 8.0)), bool(rng.randint(0, 2)),
                                                 len(dynamic), approx)
                join = (Join.Miter, Join.Bevel, Join.Round)[rng.randint(0, 3)]
                if rng.uniform() < 0.5:
                    dynamic.append(DynamicStrokeOptions.Solid(join, caps[rng.randint(0, len(caps))], caps[rng.randint(0, len(caps))]))
                else:
                    a, b = sorted(rng.uniform(0.0, 4.0, 2))
                    dynamic.append(DynamicStrokeOptions.Dashed(join, [DashInterval(float(a), float(b), caps[rng.randint(0, len(caps))], caps[rng.randint(0, len(caps))])],
                                                               float(rng.uniform(0.0, 2.0))))
            paths.append(p)
        shapes.append((dynamic, paths))
    return shapes


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("CRH_FUZZ_PATH_SEEDS", "4"))))
def test_random_paths_tessellate_to_the_same_bytes_or_fail_the_same_way(seed, oracle_lib):
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import batch_from_shapes
    from contrast_renderer_amd import renderer as R
    from oracle.binding import Oracle
    shapes = random_paths(1500, 77 + seed)
    batch = batch_from_shapes(shapes)
    oracle = Oracle(batch, 8)
    status = [oracle.shape_status(s) for s in range(len(shapes))]
    good = [s for s in range(len(shapes)) if status[s] == 0]
    assert len(good) > 0.5 * len(shapes)
    r = R.Renderer(R.Configuration(1, 4, 4), device=0)
    clean = batch_from_shapes([shapes[s] for s in good])
    scene = R.Scene(r, clean)
    assert scene.status() == 0
    layout, vb, ib = scene.all_shapes()
    olayout, ovb, oib = Oracle(clean, 8).all_shapes()
    assert np.array_equal(layout, olayout) and np.array_equal(ib, oib)
    assert np.array_equal(vb, ovb), f"{np.flatnonzero(vb != ovb).size} vertex bytes differ, first at {np.flatnonzero(vb != ovb)[0]}"
    assert vb.size > 1_000_000
    for s in [s for s in range(len(shapes)) if status[s] != 0][:8]:  # what the reference cannot tessellate fails with the same code
        assert R.Scene(r, batch.slice_shapes(s, s + 1)).status() == status[s]
    # and the pixels of a frame over them
    n = clean.n_shapes
    rng = np.random.RandomState(seed)
    t = scenes.place(256, 256, rng.uniform(0, 256, n), rng.uniform(0, 256, n), rng.uniform(5, 60, n))
    c = np.concatenate([rng.uniform(0, 1, (n, 3)), rng.uniform(0.2, 1, (n, 1))], axis=1).astype(np.float32)
    frame = R.Frame(r, 256, 256)
    frame.clear()
    scene.render(frame, t, c)
    image = frame.download()
    expect = Oracle(clean, 8).render(256, 256, 1, 4, t, c)
    assert np.array_equal(image, expect), f"{(image != expect).any(axis=2).sum()} pixels differ"


@pytest.mark.gpu
def test_two_renderers_and_reused_scenes_interleaved(oracle_lib):
    """Two Renderers on the same GPU (each with its own streams) driven alternately without synchronisation in between, each re-uploading
    scenes of changing size into the same Scene object (`existing_shape`, renderer.rs:216-221) and rendering into two frames of its own:
    every download matches the oracle."""
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    from oracle.binding import Oracle
    renderers = [R.Renderer(R.Configuration(1, 4, 4), device=0), R.Renderer(R.Configuration(4, 2, 4), device=0)]
    frames = [[R.Frame(r, 200 + 56 * k, 150 + 40 * k) for k in range(2)] for r in renderers]
    held = [None, None]
    pending = []
    for step in range(12):
        which = step % 2
        r = renderers[which]
        sc = scenes.scene_mixed(4 + (step * 7) % 23, (256, 256), seed=100 + step)
        if Oracle(sc["batch"]).status() != 0:
            continue
        held[which] = R.Scene(r, sc["batch"], existing=held[which])
        frame = frames[which][(step // 2) % 2]
        frame.clear()
        held[which].render(frame, sc["transforms"], sc["colors"])
        pending = [q for q in pending if q[0] is not frame] + [(frame, sc, r.config.msaa_sample_count)]  # the last render into each frame
    checked = 0
    for f, s, msaa in pending:
        expect = Oracle(s["batch"]).render(f.width, f.height, msaa, 4, s["transforms"], s["colors"])
        assert np.array_equal(f.download(), expect)
        checked += 1
    assert checked == 4


@pytest.mark.gpu
def test_objects_may_be_destroyed_in_any_order(oracle_lib):
    """Host bindings finalise in any order (a garbage collector, a scope exit): a Renderer destroyed before its Scenes and Frames orphans
    them, and a Scene destroyed while a Frame still refers to it (deferred tile-list check) settles that Frame first."""
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    from oracle.binding import Oracle
    sc = scenes.scene_mixed(9, (128, 128), seed=1)
    expect = Oracle(sc["batch"]).render(128, 128, 1, 4, sc["transforms"], sc["colors"])
    for order in ("renderer_first", "scene_first", "frame_first"):
        r = R.Renderer(R.Configuration(1, 4, 4), device=0)
        scene = R.Scene(r, sc["batch"])
        frame = R.Frame(r, 128, 128)
        frame.clear()
        scene.render(frame, sc["transforms"], sc["colors"])
        lib = r.lib
        if order == "renderer_first":
            lib.crh_renderer_destroy(r.handle)
            r.handle = None
            lib.crh_scene_destroy(scene.handle)
            scene.handle = None
            lib.crh_frame_destroy(frame.handle)
            frame.handle = None
        elif order == "scene_first":
            lib.crh_scene_destroy(scene.handle)  # the frame's render is still pending: it is settled here
            scene.handle = None
            assert np.array_equal(frame.download(), expect)
        else:
            lib.crh_frame_destroy(frame.handle)
            frame.handle = None
            assert scene.status() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("CRH_FUZZ_API_SEEDS", "6"))))
def test_random_api_sequences_against_a_host_model(seed, oracle_lib):
    """Stateful sweep: random sequences of upload (fresh or into an existing Scene), instance updates, plain and recorded passes, passes
    over existing content (LoadOp::Load), dynamic stroke option updates and late downloads on two Scenes and two Frames of one Renderer.
    A host model replays every pass with the oracle; every download must equal it bit for bit."""
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import Cap, DashInterval, DynamicStrokeOptions, Join
    from contrast_renderer_amd import renderer as R
    from oracle.binding import Oracle, render_pass
    rng = np.random.RandomState(4000 + seed)
    msaa = int(rng.choice([1, 4]))
    r = R.Renderer(R.Configuration(msaa, 2, 4, 2), device=0)
    sizes = [(int(rng.randint(60, 200)), int(rng.randint(60, 200))) for _ in range(2)]
    frames = [R.Frame(r, w, h) for w, h in sizes]
    model = [np.zeros((h, w, 4), dtype=np.uint8) for w, h in sizes]  # what each frame must hold
    cleared = [True, True]
    for f in frames:
        f.clear()
    gpu_scenes, host = [None, None], [None, None]  # host[i] = dict(batch, oracle, transforms, colors)

    def fresh_scene():
        while True:
            sc = scenes.scene_mixed(int(rng.randint(2, 25)), (160, 160), seed=int(rng.randint(0, 10000)))
            o = Oracle(sc["batch"])
            if o.status() == 0:
                return sc, o

    for step in range(40):
        op = rng.randint(0, 7)
        k = int(rng.randint(0, 2))
        if op == 0 or gpu_scenes[k] is None:  # upload: a fresh Scene, or new geometry into the existing one (existing_shape, renderer.rs:216-221)
            sc, o = fresh_scene()
            gpu_scenes[k] = R.Scene(r, sc["batch"], existing=gpu_scenes[k] if rng.uniform() < 0.6 else None)
            assert gpu_scenes[k].status() == 0
            host[k] = dict(batch=sc["batch"], oracle=o, transforms=sc["transforms"], colors=sc["colors"], instances_set=False)
        elif op == 1:  # new instance data
            n = host[k]["batch"].n_shapes
            host[k]["transforms"] = scenes.place(160, 160, rng.uniform(0, 160, n), rng.uniform(0, 160, n), rng.uniform(5, 70, n))
            host[k]["colors"] = np.concatenate([rng.uniform(0, 1, (n, 3)), rng.uniform(0.2, 1, (n, 1))], axis=1).astype(np.float32)
            gpu_scenes[k].set_instances(host[k]["transforms"], host[k]["colors"])
            host[k]["instances_set"] = True
        elif op in (2, 3, 4):  # a pass into frame j: cleared first or over what is there
            j = int(rng.randint(0, 2))
            w, h = sizes[j]
            if rng.uniform() < 0.6:
                frames[j].clear()
                cleared[j] = True
                model[j] = np.zeros_like(model[j])
            n = host[k]["batch"].n_shapes
            draws = [d for i in range(n) for d in ((i, i, int(Op.Stencil), 0, 0), (i, i, int(Op.Color), 0, 0))]
            if op == 4 and n >= 3:  # recorded: shape 0 clips the rest
                inner = [d for i in range(1, n) for d in ((i, i, int(Op.Stencil), 1, 0), (i, i, int(Op.Color), 1, 0))]
                draws = [(0, 0, int(Op.Stencil), 0, 0), (0, 0, int(Op.Clip), 1, 0)] + inner + [(0, 0, int(Op.UnClip), 0, 0)]
                gpu_scenes[k].render_draws(frames[j], host[k]["transforms"], host[k]["colors"], draws)
            else:
                if not host[k]["instances_set"] or rng.uniform() < 0.5:
                    gpu_scenes[k].render(frames[j], host[k]["transforms"], host[k]["colors"])
                    host[k]["instances_set"] = True
                else:
                    gpu_scenes[k].render(frames[j])
            model[j], _ = render_pass(host[k]["oracle"], w, h, msaa, 4, 2, 2, host[k]["transforms"], host[k]["colors"], draws, load=None if cleared[j] else model[j])
            cleared[j] = False
        elif op == 5:  # tessellate again (same geometry: the frames in flight must not notice), or clear a frame without drawing
            if rng.uniform() < 0.5:
                gpu_scenes[k].tessellate()
            else:
                j = int(rng.randint(0, 2))
                frames[j].clear()
                cleared[j] = True
                model[j] = np.zeros_like(model[j])
        else:  # a late download (a cleared frame is transparent)
            j = int(rng.randint(0, 2))
            assert np.array_equal(frames[j].download(), model[j]), f"seed {seed} step {step}: frame {j} differs"
    for j in range(2):
        assert np.array_equal(frames[j].download(), model[j]), f"seed {seed}: final frame {j} differs"


@pytest.mark.gpu
def test_malformed_batches_are_rejected_not_read_out_of_bounds(oracle_lib):
    """The C ABI cannot trust the index arrays of a crh_path_batch (the Rust types make these states unrepresentable): prefix arrays that
    do not start at 0, decrease or end beyond the element counts, segment types out of range, a float count that does not match the
    segment types, missing arrays — each is CRH_ERR_INVALID_ARGUMENT, and the valid batch still uploads afterwards."""
    import copy
    import ctypes as C
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import _ffi
    from contrast_renderer_amd import renderer as R
    sc = scenes.scene_mixed(6, (128, 128), seed=3)
    batch = sc["batch"]
    r = R.Renderer(R.Configuration(1, 4, 4), device=0)
    lib = r.lib

    def upload(c_struct):
        handle = C.c_void_p()
        status = lib.crh_scene_upload(r.handle, C.byref(c_struct), None, C.byref(handle))
        if status == 0:
            lib.crh_scene_destroy(handle)
        return status

    assert upload(batch.c) == 0
    keep = []  # numpy arrays behind the patched pointers

    def variant(edit):
        c = _ffi.PathBatchC()
        C.memmove(C.byref(c), C.byref(batch.c), C.sizeof(c))
        edit(c)
        return c

    def patched(field, ctype, values):
        arr = np.ascontiguousarray(values)
        keep.append(arr)
        return lambda c: setattr(c, field, arr.ctypes.data_as(C.POINTER(ctype)))

    spb, psb = batch.shape_path_begin.copy(), batch.path_segment_begin.copy()
    cases = {
        "shape prefix does not start at 0": patched("shape_path_begin", C.c_uint32, spb + 1),
        "shape prefix decreases": patched("shape_path_begin", C.c_uint32, np.concatenate([spb[:2][::-1], spb[2:]]) if spb[1] > 0 else spb[::-1]),
        "shape prefix ends beyond the paths": patched("shape_path_begin", C.c_uint32, np.concatenate([spb[:-1], [spb[-1] + 5]]).astype(np.uint32)),
        "path prefix ends beyond the segments": patched("path_segment_begin", C.c_uint32, np.concatenate([psb[:-1], [psb[-1] + 1000]]).astype(np.uint32)),
        "path prefix decreases": patched("path_segment_begin", C.c_uint32, np.concatenate([[0, psb[-1]], psb[2:]]).astype(np.uint32)),
        "segment type out of range": patched("segment_types", C.c_uint8, np.where(np.arange(len(batch.segment_types)) == 3, 9, batch.segment_types).astype(np.uint8)),
        "float count too large": lambda c: setattr(c, "n_control_floats", c.n_control_floats + 7),
        "float count too small": lambda c: setattr(c, "n_control_floats", c.n_control_floats - 2),
        "segment count too large": lambda c: setattr(c, "n_segments", c.n_segments + 1),
        "no control data": lambda c: setattr(c, "control_data", C.POINTER(C.c_float)()),
        "no start points": lambda c: setattr(c, "path_start", C.POINTER(C.c_float)()),
        "dynamic options without a prefix": lambda c: setattr(c, "shape_dynamic_begin", C.POINTER(C.c_uint32)()),
        "stroke options index out of range": patched("path_stroke_options", C.c_int32, np.where(batch.path_stroke_options >= 0, 10 ** 6, batch.path_stroke_options).astype(np.int32)),
    }
    for name, edit in cases.items():
        assert upload(variant(edit)) == _ffi.ERR_INVALID_ARGUMENT, name
    assert upload(batch.c) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("msaa", [1, 4])
def test_resubmitted_passes_with_moving_instances(msaa, oracle_lib):
    """An animation: the same recorded pass (clip nesting) and the same plain pass submitted frame after frame with new instance data,
    no host synchronisation in between, two frames alternating — the cached pass, the double-buffered instance sets and the upload
    stream. Every frame is downloaded only at the end and must equal the oracle's rendering of ITS instance data."""
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    from oracle.binding import Oracle, render_pass
    sc = scenes.scene_mixed(14, (160, 160), seed=21)
    o = Oracle(sc["batch"])
    assert o.status() == 0
    n = sc["batch"].n_shapes
    inner = [d for i in range(1, n) for d in ((i, i, int(Op.Stencil), 1, 0), (i, i, int(Op.Color), 1, 0))]
    recorded = np.array([(0, 0, int(Op.Stencil), 0, 0), (0, 0, int(Op.Clip), 1, 0)] + inner + [(0, 0, int(Op.UnClip), 0, 0)], dtype=np.uint32)
    plain = [d for i in range(n) for d in ((i, i, int(Op.Stencil), 0, 0), (i, i, int(Op.Color), 0, 0))]
    r = R.Renderer(R.Configuration(msaa, 2, 4, 0), device=0)
    scene = R.Scene(r, sc["batch"])
    rng = np.random.RandomState(5)
    for mode in ("recorded", "plain"):
        frames = [R.Frame(r, 160, 160) for _ in range(2)]
        shown = [None, None]
        for step in range(14):
            t = scenes.place(160, 160, rng.uniform(20, 140, n), rng.uniform(20, 140, n), rng.uniform(10, 60, n))
            c = np.concatenate([rng.uniform(0, 1, (n, 3)), rng.uniform(0.3, 1, (n, 1))], axis=1).astype(np.float32)
            f = frames[step & 1]
            f.clear()
            if mode == "recorded":
                scene.render_draws(f, t, c, recorded)
            else:
                scene.render(f, t, c)
            shown[step & 1] = (t, c)
            if step == 7:  # one download in the middle of the animation
                expect, _ = render_pass(o, 160, 160, msaa, 4, 2, 0, t, c, recorded.tolist() if mode == "recorded" else plain)
                assert np.array_equal(f.download(), expect)
        for k in range(2):
            t, c = shown[k]
            expect, _ = render_pass(o, 160, 160, msaa, 4, 2, 0, t, c, recorded.tolist() if mode == "recorded" else plain)
            assert np.array_equal(frames[k].download(), expect), f"{mode}: frame {k}"
