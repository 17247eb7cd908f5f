"""RenderOperation::{Clip, UnClip, SaveAlphaContext, ScaleAlphaContext, RestoreAlphaContext} and instancing:
the oracle's statement of renderer.rs:692-729,761-861 + shaders.wgsl:311-355 (CPU tests, hand-checkable), and the HIP tile rasterizer
against it through crh_scene_render_draws (GPU tests, bit-exact)."""
import os

import numpy as np
import pytest

from contrast_renderer_amd import Path, batch_from_shapes
from contrast_renderer_amd.renderer import RenderOperation as Op

IDENTITY = np.eye(4, dtype=np.float32).reshape(-1)


def rect(cx, cy, hx, hy):
    return ([], [Path.from_rect((cx, cy), (hx, hy))])


def clip_scene():
    """Shape 0: a 24-gon used as clip; shapes 1, 2: rectangles drawn inside the clip; shape 3: a nested clip; shape 4: drawn after UnClip."""
    shapes = [([], [Path.from_regular_polygon((0.0, 0.0), 0.6, 0.0, 24)]), rect(-0.3, 0.0, 0.5, 0.2), rect(0.3, 0.3, 0.5, 0.2),
              rect(0.0, 0.0, 0.25, 0.9), rect(0.0, -0.6, 0.9, 0.1)]
    colors = np.array([[1, 0, 0, 1], [0, 1, 0, 0.5], [0, 0, 1, 1], [1, 0, 1, 1], [1, 1, 0, 0.75]], dtype=np.float32)
    transforms = np.tile(IDENTITY, (5, 1))
    draws = [(0, 0, Op.Stencil, 0, 0), (0, 0, Op.Clip, 1, 0),            # clip_depth += 1; set_clip_depth; Clip
             (1, 1, Op.Stencil, 1, 0), (1, 1, Op.Color, 1, 0),
             (3, 3, Op.Stencil, 1, 0), (3, 3, Op.Clip, 2, 0),            # nested clip
             (2, 2, Op.Stencil, 2, 0), (2, 2, Op.Color, 2, 0),
             (3, 3, Op.UnClip, 1, 0),                                    # clip_depth -= 1; set_clip_depth; UnClip
             (0, 0, Op.UnClip, 0, 0),
             (4, 4, Op.Stencil, 0, 0), (4, 4, Op.Color, 0, 0)]
    return batch_from_shapes(shapes), transforms, colors, draws


def alpha_scene():
    """An opacity group: save the frame's alpha under the group shape, scale it, draw the group's content, restore."""
    shapes = [rect(0.0, 0.0, 0.9, 0.9), rect(0.0, 0.0, 0.6, 0.6), rect(-0.2, 0.1, 0.3, 0.5), rect(0.3, -0.2, 0.4, 0.2)]
    colors = np.array([[0.2, 0.4, 0.8, 0.5], [0, 0, 0, 0.25], [1, 0, 0, 1], [0, 1, 0, 0.6]], dtype=np.float32)
    transforms = np.tile(IDENTITY, (4, 1))
    draws = [(0, 0, Op.Stencil, 0, 0), (0, 0, Op.Color, 0, 0),                    # background
             (1, 1, Op.SaveAlphaContext, 0, 0), (1, 1, Op.ScaleAlphaContext, 0, 0),
             (2, 2, Op.Stencil, 0, 0), (2, 2, Op.Color, 0, 0), (3, 3, Op.Stencil, 0, 0), (3, 3, Op.Color, 0, 0),
             (1, 1, Op.RestoreAlphaContext, 0, 0)]
    return batch_from_shapes(shapes), transforms, colors, draws


def instanced_scene():
    """One Shape, many instances (instance_indices of renderer.rs:271): the same star stroked + filled at different places / colours."""
    from contrast_renderer_amd import scenes
    sc = scenes.scene_mixed(3, (256, 256), seed=3)
    rng = np.random.RandomState(9)
    n = 12
    transforms = scenes.place(256, 256, rng.uniform(30, 226, n), rng.uniform(30, 226, n), rng.uniform(10, 40, n))
    colors = np.concatenate([rng.uniform(0, 1, (n, 3)), rng.uniform(0.3, 1, (n, 1))], axis=1).astype(np.float32)
    draws = []
    for i in range(n):
        draws += [(i % 3, i, Op.Stencil, 0, 0), (i % 3, i, Op.Color, 0, 0)]
    return sc["batch"], transforms, colors, draws


def oracle_image(batch, transforms, colors, draws, size=192, msaa=1, clip_bits=4, layers=2):
    from oracle.binding import Oracle, render_draws
    o = Oracle(batch)
    assert o.status() == 0
    return render_draws(o, size, size, msaa, 4, clip_bits, layers, transforms, colors, [tuple(int(v) for v in d) for d in draws])


def pixel(img, x, y, size=192):  # scene coordinates in [-1, 1], y up
    return img[int((0.5 - y * 0.5) * size), int((x * 0.5 + 0.5) * size)]


def test_oracle_clip_nesting_known_answers(oracle_lib):
    batch, t, c, draws = clip_scene()
    img = oracle_image(batch, t, c, draws)
    assert tuple(pixel(img, -0.3, 0.0)) == (0, 128, 0, 128)       # green 50 %, inside the clip polygon
    assert tuple(pixel(img, -0.75, 0.0)) == (0, 0, 0, 0)          # the same rectangle outside the polygon: clipped away
    assert tuple(pixel(img, 0.1, 0.3)) == (0, 0, 255, 255)        # blue: inside polygon AND inside the nested clip |x| < 0.25
    assert tuple(pixel(img, 0.4, 0.3)) == (0, 0, 0, 0)            # blue rectangle outside the nested clip
    assert tuple(pixel(img, 0.0, 0.8)) == (0, 0, 0, 0)            # clip shapes themselves are never coloured
    assert tuple(pixel(img, 0.8, -0.6)) == (191, 191, 0, 191)     # drawn after both UnClips, outside the polygon: not clipped
    # without the UnClips the last shape would be confined to the clips
    clipped = oracle_image(batch, t, c, [d for d in draws if d[2] != Op.UnClip][:-2] + [(4, 4, Op.Stencil, 2, 0), (4, 4, Op.Color, 2, 0)])
    assert tuple(pixel(clipped, 0.8, -0.6)) == (0, 0, 0, 0)


def test_oracle_alpha_context_known_answers(oracle_lib):
    batch, t, c, draws = alpha_scene()
    img = oracle_image(batch, t, c, draws)
    base = oracle_image(batch, t, c, draws[:2])
    assert tuple(pixel(base, 0.8, 0.8)) == tuple(pixel(img, 0.8, 0.8)) == (26, 51, 102, 128)  # outside the group: untouched
    # inside the group, away from its content: alpha goes 0.5 -> scale: (1-a) + 0.5*a with a = 0.25 -> 0.875
    #                                          -> restore: 0.875 - (1 - 0.5) * (1 - 0.25) = 0.5 again
    assert tuple(pixel(img, 0.5, 0.5)) == (26, 51, 102, 128)
    # under the opaque red content: colour = red over background, alpha: scaled 0.875 -> over with a=1 -> 1 -> restore 1 - 0.375 = 0.625
    assert tuple(pixel(img, -0.2, 0.4)) == (255, 0, 0, 159)


def test_oracle_draw_validation(oracle_lib):
    from oracle.binding import Oracle, render_draws
    batch, t, c, draws = clip_scene()
    o = Oracle(batch)
    with pytest.raises(RuntimeError, match="2"):  # ClipStackOverflow: depth 2 does not fit 1 clip bit
        render_draws(o, 64, 64, 1, 4, 1, 0, t, c, [tuple(int(v) for v in d) for d in draws])
    with pytest.raises(RuntimeError, match="3"):  # TooManyNestedOpacityGroups
        render_draws(o, 64, 64, 1, 4, 4, 1, t, c, [(0, 0, int(Op.SaveAlphaContext), 0, 1)])


CASES = {"clip": clip_scene, "alpha": alpha_scene, "instanced": instanced_scene}


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("msaa", [1, 4])
def test_recorded_pass_matches_the_oracle(case, msaa, oracle_lib):
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    batch, t, c, draws = CASES[case]()
    size = 256 if case == "instanced" else 192
    r = R.Renderer(R.Configuration(msaa_sample_count=msaa, clip_nesting_counter_bits=4, winding_counter_bits=4, alpha_layer_count=2), device=0)
    scene = R.Scene(r, batch)
    assert scene.status() == 0
    frame = R.Frame(r, size, size)
    frame.clear()
    scene.render_draws(frame, t, c, draws)
    image = frame.download()
    expect = oracle_image(batch, t, c, draws, size=size, msaa=msaa)
    diff = (image != expect).any(axis=2)
    assert not diff.any(), f"{case} msaa {msaa}: {diff.sum()} pixels differ"
    assert (image[..., 3] > 0).mean() > 0.02
    # the plain pass afterwards still works on the same frame / scene objects
    if case != "instanced":
        frame.clear()
        scene.render(frame, t, c)
        assert np.array_equal(frame.download(), oracle_lib.Oracle(batch).render(size, size, msaa, 4, t, c))


@pytest.mark.gpu
def test_recorded_pass_errors_match_the_reference(oracle_lib):
    from contrast_renderer_amd import ContrastError, _ffi
    from contrast_renderer_amd import renderer as R
    batch, t, c, draws = clip_scene()
    r = R.Renderer(R.Configuration(msaa_sample_count=1, clip_nesting_counter_bits=1, winding_counter_bits=4, alpha_layer_count=1), device=0)
    scene = R.Scene(r, batch)
    frame = R.Frame(r, 64, 64)
    with pytest.raises(ContrastError) as e:
        scene.render_draws(frame, t, c, draws)  # depth 2 with one clip bit (renderer.rs:933-935)
    assert e.value.status == _ffi.ERR_CLIP_STACK_OVERFLOW
    with pytest.raises(ContrastError) as e:
        scene.render_draws(frame, t, c, [(0, 0, Op.SaveAlphaContext, 0, 1)])  # layer 1 of 1 (renderer.rs:947-949)
    assert e.value.status == _ffi.ERR_TOO_MANY_NESTED_OPACITY_GROUPS
    with pytest.raises(ContrastError):
        scene.render_draws(frame, t, c, [(9, 0, Op.Stencil, 0, 0)])  # no such Shape


@pytest.mark.gpu
@pytest.mark.parametrize("msaa", [1, 4])
@pytest.mark.parametrize("pin", ["CRH_EDGE_PASS", "CRH_TRIANGLE_PASS"])
def test_an_opaque_cover_that_does_not_overwrite_its_tiles(msaa, pin, oracle_lib, monkeypatch):
    """The raster kernel's late start of a tile's list must notice when the opaque cover it relies on does not paint every sample: a
    counter-clockwise rectangle is stencilled (winding -1) between an opaque background and an opaque foreground; inside it the
    foreground's +1 sums to zero, the stencil test fails, and the BACKGROUND stays visible (renderer.rs:340-354, 577-582) — in tiles that
    lie wholly inside all three shapes the kernel starts behind the background's cover, finds the foreground not overwriting, and does
    the tile again from the top. Also with a translucent background (the colour under the hole then needs the whole history)."""
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    monkeypatch.setenv(pin, "1")
    hole = Path.from_rect((0.1, -0.05), (0.45, 0.4))
    hole.reverse()
    shapes = [rect(0.0, 0.0, 0.95, 0.95), rect(0.05, 0.0, 0.9, 0.85), ([], [hole]), rect(0.0, 0.0, 0.8, 0.8)]
    batch = batch_from_shapes(shapes)
    t = np.tile(IDENTITY, (4, 1))
    draws = [(0, 0, Op.Stencil, 0, 0), (0, 0, Op.Color, 0, 0), (1, 1, Op.Stencil, 0, 0), (1, 1, Op.Color, 0, 0),
             (2, 2, Op.Stencil, 0, 0),                                       # no Color: the winding stays in the stencil
             (3, 3, Op.Stencil, 0, 0), (3, 3, Op.Color, 0, 0)]
    for background_alpha in (1.0, 0.5):
        c = np.array([[0.2, 0.3, 0.9, 1.0], [0.1, 0.8, 0.3, background_alpha], [0, 0, 0, 1], [0.9, 0.2, 0.1, 1.0]], dtype=np.float32)
        r = R.Renderer(R.Configuration(msaa_sample_count=msaa, clip_nesting_counter_bits=4, winding_counter_bits=4, alpha_layer_count=2), device=0)
        scene = R.Scene(r, batch)
        assert scene.status() == 0
        frame = R.Frame(r, 256, 256)
        frame.clear()
        scene.render_draws(frame, t, c, draws)
        image = frame.download()
        expect = oracle_image(batch, t, c, draws, size=256, msaa=msaa)
        assert np.array_equal(image, expect), f"{(image != expect).any(axis=2).sum()} pixels differ"
        inside_hole, outside = image[128 + 6, 128 + 13], image[128 + 90, 128]  # (y down) a pixel inside the hole, one below it inside the foreground
        assert tuple(outside[:3]) == (230, 51, 26) and tuple(inside_hole[:3]) != (230, 51, 26)  # the hole shows what lies under the foreground


@pytest.mark.gpu
@pytest.mark.parametrize("msaa", [1, 4])
def test_rgba8_attachment_rounds_at_every_blend(msaa, oracle_lib):
    """CRH_FORMAT_RGBA8_ATTACHMENT: the frame behaves like the Rgba8Unorm colour attachment the reference blends into (renderer.rs:736-754,
    examples/showcase/main.rs:32-43,205-215) — every blend reads and writes 8-bit components — where CRH_FORMAT_RGBA8 keeps f32 colours for
    the pass and rounds once. A stack of translucent Shapes: (a) the device equals the oracle run the same way, on all three raster
    formulations; (b) at msaa 1 it is EXACTLY what Shape-by-Shape passes over existing content give on a plain RGBA8 frame (each pass loads
    the rounded pixels and stores them rounded); (c) it differs from the single-rounding frame by a few 1/255 at most, somewhere."""
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    from contrast_renderer_amd import scenes
    sc = scenes.scene_cubic_fill(300, (256, 192), r_lo=10.0, r_hi=70.0, config_index=13)
    colors = np.asarray(sc["colors"], np.float32).copy()
    colors[:, 3] = np.random.RandomState(3).uniform(0.15, 0.6, len(colors))  # all translucent: deep stacks of blends
    oracle = oracle_lib.Oracle(sc["batch"], 4)
    expect = oracle.render(256, 192, msaa, 4, sc["transforms"], colors, attachment8=True)
    once = oracle.render(256, 192, msaa, 4, sc["transforms"], colors)
    worst = np.abs(expect.astype(int) - once.astype(int)).max()
    assert 1 <= worst <= 12, worst
    r = R.Renderer(R.Configuration(msaa_sample_count=msaa, winding_counter_bits=4), device=0)
    scene = R.Scene(r, sc["batch"])
    for pin in ("CRH_EDGE_PASS", "CRH_TRIANGLE_PASS", "CRH_ROWS"):
        os.environ[pin] = "1"
        try:
            frame = R.Frame(r, 256, 192, R.FORMAT_RGBA8_ATTACHMENT)
            frame.clear()
            scene.render(frame, sc["transforms"], colors)
            got = frame.download()
        finally:
            del os.environ[pin]
        assert np.array_equal(got, expect), f"{pin}: {(got != expect).any(axis=2).sum()} pixels differ, max {np.abs(got.astype(int) - expect.astype(int)).max()}"
    if msaa == 1:  # Shape by Shape over existing content, plain RGBA8: the same bytes
        plain = R.Frame(r, 256, 192)
        plain.clear()
        n = sc["batch"].n_shapes
        for k in range(n):
            one = R.Scene(r, sc["batch"].slice_shapes(k, k + 1))
            one.render(plain, sc["transforms"][k:k + 1], colors[k:k + 1])
        assert np.array_equal(plain.download(), expect)


# ---- pass state that spans Shape objects (renderer.rs:148-158, 257-266: the stencil attachment and the alpha layers are caller-owned, so
#      `a.render(Clip)` clips whatever Shapes are rendered afterwards, until `a.render(UnClip)`). The device draws every Shape from an object of
#      its own (Shape.from_paths, one crh_scene_render_draws per run of draws of one object); the oracle draws the same draws in one pass over one batch.
def _separate_shapes(r, shapes):
    from contrast_renderer_amd import renderer as R
    return [R.Shape.from_paths(r, opts, paths) for opts, paths in shapes]


def _submit_per_shape(r, frame, objects, t, c, draws):
    from contrast_renderer_amd import renderer as R
    rp = R.RenderPass(r, frame)
    for i in range(len(t)):
        rp.push_instance(t[i], c[i])
    for shape, instance, op, clip_depth, layer in draws:
        rp.set_clip_depth(clip_depth)
        rp.set_alpha_layer(layer)
        objects[shape].render_in(rp, range(instance, instance + 1), op)
    rp.submit()


def _shapes_of(case):
    if case == "clip":
        return [([], [Path.from_regular_polygon((0.0, 0.0), 0.6, 0.0, 24)]), rect(-0.3, 0.0, 0.5, 0.2), rect(0.3, 0.3, 0.5, 0.2), rect(0.0, 0.0, 0.25, 0.9), rect(0.0, -0.6, 0.9, 0.1)]
    return [rect(0.0, 0.0, 0.9, 0.9), rect(0.0, 0.0, 0.6, 0.6), rect(-0.2, 0.1, 0.3, 0.5), rect(0.3, -0.2, 0.4, 0.2)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["clip", "alpha"])
@pytest.mark.parametrize("msaa", [1, 4])
@pytest.mark.parametrize("fmt", ["rgba8", "attachment"])
def test_pass_state_spans_shape_objects(case, msaa, fmt, oracle_lib):
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    from oracle.binding import Oracle, render_pass
    batch, t, c, draws = CASES[case]()
    # rotate / shift a little so that edges cross sample positions of the 4x pattern
    t = t.copy().reshape(-1, 4, 4)
    t[:, 0, 1] = 0.07
    t[:, 1, 0] = -0.05
    t = t.reshape(-1, 16)
    r = R.Renderer(R.Configuration(msaa_sample_count=msaa, clip_nesting_counter_bits=4, winding_counter_bits=4, alpha_layer_count=2), device=0)
    objects = _separate_shapes(r, _shapes_of(case))
    frame = R.Frame(r, 192, 192, R.FORMAT_RGBA8_ATTACHMENT if fmt == "attachment" else R.FORMAT_RGBA8)
    for _ in range(2):  # twice: crh_frame_clear drops the state the first round left (LoadOp::Clear of the stencil, main.rs:217-230)
        frame.clear()
        _submit_per_shape(r, frame, objects, t, c, draws)
        image = frame.download()
        expect, _ = render_pass(Oracle(batch), 192, 192, msaa, 4, 4, 2, t, c, [tuple(int(v) for v in d) for d in draws], attachment8=fmt == "attachment")
        diff = (image != expect).any(axis=2)
        assert not diff.any(), f"{case} msaa {msaa} {fmt}: {diff.sum()} pixels differ"
        assert (image[..., 3] > 0).mean() > 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("msaa", [1, 4])
def test_an_open_clip_confines_later_passes_of_any_kind(msaa, oracle_lib):
    """Clip by Shape A in one pass; a whole Scene drawn at clip depth 1 in the next is confined to A; the winding a Stencil leaves without its
    cover stays for the pass after it (Shape::render(Stencil) in one pass, render(Color) in the next); after UnClip the plain Stencil + Color loop
    (crh_scene_render — the pass the benchmark times, otherwise the edge formulation) draws over what is there, unconfined."""
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    from oracle.binding import Oracle, render_pass
    rng = np.random.RandomState(11)
    content = []
    for i in range(14):
        cx, cy = rng.uniform(30, 162, 2)
        content.append(([], [Path.from_regular_polygon((cx, cy), rng.uniform(8, 40), rng.uniform(0, 1), 3 + i % 6)]) if i % 2 else rect(cx, cy, rng.uniform(5, 45), rng.uniform(5, 30)))
    n = len(content)
    clip = ([], [Path.from_regular_polygon((96.0, 96.0), 70.0, 0.3, 7)])
    late = rect(96.0, 40.0, 80.0, 12.0)
    pix = np.array([2.0 / 192, 0.003, 0, 0, -0.002, 2.0 / 192, 0, 0, 0, 0, 1, 0, -1, -1, 0, 1], dtype=np.float32)
    t_scene = np.tile(pix, (n, 1))
    c_scene = np.concatenate([rng.uniform(0, 1, (n, 3)), rng.uniform(0.3, 1, (n, 1))], axis=1).astype(np.float32)
    c_scene[::3, 3] = 1.0
    one_t, col_a, col_b = pix.reshape(1, 16), np.array([[1, 1, 1, 1]], dtype=np.float32), np.array([[0.9, 0.2, 0.1, 0.6]], dtype=np.float32)
    r = R.Renderer(R.Configuration(msaa_sample_count=msaa, clip_nesting_counter_bits=2, winding_counter_bits=4, alpha_layer_count=0), device=0)
    a, b = R.Shape.from_paths(r, *clip), R.Shape.from_paths(r, *late)
    scene = R.Scene(r, batch_from_shapes(content))
    frame = R.Frame(r, 192, 192, R.FORMAT_RGBA8_ATTACHMENT)
    frame.clear()
    a.render_draws(frame, one_t, col_a, [(0, 0, Op.Stencil, 0, 0), (0, 0, Op.Clip, 1, 0)])
    inside = []
    for i in range(n):
        inside += [(i, i, Op.Stencil, 1, 0), (i, i, Op.Color, 1, 0)]
    scene.render_draws(frame, t_scene, c_scene, inside)
    b.render_draws(frame, one_t, col_b, [(0, 0, Op.Stencil, 1, 0)])   # the Stencil alone ...
    b.render_draws(frame, one_t, col_b, [(0, 0, Op.Color, 1, 0)])     # ... its cover in the next pass
    a.render_draws(frame, one_t, col_a, [(0, 0, Op.UnClip, 0, 0)])
    scene.render(frame, t_scene, c_scene)                             # the plain loop over what is there, unconfined
    image = frame.download()
    # the oracle: everything in one batch, one pass
    batch = batch_from_shapes(content + [clip, late])
    t = np.concatenate([t_scene, one_t, one_t])
    c = np.concatenate([c_scene, col_a, col_b])
    ia, ib = n, n + 1
    draws = [(ia, ia, Op.Stencil, 0, 0), (ia, ia, Op.Clip, 1, 0)] + inside + [(ib, ib, Op.Stencil, 1, 0), (ib, ib, Op.Color, 1, 0), (ia, ia, Op.UnClip, 0, 0)]
    for i in range(n):
        draws += [(i, i, Op.Stencil, 0, 0), (i, i, Op.Color, 0, 0)]
    expect, _ = render_pass(Oracle(batch), 192, 192, msaa, 4, 2, 0, t, c, [tuple(int(v) for v in d) for d in draws], attachment8=True)
    diff = (image != expect).any(axis=2)
    assert not diff.any(), f"msaa {msaa}: {diff.sum()} pixels differ"
    # (the clip does confine something in this scene: the content drawn at depth 1 behind the Clip covers less than the same content drawn freely)
    confined = oracle_image(batch, t, c, draws[:2 + len(inside)], size=192, msaa=1, clip_bits=2, layers=0)
    free = oracle_image(batch, t, c, [(i, i, op, 0, 0) for i in range(n) for op in (Op.Stencil, Op.Color)], size=192, msaa=1, clip_bits=2, layers=0)
    assert 0 < (confined[..., 3] > 0).sum() < (free[..., 3] > 0).sum()


def test_which_recorded_passes_leave_state_with_the_frame():
    """Host logic of the C ABI (no device): a recorded pass that ends with an open Clip, a Stencil nobody covered or a saved alpha context makes the frame
    keep its pass state (crh_frame: `carry`); a pass that closes what it opens does not — the plain Stencil + Color loop stays on the fast formulation."""
    import ctypes as C
    from contrast_renderer_amd import _ffi
    lib = _ffi.load_library()
    lib.crh_debug_pass_leaves_state.restype = C.c_int
    lib.crh_debug_pass_leaves_state.argtypes = [C.c_void_p, C.c_uint32]

    def leaves(draws):
        table = np.zeros((max(1, len(draws)), 5), dtype=np.uint32)
        for i, d in enumerate(draws):
            table[i, :len(d)] = [int(v) for v in d]
        return lib.crh_debug_pass_leaves_state(table.ctypes.data, len(draws))
    S, CL, UN, CO, SA, SC, RE = (int(o) for o in (Op.Stencil, Op.Clip, Op.UnClip, Op.Color, Op.SaveAlphaContext, Op.ScaleAlphaContext, Op.RestoreAlphaContext))
    assert leaves([]) == 0
    assert leaves([(0, 0, S, 0, 0), (0, 0, CO, 0, 0), (1, 1, S, 0, 0), (1, 1, CO, 0, 0)]) == 0                       # the plain loop
    assert leaves([tuple(int(v) for v in d) for d in clip_scene()[3]]) == 0                                             # nested clips, all closed
    assert leaves([tuple(int(v) for v in d) for d in alpha_scene()[3]]) == 0                                            # an opacity group, restored
    assert leaves([(0, 0, S, 0, 0)]) == 1                                                                               # a Stencil without its cover
    assert leaves([(0, 0, S, 0, 0), (1, 0, CO, 0, 0)]) == 1                                                             # ... covered by ANOTHER Shape's hull only
    assert leaves([(0, 0, S, 0, 0), (0, 0, CL, 1, 0)]) == 1                                                             # an open Clip
    assert leaves([(0, 0, UN, 0, 0)]) == 1                                                                              # closes a level an earlier pass opened
    assert leaves([(1, 1, S, 1, 0), (1, 1, CO, 1, 0)]) == 1                                                             # drawn at a clip depth nobody opened here
    assert leaves([(0, 0, S, 0, 0), (0, 0, CL, 1, 0), (1, 1, S, 1, 0), (1, 1, CO, 1, 0), (0, 0, UN, 0, 0)]) == 0
    assert leaves([(0, 0, S, 0, 0), (0, 0, CL, 1, 0), (1, 1, S, 1, 0), (1, 1, CL, 2, 0), (0, 0, UN, 1, 0)]) == 1        # UnClip out of order
    assert leaves([(0, 0, SA, 0, 0), (0, 0, SC, 0, 0)]) == 1                                                            # a saved context nobody restored
    assert leaves([(0, 0, RE, 0, 1)]) == 1                                                                              # restores what an earlier pass saved
    assert leaves([(0, 0, S, 0, 0), (0, 0, SA, 0, 0), (0, 0, SC, 0, 0), (0, 0, RE, 0, 0)]) == 1                         # the alpha covers write no stencil: the winding stays
