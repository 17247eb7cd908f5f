"""Perspective instances, the depth attachment and face culling of the colour cover (renderer.rs:383-390,743-745,
shaders.wgsl:13-27, main.rs:162-202): hand-checkable statements about the oracle on the CPU, then the HIP tile rasterizer against the
oracle through the C ABI, bit for bit (GPU tests)."""
import math

import numpy as np
import pytest

from contrast_renderer_amd import Cap, CurveApproximation, DynamicStrokeOptions, Join, Path, StrokeOptions, batch_from_shapes, scenes, utils
from contrast_renderer_amd.renderer import Compare, Cull
from contrast_renderer_amd.renderer import RenderOperation as Op

IDENTITY = np.eye(4, dtype=np.float32).reshape(-1)
SIZE = 160


def showcase_projection(aspect=1.0):
    return utils.perspective_projection(math.pi * 0.5, aspect, 1.0, 1000.0)  # main.rs:162-167


def placed(x, y, z, tilt=0.0, axis=(1.0, 0.0, 0.0), scale=1.0):
    m = utils.matrix_multiplication(utils.translation_matrix(x, y, z), utils.rotation_matrix(tilt, axis))
    s = np.eye(4, dtype=np.float32)
    s[0, 0] = s[1, 1] = scale
    return utils.matrix_multiplication(utils.matrix_multiplication(showcase_projection(), m), s.reshape(-1))


def decal_scene():
    """Shapes as decals in a 3-D scene: a stroked + filled mixed-curve Shape, a glyph-like polygon and rectangles, each drawn at several
    depths and tilts (instancing), some reaching through the near plane."""
    star = scenes.scene_mixed(3, (256, 256), seed=5)
    shapes_batch = star["batch"]
    transforms = np.stack([placed(-0.6, 0.4, 2.5, 0.9), placed(0.5, 0.3, 1.8, -0.7, (0.0, 1.0, 0.0)), placed(0.0, -0.5, 3.0, 1.2, (0.6, 0.8, 0.0)),
                           placed(0.2, 0.1, 1.15, 1.3), placed(-0.3, -0.2, 2.0, 0.3, (0.0, 0.0, 1.0)), placed(0.0, 0.0, 0.6, 1.45)])
    rng = np.random.RandomState(4)
    colors = np.concatenate([rng.uniform(0, 1, (len(transforms), 3)), rng.uniform(0.4, 1, (len(transforms), 1))], axis=1).astype(np.float32)
    draws = []
    for i in range(len(transforms)):
        draws += [(i % 3, i, Op.Stencil, 0, 0), (i % 3, i, Op.Color, 0, 0)]
    return shapes_batch, transforms, colors, draws


def rect(cx, cy, hx, hy):
    return ([], [Path.from_rect((cx, cy), (hx, hy))])


def depth_scene():
    """Three overlapping rectangles at view distances 3, 2 and 4, drawn in that order: with LessEqual + depth write the nearest wins
    wherever it was drawn before a farther one; a mixed plain / projective pass (the last draw is a screen-space overlay at depth 0)."""
    batch = batch_from_shapes([rect(0.0, 0.0, 1.0, 1.0), rect(0.0, 0.0, 0.6, 1.2), rect(0.0, 0.0, 1.4, 0.5), rect(0.0, 0.0, 0.3, 0.3)])
    overlay = IDENTITY.copy()
    overlay[0], overlay[5] = -0.5, 0.5  # mirrored: the only back-facing instance of the pass
    overlay[12], overlay[13] = 0.55, -0.55  # m14 = 0: a plain instance in front of everything
    transforms = np.stack([placed(-0.2, 0.1, 3.0), placed(0.1, 0.0, 2.0, 0.5, (0.0, 1.0, 0.0)), placed(0.0, -0.1, 4.0), overlay])
    colors = np.array([[1, 0, 0, 1], [0, 1, 0, 1], [0, 0, 1, 1], [1, 1, 0, 0.5]], dtype=np.float32)
    draws = []
    for i in range(4):
        draws += [(i, i, Op.Stencil, 0, 0), (i, i, Op.Color, 0, 0)]
    return batch, transforms, colors, draws


def oracle_pass(batch, transforms, colors, draws, size=SIZE, msaa=1, **state):
    from oracle.binding import Oracle, render_pass
    o = Oracle(batch)
    assert o.status() == 0
    return render_pass(o, size, size, msaa, 4, 2, 0, transforms, colors, [tuple(int(v) for v in d) for d in draws], **state)


def ndc_depth(view_z, near=1.0, far=1000.0):
    return far * (view_z - near) / ((far - near) * view_z)


def exact_depth(m, column, row, size=SIZE):
    """z/w where the ray through the centre of pixel (column, row) meets the instance's plane, in float64 (independent of the oracle)."""
    m = np.asarray(m, dtype=np.float64)
    cx, cy, cz, cw = (np.array([m[r], m[4 + r], m[12 + r]]) for r in range(4))
    H = np.stack([(cx * 0.5 + cw * 0.5) * size, (cw * 0.5 - cy * 0.5) * size, cw])
    model = np.linalg.inv(H) @ np.array([column + 0.5, row + 0.5, 1.0])
    xy1 = model / model[2]
    return float(cz @ xy1) / float(cw @ xy1)


def at(image, x, y, size=SIZE):  # NDC coordinates, y up
    return image[int((0.5 - y * 0.5) * size), int((x * 0.5 + 0.5) * size)]


# ------------------------------------------------------------------------------------------------ oracle, on the CPU
def test_oracle_nearest_surface_wins_with_less_equal_and_depth_write(oracle_lib):
    batch, t, c, draws = depth_scene()
    cleared = np.ones((SIZE, SIZE, 1), dtype=np.float32)
    image, depth = oracle_pass(batch, t, c, draws[:6], depth_compare=Compare.LessEqual, depth_write=1, depth=cleared)
    # centre of the frame: red (z 3) is drawn first, green (z ~2) passes LessEqual over it, blue (z 4) fails against green
    assert tuple(at(image, 0.0, 0.0)) == (0, 255, 0, 255)
    assert abs(depth[SIZE // 2, SIZE // 2, 0] - exact_depth(t[1], SIZE // 2, SIZE // 2)) < 1e-5  # the tilted green plane
    # where only red and blue overlap, red (nearer, drawn first) survives; without the depth test blue paints over it
    assert tuple(at(image, -0.3, 0.05)) == (255, 0, 0, 255)
    no_test, _ = oracle_pass(batch, t, c, draws[:6])
    assert tuple(at(no_test, -0.3, 0.05)) == (0, 0, 255, 255)
    # the written depth of a plane parallel to the screen is the NDC depth of its distance (utils.rs:181-192)
    assert abs(at(depth, -0.3, 0.05)[0] - ndc_depth(3.0)) < 1e-5  # red
    assert depth[2, 2, 0] == 1.0


def test_oracle_depth_test_against_an_existing_scene_and_depth_fail_keeps_the_winding(oracle_lib):
    """HUD / decal occluded by the 3-D scene (README.md:8-12): the uploaded depth buffer hides the left half of a rectangle. The stencil
    of the hidden samples is kept (depth_fail_op Keep, renderer.rs:442), so a later cover of the same area still finds it."""
    batch = batch_from_shapes([rect(0.0, 0.0, 1.0, 1.0)])
    t = np.stack([placed(0.0, 0.0, 3.0), placed(0.0, 0.0, 1.5)])
    c = np.array([[1, 0, 0, 1], [0, 0, 1, 1]], dtype=np.float32)
    wall = np.ones((SIZE, SIZE, 1), dtype=np.float32)
    wall[:, : SIZE // 2] = ndc_depth(2.0)  # a wall at distance 2 covers the left half
    image, depth = oracle_pass(batch, t, c, [(0, 0, Op.Stencil, 0, 0), (0, 0, Op.Color, 0, 0)], depth_compare=Compare.LessEqual, depth_write=0, depth=wall)
    assert tuple(at(image, 0.2, 0.0)) == (255, 0, 0, 255) and tuple(at(image, -0.2, 0.0)) == (0, 0, 0, 0)
    assert np.array_equal(depth, wall)  # no depth write
    # second cover in front of the wall (z 1.5) without a stencil pass of its own: it paints exactly the samples whose winding survived
    image2, _ = oracle_pass(batch, t, c, [(0, 0, Op.Stencil, 0, 0), (0, 0, Op.Color, 0, 0), (0, 1, Op.Color, 0, 0)],
                            depth_compare=Compare.LessEqual, depth_write=0, depth=wall)
    assert tuple(at(image2, -0.2, 0.0)) == (0, 0, 255, 255) and tuple(at(image2, 0.2, 0.0)) == (255, 0, 0, 255)


@pytest.mark.parametrize("compare,expect_drawn", [(Compare.Never, False), (Compare.Less, False), (Compare.Equal, True), (Compare.LessEqual, True),
                                                  (Compare.Greater, False), (Compare.NotEqual, False), (Compare.GreaterEqual, True), (Compare.Always, True)])
def test_oracle_compare_functions_at_equal_depth(oracle_lib, compare, expect_drawn):
    batch = batch_from_shapes([rect(0.0, 0.0, 0.5, 0.5)])
    t = IDENTITY.copy()
    t[14] = 0.25  # a plain instance: constant fragment depth m14
    stored = np.full((SIZE, SIZE, 1), 0.25, dtype=np.float32)
    image, _ = oracle_pass(batch, t.reshape(1, 16), np.array([[1, 1, 1, 1]], dtype=np.float32), [(0, 0, Op.Stencil, 0, 0), (0, 0, Op.Color, 0, 0)],
                           depth_compare=compare, depth=stored)
    assert (at(image, 0.0, 0.0)[3] == 255) == expect_drawn


def test_oracle_cull_mode_drops_the_cover_by_its_screen_orientation(oracle_lib):
    """Under an orientation-preserving transform the hull strip is front-facing (counter-clockwise as displayed, FrontFace::Ccw of
    renderer.rs:477): the showcase culls Back (main.rs:46) and still sees its Shapes. A mirrored instance flips the facing. Culling acts on
    the colour cover only: the stencil passes are built with cull None (renderer.rs:565-690)."""
    batch = batch_from_shapes([rect(0.0, 0.0, 0.5, 0.5)])
    mirrored = IDENTITY.copy()
    mirrored[0] = -1.0
    c = np.array([[1, 1, 1, 1]], dtype=np.float32)
    draws = [(0, 0, Op.Stencil, 0, 0), (0, 0, Op.Color, 0, 0)]
    drawn = {}
    for name, t in (("plain", IDENTITY), ("mirrored", mirrored)):
        for cull in (Cull.Disabled, Cull.Front, Cull.Back):
            image, _ = oracle_pass(batch, t.reshape(1, 16), c, draws, cull_mode=cull)
            drawn[name, cull] = at(image, 0.0, 0.0)[3] == 255
    assert drawn["plain", Cull.Disabled] and drawn["mirrored", Cull.Disabled]
    assert drawn["plain", Cull.Back] and not drawn["plain", Cull.Front]
    assert drawn["mirrored", Cull.Front] == drawn["plain", Cull.Back] and drawn["mirrored", Cull.Back] == drawn["plain", Cull.Front]


def test_utils_match_their_definitions():
    p = utils.perspective_projection(math.pi * 0.5, 2.0, 1.0, 1000.0)
    assert p[0] == np.float32(0.5) * p[5] and p[11] == 1.0 and p[15] == 0.0
    for z in (1.0, 10.0, 1000.0):
        clip = p.reshape(4, 4).T @ np.array([0.0, 0.0, z, 1.0])
        assert abs(clip[2] / clip[3] - ndc_depth(z)) < 1e-5 and clip[3] == z
    a, b = utils.rotation_matrix(0.7, (0.0, 0.0, 1.0)), utils.translation_matrix(1.0, 2.0, 3.0)
    assert np.allclose(utils.matrix_multiplication(a, b).reshape(4, 4).T, a.reshape(4, 4).T @ b.reshape(4, 4).T, atol=1e-6)
    assert np.allclose(utils.linear_to_srgb(utils.srgb_to_linear([0.2, 0.5, 0.9, 0.3])), [0.2, 0.5, 0.9, 0.3], atol=1e-6)


# ------------------------------------------------------------------------------------------------ HIP against the oracle
def gpu_pass(batch, transforms, colors, draws, size=SIZE, msaa=1, depth=None, **state):
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    config = R.Configuration(msaa_sample_count=msaa, clip_nesting_counter_bits=2, winding_counter_bits=4, alpha_layer_count=0,
                             cull_mode=state.get("cull_mode", 0), depth_compare=state.get("depth_compare", 0), depth_write_enabled=bool(state.get("depth_write", 0)))
    r = R.Renderer(config, device=0)
    scene = R.Scene(r, batch)
    assert scene.status() == 0
    frame = R.Frame(r, size, size)
    frame.clear()
    if depth is not None:
        frame.upload_depth(np.asarray(depth, dtype=np.float32).reshape(size, size))
    scene.render_draws(frame, transforms, colors, draws)
    image = frame.download()
    has_depth = state.get("depth_compare", 0) != 0 or state.get("depth_write", 0)
    return image, (frame.download_depth() if has_depth else None), (r, scene, frame)


@pytest.mark.gpu
@pytest.mark.parametrize("msaa", [1, 4])
def test_perspective_decals_match_the_oracle(oracle_lib, msaa):
    batch, t, c, draws = decal_scene()
    expect, _ = oracle_pass(batch, t, c, draws, msaa=msaa)
    image, _, _ = gpu_pass(batch, t, c, draws, msaa=msaa)
    diff = (image != expect).any(axis=2)
    assert not diff.any(), f"msaa {msaa}: {diff.sum()} pixels differ"
    assert (image[..., 3] > 0).mean() > 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("msaa", [1, 4])
@pytest.mark.parametrize("state", [dict(depth_compare=Compare.LessEqual, depth_write=1), dict(depth_compare=Compare.Greater, depth_write=0),
                                   dict(depth_compare=Compare.Always, depth_write=1, cull_mode=Cull.Back), dict(cull_mode=Cull.Front)])
def test_depth_and_cull_match_the_oracle(oracle_lib, msaa, state):
    batch, t, c, draws = depth_scene()
    rng = np.random.RandomState(2)
    start = np.where(rng.uniform(size=(SIZE, SIZE)) < 0.3, ndc_depth(2.5), 1.0).astype(np.float32)  # a scene depth with holes
    uses_depth = state.get("depth_compare", 0) != 0 or state.get("depth_write", 0)
    expect, expect_depth = oracle_pass(batch, t, c, draws, msaa=msaa, depth=start if uses_depth else None, **state)
    image, depth, _ = gpu_pass(batch, t, c, draws, msaa=msaa, depth=start if uses_depth else None, **state)
    assert np.array_equal(image, expect), f"{(image != expect).any(axis=2).sum()} pixels differ"
    if uses_depth:
        assert np.array_equal(depth, expect_depth)
    assert (image[..., 3] > 0).mean() > 0.01


@pytest.mark.gpu
def test_plain_pass_with_perspective_instances_and_stroked_shapes(oracle_lib):
    """crh_scene_render (Stencil + Color per Shape) with perspective instance transforms: dashed, round-joined strokes and curve fills,
    whose fragment stages read perspective-correct attributes (shaders.wgsl:35-58)."""
    sc = scenes.scene_mixed(12, (SIZE, SIZE), seed=11)
    batch = sc["batch"]
    rng = np.random.RandomState(6)
    n = batch.n_shapes
    t = np.stack([placed(rng.uniform(-1.5, 1.5), rng.uniform(-1.5, 1.5), rng.uniform(1.2, 4.0), rng.uniform(-1.2, 1.2),
                         (math.cos(i), math.sin(i), 0.0)) for i in range(n)])
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    for msaa in (1, 4):
        r = R.Renderer(R.Configuration(msaa_sample_count=msaa, winding_counter_bits=4, clip_nesting_counter_bits=0), device=0)
        scene = R.Scene(r, batch)
        frame = R.Frame(r, SIZE, SIZE)
        frame.clear()
        scene.render(frame, t, sc["colors"])
        expect = oracle_lib.Oracle(batch).render(SIZE, SIZE, msaa, 4, t, sc["colors"])
        image = frame.download()
        assert np.array_equal(image, expect), f"msaa {msaa}: {(image != expect).any(axis=2).sum()} pixels differ"
        assert (image[..., 3] > 0).mean() > 0.03
        # and back to plain instances on the same objects
        frame.clear()
        scene.render(frame, sc["transforms"], sc["colors"])
        assert np.array_equal(frame.download(), oracle_lib.Oracle(batch).render(SIZE, SIZE, msaa, 4, sc["transforms"], sc["colors"]))
