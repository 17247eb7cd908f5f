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


def render_pass_rect(o, c, render_pass):
    """oracle.binding.render_pass takes (width, height); kept in one place so that the argument order cannot drift from the GPU call."""
    return render_pass(o, c["width"], c["height"], c["msaa"], c["winding_bits"], 2, 2, c["transforms"], c["colors"], [tuple(int(v) for v in d) for d in c["draws"]],
                       depth=c["depth"], **c["state"])
