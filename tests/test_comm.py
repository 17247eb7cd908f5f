"""The multi-GPU exchange behind the C ABI (include/contrast_hip.h crh_comm_*, csrc/comm.hip).

CPU: the sharding arithmetic the C ABI exports equals contrast_renderer_amd/distributed.py (which the gloo tests drive end to end).
GPU (one device): an in-process loopback group of 2 / 3 / 5 communicators runs the whole exchange — occupancy bitmaps, packed non-empty
tiles, slab all-to-all, ordered composite, gather, unpack — with device-to-device copies in place of the RCCL transfers, and an RCCL
communicator of world size 1 runs the real ncclAllGather / grouped send-recv code path against itself."""
import numpy as np
import pytest

from contrast_renderer_amd import distributed as D


def test_c_abi_sharding_matches_the_python_statement():
    from contrast_renderer_amd import renderer as R
    for n in (0, 1, 7, 10000, 100003):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                assert R.shard_range(n, rank, world) == D.shard_range(n, rank, world)
    for height in (1, 16, 40, 136, 4096, 8192):
        for world in (1, 2, 3, 8):
            rows = D.slab_rows(height, world)
            for rank in range(world):
                assert R.slab_rows(height, rank, world) == rows[rank]
    with pytest.raises(R.ContrastError):
        R.shard_range(10, 3, 3)


def _render_shards(R, scenes, world, size, n_shapes, seed):
    from contrast_renderer_amd import distributed
    sc = scenes.scene_mixed(n_shapes, size, seed=seed)
    r = R.Renderer(R.Configuration(1, 4, 4), device=0)
    layers, keep = [], []
    for rank in range(world):
        b, e = distributed.shard_range(sc["batch"].n_shapes, rank, world)
        frame = R.Frame(r, *size)
        frame.clear()
        if e > b:
            scene = R.Scene(r, sc["batch"].slice_shapes(b, e))
            scene.render(frame, sc["transforms"][b:e], sc["colors"][b:e])
            keep.append(scene)
        layers.append(frame)
    return r, sc, layers, keep


@pytest.mark.gpu
@pytest.mark.parametrize("world,size", [(2, (256, 256)), (3, (200, 136)), (5, (96, 40))])
def test_loopback_exchange_equals_the_ordered_composite_of_the_layers(world, size, oracle_lib):
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    from contrast_renderer_amd import scenes
    from oracle.binding import Oracle
    r, sc, layers, keep = _render_shards(R, scenes, world, size, 40, seed=world)
    comms = [R.Comm(r, 0, world)]
    comms += [R.Comm(r, k, world, rank0=comms[0]) for k in range(1, world)]
    result = R.Frame(r, *size)
    comms[0].local_exchange(layers, result)
    image = result.download()
    stack = np.stack([f.download() for f in layers])
    assert np.array_equal(image, D.composite_over_reference(stack))  # bit for bit: the sparse exchange is the dense composite
    whole = Oracle(sc["batch"]).render(size[0], size[1], 1, 4, sc["transforms"], sc["colors"])
    assert np.abs(image.astype(int) - whole.astype(int)).max() <= 2  # RGBA8 hand-off between ranks
    # only non-empty tiles travelled
    sent = [c.last_traffic() for c in comms]
    assert all(s <= d for s, d in sent)
    # an empty shard (a cleared layer) contributes nothing, whatever its buffer held before
    layers[-1].clear()
    comms[0].local_exchange(layers, result)
    stack[-1] = 0
    assert np.array_equal(result.download(), D.composite_over_reference(stack))
    assert comms[-1].last_traffic()[0] < sent[-1][0]  # its layer's tiles no longer travel (only its composited slab does)


@pytest.mark.gpu
def test_rccl_exchange_with_itself(oracle_lib):
    """World size 1 over RCCL: ncclCommInitRank, ncclAllGather and the grouped transfers of crh_frame_exchange run for real."""
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    from contrast_renderer_amd import scenes
    r, sc, layers, keep = _render_shards(R, scenes, 1, (320, 200), 30, seed=9)
    comm = R.Comm(r, 0, 1, unique_id=R.comm_unique_id(r.lib))
    result = R.Frame(r, 320, 200)
    comm.exchange(layers[0], result)
    assert np.array_equal(result.download(), layers[0].download())
    # and again into the same frames while the renderer already draws the next step into the layer's sibling
    comm.exchange(layers[0], result)
    assert np.array_equal(result.download(), layers[0].download())
