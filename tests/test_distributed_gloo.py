"""World-size 2 / 3 tests of the multi-GPU path over the gloo backend, on CPU.

The data path under test is exactly what bench.py --gpus N runs (contrast_renderer_amd/distributed.py): contiguous shape-range
sharding, the tile-sliced all-to-all of layers, the ordered premultiplied "over" and the gather to rank 0. No GPU exists here, so
each rank's layer comes from the oracle's software rasterizer (test infrastructure) and the composite from the numpy statement of
k_composite; on the GPU the same functions move the HIP frame through RCCL.
"""
import os
import tempfile

import numpy as np
import pytest

from contrast_renderer_amd import distributed as D


def test_shard_range_is_an_ordered_partition():
    for n in (0, 1, 7, 10000, 100003):
        for world in (1, 2, 3, 8):
            ranges = [D.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_slab_rows_are_tile_aligned_and_cover_the_frame():
    for height in (1, 16, 40, 136, 4096, 8192):
        for world in (1, 2, 3, 8):
            rows = D.slab_rows(height, world)
            assert rows[0][0] == 0 and rows[-1][1] == height
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            assert all(r0 % 16 == 0 for r0, _ in rows if r0 < height)


def test_over_is_ordered():
    rng = np.random.RandomState(3)
    a = rng.randint(0, 256, (4, 4, 4)).astype(np.uint8)
    b = rng.randint(0, 256, (4, 4, 4)).astype(np.uint8)
    a[..., :3] = np.minimum(a[..., :3], a[..., 3:4])  # premultiplied
    b[..., :3] = np.minimum(b[..., :3], b[..., 3:4])
    ab = D.composite_over_reference(np.stack([a, b]))
    ba = D.composite_over_reference(np.stack([b, a]))
    assert not np.array_equal(ab, ba)  # "over" does not commute: the rank order matters
    opaque = b.copy()
    opaque[..., 3] = 255
    assert np.array_equal(D.composite_over_reference(np.stack([a, opaque])), opaque)  # an opaque top layer hides what is below


def _worker(rank, world, init_file, out_dir, width, height, n_shapes):
    import torch
    import torch.distributed as dist
    from contrast_renderer_amd import scenes
    from oracle.binding import Oracle

    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        sc = scenes.scene_mixed(n_shapes, (width, height))
        batch, transforms, colors = sc["batch"], sc["transforms"], sc["colors"]
        begin, end = D.shard_range(n_shapes, rank, world)
        shard = batch.slice_shapes(begin, end)
        oracle = Oracle(shard)
        assert oracle.status() == 0
        layer_np = oracle.render(width, height, 1, 4, transforms[begin:end], colors[begin:end])
        np.save(os.path.join(out_dir, f"layer{rank}.npy"), layer_np)
        layer = torch.from_numpy(layer_np)

        received, (r0, r1) = D.exchange_layers(layer, rank, world)
        assert received.shape == (world, r1 - r0, width, 4)
        np.save(os.path.join(out_dir, f"received{rank}.npy"), received.numpy())
        slab = torch.from_numpy(D.composite_over_reference(received.numpy()))
        image = D.gather_slabs(slab, rank, world, height)
        if rank == 0:
            np.save(os.path.join(out_dir, "image.npy"), image.numpy())
        else:
            assert image is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 136), (3, 200), (2, 16)])
def test_sharded_render_exchange_composite_gather(world, height, oracle_lib):
    import torch.multiprocessing as mp
    from contrast_renderer_amd import scenes
    from oracle.binding import Oracle

    width, n_shapes = 160, 11
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_worker, args=(world, init_file, tmp, width, height, n_shapes), nprocs=world, join=True)
        layers = [np.load(os.path.join(tmp, f"layer{r}.npy")) for r in range(world)]
        rows = D.slab_rows(height, world)
        for r in range(world):  # rank r received slab r of every layer, in rank order
            received = np.load(os.path.join(tmp, f"received{r}.npy"))
            for p in range(world):
                assert np.array_equal(received[p], layers[p][rows[r][0]:rows[r][1]])
        image = np.load(os.path.join(tmp, "image.npy"))
    assert image.shape == (height, width, 4)
    # the gathered image is the ordered composite of the full layers
    assert np.array_equal(image, D.composite_over_reference(np.stack(layers)))
    # and it matches the single-process render of all shapes up to the RGBA8 hand-off of the layers (<= 2/255)
    sc = scenes.scene_mixed(n_shapes, (width, height))
    single = Oracle(sc["batch"]).render(width, height, 1, 4, sc["transforms"], sc["colors"])
    assert np.abs(image.astype(np.int32) - single.astype(np.int32)).max() <= 2
    assert (image[..., 3] > 0).mean() > 0.05  # something was drawn


def _tile_split_worker(rank, world, init_file, out_dir, width, height, n_shapes):
    import torch
    import torch.distributed as dist
    from contrast_renderer_amd import scenes
    from oracle.binding import Oracle

    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        sc = scenes.scene_mixed(n_shapes, (width, height))
        # every rank holds ALL the paths and draws its slab of tile rows only (crh_frame_set_tile_rows on the GPU; here the oracle's frame
        # with everything outside the slab left transparent)
        whole = Oracle(sc["batch"]).render(width, height, 1, 4, sc["transforms"], sc["colors"])
        r0, r1 = D.slab_rows(height, world)[rank]
        layer_np = np.zeros_like(whole)
        layer_np[r0:r1] = whole[r0:r1]
        received, (s0, s1) = D.exchange_layers(torch.from_numpy(layer_np), rank, world)
        assert (s0, s1) == (r0, r1)
        for peer in range(world):  # the all-to-all carried nothing but this rank's own slab
            assert (peer == rank) or not received[peer].numpy().any()
        slab = torch.from_numpy(D.composite_over_reference(received.numpy()))
        image = D.gather_slabs(slab, rank, world, height)
        if rank == 0:
            np.save(os.path.join(out_dir, "image.npy"), image.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 136), (3, 200)])
def test_tile_split_exchange_gathers_the_single_process_frame_exactly(world, height, oracle_lib):
    """The other split of SURVEY.md §8(e) through the same exchange: slabs of tile rows instead of ranges of paths. No layer is composited
    over another one, so the gathered frame EQUALS the single-process render (path sharding: within 2/255)."""
    import torch.multiprocessing as mp
    from contrast_renderer_amd import scenes
    from oracle.binding import Oracle

    width, n_shapes = 160, 11
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_tile_split_worker, args=(world, os.path.join(tmp, "rendezvous"), tmp, width, height, n_shapes), nprocs=world, join=True)
        image = np.load(os.path.join(tmp, "image.npy"))
    sc = scenes.scene_mixed(n_shapes, (width, height))
    assert np.array_equal(image, Oracle(sc["batch"]).render(width, height, 1, 4, sc["transforms"], sc["colors"]))


def test_shard_bytes_equal_the_same_shapes_of_the_full_batch(oracle_lib):
    """Sharding must not change a single emitted byte: shape i of the full batch == shape i - begin of its shard."""
    from contrast_renderer_amd import scenes
    from oracle.binding import Oracle

    sc = scenes.scene_mixed(13, (256, 256))
    full = Oracle(sc["batch"])
    for world in (2, 3):
        for rank in range(world):
            begin, end = D.shard_range(13, rank, world)
            shard = Oracle(sc["batch"].slice_shapes(begin, end))
            for i in range(begin, end):
                assert full.shape_status(i) == shard.shape_status(i - begin)
                for a, b in zip(full.shape(i), shard.shape(i - begin)):  # offsets[8], offsets[3], vertex bytes, index bytes
                    assert np.array_equal(a, b)
