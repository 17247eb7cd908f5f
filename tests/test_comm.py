"""The multi-GPU exchange behind the C ABI (include/contrast_hip.h crh_comm_*, csrc/comm.hip).

CPU: the sharding arithmetic the C ABI exports equals contrast_renderer_amd/distributed.py (which the gloo tests drive end to end).
GPU (one device): an in-process loopback group of 2 / 3 / 5 communicators runs the whole exchange — occupancy bitmaps, packed non-empty
tiles, slab all-to-all, ordered composite, gather, unpack — with device-to-device copies in place of the RCCL transfers (RGBA8 and
RGBA16F layers; BASELINE configs[3] whole: eight shards of the 100 000-path scene at 8192x8192), and an RCCL communicator of world
size 1 runs ncclCommInitRank / ncclAllGather for real (its only peer is itself, so no ncclSend / ncclRecv is issued: the first real
point-to-point transfer of this code happens on a multi-GPU node)."""
import os

import numpy as np
import pytest

from contrast_renderer_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
@pytest.mark.parametrize("world,size", [(2, (256, 256)), (3, (200, 136)), (5, (96, 40)), (4, (133, 75))])
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
@pytest.mark.parametrize("world,size,msaa,kind", [(2, (256, 256), 1, "mixed"), (3, (200, 136), 4, "mixed"), (8, (512, 384), 1, "cubic"), (5, (96, 40), 1, "mixed")])
def test_tile_split_gathers_the_single_gpu_frame_bit_for_bit(world, size, msaa, kind, oracle_lib):
    """SURVEY.md §8(e)'s other split: every rank holds ALL paths (uploads, tessellates, bins everything) and draws only the tile rows of its
    slab (crh_frame_set_tile_rows); the same crh_frame_exchange then has nothing to send in its all-to-all and nothing to composite, and the
    gathered image is the single-GPU frame — EQUAL, not within 2/255 — which is the oracle's. Several passes in a row (lists in place,
    slab-restricted tile order), a slab moved between passes, and the whole frame given back."""
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    from contrast_renderer_amd import scenes
    from oracle.binding import Oracle
    sc = scenes.scene_mixed(40, size, seed=world) if kind == "mixed" else scenes.scene_cubic_fill(300, size, r_lo=4.0, r_hi=40.0)
    r = R.Renderer(R.Configuration(msaa, 4, 4), device=0)
    scene = R.Scene(r, sc["batch"])
    scene.set_instances(sc["transforms"], sc["colors"])
    whole = Oracle(sc["batch"]).render(size[0], size[1], msaa, 4, sc["transforms"], sc["colors"])
    layers = [R.Frame(r, *size) for _ in range(world)]
    for rank, f in enumerate(layers):
        f.set_tile_rows(*R.slab_rows(size[1], rank, world))
    comms = [R.Comm(r, 0, world)]
    comms += [R.Comm(r, k, world, rank0=comms[0]) for k in range(1, world)]
    result = R.Frame(r, *size)
    for step in range(5):  # (the third pass on has its lists in place and the slab's tiles in the verified pass' order)
        for f in layers:
            f.clear()
            scene.render(f)
        comms[0].local_exchange(layers, result)
        assert np.array_equal(result.download(), whole), f"pass {step}"
    for rank, f in enumerate(layers):  # a rank's layer holds its slab and nothing else
        r0, r1 = R.slab_rows(size[1], rank, world)
        image = f.download()
        assert np.array_equal(image[r0:r1], whole[r0:r1]) and not image[:r0].any() and not image[r1:].any()
    # nothing but the gather travelled: every rank's tiles outside its own slab are empty
    for rank, c in enumerate(comms):
        assert sum(c.last_peer_bytes()) == 0, rank
    # the slabs dealt the other way round, then the first layer whole again
    for rank, f in enumerate(layers):
        f.set_tile_rows(*R.slab_rows(size[1], world - 1 - rank, world))
        f.clear()
        scene.render(f)
    comms[0].local_exchange(layers, result)
    assert np.array_equal(result.download(), whole)
    # the split's own exchange: the slabs straight from the layers' rows into the result frame (no bitmaps, packing, composite, unpacking)
    # (rank k's layer holds slab k again: that is what the gather takes from it)
    for _ in range(2):
        for rank, f in enumerate(layers):
            f.set_tile_rows(*R.slab_rows(size[1], rank, world))
            f.clear()
            scene.render(f)
        result.clear()
        comms[0].local_gather_slabs(layers, result)
        assert np.array_equal(result.download(), whole)
    layers[0].set_tile_rows(0, size[1])
    layers[0].clear()
    scene.render(layers[0])
    assert np.array_equal(layers[0].download(), whole)
    with pytest.raises(R.ContrastError):
        layers[0].set_tile_rows(8, size[1])  # not a whole tile row


@pytest.mark.gpu
def test_rccl_exchange_with_itself(oracle_lib):
    """World size 1 over RCCL: ncclCommInitRank, ncclAllGather and an (empty) ncclGroupStart / End pair run for real; the slab "transfer"
    to itself is a device-to-device copy, so ncclSend / ncclRecv are NOT reached here (they need a peer: a multi-GPU node)."""
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    from contrast_renderer_amd import scenes
    r, sc, layers, keep = _render_shards(R, scenes, 1, (320, 200), 30, seed=9)
    comm = R.Comm(r, 0, 1, unique_id=R.comm_unique_id(r.lib))
    result = R.Frame(r, 320, 200)
    comm.exchange(layers[0], result)
    assert np.array_equal(result.download(), layers[0].download())
    info = comm.info()  # RCCL's own word on the communicator
    assert info["nranks"] == 1 and info["rccl_version"] > 20000, info
    # and again into the same frames while the renderer already draws the next step into the layer's sibling
    comm.exchange(layers[0], result)
    assert np.array_equal(result.download(), layers[0].download())
    # the tile split's exchange over RCCL (world 1: the header all-gather and an empty group run for real, the slab is a device-to-device copy)
    other = R.Frame(r, 320, 200)
    comm.gather_slabs(layers[0], other)
    assert np.array_equal(other.download(), layers[0].download())
    assert comm.last_timing()["gather"] >= 0.0


@pytest.mark.gpu
def test_exchange_refuses_layers_of_different_sizes_and_formats(oracle_lib):
    """Every rank's layer has the same size and format (the collectives' counts follow from them): checked before anything is sent."""
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    r = R.Renderer(R.Configuration(1, 4, 4), device=0)
    comms = [R.Comm(r, 0, 2)]
    comms.append(R.Comm(r, 1, 2, rank0=comms[0]))
    result = R.Frame(r, 64, 64)
    for other in (R.Frame(r, 64, 48), R.Frame(r, 64, 64, R.FORMAT_RGBA16F)):
        with pytest.raises(R.ContrastError) as e:
            comms[0].local_exchange([R.Frame(r, 64, 64), other], result)
        assert e.value.status == 10  # CRH_ERR_INVALID_ARGUMENT
    with pytest.raises(R.ContrastError):  # the result frame is RGBA8 (the composite quantises once)
        comms[0].local_exchange([R.Frame(r, 64, 64), R.Frame(r, 64, 64)], R.Frame(r, 64, 64, R.FORMAT_RGBA16F))
    with pytest.raises(R.ContrastError):  # rank out of range
        R.Comm(r, 2, 2, rank0=comms[0])
    comms[0].local_exchange([R.Frame(r, 64, 64), R.Frame(r, 64, 64)], result)  # two cleared layers: a transparent image
    assert not result.download().any()


@pytest.mark.gpu
@pytest.mark.parametrize("world,size,msaa,kind", [(4, (512, 384), 1, "cubic"), (2, (270, 150), 1, "cubic"), (4, (320, 256), 1, "mixed"), (3, (200, 136), 4, "mixed")])
def test_rgba16f_layers_keep_the_exchange_within_one_255th(world, size, msaa, kind, oracle_lib):
    """layers exchanged as RGBA16F -> the composite is within 1/255 of the single-GPU render of the whole scene (RGBA8
    layers: 2/255). The layer itself is the resolved colour rounded to binary16: within half a unit of the RGBA8 layer everywhere.
    The bound is a statement about ROUNDING. Two properties of path-sharded compositing are outside it and are checked as what they are:
      * a Shape may leave a winding on a sample outside its own hull, which the reference's NEXT Shape::render(Color) then tests
        (renderer.rs:747-752 only zeroes the stencil inside the hull strip); when that next Shape belongs to another rank, the sample is not
        painted. The structured scenes (closed cubic fills) have none; the unstructured one has a single such pixel of 82 000;
      * with msaa 4 a layer is RESOLVED before it travels, and the composite of box averages is not the box average of per-sample
        composites wherever Shapes of two ranks share a pixel edge: there the exchange is exact about the layers, not about the samples."""
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    from contrast_renderer_amd import scenes
    from oracle.binding import Oracle
    sc = scenes.scene_cubic_fill(600, size, r_lo=6.0, r_hi=48.0) if kind == "cubic" else scenes.scene_mixed(60, size, seed=21 + world)
    n = sc["batch"].n_shapes
    r = R.Renderer(R.Configuration(msaa, 4, 4), device=0)
    layers16, layers8, keep = [], [], []
    for rank in range(world):
        b, e = D.shard_range(n, rank, world)
        scene = R.Scene(r, sc["batch"].slice_shapes(b, e))
        keep.append(scene)
        for fmt, out in ((R.FORMAT_RGBA16F, layers16), (R.FORMAT_RGBA8, layers8)):
            frame = R.Frame(r, *size, fmt)
            frame.clear()
            scene.render(frame, sc["transforms"][b:e], sc["colors"][b:e])
            out.append(frame)
    comms = [R.Comm(r, 0, world)]
    comms += [R.Comm(r, k, world, rank0=comms[0]) for k in range(1, world)]
    result = R.Frame(r, *size)
    comms[0].local_exchange(layers16, result)
    image = result.download()
    sent16 = comms[1].last_traffic()[0]
    halves = np.stack([f.download() for f in layers16])
    assert halves.dtype == np.float16
    assert np.array_equal(image, D.composite_over_reference(halves))  # bit for bit: sparse tiles, f32 accumulation, one quantisation
    bytes8 = np.stack([f.download() for f in layers8])
    assert np.abs(halves.astype(np.float64) * 255.0 - bytes8).max() <= 0.5 + 255.0 * 2.0 ** -11  # the same colours, rounded to binary16 instead of to 1/255
    comms[0].local_exchange(layers8, result)
    image8 = result.download()
    assert comms[1].last_traffic()[0] < sent16  # 1 KiB tiles travel instead of 2 KiB ones
    if msaa == 1:
        oracle = Oracle(sc["batch"])
        whole = oracle.render(size[0], size[1], msaa, 4, sc["transforms"], sc["colors"])
        delta = np.abs(image.astype(int) - whole.astype(int)).max(axis=2)
        delta8 = np.abs(image8.astype(int) - whole.astype(int)).max(axis=2)
        if kind == "cubic":
            assert delta.max() <= 1 and delta8.max() <= 2
        else:  # the pixels beyond the bound are exactly those where the composite of the ORACLE's own shard renders leaves the whole render
            shard_layers = [oracle.render(size[0], size[1], msaa, 4, sc["transforms"], sc["colors"], *D.shard_range(n, k, world)) for k in range(world)]
            structural = np.abs(D.composite_over_reference(np.stack(shard_layers)).astype(int) - whole.astype(int)).max(axis=2) > 2
            assert structural.sum() <= 4 and not (delta[~structural] > 1).any() and not (delta8[~structural] > 2).any()
    # LoadOp::Load on a 16F layer reads the halves back: rank 1's Shapes drawn over rank 0's layer = the "over" of the two layers (up to
    # the rounding of the intermediate layer, which this way does not exist)
    if msaa == 1:
        b, e = D.shard_range(n, 1, world)
        keep[1].render(layers16[0], sc["transforms"][b:e], sc["colors"][b:e])
        over = halves[1].astype(np.float32) + halves[0].astype(np.float32) * (1.0 - halves[1].astype(np.float32)[..., 3:4])
        assert np.abs(layers16[0].download().astype(np.float32) - over).max() <= 3e-3
    layers16[0].clear()
    keep[0].render(layers16[0])
    assert np.array_equal(layers16[0].download(), halves[0])
    timing = comms[0].last_timing()
    assert set(timing) == set(R.Comm.PHASES) and all(v >= 0.0 for v in timing.values()) and timing["pack"] > 0.0


@pytest.mark.gpu
def test_config4_whole_scene_eight_shards_through_the_exchange(oracle_lib):
    """BASELINE.json configs[3] as a whole on one GPU: the 100 000-path scene at 8192x8192 split into the eight contiguous shards the
    eight ranks would draw, each rendered into its own full-size layer, the eight layers through crh_comm_local_exchange (the C-ABI
    exchange: occupancy bitmaps, sparse slab all-to-all, ordered composite, gather) — against (a) the ordered composite of the eight
    layers computed on the host, bit for bit, and (b) the oracle's single render of all 100 000 paths, within the 2/255 the RGBA8
    hand-off between ranks costs."""
    import torch
    assert torch.cuda.is_available()
    from contrast_renderer_amd import renderer as R
    from contrast_renderer_amd import scenes
    from oracle.binding import Oracle
    world, size, n = 8, (8192, 8192), 100000
    sc = scenes.scene_cubic_fill(n, size, config_index=2)
    r = R.Renderer(R.Configuration(1, 4, 4), device=0)
    layers = []
    for rank in range(world):
        b, e = R.shard_range(n, rank, world)
        assert e - b == 12500
        scene = R.Scene(r, sc["batch"].slice_shapes(b, e))
        assert scene.status() == 0
        frame = R.Frame(r, *size)
        frame.clear()
        scene.render(frame, sc["transforms"][b:e], sc["colors"][b:e])
        frame.synchronize()
        layers.append(frame)
        del scene
    comms = [R.Comm(r, 0, world)]
    comms += [R.Comm(r, k, world, rank0=comms[0]) for k in range(1, world)]
    result = R.Frame(r, *size)
    comms[0].local_exchange(layers, result)
    image = result.download()
    sent, dense = zip(*[c.last_traffic() for c in comms])
    assert all(s < d for s, d in zip(sent, dense))  # a shard of 1/8 of the paths leaves empty tiles, and they do not travel
    # (a) the ordered composite of the eight layers, in row bands (eight full layers as f32 would be 8.6 GB)
    band = 1024
    host_layers = [f.download() for f in layers]
    for y in range(0, size[1], band):
        expect = D.composite_over_reference(np.stack([layer[y:y + band] for layer in host_layers]))
        assert np.array_equal(image[y:y + band], expect), f"rows {y}..{y + band}: the exchange differs from the ordered composite of the layers"
    del host_layers
    # (b) the oracle's render of the whole scene
    oracle = Oracle(sc["batch"], 16)
    assert oracle.status() == 0
    whole = oracle.render(size[0], size[1], 1, 4, sc["transforms"], sc["colors"])
    worst = 0
    for y in range(0, size[1], band):
        worst = max(worst, int(np.abs(image[y:y + band].astype(np.int16) - whole[y:y + band].astype(np.int16)).max()))
    assert worst <= 2, worst
    assert (image[..., 3] > 0).mean() > 0.5


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_and_reports_the_metrics_scene():
    """`python bench.py --gpus 2` with NO launcher (WORLD_SIZE unset) must spawn its two ranks itself and print one line whose `value` is the
    BASELINE scene split two ways (strong), with the weak figure in a side block. One GPU here: both ranks on cuda:0, gloo for the
    barrier, the torch statement of the exchange (RCCL refuses two ranks on one device) — the launch path is what this test is about."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "3", "--paths", "2000", "--size", "1024"],
                          capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [l for l in done.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, done.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "strong"
    assert line["config"]["paths_total"] == 2000 and line["config"]["paths_per_gpu"] == 1000
    assert line["value"] > 0 and abs(line["value"] - 2000 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    weak = line["weak_scaling"]
    assert weak["scaling"] == "weak" and weak["paths_total"] == 4000 and weak["value"] > 0


@pytest.mark.gpu
def test_bench_goes_on_with_the_torch_exchange_when_the_first_c_abi_exchange_fails():
    """The first multi-GPU run on real links is the first execution of the RCCL point-to-point path: if crh_frame_exchange FAILS there, every
    rank must hear of it (one all-reduce of a flag) and the run must go on with the torch.distributed statement of the exchange, the
    line saying which path produced the number. Injected here: both ranks hold a communicator whose exchange raises."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["CRH_BENCH_FAIL_FIRST_EXCHANGE"] = "1"
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "3", "--paths", "2000", "--size", "1024"],
                          capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [l for l in done.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, done.stdout[-2000:]
    line = json.loads(lines[0])
    assert "FALLBACK to torch.distributed (the first crh_frame_exchange failed" in line["config"]["parallelism"], line["config"]["parallelism"]
    assert line["n_gpus"] == 2 and line["value"] > 0
