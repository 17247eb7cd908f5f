#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: paths/sec (+ Mpixel/s) of the tessellate + tile-raster hot path.

One step = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
crh_scene_tessellate (count / scan / emit / hull kernels) + crh_scene_render_resident (bin + tile raster kernels) of
BASELINE.json configs[1]: 10 000 mixed integral / rational cubic paths at 4096x4096 on one MI355X.
With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) the paths shard by contiguous index range — every rank
renders its own 10 000-path shard (weak scaling) into a private layer — followed by the tile-sliced RCCL exchange, the ordered
"over" composite and the gather to rank 0.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

# timing mark (HIP events inside the library) -> kernel name as rocprofv3 prints it
MARK_TO_KERNEL = {
    "raster_tiles": "crh::k_raster_tile<1, 4, false, false>",
    "raster_tile_fill": "crh::k_tile_walk<1, true>",
    "raster_tile_count": "crh::k_tile_walk<1, false>",
    "raster_prim_setup": "crh::k_prim_setup<1>",
    "tess_emit": "crh::k_emit",
    "tess_count": "crh::k_count",
    "tess_hull": "crh::k_hull_small",
}


def valu_issue(mark, avg_launch_ms):
    """Secondary roofline of the dominant kernel: VALU issue utilisation = wave-level VALU instructions (SQ_INSTS_VALU of the newest committed
    PMC summary) x 4 cycles per wave64 instruction / (1024 SIMDs x 2.4 GHz x launch time). None when the counters do not cover the kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_counters.json")))
    if not files or mark not in MARK_TO_KERNEL or avg_launch_ms <= 0:
        return None
    with open(files[-1]) as f:
        doc = json.load(f)
    k = doc.get("per_launch", {}).get(MARK_TO_KERNEL[mark])
    if not k or "SQ_INSTS_VALU" not in k:
        return None
    simds, clock_hz = 256 * 4, 2.4e9
    return {"valu_wave_instructions": int(k["SQ_INSTS_VALU"]), "salu_wave_instructions": int(k.get("SQ_INSTS_SALU", 0)),
            "frac_of_valu_issue_peak": k["SQ_INSTS_VALU"] * 4.0 / (simds * clock_hz * avg_launch_ms * 1e-3), "source": os.path.basename(files[-1]),
            "note": "one wave64 VALU instruction per 4 cycles per SIMD; 256 CUs x 4 SIMDs at 2.4 GHz"}


def measured_traffic(mark):
    """HBM bytes per launch of the kernel behind `mark`, from the newest committed PMC summary (profiles/rNN_traffic.json, produced by
    tools/profile.sh: separate FETCH_SIZE / WRITE_SIZE passes with the gfx950 corrections of the microarchitecture guide). The
    counters cannot be collected from inside this process, so the committed file is the source; None when it does not cover the kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files or mark not in MARK_TO_KERNEL:
        return None, None
    with open(files[-1]) as f:
        doc = json.load(f)
    k = doc.get("kernels", {}).get(MARK_TO_KERNEL[mark])
    return (k["hbm_bytes_per_launch"], os.path.basename(files[-1])) if k else (None, None)


class _DeviceArray:
    """Wraps a raw HIP pointer for torch.as_tensor (zero copy)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--paths", type=int, default=10000, help="paths per GPU (configs[1] = 10000)")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"), help="nccl = RCCL over xGMI (the real thing); gloo validates the multi-rank flow "
                    "where RCCL cannot run (e.g. two ranks on one GPU with --same-device): layers are staged through host memory")
    ap.add_argument("--same-device", action="store_true", help="all ranks use cuda:0 (validation only)")
    ap.add_argument("--check", action="store_true", help="N > 1: rank 0 also renders every shard itself and compares the composite of those layers with the gathered image")
    ap.add_argument("--workload", default="cubic", choices=("cubic", "glyphs", "dashed"),
                    help="cubic = BASELINE configs[1] (the metric's configuration, default); glyphs = configs[2] (50 000 glyphs @ 2048^2); "
                         "dashed = configs[4] (2 000 dashed rational-cubic strokes @ 4096^2, msaa 4). Only `cubic` is the headline line.")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    from contrast_renderer_amd import distributed as D
    from contrast_renderer_amd import scenes
    from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene

    size = (args.size, args.size)
    if args.workload == "cubic":
        sc = scenes.scene_cubic_fill(args.paths, size, config_index=2, first_path=rank * args.paths)
        workload = (f"BASELINE configs[1]: {args.paths} filled closed paths x 8 cubic segments (alternating integral / rational), {size[0]}x{size[1]}, "
                    "msaa 1, winding_counter_bits 4")
    elif args.workload == "glyphs":
        size = (2048, 2048)
        sc = scenes.scene_glyphs(50000, size)
        args.paths = sc["n_paths"]
        workload = f"BASELINE configs[2]: 50000 glyph instances ({sc['n_paths']} line/quadratic paths) via text::paths_of_text, 2048x2048, msaa 1"
    else:
        size = (4096, 4096)
        sc = scenes.scene_dashed_strokes(2000, size)
        args.paths = 2000
        workload = "BASELINE configs[4]: 2000 dashed rational-cubic strokes (UniformTangentAngle 0.1, miter/round joins), 4096x4096, msaa 4"
    batch = sc["batch"]
    renderer = Renderer(Configuration(msaa_sample_count=sc["msaa"], clip_nesting_counter_bits=4, winding_counter_bits=4), device=local_rank)
    t_up = time.perf_counter()
    scene = Scene(renderer, batch, tessellate=True)  # host -> HBM + first tessellation (sizes the output buffers): outside the timed region
    renderer.synchronize()
    upload_s = time.perf_counter() - t_up  # validation + element stream + H2D + first tessellation, once per scene
    scene.check()
    scene.set_instances(sc["transforms"], sc["colors"])
    frame = Frame(renderer, *size)
    lib = renderer.lib
    import ctypes as C

    # N > 1: two frames, so that the framebuffer exchange of step i (RCCL + composite + gather) runs while step i + 1 is being
    # tessellated and rasterized into the other frame
    frames = [frame] + ([Frame(renderer, *size)] if world > 1 else [])
    layer_views, slab = [], None
    if world > 1:
        layer_views = [torch.as_tensor(_DeviceArray(f.device_pointer(), (size[1], size[0], 4)), device=f"cuda:{local_rank}") for f in frames]
        r0, r1 = D.slab_rows(size[1], world)[rank]
        slab = torch.empty((r1 - r0, size[0], 4), dtype=torch.uint8, device=f"cuda:{local_rank}")

    def launch(i):
        """Enqueues step i's tessellation + render (asynchronous on the renderer's streams)."""
        f = frames[i % len(frames)]
        scene.tessellate()
        f.clear()
        scene.render(f)

    def finish(i):
        """The exchange step of the path for step i's layer; the renderer may already be working on step i + 1."""
        f = frames[i % len(frames)]
        f.synchronize()  # step i's raster kernel only
        received, _ = D.exchange_layers(layer_views[i % len(frames)], rank, world)
        if received.is_cuda:
            torch.cuda.current_stream().synchronize()  # the RCCL transfers (not the renderer's streams: step i + 1 keeps running)
        ptrs = (C.c_void_p * world)(*[received[k].data_ptr() for k in range(world)])
        rc = lib.crh_composite_over(renderer.handle, ptrs, world, received[0].numel() // 4, C.c_void_p(slab.data_ptr()))
        assert rc == 0, rc
        return D.gather_slabs(slab, rank, world, size[1])

    def run(n):
        """n steps; with N > 1 the exchange of step i overlaps the rendering of step i + 1. Returns the last gathered frame (rank 0)."""
        out = None
        for i in range(n):
            launch(i)
            if world > 1 and i > 0:
                out = finish(i - 1)
        if world > 1 and n > 0:
            out = finish(n - 1)
        return out

    def step():
        return run(1)

    def sync():
        renderer.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    run(args.warmup)
    sync()
    scene.check()
    renderer.enable_timing(True)  # HIP events on the renderer's stream between kernels; drained once after the timed region
    sync()
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.backend == "gloo" else f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_times = renderer.kernel_times()
    renderer.enable_timing(False)
    scene.check()
    image = frame.download()
    covered = float((image[..., 3] > 0).mean())
    check = None
    if args.check and world > 1:
        gathered = step()  # one more pass outside the timed region: the gathered frame on rank 0
        sync()
        if rank == 0:
            layers = []
            for other in range(world):  # the same shards, rendered one after the other by this rank alone
                shard = scenes.scene_cubic_fill(args.paths, size, config_index=2, first_path=other * args.paths)
                shard_scene = Scene(renderer, shard["batch"], tessellate=True)
                shard_frame = Frame(renderer, *size)
                shard_frame.clear()
                shard_scene.render(shard_frame, shard["transforms"], shard["colors"])
                layers.append(shard_frame.download())
            expect = D.composite_over_reference(np.stack(layers))
            got = gathered.cpu().numpy()
            check = {"gathered_equals_ordered_composite_of_all_shards": bool(np.array_equal(got, expect)),
                     "max_abs_difference": int(np.abs(got.astype(np.int32) - expect.astype(np.int32)).max()),
                     "own_layer_unchanged": bool(np.array_equal(image, layers[0]))}

    # per-kernel averages
    agg = {}
    for name, ms, nbytes in kernel_times:
        a = agg.setdefault(name, [0.0, 0, nbytes])
        a[0] += ms
        a[1] += 1
        a[2] = max(a[2], nbytes)
    kernels = {k: {"avg_ms": v[0] / v[1], "launches": v[1], "algorithmic_bytes": v[2]} for k, v in agg.items()}
    dominant = max(kernels, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"])
    dk = kernels[dominant]
    achieved = dk["algorithmic_bytes"] / (dk["avg_ms"] * 1e-3) / 1e9 if dk["avg_ms"] > 0 else 0.0
    default_workload = args.workload == "cubic" and args.paths == 10000 and args.size == 4096
    traffic, traffic_source = measured_traffic(dominant) if default_workload else (None, None)

    total_paths = args.paths * world
    ms_per_step = elapsed / args.steps * 1e3
    out = {
        "metric": "paths/sec, 10k mixed-Bezier paths @ 4096^2 (tessellate + tile raster)",
        "value": total_paths / (elapsed / args.steps),
        "unit": "paths/s",
        "mpixel_per_s": size[0] * size[1] / (elapsed / args.steps) / 1e6,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": workload + "; step = tessellate (count/scan/emit/hull) + bin + tile raster, inputs resident in HBM",
            "paths_per_gpu": args.paths,
            "segments_per_gpu": int(batch.n_segments),
            "parallelism": "single GPU" if world == 1 else f"path-index sharding x{world} + tile-sliced RCCL all-to-all + ordered over-composite + gather (exchange of step i overlaps the rendering of step i + 1)",
            "covered_fraction": covered,
        },
        "roofline": {
            "kernel": dominant,
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "algorithmic_bytes": dk["algorithmic_bytes"],
            "valu_issue": valu_issue(dominant, dk["avg_ms"]) if default_workload else None,
            "avg_launch_ms": dk["avg_ms"],
            "note": "achieved = algorithmic bytes (SURVEY.md §8(d): emitted vertex/index bytes read once + 80 B per shape + W*H*4 written once) / "
                    "HIP-event launch time of the dominant kernel; traffic = HBM bytes per launch from rocprofv3 PMC passes (committed under "
                    "profiles/). The kernel is VALU-issue bound (per-sample edge functions), not HBM bound: see DESIGN.md",
        },
        "kernels": kernels,
        "check": check,
        # the boundary hands over host buffers once per scene (crh_scene_upload); never part of `value`
        "host_inclusive": {"upload_ms": upload_s * 1e3, "paths_per_s_first_frame": args.paths / (upload_s + elapsed / args.steps),
                           "input_bytes": int(batch.input_bytes())},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.binding import time_tessellate
        t1 = time_tessellate(batch, 1, 1)
        repeats = max(2, min(200, int(math.ceil(10.0 / max(t1, 1e-3)))))
        ts = time_tessellate(batch, 1, repeats)
        cores = os.cpu_count() or 1
        tall = time_tessellate(batch, cores, max(2, repeats))
        out["cpu_baseline"] = {
            "value": args.paths * repeats / ts,
            "unit": "paths/s",
            "cores": 1,
            "kind": "port",
            "sample": f"{repeats} x full tessellation of the same {args.paths}-path scene by the C++ restatement of the reference's CPU tessellation "
                      f"(Shape::from_paths minus the wgpu upload; the reference itself cannot be built here), single thread as in renderer.rs:187; "
                      "tessellation only — the reference rasterizes on a GPU",
            "all_cores": {"value": args.paths * max(2, repeats) / tall, "cores": cores},
        }
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
