#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: paths/sec (+ Mpixel/s) of the tessellate + tile-raster hot path.

One step = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
crh_scene_tessellate (count / scan / emit / hull kernels) + crh_scene_render_resident (bin + tile raster kernels) of
BASELINE.json configs[1]: 10 000 mixed integral / rational cubic paths at 4096x4096 on one MI355X.

With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) the Shapes shard by contiguous index range and every rank
renders its shard into a private layer; the exchange step — occupancy bitmaps, slab all-to-all of the non-empty tiles, ordered "over"
composite, gather to rank 0 — runs behind the C ABI (crh_frame_exchange, csrc/comm.hip) on RCCL, overlapped with the rendering of the
next step. torch.distributed only carries the 128-byte RCCL id, the barrier and the max-over-ranks of the elapsed time.
  --scaling strong (default at N > 1) THE scene of `--paths` shapes — BASELINE's 10 000-path scene — is split over the ranks: total work
                   fixed, value = paths / step time; the default run then also measures the weak figure and reports it in `weak_scaling`
  --scaling weak   every rank renders its own `--paths` shapes: per-GPU work fixed, value = N x paths / step time
  `python bench.py --gpus N` without a launcher (WORLD_SIZE unset) starts its N ranks itself through torch.distributed.run.
  --workload s100k BASELINE configs[3]: 100 000 paths at 8192x8192 split over the ranks (strong)

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

# timing mark (HIP events inside the library) -> kernel name as rocprofv3 prints it
def mark_to_kernel(workload, triangle_pass=False):
    """`triangle_pass`: the library measured the strip-triangle formulation (raster.hip) to be the faster one for this scene and drew with it"""
    msaa4_strokes = workload == "dashed"
    if triangle_pass:
        return {"raster_tiles": "crh::k_raster_tile<4, 1, false, true>" if msaa4_strokes else "crh::k_raster_tile<1, 4, false, false>",
                "raster_prim_setup": "crh::k_prim_setup<4, false>" if msaa4_strokes else "crh::k_prim_setup<1, false>",
                "tess_emit": "crh::k_emit", "tess_count": "crh::k_count", "tess_hull": "crh::k_hull_small"}
    return {
        # (the last argument: the variant that looks for its late start across the chunks of long tile lists — the host picks it for frames with
        # many entries per tile and opaque whole-tile covers: the 100 000 path scene)
        # (k_raster_fill<LONG, WAVES>: one build at seven wavefronts per SIMD since round 6)
        "raster_tiles": "crh::k_raster_edges<4, 1, true, false>" if msaa4_strokes else "crh::k_raster_fill<true, 7>",
        # a pass whose average item is beyond a batch of k_bin_flat (the dashed strokes) is binned item by item
        "raster_rows": "crh::k_raster_rows<true>" if workload == "s100k" else "crh::k_raster_rows<false>",  # the row-span kernel, where the library's trial picked it
        "raster_bin": "crh::k_bin_edges<4, false>" if msaa4_strokes else ("crh::k_bin_flat<1, 64u>" if workload == "s100k" else "crh::k_bin_flat<1, 128u>"),
        "raster_scatter": "crh::k_scatter",
        "tess_fused": "crh::k_tess_runs<true, 256>" if msaa4_strokes else ("crh::k_tess_runs<false, 128>" if workload == "glyphs" else "crh::k_tess_runs<false, 256>"),
        "tess_emit": "crh::k_emit",
        "tess_count": "crh::k_count",
        "tess_hull": "crh::k_hull_small",
    }


def kernel_source_hash():
    """Content hash of the kernel sources: PMC summaries under profiles/ carry the hash they were measured at, and are reported only while
    it still matches (a committed counter file silently goes stale with the next kernel change otherwise)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "contrast_renderer_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp")):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    with open(os.path.join(ROOT, "contrast_renderer_amd", "build.py"), "rb") as f:  # (the compile flags make the kernels as much as their sources do: round 6's -fno-slp-vectorize)
        h.update(b"build.py\0" + f.read())
    with open(os.path.join(ROOT, "include", "crh_fmath.h"), "rb") as f:
        h.update(b"crh_fmath.h\0" + f.read())
    return h.hexdigest()[:16]


def _newest_profile(kind, workload):
    """profiles/rNN_<kind>.json (the metric's workload) or profiles/rNN_<kind>_<workload>.json, newest round, if measured on these sources"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{kind}.json" if workload == "cubic" else f"r[0-9][0-9]_{kind}_{workload}.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        doc = json.load(f)
    if doc.get("kernel_source_hash") != kernel_source_hash():
        return None, None  # measured on other kernels: not this run's traffic
    return doc, os.path.basename(files[-1])


# ISA class (tools/isa_histogram.py) -> the class of tools/valu_rate.hip that prices it
_RATE_OF_CLASS = {"simple": "v_fma_f32", "packed_f32": "v_pk_fma_f32", "compare": "v_cmp_ge_i32 -> sgpr pair", "cndmask": "v_cndmask_b32 (sgpr mask)", "lane": "v_readlane_b32",
                  "quarter_rate_int": "v_mul_lo_u32", "convert": "v_cvt_i32_f32", "transcendental": "v_rcp_f32", "f64": "v_rcp_f32"}


def _valu_cycles_per_instruction(kernel):
    """Measured issue cost of one wave64 VALU instruction of `kernel`, in cycles of a SIMD: the chip-wide rates of profiles/rNN_valu_rate.json
    (tools/valu_rate.hip: independent inline-asm chains, 8 wavefronts per SIMD, wall clock) weighted with the static instruction mix of the
    kernel's ISA (profiles/rNN_isa_histogram.json, measured on THESE sources). None when either file is missing or stale."""
    import glob
    rates = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_valu_rate.json")))
    hists = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_isa_histogram.json")))
    if not rates or not hists:
        return None
    with open(rates[-1]) as f:
        rate_doc = json.load(f)
    with open(hists[-1]) as f:
        hist_doc = json.load(f)
    if hist_doc.get("kernel_source_hash") != kernel_source_hash():
        return None
    mix = hist_doc["kernels"].get(kernel.replace("crh::", "crh::", 1))
    if not mix or not mix.get("valu_total"):
        return None

    def chip_rate(name):
        c = rate_doc["classes"][name]["chip"]
        return [v for k, v in c.items() if k.startswith("cycles_per_wave_instruction")][0]
    cycles = sum(n * chip_rate(_RATE_OF_CLASS[c]) for c, n in mix["valu"].items())
    return {"cycles": cycles / mix["valu_total"], "mix": mix["valu"], "rates": {c: chip_rate(_RATE_OF_CLASS[c]) for c in mix["valu"]},
            "sources": [os.path.basename(rates[-1]), os.path.basename(hists[-1])]}


def _salu_cycles_per_instruction():
    """Measured issue cost of a scalar ALU instruction per SIMD (mean of the s_* classes of profiles/rNN_valu_rate.json), or None"""
    import glob
    rates = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_valu_rate.json")))
    if not rates:
        return None
    with open(rates[-1]) as f:
        classes = json.load(f)["classes"]
    values = [[v for k, v in c["chip"].items() if k.startswith("cycles_per_wave_instruction")][0] for name, c in classes.items() if name.startswith("s_")]
    return sum(values) / len(values) if values else None


def valu_issue(mark, avg_launch_ms, workload, triangle_pass=False):
    """Secondary roofline of the dominant kernel: VALU issue utilisation = wave-level VALU instructions (SQ_INSTS_VALU of the committed PMC
    summary measured on THESE kernel sources) x the MEASURED cycles per wave64 instruction of the kernel's instruction mix
    (_valu_cycles_per_instruction; round 2 assumed 4 cycles for everything) / (1024 SIMDs x 2.4 GHz x launch time)."""
    doc, source = _newest_profile("sq_counters", workload)
    names = mark_to_kernel(workload, triangle_pass)
    if not doc or mark not in names or avg_launch_ms <= 0:
        return None
    k = doc.get("per_launch", {}).get(names[mark])
    if not k or "SQ_INSTS_VALU" not in k:
        return None
    simds, clock_hz = 256 * 4, 2.4e9
    priced = _valu_cycles_per_instruction(names[mark])
    cycles = priced["cycles"] if priced else 4.0
    scalar_cycles = _salu_cycles_per_instruction()
    out = {"valu_wave_instructions": int(k["SQ_INSTS_VALU"]), "salu_wave_instructions": int(k.get("SQ_INSTS_SALU", 0)),
           "cycles_per_valu_instruction": cycles, "priced_by": priced if priced else "assumed 4 cycles (no current profiles/rNN_valu_rate.json + rNN_isa_histogram.json)",
           "frac_of_valu_issue_peak": k["SQ_INSTS_VALU"] * cycles / (simds * clock_hz * avg_launch_ms * 1e-3), "source": source,
           "note": "256 CUs x 4 SIMDs at 2.4 GHz; measured per class on this part: simple f32 / int 2.5 cycles, v_pk_fma_f32 4.6, v_cmp 4.3, v_cndmask 4.2 "
                   "(profiles/r03_valu_rate.json): the raster kernels' mix of packed fma + compare + select averages 3.3 - 3.7; a scalar instruction issues "
                   "every 4.3 - 4.4 cycles per SIMD (the CU's scalar unit serves its four SIMDs in turn) and overlaps the vector stream only partly"}
    if scalar_cycles and k.get("SQ_INSTS_SALU"):
        out["cycles_per_salu_instruction"] = scalar_cycles
        out["frac_of_salu_issue_peak"] = k["SQ_INSTS_SALU"] * scalar_cycles / (simds * clock_hz * avg_launch_ms * 1e-3)
    return out


def measured_traffic(mark, workload, triangle_pass=False):
    """HBM bytes per launch of the kernel behind `mark` from the committed PMC summary (profiles/rNN_traffic*.json: separate FETCH_SIZE /
    WRITE_SIZE passes with the gfx950 corrections of the microarchitecture guide), or None when that file was measured on other sources."""
    doc, source = _newest_profile("traffic", workload)
    names = mark_to_kernel(workload, triangle_pass)
    if not doc or mark not in names:
        return None, None
    k = doc.get("kernels", {}).get(names[mark])
    return (k["hbm_bytes_per_launch"], source) if k else (None, None)


def usable_cores():
    """CPUs this process can actually run on: the affinity mask, capped by the cgroup's CPU quota (cpu.max / cfs_quota_us)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                quota, period = int(f.read()), int(g.read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


class _DeviceArray:
    """Wraps a raw HIP pointer for torch.as_tensor (zero copy)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _rocminfo_gpu():
    """Compute units and clock of the first GPU agent as `rocminfo` prints them (torch's device properties leave the clocks at 0 on this stack)."""
    import re
    import subprocess
    try:
        text = subprocess.run(["/opt/rocm/bin/rocminfo"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}
    for block in text.split("*******")[1:]:
        if "gfx" in block and re.search(r"Device Type:\s+GPU", block):
            grab = lambda pat: (re.search(pat, block) or [None, None])[1]
            return {"name": grab(r"Name:\s+(gfx\w+)"), "marketing_name": (grab(r"Marketing Name:[ \t]*([^\n]*)") or "").strip(), "compute_units": int(grab(r"Compute Unit:\s+(\d+)") or 0),
                    "max_clock_mhz": int(grab(r"Max Clock Freq\. \(MHz\):\s+(\d+)") or 0), "wavefront_size": int(grab(r"Wavefront Size:\s+(\d+)") or 0)}
    return {"error": "no GPU agent in rocminfo's output"}


def _frame_crc(workload, image):
    """-> (crc32 of the frame, the oracle's committed crc32 for this workload's default scene or None)"""
    import zlib
    expected = None
    crc_file = os.path.join(ROOT, "tests", "golden", "bench_frame_crc.json")
    if os.path.exists(crc_file):
        with open(crc_file) as fh:
            expected = json.load(fh).get(workload)
    return zlib.crc32(image.tobytes()), (expected["crc32"] if expected else None)


def side_workload(name, device, steps, warmup):
    """The step of one of the other BASELINE configs at N = 1, through the same loop as `value` (tessellate + clear + render of resident inputs, 20 set-up steps
    + warm-up in front, barrier-free: one GPU) — a side block of the default line, never `value`: glyphs = configs[2], dashed = configs[4], s100k = configs[3] whole."""
    from contrast_renderer_amd import scenes
    from contrast_renderer_amd.renderer import Configuration, Frame, Renderer, Scene
    t_gen = time.perf_counter()
    if name == "glyphs":
        size, sc = (2048, 2048), scenes.scene_glyphs(50000, (2048, 2048))
    elif name == "dashed":
        size, sc = (4096, 4096), scenes.scene_dashed_strokes(2000, (4096, 4096))
    else:
        size, sc = (8192, 8192), scenes.scene_cubic_fill(100000, (8192, 8192), config_index=2)
    gen_s = time.perf_counter() - t_gen
    renderer = Renderer(Configuration(msaa_sample_count=sc["msaa"], clip_nesting_counter_bits=4, winding_counter_bits=4), device=device)
    scene = Scene(renderer, sc["batch"], tessellate=True)
    scene.check()
    scene.set_instances(sc["transforms"], sc["colors"])
    frame = Frame(renderer, *size)

    def run(n):
        for _ in range(n):
            scene.tessellate()
            frame.clear()
            scene.render(frame)
    run(20 + warmup)
    renderer.synchronize()
    t0 = time.perf_counter()
    run(steps)
    renderer.synchronize()
    step_s = (time.perf_counter() - t0) / steps
    scene.check()
    got, expected = _frame_crc(name, frame.download())
    n = int(sc["batch"].n_shapes)
    return {"workload": name, "value": n / step_s, "unit": "paths/s", "ms_per_step": step_s * 1e3, "paths": n, "size": list(size), "msaa": int(sc["msaa"]), "steps": steps,
            "mpixel_per_s": size[0] * size[1] / step_s / 1e6, "frame_crc32": got, "expected_crc32": expected, "frame_equals_oracle": (got == expected) if expected is not None else None,
            "scene_generation_s": gen_s}


def side_reupload(renderer, batch, transforms, colors, size, steps, warmup):
    """New paths every step (never `value`): crh_scene_upload into an existing Scene — what Shape::from_paths does, paths in, buffers out, nothing carried
    between calls (renderer.rs:177-249) — then tessellate + clear + render; two Scenes and two targets in turn, as an application double-buffers both.
    Host-inclusive: the caller's arrays are host memory. Also: the host time of the upload call by itself."""
    from contrast_renderer_amd.renderer import Frame, Scene
    scenes_ = [Scene(renderer, batch, tessellate=True), Scene(renderer, batch, tessellate=True)]
    frames = [Frame(renderer, *size), Frame(renderer, *size)]
    host = []

    def run(n, first):
        for i in range(first, first + n):
            k = i % 2
            t0 = time.perf_counter()
            scenes_[k] = Scene(renderer, batch, tessellate=False, existing=scenes_[k])
            host.append(time.perf_counter() - t0)
            scenes_[k].set_instances(transforms, colors)
            scenes_[k].tessellate()
            frames[k].clear()
            scenes_[k].render(frames[k])
    run(40 + warmup + (warmup % 2), 0)  # (an even number: target k stays with Scene k)
    renderer.synchronize()
    del host[:]
    t0 = time.perf_counter()
    run(steps, 0)
    renderer.synchronize()
    step_s = (time.perf_counter() - t0) / steps
    for sc in scenes_:
        sc.check()
    got, expected = _frame_crc("cubic", frames[(steps - 1) % 2].download())
    n = int(batch.n_shapes)
    return {"value": n / step_s, "unit": "paths/s", "ms_per_step": step_s * 1e3, "steps": steps, "upload_call_host_ms": sorted(host)[len(host) // 2] * 1e3,
            "upload_call_host_ms_max": max(host) * 1e3, "input_bytes": int(batch.input_bytes()),
            "frame_crc32": got, "expected_crc32": expected,
            "note": "every step: crh_scene_upload of the same host arrays into an existing Scene (finiteness / -0 checks, element stream, H2D), crh_scene_set_instances, "
                    "crh_scene_tessellate, clear + render — host calls and PCIe inside the clock; two Scenes and two targets in turn"}


def run_loopback(args, size, scaling, np):
    """ONE GPU plays all N ranks: shard k of the scene -> its own Scene and full-size layer, the N layers -> crh_comm_local_exchange.
    A step = tessellate + render of every shard + the exchange; what a rank of an N-GPU node does per step is 1/N of the former and its
    part of the latter, so the line also reports the exchange alone: GPU time per phase (HIP events on each rank's stream), bytes each
    rank puts on the wire against dense slabs, and the transfer time those bytes cost at one xGMI link per peer."""
    from contrast_renderer_amd import scenes
    from contrast_renderer_amd.renderer import FORMAT_RGBA8, FORMAT_RGBA16F, Comm, Configuration, Frame, Renderer, Scene, shard_range
    n = args.loopback
    fmt = FORMAT_RGBA16F if args.layers == "rgba16f" else FORMAT_RGBA8
    renderer = Renderer(Configuration(msaa_sample_count=1, clip_nesting_counter_bits=4, winding_counter_bits=4), device=0)
    shards = []
    tile_split = args.split == "tile"
    if tile_split:  # every rank holds the whole scene and draws its slab of tile rows
        sc = scenes.scene_cubic_fill(args.paths, size, config_index=2)
        scaling = "strong"
    elif scaling == "strong":
        sc = scenes.scene_cubic_fill(args.paths, size, config_index=2)
        for k in range(n):
            lo, hi = shard_range(args.paths, k, n)
            shards.append((sc["batch"].slice_shapes(lo, hi), sc["transforms"][lo:hi], sc["colors"][lo:hi]))
    else:
        for k in range(n):
            one = scenes.scene_cubic_fill(args.paths, size, config_index=2, first_path=k * args.paths)
            shards.append((one["batch"], one["transforms"], one["colors"]))
    scene_objs, layers = [], []
    if tile_split:
        from contrast_renderer_amd.renderer import slab_rows
        scene = Scene(renderer, sc["batch"], tessellate=True)
        scene.check()
        scene.set_instances(sc["transforms"], sc["colors"])
        for k in range(n):  # (ONE Scene object stands for the n identical copies the ranks of a node would hold: each `draw` of it is one rank's)
            scene_objs.append(scene)
            layers.append(Frame(renderer, *size, fmt))
            layers[-1].set_tile_rows(*slab_rows(size[1], k, n))
    for batch, t, c in shards:
        scene = Scene(renderer, batch, tessellate=True)
        scene.check()
        scene.set_instances(t, c)
        scene_objs.append(scene)
        layers.append(Frame(renderer, *size, fmt))
    comms = [Comm(renderer, 0, n)]
    comms += [Comm(renderer, k, n, rank0=comms[0]) for k in range(1, n)]
    result = Frame(renderer, *size)

    def draw():
        for scene, layer in zip(scene_objs, layers):
            scene.tessellate()
            layer.clear()
            scene.render(layer)

    exchange = (lambda: comms[0].local_gather_slabs(layers, result)) if tile_split else (lambda: comms[0].local_exchange(layers, result))

    def step():
        draw()
        exchange()

    for _ in range(20 + args.warmup):  # (the library's pass trial, as in the default run)
        step()
    renderer.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    renderer.synchronize()
    for c in comms:
        c.last_timing()  # waits for the communicator's stream: the tail of the last exchange is inside the clock
    elapsed = time.perf_counter() - t0
    # the exchange alone: layers final, nothing else on the GPU
    wall, phases = [], []
    for _ in range(5):
        renderer.synchronize()
        t1 = time.perf_counter()
        exchange()
        per_rank = [c.last_timing() for c in comms]
        wall.append(time.perf_counter() - t1)
        phases.append(per_rank)
    last = phases[-1]
    serial = os.environ.get("CRH_LOOPBACK_SERIAL") is not None
    phase_ms = {name: {"max_over_ranks": max(p[name] for p in last), "sum_over_ranks": sum(p[name] for p in last)} for name in Comm.PHASES}
    # the drawing alone (all N shards, no exchange)
    renderer.synchronize()
    t2 = time.perf_counter()
    for _ in range(args.steps):
        draw()
    renderer.synchronize()
    draw_s = (time.perf_counter() - t2) / max(1, args.steps)
    traffic = [c.last_traffic() for c in comms]
    peers = [c.last_peer_bytes() for c in comms]
    link_gbs = 153.0  # one xGMI link between every pair of GPUs of the node, per direction (MI355X_MICROARCH.md)
    tile_bytes = 2048 if fmt == FORMAT_RGBA16F else 1024
    n_tiles = ((size[0] + 15) // 16) * ((size[1] + 15) // 16)
    gather_bytes = [t[0] - sum(p) for t, p in zip(traffic, peers)]  # what a rank sends to rank 0 in the gather
    total_paths = args.paths if scaling == "strong" else args.paths * n
    step_s = elapsed / max(1, args.steps)
    image = result.download()
    equals_single = None
    if tile_split:  # the gathered frame against ONE render of the whole scene into a whole frame: equal, bit for bit
        single = Frame(renderer, *size)
        single.clear()
        scene_objs[0].render(single)
        equals_single = bool(np.array_equal(image, single.download()))
    out = {
        "metric": "paths/sec, 10k mixed-Bezier paths @ 4096^2 (tessellate + tile raster)",
        "value": total_paths / step_s,
        "unit": "paths/s",
        "mpixel_per_s": size[0] * size[1] / step_s / 1e6,
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": step_s * 1e3,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"LOOPBACK x{n} on ONE GPU — {'BASELINE configs[3]' if args.workload == 's100k' else 'BASELINE configs[1]'}: {total_paths} filled closed cubic paths, "
                         f"{size[0]}x{size[1]}, msaa 1; " + (f"TILE SPLIT: the whole scene is tessellated, binned and drawn {n} times, each time into the slab of tile rows of one of {n} {args.layers} layers, then "
                                                            "crh_comm_local_gather_slabs (the slabs straight into the result frame); for comparison: "
                                                            if tile_split else f"the {n} contiguous shards ({scaling}) are rendered one after the other into {n} {args.layers} layers, then ") +
                         "crh_comm_local_exchange (occupancy bitmaps, sparse slab all-to-all, ordered over-composite, gather: csrc/comm.hip with D2D copies in place of RCCL)"),
            "paths_total": int(total_paths),
            "parallelism": f"one GPU plays {n} ranks; `value` = {total_paths} paths / (all {n} shards drawn + one exchange) — NOT an {n}-GPU number",
            "covered_fraction": float((image[..., 3] > 0).mean()),
        },
        "loopback": {
            "ranks": n,
            "split": args.split,
            "gathered_equals_single_gpu_frame": equals_single,
            "draw_all_shards_ms": draw_s * 1e3,
            "draw_per_rank_ms": draw_s * 1e3 / n,
            "exchange_wall_ms": {"median": sorted(wall)[len(wall) // 2] * 1e3, "min": min(wall) * 1e3,
                                 "note": f"host clock around crh_comm_local_exchange + its completion, all {n} ranks' kernels and copies on one GPU"},
            "exchange_phase_ms": phase_ms,
            "exchange_phase_mode": ("CRH_LOOPBACK_SERIAL=1: every rank's part of a phase ran with the GPU to itself — max_over_ranks is what ONE rank's kernels and copies cost on a GPU of its own"
                                    if serial else f"all {n} ranks' kernels and copies share the one GPU: the per-rank times are stretched by the other ranks' work"),
            "bytes_sent_per_rank": [t[0] for t in traffic],
            "bytes_dense_per_rank": [t[1] for t in traffic],
            "sent_over_dense": sum(t[0] for t in traffic) / max(1, sum(t[1] for t in traffic)),
            "tile_bytes": tile_bytes,
            "n_tiles": n_tiles,
            "xgmi_estimate": {
                "link_GBps": link_gbs,
                "alltoall_ms": max(max(p) for p in peers) / (link_gbs * 1e9) * 1e3,
                "gather_ms": max(gather_bytes) / (link_gbs * 1e9) * 1e3,
                "note": "largest single rank->peer transfer of the all-to-all / rank->0 transfer of the gather at one link's bandwidth: every pair of GPUs "
                        "has its own link, all transfers of a phase run at once — an estimate from measured bytes, NOT a measurement (no multi-GPU box here)",
            },
            "per_rank_estimate_ms": {
                "draw": draw_s * 1e3 / n,
                "exchange_gpu_phases_max": sum(v["max_over_ranks"] for v in phase_ms.values()),
                "note": f"what ONE rank of an {n}-GPU node spends per step: 1/{n} of the drawing + its own exchange phases (kernels measured here; transfers per xgmi_estimate); "
                        "the exchange of step i overlaps the drawing of step i + 1 (bench.py --gpus N)",
            },
        },
        "roofline": None,
    }
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--paths", type=int, default=10000, help="shapes of the scene: per GPU (weak) or in total (strong); configs[1] = 10000")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--scaling", default=None, choices=("weak", "strong"), help="default: strong at N > 1 (the metric's scene split N ways) with the weak figure in a side block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-animated", action="store_true", help="N = 1: skip the `animated` side block (new instance transforms every step)")
    ap.add_argument("--repeats", type=int, default=5, help="blocks of --steps steps timed again behind the timed region (never `value`): their min / median / max ms per step go to `spread`")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"), help="torch.distributed backend for the barrier / id broadcast (nccl = RCCL)")
    ap.add_argument("--exchange", default="cabi", choices=("cabi", "torch"), help="cabi: crh_frame_exchange over RCCL (the product path); torch: the "
                    "torch.distributed statement of the same exchange (contrast_renderer_amd/distributed.py), dense slabs — validation only")
    ap.add_argument("--same-device", action="store_true", help="all ranks use cuda:0 (validation only; implies --exchange torch)")
    ap.add_argument("--check", action="store_true", help="N > 1: rank 0 also renders every shard itself and compares the composite of those layers with the gathered image "
                    "(N = 1 always checks: CRC-32 of the downloaded frame against tests/golden/bench_frame_crc.json, the oracle's frame of the same scene)")
    ap.add_argument("--no-check", action="store_true", help="N = 1: skip the frame CRC")
    ap.add_argument("--no-side-workloads", action="store_true", help="N = 1, default invocation: skip the `other_workloads` (glyphs, dashed, s100k) and `reupload` side blocks")
    ap.add_argument("--side-budget-s", type=float, default=150.0, help="the side blocks stop starting new workloads once the run has taken this long (the line says which were left out)")
    ap.add_argument("--reupload", action="store_true", help="every step uploads the paths again into the existing Scene before it tessellates and renders — the "
                    "reference's animated-path use (new geometry every frame, Shape::from_paths with existing_shape): host marshalling + PCIe are inside the "
                    "step, so this is a host-inclusive figure, reported as such and never as the metric's `value`")
    ap.add_argument("--loopback", type=int, default=0, help="N > 1 (with --gpus 1): ONE GPU plays all N ranks of the sharded path — the N shards are rendered one after "
                    "the other into N layers and exchanged through crh_comm_local_exchange (csrc/comm.hip with device-to-device copies in place of RCCL): "
                    "the whole of BASELINE configs[3] on one box, with the exchange's per-phase GPU time and bytes on the wire")
    ap.add_argument("--layers", default="rgba8", choices=("rgba8", "rgba16f"), help="--loopback: storage format of the per-rank layers")
    ap.add_argument("--split", default="path", choices=("path", "tile"),
                    help="N > 1 / --loopback: path = contiguous path-index shards, every rank draws a full-size layer, ordered composite (north_star; RGBA8 layers: <= 2/255 "
                         "against one GPU); tile = SURVEY.md §8(e)'s other split: every rank holds ALL paths and draws its slab of tile rows "
                         "(crh_frame_set_tile_rows), nothing is composited, the gathered frame is bit-equal to one GPU's. The default N > 1 run times path as "
                         "`value` and tile in the `tile_split` side block")
    ap.add_argument("--workload", default="cubic", choices=("cubic", "glyphs", "dashed", "s100k"),
                    help="cubic = BASELINE configs[1] (the metric's configuration); glyphs = configs[2]; dashed = configs[4]; s100k = configs[3] (100k paths @ 8192^2, split over the ranks)")
    args = ap.parse_args()
    t_process = time.perf_counter()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare: this process becomes the launcher of its own N ranks (one per GPU, rendezvous on the loopback address)
        import socket
        with socket.socket() as probe:
            probe.bind(("127.0.0.1", 0))
            port = probe.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    default_scaling = args.scaling is None
    if default_scaling:
        args.scaling = "strong" if args.gpus > 1 else "single"  # (N = 1: nothing is split; the scene generator treats it as weak, which is the same thing)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    if args.same_device:
        args.exchange = "torch"  # RCCL refuses two ranks on one device ...
        args.backend = "gloo"    # ... for the process group as well ("Duplicate GPU detected")
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
    from contrast_renderer_amd.renderer import Comm, Configuration, Frame, Renderer, Scene, comm_unique_id, shard_range

    size = (args.size, args.size)
    scaling = args.scaling
    if args.workload == "s100k":
        args.paths, size, scaling = 100000, (8192, 8192), "strong"
    if world == 1 and args.loopback <= 1:
        scaling = "single"  # nothing is split
    if args.loopback > 1:
        if world != 1 or args.workload not in ("cubic", "s100k"):
            raise SystemExit("--loopback N runs in one process on one GPU (--gpus 1), workloads cubic / s100k")
        return run_loopback(args, size, scaling, np)
    if args.workload in ("cubic", "s100k"):
        label = "BASELINE configs[1]" if args.workload == "cubic" else "BASELINE configs[3]"
        if args.split == "tile" and world > 1:  # THE scene on every rank, split by tile rows (the frames' slabs, below)
            sc = scenes.scene_cubic_fill(args.paths, size, config_index=2)
            shard = (0, args.paths)
            scaling = "strong"
        elif scaling == "weak" or world == 1:
            sc = scenes.scene_cubic_fill(args.paths, size, config_index=2, first_path=rank * args.paths)
            shard = (0, args.paths)
        else:  # THE scene, split by index
            sc = scenes.scene_cubic_fill(args.paths, size, config_index=2)
            shard = shard_range(args.paths, rank, world)
        workload = (f"{label}: {args.paths} filled closed paths x 8 cubic segments (alternating integral / rational), {size[0]}x{size[1]}, msaa 1, "
                    "winding_counter_bits 4")
    elif args.workload == "glyphs":
        size = (2048, 2048)
        sc = scenes.scene_glyphs(50000, size)
        args.paths = sc["batch"].n_shapes
        shard = (0, args.paths) if (scaling == "weak" or world == 1) else shard_range(args.paths, rank, world)
        workload = f"BASELINE configs[2]: 50000 glyph instances ({sc['n_paths']} line/quadratic paths) via text::paths_of_text, 2048x2048, msaa 1"
    else:
        size = (4096, 4096)
        sc = scenes.scene_dashed_strokes(2000, size)
        args.paths = 2000
        shard = (0, args.paths) if (scaling == "weak" or world == 1) else shard_range(args.paths, rank, world)
        workload = "BASELINE configs[4]: 2000 dashed rational-cubic strokes (UniformTangentAngle 0.1, miter/round joins), 4096x4096, msaa 4"
    batch = sc["batch"] if shard == (0, sc["batch"].n_shapes) else sc["batch"].slice_shapes(*shard)
    transforms, colors = sc["transforms"][shard[0]:shard[1]], sc["colors"][shard[0]:shard[1]]
    renderer = Renderer(Configuration(msaa_sample_count=sc["msaa"], clip_nesting_counter_bits=4, winding_counter_bits=4), device=local_rank)
    t_up = time.perf_counter()
    scene = Scene(renderer, batch, tessellate=True)  # host -> HBM + first tessellation (sizes the output buffers): outside the timed region
    renderer.synchronize()
    upload_s = time.perf_counter() - t_up  # validation + element stream + H2D + first tessellation, once per scene
    scene.check()
    scene.set_instances(transforms, colors)
    frame = Frame(renderer, *size)
    lib = renderer.lib
    import ctypes as C

    # N > 1: two layers, so that the exchange of step i runs while step i + 1 is being tessellated and rasterized into the other one
    # (--reupload: two targets as well — an application that re-uploads its paths every frame double-buffers Scene AND target: Scene k draws into target k,
    # so the list places and batch runs a target keeps for "its" Scene survive the upload of same-structure paths, csrc/api.hip crh_scene::lineage)
    frames = [frame] + ([Frame(renderer, *size)] if (world > 1 or args.reupload) else [])
    tile_split = args.split == "tile" and world > 1
    gather_mode = tile_split  # (finish() reads it: the `tile_split` side block of a path-sharded run switches it on for its own steps)
    if tile_split:
        exchange_note_tile = "crh_frame_gather_slabs (C ABI): every rank's slab of rows straight into rank 0's frame, one grouped ncclSend / ncclRecv per rank"
        if args.workload not in ("cubic", "s100k"):
            raise SystemExit("--split tile: workloads cubic / s100k")
        from contrast_renderer_amd.renderer import slab_rows
        for f in frames:
            f.set_tile_rows(*slab_rows(size[1], rank, world))
    comm, result, exchange_note = None, None, None
    layer_views, slab = [], None
    if world > 1 and args.exchange == "cabi":
        try:
            ident = [comm_unique_id(lib) if rank == 0 else None]
            dist.broadcast_object_list(ident, src=0)  # the 128-byte RCCL id: the only payload torch.distributed carries
            comm = Comm(renderer, rank, world, unique_id=ident[0])
            result = Frame(renderer, *size) if rank == 0 else None
            exchange_note = "crh_frame_exchange (C ABI): occupancy bitmaps all-gathered, non-empty tiles of row slabs all-to-all (grouped ncclSend/ncclRecv), ordered over-composite, gather to rank 0"
        except Exception as e:  # reported, never silent: the line then says which path produced the number
            comm = None
            exchange_note = f"FALLBACK to torch.distributed (crh_comm_create failed: {e})"
        # every rank takes the same branch from here on: a communicator that only SOME ranks hold would leave those in the first exchange's
        # collectives while the others are already in the torch statement of it
        created = torch.tensor([0 if comm is None else 1], dtype=torch.int32, device="cpu" if args.backend == "gloo" else f"cuda:{local_rank}")
        dist.all_reduce(created, op=dist.ReduceOp.MIN)
        if int(created.item()) == 0 and comm is not None:
            comm, result = None, None
            exchange_note = "FALLBACK to torch.distributed (crh_comm_create failed on another rank)"
        if comm is not None and tile_split:
            exchange_note = exchange_note_tile
    if world > 1 and os.environ.get("CRH_BENCH_FAIL_FIRST_EXCHANGE") is not None:  # (tests: every rank holds a communicator whose exchange fails)
        class _Failing:
            def exchange(self, *a):
                raise RuntimeError("injected failure of the first exchange")
        comm = _Failing()

    def torch_path():
        nonlocal layer_views, slab
        layer_views = [torch.as_tensor(_DeviceArray(f.device_pointer(), (size[1], size[0], 4)), device=f"cuda:{local_rank}") for f in frames]
        r0, r1 = D.slab_rows(size[1], world)[rank]
        slab = torch.empty((r1 - r0, size[0], 4), dtype=torch.uint8, device=f"cuda:{local_rank}")

    if world > 1 and comm is None:
        exchange_note = exchange_note or "torch.distributed statement of the exchange (dense slabs, validation only)"
        torch_path()

    reupload_other = [None]
    reupload_steps = [0]

    def launch(i):
        """Enqueues step i's tessellation + render (asynchronous on the renderer's streams)."""
        nonlocal scene
        f = frames[i % len(frames)]
        if args.reupload:
            # (target k goes with Scene k across run() calls — an odd number of warm-up steps must not swap the pairs: a target keeps its list
            # places and batch runs for "its" Scene)
            f = frames[reupload_steps[0] % len(frames)]
            reupload_steps[0] += 1
            # crh_scene_upload into an existing Scene: validation, element stream, H2D. TWO Scenes in turn — geometry double-buffered as an application
            # double-buffers its vertex buffers: an upload into the Scene of the frame still in flight has to wait for that frame (it may have
            # to be drawn again from the old paths), an upload into the other one does not
            if reupload_other[0] is None:
                reupload_other[0] = Scene(renderer, batch, tessellate=True)
            scene, reupload_other[0] = reupload_other[0], scene
            scene = Scene(renderer, batch, tessellate=False, existing=scene)
            scene.set_instances(transforms, colors)
        scene.tessellate()
        f.clear()
        scene.render(f)

    def finish(i):
        """The exchange step of the path for step i's layer; the renderer may already be working on step i + 1."""
        f = frames[i % len(frames)]
        if comm is not None:
            if gather_mode:
                comm.gather_slabs(f, result)  # the tile split: the slabs travel straight from the layers' rows into the result frame
            else:
                comm.exchange(f, result)  # waits for step i's raster kernel only, then runs on the communicator's own stream
            return result
        f.synchronize()
        received, _ = D.exchange_layers(layer_views[i % len(frames)], rank, world)
        if received.is_cuda:
            torch.cuda.current_stream().synchronize()
        ptrs = (C.c_void_p * world)(*[received[k].data_ptr() for k in range(world)])
        rc = lib.crh_composite_over(renderer.handle, ptrs, world, received[0].numel() // 4, C.c_void_p(slab.data_ptr()))
        assert rc == 0, rc
        return D.gather_slabs(slab, rank, world, size[1])

    def run(n):
        """n steps; with N > 1 the exchange of step i overlaps the rendering of step i + 1. Returns the last gathered image (rank 0)."""
        out = None
        for i in range(n):
            launch(i)
            if world > 1 and i > 0:
                out = finish(i - 1)
        if world > 1 and n > 0:
            out = finish(n - 1)
        return out

    def sync():
        renderer.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # Scene set-up, before the W warm-up steps: the library draws a Scene's first frames with both raster formulations (boundary edges /
    # strip triangles: same pixels), times a group of three frames of each on the GPU and keeps the faster one from the thirteenth frame on
    watchdog = None
    if world > 1:  # the first exchanges of a multi-rank run: a rank that never answers must end the job with a message, not hang it
        import threading

        def _stuck():
            sys.stderr.write(f"[bench] rank {rank}: no progress in the first exchange steps after 180 s ({exchange_note}); aborting\n")
            sys.stderr.flush()
            os._exit(3)
        watchdog = threading.Timer(180.0, _stuck)
        watchdog.daemon = True
        watchdog.start()
    if world > 1 and comm is not None:
        # The first exchange through the C ABI, by itself: if it FAILS on any rank (an RCCL error — the point-to-point path has never run on
        # real links), every rank hears of it and the run goes on with the torch.distributed statement of the exchange; the line says so.
        # (A rank that HANGS in it is the watchdog's business.)
        failure = ""
        try:
            launch(0)
            finish(0)
            renderer.synchronize()
            comm.last_timing()
        except Exception as e:
            failure = f"{type(e).__name__}: {e}"
        flag = torch.tensor([0 if failure else 1], dtype=torch.int32, device="cpu" if args.backend == "gloo" else f"cuda:{local_rank}")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            sys.stderr.write(f"[bench] rank {rank}: the first crh_frame_exchange failed on some rank ({failure or 'not this one'}): continuing with torch.distributed\n")
            comm, result = None, None
            exchange_note = "FALLBACK to torch.distributed (the first crh_frame_exchange failed" + (f": {failure}" if failure else " on another rank") + ")"
            torch_path()
    run(40 if args.reupload else 20)  # (--reupload: two Scenes and two targets in turn — twenty set-up steps for each pair)
    sync()
    if watchdog is not None:
        watchdog.cancel()
    # The timing marks of the timed region (below) take their HIP events from a pool the library grows on first use: one untimed pass of the same
    # length with the marks on creates them here, outside the clock (until round 5 the first timed block created its forty events itself: it was
    # 2 - 3 % slower than the blocks behind it — `spread` showed it).
    if os.environ.get("CRH_BENCH_NO_MARKS") is None:
        renderer.enable_timing(2)
        run(args.steps)
        sync()
        renderer.kernel_times()
        renderer.enable_timing(0)
    run(args.warmup)
    sync()
    scene.check()
    # HIP events around the raster lane's kernels — the dominant kernel's launches — inside the timed region, drained once after it. (Events
    # around EVERY kernel of a step, a dozen per step on three streams, cost the loop 4 %: 0.437 against 0.415 ms per step. The other
    # kernels' times in the run come from a second loop of the same length right after, outside the clock.)
    renderer.enable_timing(0 if os.environ.get("CRH_BENCH_NO_MARKS") is not None else 2)
    sync()
    if os.environ.get("CRH_PASS_VERBOSE"):
        sys.stderr.write("[bench] the timed block begins\n")
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if os.environ.get("CRH_PASS_VERBOSE"):
        sys.stderr.write("[bench] the timed block ends\n")
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.backend == "gloo" else f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if os.environ.get("CRH_BENCH_NO_MARKS") is not None:  # (measurement of the measurement: what do the timing marks cost the timed region?)
        raise SystemExit(f"[bench] without timing marks: {elapsed / args.steps * 1e3:.4f} ms/step")
    kernel_times = renderer.kernel_times()  # the raster lane of the timed steps
    # the spread of the timed region: the same block of K steps again, R times, each bracketed like the timed one (never `value`)
    blocks = []
    for _ in range(max(0, args.repeats)):
        renderer.kernel_times()  # (drained: the marks of these blocks are not reported)
        sync()
        tb = time.perf_counter()
        run(args.steps)
        sync()
        blocks.append(time.perf_counter() - tb)
    if dist is not None and blocks:
        tb = torch.tensor(blocks, dtype=torch.float64, device="cpu" if args.backend == "gloo" else f"cuda:{local_rank}")
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        blocks = [float(x) for x in tb.tolist()]
    renderer.kernel_times()
    renderer.enable_timing(1)
    sync()
    run(args.steps)
    sync()
    timed_lane = {name for name, _, _ in kernel_times}
    kernel_times += [k for k in renderer.kernel_times() if k[0] not in timed_lane]  # the other lanes, from the same loop run once more
    renderer.enable_timing(False)
    scene.check()
    # latency of ONE step, nothing overlapped (the timed loop above keeps up to three steps in flight)
    latency = []
    renderer.enable_timing(True)
    for _ in range(min(5, max(1, args.steps))):
        sync()
        t1 = time.perf_counter()
        run(1)
        renderer.synchronize()
        if comm is not None:
            comm.last_timing()  # the tail of the exchange runs on the communicator's stream: inside the clock
        latency.append(time.perf_counter() - t1)
    latency_ms = sorted(latency)[len(latency) // 2] * 1e3
    alone = {}  # the kernels of those steps, each with the GPU to itself
    for name, ms, _ in renderer.kernel_times():
        alone.setdefault(name, []).append(ms)
    alone = {k: sum(v) / len(v) for k, v in alone.items()}
    renderer.enable_timing(False)
    image = frame.download()
    covered = float((image[..., 3] > 0).mean())
    traffic_sent = comm.last_traffic() if comm is not None else None
    # what the transport and every rank say about the last exchange (outside the clock): RCCL's own rank count and version, this step's
    # GPU time per phase and the bytes each rank put on the wire against dense slabs
    exchange_stats = None
    if comm is not None:
        mine = {"rank": rank, "rccl": comm.info(), "phase_ms": comm.last_timing(), "bytes_sent": traffic_sent[0], "bytes_dense": traffic_sent[1],
                "peer_bytes": comm.last_peer_bytes()}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        exchange_stats = {"rccl": {"nranks": everyone[0]["rccl"]["nranks"], "version": everyone[0]["rccl"]["rccl_version"],
                                   "nranks_seen_by_every_rank": [e["rccl"]["nranks"] for e in everyone]},
                          "phase_ms_max_over_ranks": {k: max(e["phase_ms"][k] for e in everyone) for k in Comm.PHASES},
                          "phase_ms_by_rank": [e["phase_ms"] for e in everyone],
                          "bytes_sent_by_rank": [e["bytes_sent"] for e in everyone], "bytes_dense_by_rank": [e["bytes_dense"] for e in everyone],
                          "sent_over_dense": sum(e["bytes_sent"] for e in everyone) / max(1, sum(e["bytes_dense"] for e in everyone)),
                          "peer_bytes_by_rank": [e["peer_bytes"] for e in everyone]}
    # The weak figure beside the strong default (N > 1): every rank now draws a 10 000-path scene of its own — the generator's streams
    # rank * paths ... — through the same loop, exchange included.
    weak_side = None
    if world > 1 and default_scaling and scaling == "strong" and args.workload == "cubic" and not args.reupload:
        strong_scene = scene
        one = scenes.scene_cubic_fill(args.paths, size, config_index=2, first_path=rank * args.paths)
        scene = Scene(renderer, one["batch"], tessellate=True)
        scene.check()
        scene.set_instances(one["transforms"], one["colors"])
        run(20 + args.warmup)
        sync()
        tw = time.perf_counter()
        run(args.steps)
        sync()
        weak_elapsed = time.perf_counter() - tw
        tmax = torch.tensor([weak_elapsed], dtype=torch.float64, device="cpu" if args.backend == "gloo" else f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        weak_elapsed = float(tmax.item())
        scene.check()
        weak_side = {"scaling": "weak", "value": args.paths * world / (weak_elapsed / args.steps), "unit": "paths/s", "ms_per_step": weak_elapsed / args.steps * 1e3,
                     "paths_per_gpu": int(args.paths), "paths_total": int(args.paths * world),
                     "note": f"every rank draws its OWN {args.paths} paths (generator streams rank x {args.paths} ...): {world} x the metric's scene per step, same loop and exchange"}
        scene = strong_scene
    # The other split beside the path-sharded default (N > 1): every rank holds the WHOLE scene and draws its slab of tile rows — same loop, same
    # exchange (which then moves nothing in its all-to-all and composites nothing): the first run on real links decides between the two.
    tile_side = None
    if world > 1 and default_scaling and scaling == "strong" and args.split == "path" and args.workload == "cubic" and not args.reupload:
        from contrast_renderer_amd.renderer import slab_rows
        strong_scene = scene
        whole = scenes.scene_cubic_fill(args.paths, size, config_index=2)
        scene = Scene(renderer, whole["batch"], tessellate=True)
        scene.check()
        scene.set_instances(whole["transforms"], whole["colors"])
        renderer.synchronize()
        for f in frames:
            f.set_tile_rows(*slab_rows(size[1], rank, world))
        gather_mode = True
        tile_error = None
        tile_elapsed = float("nan")
        try:  # (a side block: a failure of the slab gather — it fails on every rank together, csrc/comm.hip — is reported in the block, the line's `value` stands)
            run(20 + args.warmup)
            sync()
            tt = time.perf_counter()
            run(args.steps)
            sync()
            tile_elapsed = time.perf_counter() - tt
        except Exception as e:
            tile_error = f"{type(e).__name__}: {e}"
        tmax = torch.tensor([tile_elapsed], dtype=torch.float64, device="cpu" if args.backend == "gloo" else f"cuda:{local_rank}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tile_elapsed = float(tmax.item())
        scene.check()
        sent = comm.last_traffic() if comm is not None else None
        gathered_equal = None
        try:  # the gathered frame once against this rank's own render of the whole scene into a whole frame (the side block had no pixel check)
            gathered = run(1)
            sync()
            if rank == 0:
                got = gathered.download() if comm is not None else gathered.cpu().numpy()
                whole_frame = Frame(renderer, *size)
                whole_frame.clear()
                scene.render(whole_frame)
                gathered_equal = bool(np.array_equal(got, whole_frame.download()))
                del whole_frame
        except Exception as e:
            gathered_equal = f"{type(e).__name__}: {e}"
        tile_side = {"split": "tile", "scaling": "strong", "error": tile_error, "gathered_equals_single_gpu_frame": gathered_equal, "value": args.paths / (tile_elapsed / args.steps), "unit": "paths/s", "ms_per_step": tile_elapsed / args.steps * 1e3,
                     "paths_per_gpu": int(args.paths), "bytes_sent_by_rank0_last_step": sent[0] if sent else None,
                     "note": f"every rank tessellates and bins all {args.paths} paths and draws 1/{world} of the tile rows (crh_frame_set_tile_rows); the exchange gathers the slabs — "
                             "no compositing, the frame is bit-equal to one GPU's (tests/test_comm.py::test_tile_split_gathers_the_single_gpu_frame_bit_for_bit)"}
        renderer.synchronize()
        gather_mode = False
        for f in frames:
            f.set_tile_rows(0, size[1])
        scene = strong_scene
    # A scene that MOVES (never `value`): every step new instance transforms — a zoom about the frame's centre by 1 % per frame, in and out over
    # twenty frames, what the reference's own loop does with its view (examples/showcase/main.rs:154-161, 236-250) — uploaded with
    # crh_scene_set_instances in front of the step; the dashed workload also moves one Shape's dash phase per frame (main.rs:243-250 through
    # crh_scene_set_dynamic_stroke_options). The tile lists change from frame to frame: what the frame-coherent state of the steady figure
    # (lists in place, batches cut by cost, heavy tiles first) is worth when the frames are not identical shows here.
    # `recount` (a side block, never `value`): the one-pass tessellation keeps what every element emits and where every run of Shapes begins from the
    # FIRST tessellation of the uploaded paths (properties of the paths: csrc/tessellate.hip k_tess_count_runs / k_scan_runs) — the timed steps re-analyse
    # and re-emit every vertex, but do not count again. Here they do: the counting pass and the scan over the runs in front of every step's kernel,
    # as for paths that have just been uploaded.
    recount = None
    if world == 1 and not args.reupload and not args.no_animated:
        try:
            os.environ["CRH_TESS_COUNT_EVERY_RUN"] = "1"
            run(args.warmup + 2)
            sync()
            tr = time.perf_counter()
            run(args.steps)
            sync()
            recount_s = (time.perf_counter() - tr) / args.steps
            recount = {"ms_per_step": recount_s * 1e3, "value": args.paths / recount_s, "unit": "paths/s", "steps": args.steps,
                       "note": "the timed loop with k_tess_count_runs + k_scan_runs in front of every step's k_tess_runs (CRH_TESS_COUNT_EVERY_RUN=1): nothing about "
                               "the paths is kept from an earlier tessellation but the streams' capacities"}
        except Exception as e:  # (reported, never fatal: a side block)
            recount = {"error": f"{type(e).__name__}: {e}"}
        finally:
            os.environ.pop("CRH_TESS_COUNT_EVERY_RUN", None)
            sync()
    animated = None
    if world == 1 and not args.reupload and not args.no_animated:
        n_sets = 20
        zoom = [1.01 ** (k if k <= n_sets // 2 else n_sets - k) for k in range(n_sets)]
        moved = []
        for z in zoom:
            t = np.array(transforms, dtype=np.float32, copy=True).reshape(-1, 16)
            t[:, [0, 1, 4, 5, 12, 13]] *= np.float32(z)
            moved.append(t)
        phase_of = None
        if args.workload == "dashed":
            from contrast_renderer_amd.path import DashInterval, DynamicStrokeOptions, Join, Cap
            pattern = [DashInterval(2.0, 3.0, Cap(0), Cap(3)), DashInterval(5.0, 6.0, Cap(3), Cap(0))]  # Shape 0's own pattern (scenes.scene_dashed_strokes), its phase moving
            phase_of = lambda i: DynamicStrokeOptions.Dashed(Join.Miter, pattern, 0.25 * (i % n_sets))

        shown = [frame, Frame(renderer, *size)]  # two targets, as a swap chain has: frame i is CONSUMED (crh_frame_synchronize: the pixels are final,
                                                 # a pass whose optimistic list places were outgrown has been drawn again) before its target is drawn into again

        def run_animated(n):
            for i in range(n):
                target = shown[i % 2]
                target.synchronize()
                scene.tessellate()  # (first: it does not depend on the instances, and started now it runs in the gap behind the raster kernel of the frame before)
                scene.set_instances(moved[i % n_sets], colors)
                if phase_of is not None:
                    scene.set_dynamic_stroke_options(0, 0, phase_of(i))
                target.clear()
                scene.render(target)
        try:
            run_animated(2 * n_sets)  # (untimed: the first pass over the twenty views)
            sync()
            ta = time.perf_counter()
            run_animated(max(args.steps, n_sets))
            sync()
            animated_s = (time.perf_counter() - ta) / max(args.steps, n_sets)
            scene.check()
            animated = {"ms_per_step": animated_s * 1e3, "value": args.paths / animated_s, "unit": "paths/s", "steps": max(args.steps, n_sets),
                        "views": n_sets, "zoom_per_frame": 0.01, "dash_phase_moves": phase_of is not None,
                        "note": "every step: crh_frame_synchronize of the target (two targets in turn: the frame drawn two steps ago is consumed — final pixels — before "
                                "its target is reused), crh_scene_tessellate, crh_scene_set_instances with the next of twenty views (zoom about the centre, 1 % per frame, in and out)"
                                + (" + crh_scene_set_dynamic_stroke_options of Shape 0 with a new dash phase" if phase_of is not None else "")
                                + ", then clear + render: the step of the metric with the instances renewed; host calls and the instance upload inside the clock"}
        except Exception as e:  # (reported, never fatal: a side block)
            animated = {"error": f"{type(e).__name__}: {e}"}
        scene.set_instances(transforms, colors)
        renderer.synchronize()
        del shown
    check = None
    if world == 1 and not args.no_check:  # the pixels this run timed, against the oracle's frame of the same scene (its CRC-32 is committed: the oracle takes minutes at this size)
        import zlib
        crc_file = os.path.join(ROOT, "tests", "golden", "bench_frame_crc.json")
        expected = None
        default_scene = (args.workload == "cubic" and args.paths == 10000 and args.size == 4096) or args.workload in ("glyphs", "dashed", "s100k")
        if default_scene and os.path.exists(crc_file):
            with open(crc_file) as fh:
                expected = json.load(fh).get(args.workload)
        got = zlib.crc32(image.tobytes())
        check = {"frame_crc32": got, "expected_crc32": expected["crc32"] if expected else None,
                 "frame_equals_oracle": (got == expected["crc32"]) if expected else None,
                 "source": "tests/golden/bench_frame_crc.json (CRC-32 of the oracle's frame of this scene; tests/golden/make_golden.py --crc)" if expected
                           else "no committed CRC for this invocation (non-default --paths / --size)"}
        if expected and got != expected["crc32"]:
            sys.stderr.write(f"[bench] FRAME CHECK FAILED: crc32 {got} != {expected['crc32']} (oracle)\n")
    if args.check and world > 1:  # (scene is the strong / chosen one again)
        gathered = run(1)  # one more pass outside the timed region: the gathered image on rank 0
        sync()
        if rank == 0 and tile_split:  # the gathered frame against this rank's own render of the whole scene into a whole frame: equal
            got = gathered.download() if comm is not None else gathered.cpu().numpy()
            whole_frame = Frame(renderer, *size)
            whole_frame.clear()
            scene.render(whole_frame)
            expect = whole_frame.download()
            check = {"gathered_equals_single_gpu_frame": bool(np.array_equal(got, expect)),
                     "max_abs_difference": int(np.abs(got.astype(np.int32) - expect.astype(np.int32)).max())}
        elif rank == 0:
            got = gathered.download() if comm is not None else gathered.cpu().numpy()
            layers = []
            for other in range(world):  # the same shards, rendered one after the other by this rank alone
                if scaling == "weak":
                    one = scenes.scene_cubic_fill(args.paths, size, config_index=2, first_path=other * args.paths)
                    b, t, c = one["batch"], one["transforms"], one["colors"]
                else:
                    lo, hi = shard_range(args.paths, other, world)
                    b, t, c = sc["batch"].slice_shapes(lo, hi), sc["transforms"][lo:hi], sc["colors"][lo:hi]
                shard_scene = Scene(renderer, b, tessellate=True)
                shard_frame = Frame(renderer, *size)
                shard_frame.clear()
                shard_scene.render(shard_frame, t, c)
                layers.append(shard_frame.download())
            expect = D.composite_over_reference(np.stack(layers))
            check = {"gathered_equals_ordered_composite_of_all_shards": bool(np.array_equal(got, expect)),
                     "max_abs_difference": int(np.abs(got.astype(np.int32) - expect.astype(np.int32)).max())}

    # per-kernel averages
    agg = {}
    for name, ms, nbytes in kernel_times:
        a = agg.setdefault(name, [0.0, 0, nbytes])
        a[0] += ms
        a[1] += 1
        a[2] = max(a[2], nbytes)
    kernels = {k: {"avg_ms": v[0] / v[1], "launches": v[1], "algorithmic_bytes": v[2], "alone_ms": alone.get(k)} for k, v in agg.items()}
    # the dominant kernel is the one that needs the GPU longest when it has it to itself: inside the pipelined run a small kernel that shares
    # the GPU with the raster kernel of the frame before is stretched to that kernel's length without doing more work
    longest_alone = max(kernels, key=lambda k: (kernels[k]["alone_ms"] if kernels[k]["alone_ms"] is not None else kernels[k]["avg_ms"]) * kernels[k]["launches"])
    # ... which is why the line says both: `roofline` is the kernel that needs the GPU longest when it has it to itself (its launch time in the run is
    # what `achieved` is computed from), `roofline_longest_in_run` the one whose launches last longest inside the pipelined run (waiting for wave
    # slots beside the others included) when that is another one — the binning kernel, which has no algorithmic bytes at all
    in_run_longest = max(kernels, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"])
    dominant = longest_alone
    dk = kernels[dominant]
    achieved = dk["algorithmic_bytes"] / (dk["avg_ms"] * 1e-3) / 1e9 if dk["avg_ms"] > 0 else 0.0
    la = kernels[in_run_longest]
    # the PMC summaries under profiles/ were measured on the default invocation of each workload
    default_workload = world == 1 and ((args.workload == "cubic" and args.paths == 10000 and args.size == 4096) or args.workload in ("glyphs", "dashed", "s100k"))
    # which formulation the library settled on for this scene (it measures both on the first frames): the marks tell
    launches_of = lambda k: kernels.get(k, {}).get("launches", 0)
    triangle_pass = launches_of("raster_prim_setup") > launches_of("raster_bin")
    traffic, traffic_source = measured_traffic(dominant, args.workload, triangle_pass) if default_workload else (None, None)

    step_s = elapsed / args.steps
    total_paths = args.paths * world if scaling == "weak" else args.paths
    ms_per_step = step_s * 1e3
    # whole-step algorithmic bytes: the tessellation reads the control data and writes the emitted bytes, the raster reads the
    # emitted bytes and writes the frame (the raster mark already carries emitted + 80 B / shape + W * H * 4); binning has none
    # (every mark of the tessellation lane carries what its kernels emit — tess_fused / tess_emit the vertex and index streams, tess_hull the hull vertices —; the
    # control data they read is the batch's input bytes. Until round 5 this line looked up the mark `tess_emit` only, which the one-pass tessellation no longer
    # sets, and silently dropped the tessellation's bytes: VERDICT r05)
    tess_bytes = sum(v.get("algorithmic_bytes", 0) for k, v in kernels.items() if k.startswith("tess_"))
    step_bytes = int(batch.input_bytes()) + tess_bytes + max(kernels.get("raster_tiles", {}).get("algorithmic_bytes", 0), kernels.get("raster_rows", {}).get("algorithmic_bytes", 0))
    out = {
        "metric": "paths/sec, 10k mixed-Bezier paths @ 4096^2 (tessellate + tile raster)" + (" — HOST-INCLUSIVE: new geometry uploaded every step (--reupload)" if args.reupload else ""),
        "value": total_paths / step_s,
        "unit": "paths/s",
        "mpixel_per_s": size[0] * size[1] / step_s / 1e6,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "latency_ms_per_step": latency_ms,
        "spread": None if not blocks else {"blocks": len(blocks), "steps_per_block": args.steps,
                                           "ms_per_step_min": min(blocks) / args.steps * 1e3, "ms_per_step_median": sorted(blocks)[len(blocks) // 2] / args.steps * 1e3,
                                           "ms_per_step_max": max(blocks) / args.steps * 1e3,
                                           "note": "the timed block of K steps run again R times behind the timed region, each between barrier + synchronize (max over ranks); `value` is the first block alone"},
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": workload + "; step = tessellate (count/scan/emit/hull) + bin + tile raster, inputs resident in HBM",
            "paths_per_gpu": int(batch.n_shapes),
            "paths_total": int(total_paths),
            "segments_per_gpu": int(batch.n_segments),
            "parallelism": "single GPU" if world == 1 else ((f"TILE split x{world}: every rank holds all paths and draws its slab of tile rows" if tile_split else f"path-index sharding x{world} ({scaling})")
                                                               + f"; {exchange_note}; the exchange of step i overlaps the rendering of step i + 1"),
            "covered_fraction": covered,
        },
        "setup": "20 untimed steps before the warm-up: the library times its raster formulations (boundary edges per sample / strip triangles / boundary edges as row spans: same pixels) on this scene and keeps the fastest",
        "steady_state": "`value` is the steady state of IDENTICAL frames (an animation that re-tessellates and re-draws the same geometry): the tile lists keep the "
                        "places the verified first passes left them, no read-back; `latency_ms_per_step` is one host-synchronised step, `bench.py --reupload` new geometry every step",
        "pipelining": "ms_per_step: up to three steps in flight on three HIP streams (tessellate / bin / raster); latency_ms_per_step: one step, host synchronised before and after",
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
            "valu_issue": valu_issue(dominant, dk["avg_ms"], args.workload, triangle_pass) if default_workload else None,
            "pass": "strip triangles (raster.hip)" if triangle_pass else ("boundary edges + backdrops as row spans (raster_edges.hip, k_raster_rows)" if "raster_rows" in kernels else "boundary edges + backdrops (raster_edges.hip)"),
            "avg_launch_ms": dk["avg_ms"],
            "avg_launch_ms_alone": dk["alone_ms"],
            "kernel_source_hash": kernel_source_hash(),
            "note": "kernel = the one that needs the GPU longest per launch with the GPU to itself (avg_launch_ms_alone); avg_launch_ms = its launches in the timed run "
                    "(HIP events on its own stream, sharing the GPU with the other lanes of the pipeline); "
                    "achieved = its algorithmic bytes (SURVEY.md §8(d): the raster kernel reads the emitted vertex/index bytes once + 80 B per shape and writes W*H*4 once; "
                    "binning has none) / that launch time; traffic / "
                    "valu_issue = rocprofv3 PMC passes committed under profiles/, reported only while their kernel_source_hash equals this run's. "
                    "The raster kernels are issue- and latency-bound (per-sample edge functions, tiny algorithmic traffic), not HBM bound: see DESIGN.md §4",
        },
        "roofline_longest_in_run": None if in_run_longest == dominant else {
            "kernel": in_run_longest, "bound": "hbm", "achieved": la["algorithmic_bytes"] / (la["avg_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": la["algorithmic_bytes"] / (la["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": la["algorithmic_bytes"],
            "avg_launch_ms": la["avg_ms"], "avg_launch_ms_alone": la["alone_ms"],
            "note": "the kernel whose launches last longest INSIDE the pipelined run (HIP events on its stream: the time it waits for wave slots beside the other "
                    "lanes is in it), when that is not the kernel `roofline` reports (the longest with the GPU to itself); binning has no algorithmic bytes"},
        "roofline_step": {"algorithmic_bytes": step_bytes, "achieved": step_bytes / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": step_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                          "tessellation_bytes": int(batch.input_bytes()) + tess_bytes,
                          "note": "all kernels of a step: control data read + emitted bytes written (tessellation: the batch's input bytes + the tess_* marks), emitted bytes + 80 B / shape read and W*H*4 written (raster)"},
        "kernels": kernels,
        "animated": animated,
        "recount": recount,
        "check": check,
        # the boundary hands over host buffers once per scene (crh_scene_upload); never part of `value`
        "host_inclusive": {"upload_ms": upload_s * 1e3, "paths_per_s_first_frame": batch.n_shapes / (upload_s + step_s),
                           "input_bytes": int(batch.input_bytes())},
    }
    if traffic_sent is not None:
        out["exchange"] = {"bytes_sent_by_rank0_last_step": traffic_sent[0], "dense_slabs_would_be": traffic_sent[1]}
        out["exchange"].update(exchange_stats or {})
    if weak_side is not None:
        out["weak_scaling"] = weak_side
    if tile_side is not None:
        out["tile_split"] = tile_side
    # what the normaliser of `roofline` is held against: the device as HIP reports it
    try:
        props = torch.cuda.get_device_properties(local_rank)
        out["device"] = {"name": props.name, "arch": getattr(props, "gcnArchName", None), "compute_units": int(props.multi_processor_count),
                         "engine_clock_mhz": getattr(props, "clock_rate", 0) / 1e3, "memory_clock_mhz": getattr(props, "memory_clock_rate", 0) / 1e3,
                         "memory_bus_bits": int(getattr(props, "memory_bus_width", 0)), "memory_bytes": int(props.total_memory),
                         "hbm_peak_gbs_used": HBM_PEAK_GBS, "rocminfo": _rocminfo_gpu(),
                         "note": "hipDeviceProp as torch reports it; `roofline.peak` is the MI355X HBM3E figure of /opt/skills/guides/MI355X_MICROARCH.md (8 TB/s), not derived from these clocks"}
    except Exception as e:
        out["device"] = {"error": f"{type(e).__name__}: {e}"}
    # Side blocks of the default line (N = 1, the metric's own invocation; never `value`): new paths every step, and the other BASELINE configs through the same loop
    default_line = world == 1 and args.workload == "cubic" and args.paths == 10000 and args.size == 4096 and not args.reupload and not args.no_side_workloads
    if default_line:
        try:
            out["reupload"] = side_reupload(renderer, batch, transforms, colors, size, args.steps, args.warmup)
        except Exception as e:  # (reported, never fatal: a side block)
            out["reupload"] = {"error": f"{type(e).__name__}: {e}"}
        others, skipped = [], []
        for name in ("dashed", "glyphs", "s100k"):
            if time.perf_counter() - t_process > args.side_budget_s:
                skipped.append(name)
                continue
            try:
                others.append(side_workload(name, local_rank, args.steps, args.warmup))
            except Exception as e:
                others.append({"workload": name, "error": f"{type(e).__name__}: {e}"})
        out["other_workloads"] = {"runs": others, "left_out_for_time": skipped,
                                  "note": "BASELINE configs[4] / [2] / [3] at N = 1 through the timed loop of `value` (20 set-up steps + warm-up untimed, K steps between two synchronisations); "
                                          "frame_equals_oracle: CRC-32 of the timed frame against tests/golden/bench_frame_crc.json (the oracle's frame of the same scene)"}
    # The CPU baseline (the oracle as the checker's clock, on rank 0 at N = 1 only) — after the GPU part: run before it, its sixteen busy
    # threads left the process with a ~20 ms host stall inside the timed region in one run out of three (the GPU idle, ms_per_step doubled)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.binding import time_tessellate
        t1 = time_tessellate(batch, 1, 1)
        repeats = max(2, min(200, int(math.ceil(10.0 / max(t1, 1e-3)))))
        ts = time_tessellate(batch, 1, repeats)
        cores = usable_cores()
        # (the result scenes of all repeats stay alive until the clock stops — first-touch page faults on one address space serialise in
        # the kernel, so a long run with gigabytes of live output measures the kernel's mm lock: 40 repeats at most)
        repeats_all = max(2, min(repeats, 40))
        time_tessellate(batch, cores, repeats_all)  # untimed: the first threaded run creates the threads' malloc arenas and faults their pages in
        tall = min(time_tessellate(batch, cores, repeats_all) for _ in range(2))
        cpu_baseline = {
            "value": batch.n_shapes * repeats / ts,
            "unit": "paths/s",
            "cores": 1,
            "kind": "port",
            "sample": f"{repeats} x full tessellation of the same {batch.n_shapes}-path scene by the C++ restatement of the reference's CPU tessellation "
                      f"(Shape::from_paths minus the wgpu upload; the reference itself cannot be built here), single thread as in renderer.rs:187; "
                      "tessellation only — the reference rasterizes on a GPU",
            "all_cores": {"value": batch.n_shapes * repeats_all / tall, "cores": cores, "repeats": repeats_all,
                          "note": "persistent thread pool, one malloc arena per thread, destruction outside the timed region, best of two runs after an "
                                  "untimed one (a cold run is 5x slower: arena creation and first-touch page faults); threads = the CPUs this process may "
                                  "use (affinity mask and cgroup CPU quota, not the host's core count: more threads than that only time-slice)"},
        }
    # like for like with cpu_baseline (which is tessellation only: the reference rasterizes on a GPU): the tessellation kernels of a step, each
    # with the GPU to itself
    tess_alone_ms = sum(v for k, v in alone.items() if k.startswith("tess_"))
    if tess_alone_ms > 0:
        out["gpu_tessellation"] = {"value": batch.n_shapes / (tess_alone_ms * 1e-3), "unit": "paths/s", "ms": tess_alone_ms,
                                   "note": "count / scan / emit / hull / range kernels of one step, stand-alone HIP-event times summed: the GPU side of what cpu_baseline times"}
    if cpu_baseline is not None:
        out["cpu_baseline"] = cpu_baseline
    # (the tile split's check has other keys — a gathered frame that differs from one GPU's is as invalid as a frame that differs from the oracle's;
    # the side blocks' frames count as well)
    failed = []
    if check:
        failed += [k for k in ("frame_equals_oracle", "gathered_equals_single_gpu_frame", "gathered_equals_ordered_composite_of_all_shards") if check.get(k) is False]
    if tile_side is not None and tile_side.get("gathered_equals_single_gpu_frame") is False:
        failed.append("tile_split.gathered_equals_single_gpu_frame")
    for run_ in (out.get("other_workloads") or {}).get("runs", []):
        if run_.get("frame_equals_oracle") is False:
            failed.append(f"other_workloads.{run_['workload']}.frame_equals_oracle")
    if (out.get("reupload") or {}).get("expected_crc32") is not None and out["reupload"]["frame_crc32"] != out["reupload"]["expected_crc32"]:
        failed.append("reupload.frame_crc32")
    wrong_pixels = bool(failed)
    if wrong_pixels:  # a run whose pixels differ from the oracle's frame is not a result: the line says so and the process fails
        out["invalid"] = True
        out["invalid_reason"] = "pixel checks failed: " + ", ".join(failed)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if wrong_pixels:
        sys.stdout.flush()
        raise SystemExit(4)


if __name__ == "__main__":
    main()
