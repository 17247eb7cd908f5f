#!/usr/bin/env python3
"""Turns the scratch output of tools/profile.sh <tag> and tools/pmc.sh <tag> (under gpurun_out/) into the committed evidence files
profiles/<round>_*[_<workload>]: usage  tools/collect_profiles.py <tag> <round> [bench_line.json] [workload]   (workload: glyphs | dashed; the
metric's own workload has no suffix)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, rnd = sys.argv[1], sys.argv[2]
workload = sys.argv[4] if len(sys.argv) > 4 else "cubic"
sfx = "" if workload == "cubic" else "_" + workload
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import kernel_source_hash  # noqa: E402  (the summaries are reported by bench.py only while this hash still matches)
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


shutil.copy(os.path.join(src, f"prof_{tag}", "stats", "r_kernel_stats.csv"), os.path.join(dst, f"{rnd}_bench_kernel_stats{sfx}.csv"))
summary = json.load(open(os.path.join(src, f"prof_{tag}", "summary.json")))
json.dump(summary, open(os.path.join(dst, f"{rnd}_bench_summary{sfx}.json"), "w"), indent=1)
traffic = {
    "kernel_source_hash": kernel_source_hash(),
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline" + ("" if workload == "cubic" else " --workload " + workload),
    "units": "FETCH_SIZE / WRITE_SIZE are KiB per launch; hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the x2 is the gfx950 FETCH_SIZE "
             "correction of MI355X_MICROARCH.md (HBM section); WRITE_SIZE is taken 1:1 (KiB). A raster kernel's WRITE_SIZE is the frame (W * H * 4 B = 65536 KiB at 4096^2) "
             "PLUS its scratch stores: a build that spills writes more than the frame, which is how the spills show",
    "kernels": {},
}
for name, v in summary.items():
    if "fetch_size_per_launch_raw" in v and "write_size_per_launch_raw" in v:
        traffic["kernels"][name] = {"fetch_kib_raw": v["fetch_size_per_launch_raw"], "write_kib_raw": v["write_size_per_launch_raw"],
                                    "hbm_bytes_per_launch": int((2 * v["fetch_size_per_launch_raw"] + v["write_size_per_launch_raw"]) * 1024),
                                    "avg_ns": v.get("avg_ns")}
json.dump(traffic, open(os.path.join(dst, f"{rnd}_traffic{sfx}.json"), "w"), indent=1)
per_launch = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(src, f"pmc_{tag}", "p*", "*counter_collection.csv"))):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        acc[(short(row["Kernel_Name"]), row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (kernel, counter), values in acc.items():
        per_launch[kernel][counter] = sum(values) / len(values)
json.dump({"kernel_source_hash": kernel_source_hash(), "command": "rocprofv3 --kernel-trace --pmc <set> -f csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline" + ("" if workload == "cubic" else " --workload " + workload) + " (one pass per counter set, "
                      "CRH_NO_PIPELINE unset)", "per_launch": per_launch}, open(os.path.join(dst, f"{rnd}_sq_counters{sfx}.json"), "w"), indent=1)
if len(sys.argv) > 3 and sys.argv[3] != "-":
    line = [l for l in open(sys.argv[3]).read().splitlines() if l.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(dst, f"{rnd}_bench_line{sfx}.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))
