#!/usr/bin/env python3
"""Host: one line per bench line under gpurun_out/r05b (value, ms per step, latency, the tessellation lane's kernels alone / in the run)."""
import glob
import json
import sys

for f in sorted(glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05b") + "/bench*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as e:  # noqa: BLE001
        print(f, "unreadable:", e)
        continue
    k = d.get("kernels", {})
    lane = " ".join(f"{n}={v.get('alone_ms', 0):.3f}/{v['avg_ms']:.3f}" for n, v in k.items())
    print(f"{f.split('/')[-1]:34s} value {d.get('value', 0) / 1e6:7.2f} M  step {d.get('ms_per_step', 0):.4f}  latency {d.get('latency_ms_per_step', 0):.4f}  crc_ok {d.get('check', {}).get('frame_equals_oracle')}\n      {lane}")
