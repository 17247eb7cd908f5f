#!/bin/bash
# GPU box: one line per workload (pipelined step, kernels in the run / alone) — the quick look between two kernel edits
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d.get("latency_ms_per_step") or 0,3), {k:(round(v["avg_ms"],3), round(v["alone_ms"],3) if v.get("alone_ms") else None) for k,v in d["kernels"].items() if k.startswith("raster")})'
for w in ${WORKLOADS:-cubic glyphs dashed s100k}; do
  echo "== $w $EXTRA"
  python bench.py --no-cpu-baseline --workload $w $EXTRA 2>/dev/null | tail -1 | python -c "$fmt"
done
