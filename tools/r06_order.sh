#!/bin/bash
# GPU box: the raster kernels' tile order (CRH_HEAVY_FIRST: default = every XCD's tiles by falling count; q<k> = squares of 2^k x 2^k tiles by their
# heaviest tile, a square's tiles back to back) — step, raster kernel in the run / alone; PMC=1: also FETCH_SIZE of the raster kernel per launch (its own pass)
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["spread"]["ms_per_step_median"],4) if d.get("spread") else None, round(d.get("latency_ms_per_step") or 0,3), {k:(round(v["avg_ms"],4), round(v["alone_ms"],4) if v.get("alone_ms") else None) for k,v in d["kernels"].items() if k.startswith("raster")})'
for w in ${WORKLOADS:-cubic}; do
for v in ${VARIANTS:-sort q1 q2 q3}; do
  echo "== $w CRH_HEAVY_FIRST=$v"
  CRH_HEAVY_FIRST=$v timeout 200 python bench.py --no-cpu-baseline --no-animated --no-side-workloads --repeats 3 --workload $w 2>/dev/null | tail -1 | python -c "$fmt"
  if [ -n "$PMC" ]; then CRH_HEAVY_FIRST=$v WORKLOAD=$w bash tools/pmc.sh order_${w}_$v "FETCH_SIZE" 2>&1 | grep -i "raster_fill\|raster_rows\|raster_edges"; fi
done
done
