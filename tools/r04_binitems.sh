#!/bin/bash
# GPU box: k_bin_flat with the given numbers of items per workgroup (CRH_BIN_ITEMS), in the run and alone
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), d.get("latency_ms_per_step") and round(d["latency_ms_per_step"],4), d["check"] and d["check"]["frame_equals_oracle"], {k:(round(v["avg_ms"],4), v["alone_ms"] and round(v["alone_ms"],4)) for k,v in d["kernels"].items() if k.startswith("raster_bin") or k.startswith("tess")})'
for w in ${1:-cubic}; do
  for n in ${2:-0 4 6 8 10}; do
    echo "== $w items $n"
    if [ "$n" = 0 ]; then env ${PIN:-X=1} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
    else env ${PIN:-X=1} CRH_BIN_ITEMS=$n timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"; fi
  done
done
