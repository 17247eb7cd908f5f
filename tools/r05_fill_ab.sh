#!/bin/bash
# GPU box: k_raster_fill (round 5) against k_raster_edges<1,4,false,*> (CRH_FILL_KERNEL=0) on the fill workloads, raster kernel in the run and alone
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["latency_ms_per_step"],3), d["check"] and d["check"]["frame_equals_oracle"], {k:(round(v["avg_ms"],4), v["alone_ms"] and round(v["alone_ms"],4)) for k,v in d["kernels"].items() if k.startswith("raster")})'
for w in ${1:-cubic s100k}; do
  for v in 1 0 1 0; do
    echo "== $w CRH_FILL_KERNEL=$v"
    env CRH_FILL_KERNEL=$v ${PIN:-CRH_EDGE_PASS=1} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --repeats 2 --workload $w 2>&1 | tail -1 | python -c "$fmt"
  done
done
