#!/bin/bash
# GPU box: raster kernel of the given workloads, in the run and alone (the pass pinned to the per-sample edge kernel unless PIN is set otherwise)
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), d["check"] and d["check"]["frame_equals_oracle"], {k:(round(v["avg_ms"],4), v["alone_ms"] and round(v["alone_ms"],4)) for k,v in d["kernels"].items() if k.startswith("raster")})'
for w in ${1:-cubic}; do
  echo "== $w"
  env ${PIN:-CRH_EDGE_PASS=1} timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
done
