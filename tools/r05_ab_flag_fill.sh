#!/bin/bash
# GPU box: the fill raster kernel with a compile-time flag against the default build, same box: tools/r05_ab_flag_fill.sh "<flags>" [workloads]
cd $GRAFT_REPO_ROOT
flag=$1; shift
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["latency_ms_per_step"],3), d["check"] and d["check"]["frame_equals_oracle"], {k:(round(v["avg_ms"],4), v["alone_ms"] and round(v["alone_ms"],4)) for k,v in d["kernels"].items() if k.startswith("raster_tiles") or k.startswith("raster_bin")})'
for rep in 1 2; do
  for f in "$flag" ""; do
    CRH_EXTRA_FLAGS="$f" python contrast_renderer_amd/build.py --force > /dev/null 2>&1
    for w in ${*:-cubic}; do
      echo "== $w flags: '$f'"
      CRH_EDGE_PASS=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --repeats 2 --workload $w 2>&1 | tail -1 | python -c "$fmt"
    done
  done
done
