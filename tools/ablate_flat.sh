#!/bin/bash
# GPU box: k_bin_flat cut off behind phase k (library rebuilt with -DCRH_ABLATE, restored afterwards; pixels are wrong in these runs):
# the stand-alone kernel time as a function of how far it runs = what each phase costs in wall time
cd $GRAFT_REPO_ROOT
CRH_EXTRA_FLAGS=-DCRH_ABLATE python contrast_renderer_amd/build.py --force > /dev/null 2>&1
fmt='import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k.startswith("raster_bin")})'
for k in ${PHASES:-1 2 3 4 5 6 7 8 0}; do
  echo "stop behind phase $((k-1)) (0 item records, 1 edge loads, 2 triangle set-up + record stores, 3 synthetic records, 4 rectangles + row table, 5 pass 1, 6 pass 2, 7 pass 3; -1 = the whole kernel)"
  CRH_RASTER_DEBUG=$((k << 24)) CRH_EDGE_PASS=1 CRH_NO_PIPELINE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload ${1:-cubic} 2>&1 | tail -1 | python -c "$fmt"
done
python contrast_renderer_amd/build.py --force > /dev/null 2>&1
