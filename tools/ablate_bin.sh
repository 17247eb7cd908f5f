#!/bin/bash
# GPU box: where the time of k_bin_edges goes (library rebuilt with -DCRH_ABLATE, restored afterwards; pixels are wrong in these runs)
cd $GRAFT_REPO_ROOT
CRH_EXTRA_FLAGS=-DCRH_ABLATE python contrast_renderer_amd/build.py --force > /dev/null 2>&1
fmt='import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k.startswith("raster_bin")})'
for dbg in ${DBGS:-0 512 1024 2048 3072 9216 17408}; do
  echo "debug=$dbg (512: no stage, 1024: no triangle wave, 2048: no edge wave, 8192: no edge passes, 16384: no edge loop, 131072: no tile-count atomics, 262144: no cursor atomics)"
  CRH_RASTER_DEBUG=$dbg CRH_NO_PIPELINE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload ${1:-cubic} 2>&1 | tail -1 | python -c "$fmt"
done
python contrast_renderer_amd/build.py --force > /dev/null 2>&1
