#!/bin/bash
# GPU box: cost of each entry class of k_raster_edges (rebuilds the library with -DCRH_ABLATE, restores it afterwards)
cd $GRAFT_REPO_ROOT
CRH_EXTRA_FLAGS=-DCRH_ABLATE python contrast_renderer_amd/build.py --force > /dev/null 2>&1
fmt='import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k.startswith("raster")})'
python tools/count_entries.py ${1:-cubic}
for dbg in ${DBGS:-0 8 16 32 56 64 128}; do
  echo "debug=$dbg (8: no edges, 16: no synth, 32: no triangles, 64: empty lists, 128: sort + setup only, 512: stroke triangles without their fragment stages)"
  CRH_RASTER_DEBUG=$dbg CRH_NO_PIPELINE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload ${1:-cubic} 2>&1 | tail -1 | python -c "$fmt"
done
python contrast_renderer_amd/build.py --force > /dev/null 2>&1
