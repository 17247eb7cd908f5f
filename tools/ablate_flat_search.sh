#!/bin/bash
# GPU box: k_bin_flat with the ten-step search of the (edge, tile row) items replaced by a modulo (wrong edges: timing only), whole kernel and
# cut off behind pass 1
cd $GRAFT_REPO_ROOT
CRH_EXTRA_FLAGS=-DCRH_ABLATE python contrast_renderer_amd/build.py --force > /dev/null 2>&1
fmt='import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k.startswith("raster_bin")})'
for dbg in 0 4096 $((6 << 24)) $(((6 << 24) + 4096)) $((5 << 24)) ; do
  echo "debug $dbg"
  CRH_RASTER_DEBUG=$dbg CRH_EDGE_PASS=1 CRH_NO_PIPELINE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt"
done
python contrast_renderer_amd/build.py --force > /dev/null 2>&1
