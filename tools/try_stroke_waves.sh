#!/bin/bash
# GPU box: register budget of the stroke / msaa 4 variants of k_raster_edges
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k.startswith("raster")})'
for sw in ${SW:-1 4 5}; do
  CRH_EXTRA_FLAGS=-DCRH_STROKE_TILE_WAVES=$sw python contrast_renderer_amd/build.py --force 2>&1 | grep -c "spill"
  echo "== stroke waves $sw dashed stand-alone"
  CRH_NO_PIPELINE=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload dashed 2>&1 | tail -1 | python -c "$fmt"
done
python contrast_renderer_amd/build.py --force > /dev/null 2>&1
