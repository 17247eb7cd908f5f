#!/bin/bash
# GPU box: the dashed workload (config 5) with the stroke / msaa-4 raster kernel at 5 (default), 4 and 3 wavefronts per SIMD
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), {k:(round(v["avg_ms"],4), round(v["alone_ms"],4) if v["alone_ms"] else None) for k,v in d["kernels"].items() if k.startswith("raster_tiles")})'
for w in 5 4 3; do
  CRH_EXTRA_FLAGS=-DCRH_STROKE_TILE_WAVES=$w python contrast_renderer_amd/build.py --force > /dev/null 2>&1
  echo "== CRH_STROKE_TILE_WAVES=$w"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload dashed 2>&1 | tail -1 | python -c "$fmt"
done
python contrast_renderer_amd/build.py --force > /dev/null 2>&1
