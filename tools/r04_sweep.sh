#!/bin/bash
# GPU box: build variants of k_raster_rows (-D flags) and time the raster kernel alone on a workload. Usage: tools/r04_sweep.sh <workload> "<flags>" "<flags>" ...
cd $GRAFT_REPO_ROOT
w=$1; shift
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), d["check"] and d["check"]["frame_equals_oracle"], "raster in run / alone", round(d["kernels"]["raster_tiles"]["avg_ms"],4), round(d["kernels"]["raster_tiles"]["alone_ms"],4))'
for flags in "$@"; do
  CRH_EXTRA_FLAGS="$flags" python contrast_renderer_amd/build.py --force > /dev/null 2>&1
  echo "== $w [$flags]"
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
done
python contrast_renderer_amd/build.py --force > /dev/null 2>&1
