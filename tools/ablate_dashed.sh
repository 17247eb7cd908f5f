#!/bin/bash
for flags in "$@"; do
  touch contrast_renderer_amd/csrc/raster.hip
  CRH_EXTRA_FLAGS="$flags" python contrast_renderer_amd/build.py > /dev/null 2>&1
  CRH_NO_PIPELINE=1 python bench.py --workload dashed --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-40s ms/step %.3f  tiles %.4f' % (sys.argv[1], d['ms_per_step'], d['kernels']['raster_tiles']['avg_ms']))
" "$flags"
done
