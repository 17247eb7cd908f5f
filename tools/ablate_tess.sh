#!/bin/bash
# GPU box: rebuild everything with each flag set (flags reach all .hip files) and print the tessellation kernel times.
for flags in "$@"; do
  CRH_EXTRA_FLAGS="$flags" python contrast_renderer_amd/build.py --force > /dev/null 2>&1
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernels']
print('%-40s ms/step %.3f  count %.4f scan %.4f emit %.4f hull %.4f setup %.4f' % (sys.argv[1], d['ms_per_step'], k['tess_count']['avg_ms'], k['tess_scan']['avg_ms'], k['tess_emit']['avg_ms'], k['tess_hull']['avg_ms'], k['raster_prim_setup']['avg_ms']))
" "$flags"
done
