#!/bin/bash
# GPU box: rebuild raster.hip with each flag set and print the raster kernel times. Usage: tools/ablate.sh "<flags A>" "<flags B>" ...
for flags in "$@"; do
  touch contrast_renderer_amd/csrc/raster.hip
  CRH_EXTRA_FLAGS="$flags" python contrast_renderer_amd/build.py > /dev/null 2>&1
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernels']
print('%-60s ms/step %.3f  tiles %.4f  fill %.4f  count %.4f' % (sys.argv[1], d['ms_per_step'], k['raster_tiles']['avg_ms'], k['raster_tile_fill']['avg_ms'], k['raster_tile_count']['avg_ms']))
" "$flags"
done
