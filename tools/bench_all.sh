#!/bin/bash
# GPU box: one bench line per workload (cubic = headline). Usage: tools/bench_all.sh <tag>
tag=${1:-run}
mkdir -p gpurun_out
for w in cubic glyphs dashed; do
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_${tag}_$w.json
  python - gpurun_out/bench_${tag}_$w.json $w <<'PY'
import sys, json
d = json.load(open(sys.argv[1]))
print('%-7s ms/step %.3f  paths/s %.3e  Mpix/s %.0f' % (sys.argv[2], d['ms_per_step'], d['value'], d['mpixel_per_s']))
for k, v in d['kernels'].items(): print('   %-22s %.4f ms' % (k, v['avg_ms']))
PY
done
