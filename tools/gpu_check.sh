#!/bin/bash
# GPU box: parity tests, then a bench line. Usage: tools/gpu_check.sh [tag]
tag=${1:-run}
mkdir -p gpurun_out
timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/pytest_$tag.log
python bench.py --steps 20 --warmup 3 2>&1 | tail -1 | tee gpurun_out/bench_$tag.json | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms/step %.3f  paths/s %.3e  Mpix/s %.0f  roofline frac %.4f' % (d['ms_per_step'], d['value'], d['mpixel_per_s'], d['roofline']['frac']))
for k, v in d['kernels'].items(): print('   %-22s %.4f ms' % (k, v['avg_ms']))
print('cpu', d.get('cpu_baseline', {}).get('value'))
"
