#!/bin/bash
# GPU box: parity subset + kernel times (pipelined and stand-alone). Usage: bash tools/quick.sh [pytest -k expression]
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -q --timeout 900 -x -k "${1:-random_scene or random_paths or quadratic or mixed or cubic or glyphs_600 or config}" 2>&1 | tail -4
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items()})'
for w in cubic glyphs dashed; do
  echo "== $w pipelined / stand-alone"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
  CRH_NO_PIPELINE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
done
