#!/bin/bash
# GPU box: kTessBlock (elements per workgroup of k_count / k_emit) 256 / 128 / 64, pipelined and stand-alone
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k.startswith("tess")})'
for b in 64 128; do
  CRH_EXTRA_FLAGS=-DCRH_TESS_BLOCK=$b python contrast_renderer_amd/build.py --force > /dev/null 2>&1
  python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config or glyphs_600 or mixed" 2>&1 | tail -1
  for w in cubic glyphs dashed; do
    echo "== block $b $w pipelined / stand-alone"
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
    CRH_NO_PIPELINE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
  done
done
