#!/bin/bash
# GPU box, round 3 first trip: issue-rate microbenchmark, parity suite, bench lines incl. config 4 whole (single GPU and 8-rank loopback)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate 2> gpurun_out/valu_rate.err && timeout 300 /tmp/valu_rate > gpurun_out/valu_rate.json 2>> gpurun_out/valu_rate.err
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -25 | tee gpurun_out/pytest_r03a.log
timeout 600 python bench.py 2> gpurun_out/bench_r03a_cubic.err | tail -1 > gpurun_out/bench_r03a_cubic.json
timeout 600 python bench.py --workload s100k --no-cpu-baseline 2> gpurun_out/bench_r03a_s100k.err | tail -1 > gpurun_out/bench_r03a_s100k.json
timeout 900 python bench.py --workload s100k --loopback 8 --steps 5 --warmup 1 2> gpurun_out/bench_r03a_s100k_loop8.err | tail -1 > gpurun_out/bench_r03a_s100k_loop8.json
timeout 900 python bench.py --workload s100k --loopback 8 --layers rgba16f --steps 5 --warmup 1 2> gpurun_out/bench_r03a_s100k_loop8_16f.err | tail -1 > gpurun_out/bench_r03a_s100k_loop8_16f.json
timeout 600 python bench.py --loopback 8 --scaling strong --steps 10 2> gpurun_out/bench_r03a_s10k_strong_loop8.err | tail -1 > gpurun_out/bench_r03a_s10k_strong_loop8.json
timeout 600 python bench.py --loopback 8 --scaling weak --steps 5 2> gpurun_out/bench_r03a_s10k_weak_loop8.err | tail -1 > gpurun_out/bench_r03a_s10k_weak_loop8.json
for f in gpurun_out/bench_r03a_*.json; do echo "== $f"; python - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("unreadable:", e); sys.exit(0)
print("ms/step %.3f value %.3e" % (d["ms_per_step"], d["value"]))
if d.get("kernels"): print({k: (round(v["avg_ms"], 4), round(v["alone_ms"], 4) if v["alone_ms"] else None) for k, v in d["kernels"].items()})
if d.get("loopback"): print(json.dumps(d["loopback"])[:1500])
PY
done
tail -3 gpurun_out/*_r03a_*.err
