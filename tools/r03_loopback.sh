#!/bin/bash
# GPU box: the eight-rank loopback lines of the round (config 4 whole, RGBA8 and RGBA16F layers, S10k strong and weak); CRH_LOOPBACK_SERIAL=1
# times every rank's part of a phase with the GPU to itself
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --workload s100k --loopback 8 --steps 5 --warmup 1 > gpurun_out/bench_r03_s100k_loop8.json 2> gpurun_out/bench_r03_s100k_loop8.err
CRH_LOOPBACK_SERIAL=1 python bench.py --workload s100k --loopback 8 --steps 5 --warmup 1 > gpurun_out/bench_r03_s100k_loop8_serial.json 2>> gpurun_out/bench_r03_s100k_loop8.err
CRH_LOOPBACK_SERIAL=1 python bench.py --workload s100k --loopback 8 --layers rgba16f --steps 5 --warmup 1 > gpurun_out/bench_r03_s100k_loop8_16f_serial.json 2>> gpurun_out/bench_r03_s100k_loop8.err
CRH_LOOPBACK_SERIAL=1 python bench.py --loopback 8 --scaling strong --steps 10 > gpurun_out/bench_r03_s10k_strong_loop8_serial.json 2>> gpurun_out/bench_r03_s100k_loop8.err
CRH_LOOPBACK_SERIAL=1 python bench.py --loopback 8 --scaling weak --steps 5 > gpurun_out/bench_r03_s10k_weak_loop8_serial.json 2>> gpurun_out/bench_r03_s100k_loop8.err
for f in gpurun_out/bench_r03_*loop8*.json; do echo "== $f"; python - $f <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
lb = d["loopback"]
print("ms/step %.3f" % d["ms_per_step"], {k: lb[k] for k in ("draw_per_rank_ms", "sent_over_dense")}, lb["exchange_wall_ms"]["median"], lb["xgmi_estimate"]["alltoall_ms"], lb["xgmi_estimate"]["gather_ms"])
print({k: round(v["max_over_ranks"], 3) for k, v in lb["exchange_phase_ms"].items()}, lb.get("per_rank_estimate_ms"))
PY
done
