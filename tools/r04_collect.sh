#!/bin/bash
# GPU box: the round's evidence — per workload kernel stats / traffic / SQ counters / bench lines (tools/collect_all.sh), the eight-rank loopback
# of config 4 with its exchange, the host-inclusive mode, the two-rank self-launched line, the parity suite.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/pytest_r04.log
for w in cubic glyphs dashed s100k; do bash tools/collect_all.sh r04 $w > /dev/null 2>&1; done
python bench.py --workload s100k --loopback 8 --steps 5 --warmup 1 > gpurun_out/bench_r04_s100k_loop8.json 2> gpurun_out/bench_r04_loop8.err
CRH_LOOPBACK_SERIAL=1 python bench.py --workload s100k --loopback 8 --steps 5 --warmup 1 > gpurun_out/bench_r04_s100k_loop8_serial.json 2>> gpurun_out/bench_r04_loop8.err
python bench.py --reupload --no-cpu-baseline > gpurun_out/bench_r04_reupload.json 2> gpurun_out/bench_r04_reupload.err
python bench.py --gpus 2 --backend gloo --same-device --steps 10 --no-cpu-baseline > gpurun_out/bench_r04_gpus2_same_device.json 2> gpurun_out/bench_r04_gpus2.err
cat gpurun_out/pytest_r04.log
for f in gpurun_out/bench_r04_*.json; do echo "== $f"; python - $f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
except Exception as e:
    print("unreadable:", e); sys.exit(0)
print("ms/step %.3f value %.3e latency %s check %s" % (d["ms_per_step"], d["value"], d.get("latency_ms_per_step"), (d.get("check") or {}).get("frame_equals_oracle")))
if d.get("roofline"): print({k: d["roofline"][k] for k in ("kernel", "frac", "traffic", "avg_launch_ms", "avg_launch_ms_alone", "pass")}, (d.get("roofline_longest_in_run") or d.get("roofline_longest_alone")) and {k: (d.get("roofline_longest_in_run") or d.get("roofline_longest_alone"))[k] for k in ("kernel", "frac", "avg_launch_ms")})
if d.get("weak_scaling"): print("weak:", d["weak_scaling"]["value"], d["weak_scaling"]["ms_per_step"])
PY
done
