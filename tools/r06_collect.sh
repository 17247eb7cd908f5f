#!/bin/bash
# GPU box: the round's evidence — per workload kernel stats / traffic / SQ counters / bench lines (tools/collect_all.sh), the eight-rank loopbacks of
# both splits (path / tile) on the metric's scene and on config 4, the host-inclusive mode, the two-rank self-launched line, the parity suite.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/pytest_r06.log
for w in cubic glyphs dashed s100k; do timeout 1500 bash tools/collect_all.sh r06 $w > /dev/null 2>&1; done
for w in cubic s100k; do for sp in path tile; do
  CRH_LOOPBACK_SERIAL=1 python bench.py --workload $w --loopback 8 --split $sp --scaling strong --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r06_${w}_loop8_${sp}_serial.json 2>> gpurun_out/bench_r06_loop8.err
done; done
timeout 300 python bench.py --reupload --no-cpu-baseline > gpurun_out/bench_r06_reupload.json 2> gpurun_out/bench_r06_reupload.err
timeout 300 python bench.py --gpus 2 --backend gloo --same-device --steps 10 --no-cpu-baseline > gpurun_out/bench_r06_gpus2_same_device.json 2> gpurun_out/bench_r06_gpus2.err
for w in cubic s100k; do python tools/r05_slab_step.py $w > gpurun_out/r06_slab_step_$w.txt 2>/dev/null; done
CRH_EDGE_PASS=1 python tools/r05_animated_check.py cubic > gpurun_out/r06_animated_check.txt 2>/dev/null
python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/r06_smoke.txt 2>&1; tail -1 gpurun_out/r06_smoke.txt
cat gpurun_out/pytest_r06.log
for f in gpurun_out/bench_r06_*.json; do echo "== $f"; python - $f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
except Exception as e:
    print("unreadable:", e); sys.exit(0)
print("ms/step %.3f value %.3e latency %s check %s animated %s" % (d["ms_per_step"], d["value"], d.get("latency_ms_per_step"), (d.get("check") or {}).get("frame_equals_oracle"), (d.get("animated") or {}).get("ms_per_step")))
if d.get("roofline"): print({k: d["roofline"][k] for k in ("kernel", "frac", "traffic", "avg_launch_ms", "avg_launch_ms_alone", "pass")}, d.get("roofline_longest_in_run") and {k: d["roofline_longest_in_run"][k] for k in ("kernel", "frac", "avg_launch_ms")})
if d.get("weak_scaling"): print("weak:", d["weak_scaling"]["value"], d["weak_scaling"]["ms_per_step"])
if d.get("tile_split"): print("tile split:", d["tile_split"]["value"], d["tile_split"]["ms_per_step"])
PY
done
