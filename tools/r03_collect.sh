#!/bin/bash
# GPU box: the round's evidence — per workload kernel stats / traffic / SQ counters / bench lines (tools/collect_all.sh), the whole of
# config 4 on one GPU (single frame and the eight-rank loopback with its exchange), the host-inclusive mode, the parity suite.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/pytest_r03.log
for w in cubic glyphs dashed s100k; do bash tools/collect_all.sh r03 $w > /dev/null 2>&1; done
bash tools/r03_loopback.sh > /dev/null 2>&1
python bench.py --reupload --no-cpu-baseline > gpurun_out/bench_r03_reupload.json 2> gpurun_out/bench_r03_reupload.err
cat gpurun_out/pytest_r03.log
for f in gpurun_out/bench_r03_*.json; do echo "== $f"; python - $f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
except Exception as e:
    print("unreadable:", e); sys.exit(0)
print("ms/step %.3f value %.3e latency %s" % (d["ms_per_step"], d["value"], d.get("latency_ms_per_step")))
if d.get("roofline"): print({k: d["roofline"][k] for k in ("kernel", "frac", "traffic", "avg_launch_ms", "avg_launch_ms_alone")}, (d["roofline"].get("valu_issue") or {}).get("frac_of_valu_issue_peak"))
if d.get("loopback"): print({k: d["loopback"][k] for k in ("draw_per_rank_ms", "exchange_wall_ms", "sent_over_dense", "xgmi_estimate")}, {k: round(v["max_over_ranks"], 3) for k, v in d["loopback"]["exchange_phase_ms"].items()})
PY
done
