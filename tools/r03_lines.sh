#!/bin/bash
# GPU box: the bench lines of the round with the committed PMC summaries in place (bench.py reports traffic / valu_issue from profiles/ while
# their kernel-source hash matches): default workload = the driver's command, then the other workloads, pipelined and stand-alone
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py > gpurun_out/line_r03.json 2> gpurun_out/line_r03.err
for w in glyphs dashed s100k; do python bench.py --workload $w > gpurun_out/line_r03_$w.json 2>> gpurun_out/line_r03.err; done
for w in cubic glyphs dashed; do CRH_NO_PIPELINE=1 python bench.py --workload $w --no-cpu-baseline > gpurun_out/line_r03_standalone_$w.json 2>> gpurun_out/line_r03.err; done
python bench.py --reupload --no-cpu-baseline > gpurun_out/line_r03_reupload.json 2>> gpurun_out/line_r03.err
for f in gpurun_out/line_r03*.json; do python - $f <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
r = d["roofline"]
print(sys.argv[1], "%.4f ms/step, %.3e" % (d["ms_per_step"], d["value"]), r["kernel"], "frac %.4f" % r["frac"], "traffic", r.get("traffic"), (r.get("valu_issue") or {}).get("frac_of_valu_issue_peak"), d.get("cpu_baseline", {}).get("value"))
PY
done
