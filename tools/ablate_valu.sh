#!/bin/bash
# GPU box: VALU / SALU instruction counts of k_raster_edges per entry class (ablation build + rocprofv3 --pmc)
cd $GRAFT_REPO_ROOT
CRH_EXTRA_FLAGS=-DCRH_ABLATE python contrast_renderer_amd/build.py --force > /dev/null 2>&1
out=$GRAFT_REPO_ROOT/gpurun_out/ablate_valu; rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for dbg in ${DBGS:-0 8 16 32 56 64 128}; do
  CRH_RASTER_DEBUG=$dbg CRH_NO_PIPELINE=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -f csv -d $out/d$dbg -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload ${1:-cubic} > $out/d$dbg.log 2>&1
  f=$(find $out/d$dbg -name "*counter_collection.csv" | head -1)
  python - "$f" $dbg <<'PY'
import sys, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    if "k_raster_edges" in row["Kernel_Name"] or "k_bin_edges" in row["Kernel_Name"]:
        agg[row["Kernel_Name"].split("(")[0][-28:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print("debug", sys.argv[2], k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()})
PY
done
cd $GRAFT_REPO_ROOT; python contrast_renderer_amd/build.py --force > /dev/null 2>&1
