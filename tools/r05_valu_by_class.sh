#!/bin/bash
# GPU box: vector / scalar instructions of the fill raster kernels with each class of entries left out (library built with -DCRH_ABLATE here):
# CRH_RASTER_DEBUG 8 no edges, 16 no synthetic entries, 32 no triangles, 64 empty lists, 128 sort + set-up only; CRH_FILL_KERNEL 1 / 0
cd $GRAFT_REPO_ROOT
CRH_EXTRA_FLAGS=-DCRH_ABLATE python contrast_renderer_amd/build.py --force > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
for fk in ${KERNELS:-1 0}; do
for dbg in ${DBGS:-0 8 16 32 64 128}; do
  out=$GRAFT_REPO_ROOT/gpurun_out/valu_class_$dbg; rm -rf $out; mkdir -p $out
  CRH_FILL_KERNEL=$fk CRH_EDGE_PASS=1 CRH_NO_PIPELINE=1 CRH_RASTER_DEBUG=$dbg rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -f csv -d $out -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check --repeats 0 --workload ${1:-cubic} > $out/log 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python - "$f" $dbg $fk <<'PY'
import sys, csv, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    if "k_raster_edges" in row["Kernel_Name"] or "k_raster_fill" in row["Kernel_Name"]:
        agg[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
print("fill_kernel", sys.argv[3], "debug", sys.argv[2], {c: round(v / cnt[c] / 1e6, 2) for c, v in agg.items()}, "M per launch")
PY
done
done
cd $GRAFT_REPO_ROOT; python contrast_renderer_amd/build.py --force > /dev/null 2>&1
