#!/bin/bash
# GPU box: all evidence of one workload for profiles/ (tools/collect_all.sh <round> <workload>): rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE,
# the SQ counter sets, and the bench line of the same tree. The summaries are assembled by tools/collect_profiles.py back in the container.
rnd=$1; w=${2:-cubic}
cd $GRAFT_REPO_ROOT
WORKLOAD=$w bash tools/profile.sh ${rnd}_$w > gpurun_out/profile_${rnd}_$w.log 2>&1
WORKLOAD=$w bash tools/pmc.sh ${rnd}_$w "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAIT_INST_LDS" > gpurun_out/pmc_${rnd}_$w.log 2>&1
if [ "$w" = cubic ]; then python bench.py > gpurun_out/bench_${rnd}_$w.json 2> gpurun_out/bench_${rnd}_$w.err
else python bench.py --workload $w > gpurun_out/bench_${rnd}_$w.json 2> gpurun_out/bench_${rnd}_$w.err; fi
CRH_NO_PIPELINE=1 python bench.py --workload $w --no-cpu-baseline > gpurun_out/bench_${rnd}_${w}_standalone.json 2>/dev/null
tail -c 600 gpurun_out/bench_${rnd}_$w.json
