#!/bin/bash
# GPU box: same-box A/B of a compile-time switch: tools/ab_flag.sh -DSOMETHING [workloads...] — library with the flag, without, with, without
cd $GRAFT_REPO_ROOT
flag=$1; shift
export WORKLOADS="${*:-cubic}"
for rep in 1 2; do
  CRH_EXTRA_FLAGS=$flag python contrast_renderer_amd/build.py --force > /dev/null 2>&1
  echo "#### with $flag"; bash tools/quick_bench.sh
  python contrast_renderer_amd/build.py --force > /dev/null 2>&1
  echo "#### default build"; bash tools/quick_bench.sh
done
