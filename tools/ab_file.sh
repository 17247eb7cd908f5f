#!/bin/bash
# GPU box: same-box A/B of two versions of one source file: tools/ab_file.sh <current file> <old copy> [workloads...]
cd $GRAFT_REPO_ROOT
cur=$1; old=$2; shift; shift
export WORKLOADS="${*:-cubic}"
cp $cur /tmp/_new_version
for rep in 1 2; do
  cp $old $cur; python contrast_renderer_amd/build.py --force > /dev/null 2>&1
  echo "#### old"; bash tools/quick_bench.sh
  cp /tmp/_new_version $cur; python contrast_renderer_amd/build.py --force > /dev/null 2>&1
  echo "#### new"; bash tools/quick_bench.sh
done
