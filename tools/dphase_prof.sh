#!/bin/bash
# kernel list of ONE critics' phase (run on the GPU box from the repo root): usage  bash tools/dphase_prof.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp
for n in 4 24; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp_${tag}_$n -o d$n -- python $GRAFT_REPO_ROOT/tools/dphase_prof.py $n > $GRAFT_REPO_ROOT/gpurun_out/dphase_${tag}_$n.log 2>&1 || exit 1
  find /tmp/dp_${tag}_$n -name "*kernel_stats*" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/dphase_${tag}_$n.csv \;
done
cd $GRAFT_REPO_ROOT && python tools/dphase_diff.py gpurun_out/dphase_${tag}_4.csv 4 gpurun_out/dphase_${tag}_24.csv 24 80 > gpurun_out/dphase_${tag}.txt
grep "D phase" gpurun_out/dphase_${tag}_24.log >> gpurun_out/dphase_${tag}.txt
