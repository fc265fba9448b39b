#!/bin/bash
# usage: scratch/prof.sh <tag> [bench args...]   (run on the GPU box from the repo root); TRACE=1 keeps the kernel trace
tag=$1; shift
args=${@:---steps 3 --warmup 1}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py $args --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
rc=$?
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
find /tmp/prof_$tag -name "*stats*" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof_$tag/ \;
if [ -n "$TRACE" ]; then find /tmp/prof_$tag -name "*kernel_trace*" -exec cp {} /tmp/trace_$tag.csv \; ; fi
exit $rc
