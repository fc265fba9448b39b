#!/bin/bash
# measurement cases of SURVEY 8d, one JSON line each -> gpurun_out/workloads_<tag>.jsonl ; usage: workloads.sh tag w1 w2 ...
set -o pipefail
tag=$1; shift
out=gpurun_out/workloads_$tag.jsonl
: > $out
for w in "$@"; do
  case $w in
    cycle) a="--steps 3 --warmup 1 --no-roofline"; t=240;;
    long) a="--steps 3 --warmup 1"; t=500;;
    *) a="--steps 20 --warmup 5"; t=240;;
  esac
  timeout -k 10 $t python bench.py --workload $w $a --no-cpu-baseline 2> gpurun_out/wl_$w.err | tee -a $out || { echo "FAILED $w"; tail -8 gpurun_out/wl_$w.err; exit 1; }
done
