#!/bin/bash
# usage (GPU box, repo root): bash tools/replication_prof.sh <tag>  -> gpurun_out/<tag>_replication.txt
# microseconds (kernel trace) and FETCH_SIZE (its own --pmc pass) per launch of the 240x1536x512 / 240x512x1536 and
# 160-row products, with the operands staged by LDS-DMA (round 3) and as register fragments (round 2: MSSR_SKINNY_DMA=0)
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_replication.txt
: > $out
cd /tmp && export TMPDIR=/tmp
for M in 240 160; do for dma in 1 0; do
  d=/tmp/rep_${M}_${dma}
  MSSR_SKINNY_DMA=$dma timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d/t -o t -- python $GRAFT_REPO_ROOT/tools/replication_bench.py $M 100 > $GRAFT_REPO_ROOT/gpurun_out/rep_last.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/rep_last.log; exit 1; }
  MSSR_SKINNY_DMA=$dma timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/p -o p -- python $GRAFT_REPO_ROOT/tools/replication_bench.py $M 100 > $GRAFT_REPO_ROOT/gpurun_out/rep_last.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/rep_last.log; exit 1; }
  python3 - $(find $d/t -name "*kernel_stats.csv" | head -1) $(find $d/p -name "*counter_collection.csv" | head -1) $M $dma >> $out <<'PY'
import csv, sys, collections
stats, pmc, M, dma = sys.argv[1:5]
us = {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(stats)) if "gemm_skinny" in r["Name"]}
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(pmc)):
    if "gemm_skinny" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        a = agg[r["Kernel_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (calls, t) in sorted(us.items()):
    n, kib = agg.get(k, (0, 0.0))
    print("M=%s dma=%s  %5d launches  %6.2f us  FETCH_SIZE %8.1f KiB raw/launch (x2 on gfx950 = %6.2f MB)  %s" % (
        M, dma, calls, t, kib / max(n, 1), 2 * kib / max(n, 1) / 1024, k[:90]))
PY
done; done
cat $out
