#!/bin/bash
# usage: tools/pmc.sh <counter> <tag> [bench args, e.g. --workload long]  -- one counter pass over a short bench run;
# per-kernel means to gpurun_out/pmc_<tag>.csv
ctr=$1; tag=$2; shift 2
cd /tmp && export TMPDIR=/tmp
# counter collection serialises every dispatch (minutes for a 7 000-kernel step) and prints nothing meanwhile: a
# heartbeat file tells the harness the run is alive
( while sleep 30; do date >> $GRAFT_REPO_ROOT/gpurun_out/pmc_heartbeat.log; done ) &
hb=$!
trap 'kill $hb 2>/dev/null' EXIT
timeout -s KILL 900 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc_$tag -o $tag -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
rc=$?
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
ls -la /tmp/pmc_$tag/* | head -5
head -3 "$f"
python3 - "$f" $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.csv <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        k = (r["Kernel_Name"][:100], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
with open(sys.argv[2], "w") as out:
    out.write("kernel,counter,dispatches,total,mean_per_dispatch\n")
    for (k, c), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.write('"%s",%s,%d,%.1f,%.3f\n' % (k, c, n, t, t / n))
PY
exit $rc
