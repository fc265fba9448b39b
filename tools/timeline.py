"""Summarise a rocprofv3 kernel trace of bench.py: per-kernel totals for ONE step (the last one) and the sequence of
kernels in issue order, run-length compressed, with the time of every run -- shows which phase spends what where.
usage: python tools/timeline.py <kernel_trace.csv> <steps_in_trace> [out.txt]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"mssr::", "", name)
    name = re.sub(r"at::native::\(anonymous namespace\)::", "at::", name)
    name = re.sub(r"at::native::", "at::", name)
    return name[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = int(sys.argv[2])
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(rows) // steps
    last = rows[-n:]
    out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
    t0, t1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
    out.write("last step: %d kernels, span %.2f ms, busy %.2f ms\n" % (n, (t1 - t0) / 1e6, busy / 1e6))
    runs = []
    for r in last:
        k = short(r["Kernel_Name"])
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if runs and runs[-1][0] == k:
            runs[-1][1] += 1
            runs[-1][2] += d
        else:
            runs.append([k, 1, d, int(r["Start_Timestamp"]) - t0])
    for k, c, d, at in runs:
        out.write("%9.3f ms  %4d x %-70s %8.1f us\n" % (at / 1e6, c, k, d / 1e3))


if __name__ == "__main__":
    main()
