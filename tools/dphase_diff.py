"""Per-call kernel list from two rocprofv3 kernel-stats files of the same program at N_a < N_b repetitions.
usage: python tools/dphase_diff.py stats_a.csv N_a stats_b.csv N_b [rows]"""
import csv, sys
a = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
na, nb = int(sys.argv[2]), int(sys.argv[4])
rows = []
for r in csv.DictReader(open(sys.argv[3])):
    o = a.get(r["Name"])
    c = (int(r["Calls"]) - (int(o["Calls"]) if o else 0)) / (nb - na)
    t = (float(r["TotalDurationNs"]) - (float(o["TotalDurationNs"]) if o else 0.0)) / (nb - na)
    if c > 0.01:
        rows.append((t, c, r["Name"]))
rows.sort(reverse=True)
ours = lambda n: "mssr" in n
print("per call: %.0f kernels, %.3f ms busy; not ours: %.0f kernels, %.3f ms" % (
    sum(r[1] for r in rows), sum(r[0] for r in rows) / 1e6, sum(r[1] for r in rows if not ours(r[2])),
    sum(r[0] for r in rows if not ours(r[2])) / 1e6))
for t, c, n in rows[:int(sys.argv[5]) if len(sys.argv) > 5 else 60]:
    print("%7.1f/call %8.3f ms/call avg %7.2f us  %s" % (c, t / 1e6, t / 1e3 / c, n[:110]))
