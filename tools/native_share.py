"""Share of kernels that are not this library's (at::native elementwise / reduce / cat, rocclr copies and fills,
hipBLASLt) in a rocprofv3 kernel-stats CSV: launches and GPU time per step.
usage: python tools/native_share.py <kernel_stats.csv> <steps_in_trace>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
nat = [r for r in rows if "mssr" not in r["Name"]]
tn = sum(float(r["TotalDurationNs"]) for r in nat)
cn = sum(int(r["Calls"]) for r in nat)
print("per step: %.0f kernels, %.2f ms busy; not ours: %.0f kernels (%.1f %%), %.2f ms (%.1f %% of the GPU time)" % (
    calls / steps, tot / 1e6 / steps, cn / steps, 100.0 * cn / calls, tn / 1e6 / steps, 100.0 * tn / tot))
for r in nat[:25]:
    print("%6.0f/step %7.3f ms/step  avg %6.2f us  %s" % (int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps,
                                                        float(r["AverageNs"]) / 1e3, r["Name"][:120]))
