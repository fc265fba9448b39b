import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 4
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = sum(int(r['Calls']) for r in rows)
print("GPU busy per step %.1f ms, kernels per step %.0f" % (tot/1e6/steps, calls/steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    print("%-80s %7.0f/step %8.2fms/step avg %7.2fus %5.1f%%" % (r['Name'][:80], int(r['Calls'])/steps, float(r['TotalDurationNs'])/1e6/steps, float(r['AverageNs'])/1e3, float(r['Percentage'])))
