"""Per-dispatch durations of the step-attention kernels from a rocprofv3 kernel trace (csv), in launch order:
   python tools/attn_trace.py <kernel_trace.csv> <out.txt>   (TRACE=1 bash tools/prof.sh <tag> ... leaves /tmp/trace_<tag>.csv);
tools/attn_trace_show.py prints the per-decode-step means -> profiles/r03_long_attention_trace.txt."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    rd = csv.DictReader(f)
    for r in rd:
        n = r["Kernel_Name"]
        if "attn_step" in n:
            kind = ("f" if "fwd" in n else "b") + ("8" if "Li8E" in n or ", 8>" in n else "4")
            rows.append((int(r["Start_Timestamp"]), kind, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size", ""))))
rows.sort()
with open(sys.argv[2], "w") as o:
    for t, k, d, g in rows:
        o.write(f"{k} {d} {g}\n")
