#!/usr/bin/env python
"""profiles/rNN_pmc_fetch_size.csv + rNN_pmc_write_size.csv  ->  profiles/rNN_pmc_traffic.json

The two inputs are the per-kernel summaries tools/pmc.sh writes from two separate `rocprofv3 --pmc` passes of
`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline [--workload W]` (FETCH_SIZE and WRITE_SIZE do
not fit one pass: 3 + 2 of the 4 TCC counters).  Both counters are in KiB.  On gfx950 FETCH_SIZE tallies a 128-byte
request as 64 bytes, so MI355X_MICROARCH.md prescribes doubling it before comparing with a byte count:

    traffic_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / dispatches

bench.py's kernel names (ops.stop_profile) group the instantiations rocprofv3 lists separately (the G template
parameter = rows per block is not part of the bench name); the groups are summed dispatch-weighted.

    python tools/pmc_traffic.py wgan profiles/r02_pmc_fetch_size.csv profiles/r02_pmc_write_size.csv profiles/r02_pmc_traffic.json [sha16]

(sha16: the source hash recorded on the box when the counters were collected, tools/collect_profiles.sh.)
"""
import csv
import json
import os
import re
import sys

def _tile(fm, fn):
    return "%dx%d" % (16 * int(fm), 16 * int(fn))


_LN = {"0": "noLN", "1": "LN", "false": "noLN", "true": "LN"}

RULES = [
    # rocprofv3 leaves names with a __hip_bfloat16 template argument mangled (DF16b) unless LN = true, which it demangles,
    # wrongly, as "<bool _Accum, bool, E, ...>".  K-split GEMM family: <TO, LN, FM, FN, group | k-blocks per wave[, VEC]>
    (re.compile(r"gemm_skinny_dma_kernelIDF16bLb([01])ELi(\d)ELi(\d)ELi\d+ELb[01]E"),
     lambda m: "gemm_skinny_dma_kernel<bf16,%s,%s>" % (_LN[m.group(1)], _tile(m.group(2), m.group(3)))),
    (re.compile(r"gemm_skinny_dma_kernel<bool _Accum, bool, E, (\d), (\d), \d+, (?:true|false)>"),
     lambda m: "gemm_skinny_dma_kernel<bf16,LN,%s>" % _tile(m.group(1), m.group(2))),
    (re.compile(r"gemm_skinny_dma_kernel<float, (true|false), (\d), (\d), \d+, (?:true|false)>"),
     lambda m: "gemm_skinny_dma_kernel<f32,%s,%s>" % (_LN[m.group(1)], _tile(m.group(2), m.group(3)))),
    (re.compile(r"gemm_skinny_kernelIDF16bLb([01])ELi(\d)ELi(\d)ELi\d+EEE"),
     lambda m: "gemm_skinny_kernel<bf16,%s,%s>" % (_LN[m.group(1)], _tile(m.group(2), m.group(3)))),
    (re.compile(r"gemm_skinny_kernel<bool _Accum, bool, E, (\d), (\d), \d+>"),
     lambda m: "gemm_skinny_kernel<bf16,LN,%s>" % _tile(m.group(1), m.group(2))),
    (re.compile(r"gemm_skinny_kernel<float, (true|false), (\d), (\d), \d+>"),
     lambda m: "gemm_skinny_kernel<f32,%s,%s>" % (_LN[m.group(1)], _tile(m.group(2), m.group(3)))),
    (re.compile(r"gemm_skinny_f32_kernel<\d+>"), lambda m: "gemm_skinny_f32_kernel"),
    (re.compile(r"gemm_lnb_kernelIDF16bLi\d+E"), lambda m: "gemm_lnb_kernel<bf16>"),
    (re.compile(r"gemm_lnb_kernel<float, \d+>"), lambda m: "gemm_lnb_kernel<f32>"),
    (re.compile(r"gemm_pipe_kernel<float,"), lambda m: "gemm_pipe_kernel<f32>"),
    (re.compile(r"gemm_pipe_kernel(?:IDF16b|<bool _Accum)"), lambda m: "gemm_pipe_kernel<bf16>"),
    (re.compile(r"gemm_wide_kernelIDF16bLi\d+ELb([01])E"), lambda m: "gemm_wide_kernel<bf16,%s>" % _LN[m.group(1)]),
    (re.compile(r"gemm_wide_kernel<float, \d+, (true|false)>"), lambda m: "gemm_wide_kernel<f32,%s>" % _LN[m.group(1)]),
    (re.compile(r"gemm_tn_kernel<float, \d+>"), lambda m: "gemm_tn_kernel<f32>"),
    (re.compile(r"gemm_tn_kernelIDF16bLi\d+E"), lambda m: "gemm_tn_kernel<bf16>"),
    # not GEMMs, but the kernels config 5 spends 40 % of its time in (their HBM rate = this traffic / rocprofv3 duration)
    (re.compile(r"attn_step_(fwd|bwd)_tiled_kernelIDF16bLi\d+ELi(\d+)E"),
     lambda m: "attn_step_%s_tiled_kernel<bf16,%s>" % (m.group(1), m.group(2))),
    (re.compile(r"attn_step_bwd_stream_kernelIDF16bLi\d+ELi(\d+)E"),
     lambda m: "attn_step_bwd_stream_kernel<bf16,%s>" % m.group(1)),
    (re.compile(r"gemm_kernelIDF16bDF16bLi(\d+)ELi\d+E"), lambda m: "gemm_kernel<bf16,bf16,%s>" % m.group(1)),
    (re.compile(r"gemm_kernelIDF16bfLi(\d+)ELi\d+E"), lambda m: "gemm_kernel<bf16,f32,%s>" % m.group(1)),
    (re.compile(r"gemm_kernel<float, float, (\d+), \d+>"), lambda m: "gemm_kernel<f32,f32,%s>" % m.group(1)),
]


def bench_name(kernel):
    for rx, fmt in RULES:
        m = rx.search(kernel)
        if m:
            return fmt(m)
    return None


def read(path, counter):
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["counter"] != counter:
                continue
            name = bench_name(r["kernel"])
            if name is None:
                continue
            n, tot = out.get(name, (0, 0.0))
            out[name] = (n + int(r["dispatches"]), tot + float(r["total"]))
    return out


def source_sha16(root):
    """Hash of everything that decides which kernels a bench step launches and what they do (kernel sources, the
    C ABI header, the Python host path, bench.py): bench.py recomputes it and flags `traffic` as stale when the tree
    it runs from is not the tree these counters were collected on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "mssrnet_style_transfer_amd", "csrc", "*.h*")) +
                   glob.glob(os.path.join(root, "mssrnet_style_transfer_amd", "**", "*.py"), recursive=True) +
                   [os.path.join(root, "include", "mssr_hip.h"), os.path.join(root, "bench.py")])
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    workload, fetch_csv, write_csv, out_json = sys.argv[1:5]
    sha = sys.argv[5] if len(sys.argv) > 5 else None      # recorded on the box by the stage that collected the counters
    fetch, write = read(fetch_csv, "FETCH_SIZE"), read(write_csv, "WRITE_SIZE")
    kernels = {}
    for name, (n, f_kib) in sorted(fetch.items()):
        wn, w_kib = write.get(name, (0, 0.0))
        if wn != n:
            print("skip %s: %d dispatches in the fetch pass, %d in the write pass" % (name, n, wn))
            continue
        kernels[name] = dict(
            dispatches=n,
            fetch_size_raw_kib_per_launch=round(f_kib / n, 2),
            fetch_bytes_per_launch_gfx950_corrected=int(2 * f_kib * 1024 / n),
            write_bytes_per_launch=int(w_kib * 1024 / n),
            traffic_bytes_per_launch=int((2 * f_kib + w_kib) * 1024 / n))
    doc = {}
    if os.path.exists(out_json):
        with open(out_json) as f:
            doc = json.load(f)
    doc.setdefault("note", "HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB, two separate rocprofv3 --pmc passes "
                           "(tools/pmc.sh), gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md applied; grouped by "
                           "bench.py kernel name (tools/pmc_traffic.py)")
    doc.setdefault("workloads", {})[workload] = kernels
    doc.setdefault("sources", {})[workload] = [os.path.basename(fetch_csv), os.path.basename(write_csv)]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the counters must be parsed from the tree they were collected on (collect, then `copy`, before editing anything)
    doc.setdefault("source_sha16", {})[workload] = sha or source_sha16(root)
    with open(out_json, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    for k, v in kernels.items():
        print("%-44s %6d launches  %9d B traffic" % (k, v["dispatches"], v["traffic_bytes_per_launch"]))


if __name__ == "__main__":
    main()
