"""Dependent chains of the decode sweep's K-split GEMMs as hipGraph replays: microseconds per launch by shape and
variant (LayerNorm prologue on / off, K = 512 / 1024 / 1536).  MSSR_SKINNY_W8=0/1 selects 4 / 8 waves per block for
K >= 1024.   python tools/skinny_bench.py [M]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from mssrnet_style_transfer_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 160
dev = torch.device("cuda")
ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)


def weights(n, k, count=4):
    return [(torch.randn(n, k, device=dev) / k ** 0.5).bfloat16() for _ in range(count)]


def chain(dims, ln=False, n_launch=240):
    """dims: list of widths d0 -> d1 -> ... -> d0 (cycle); every product reads the previous one's output"""
    ws = [weights(dims[(i + 1) % len(dims)], dims[i]) for i in range(len(dims))]
    bufs = [torch.randn(M, d, device=dev).bfloat16() for d in dims]
    cs = [[w.float().sum(-1).contiguous() for w in wl] for wl in ws]
    stats = torch.empty(2, M, device=dev)

    def run():
        for i in range(n_launch):
            j = i % len(dims)
            w = ws[j][(i // len(dims)) % 4]
            out = bufs[(j + 1) % len(dims)]
            if ln:
                ops.gemm(bufs[j], w, True, torch.bfloat16, out=out, ln=(cs[j][(i // len(dims)) % 4], stats[0], stats[1], 1e-6))
            else:
                ops.gemm(bufs[j], w, True, torch.bfloat16, out=out)
    return run, n_launch


def graphed(fn):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    return g.replay


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


import os
print("M=%d  MSSR_SKINNY_W8=%s" % (M, os.environ.get("MSSR_SKINNY_W8", "1")))
for name, dims, ln in (("512->512", [512], False), ("512->512 LN", [512], True), ("512->1536->512", [512, 1536], False),
                       ("512->1536(LN)->512(LN)", [512, 1536], True), ("512->1024->512", [512, 1024], False),
                       ("512->1024(LN)->512(LN)", [512, 1024], True)):
    run, n = chain(dims, ln)
    t = timeit(graphed(run))
    print("%-28s %.2f us/launch" % (name, t / n))
