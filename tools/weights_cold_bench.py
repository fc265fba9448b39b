"""Is a decode-chain GEMM slowed by where its WEIGHTS come from?  A chain of dependent M x 512 x 512 K-split GEMMs
(each launch reads the previous launch's output) replayed as a hipGraph (no host in the loop), cycling through W
distinct 512 x 512 bf16 weight matrices: W = 8 (4 MB, stays in the XCD L2s) ... W = 320 (160 MB: every launch's
weights come from the Infinity Cache / HBM, as in a decode sweep over 12 layers + vocabulary matrices).
usage: python tools/weights_cold_bench.py [M]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from mssrnet_style_transfer_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 160
N = 640
dev = torch.device("cuda")
ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


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


print("M=%d, %d dependent launches, graph replay" % (M, N))
for W in (8, 16, 32, 64, 160, 320):
    w = [(torch.randn(512, 512, device=dev) / 512 ** 0.5).bfloat16() for _ in range(W)]
    a = torch.randn(M, 512, device=dev).bfloat16()
    b = torch.empty_like(a)

    def chain():
        x, y = a, b
        for i in range(N):
            ops.gemm(x, w[i % W], True, torch.bfloat16, out=y)
            x, y = y, x

    t = timeit(graphed(chain))
    print("W = %3d distinct weights (%5.1f MB): %.3f ms  (%.2f us/launch)" % (W, W * 0.5, t, t * 1e3 / N))
