"""Vocabulary-sized products of the decode sweep (generator: M x 9606 x 512; its input gradient M x 512 x 9632) on the
K-split kernel vs the 128x128 LDS-DMA kernel (MSSR_PIPE_FIRST_MINM=128 sends M >= 128 there first)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssrnet_style_transfer_amd import ops  # noqa: E402


def timed(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ops.set_compute_dtype(torch.bfloat16)
print("MSSR_PIPE_FIRST_MINM =", os.environ.get("MSSR_PIPE_FIRST_MINM", "2048"))
V, Vp, d = 9606, 9632, 512
W = torch.randn(V, d, device="cuda").bfloat16()
Wt = torch.zeros(d, Vp, device="cuda").bfloat16()
for M in (160, 240, 512):
    x = torch.randn(M, d, device="cuda").bfloat16()
    out16 = torch.empty(M, Vp, device="cuda").bfloat16()[:, :V]
    out32 = torch.empty(M, V, device="cuda")
    dl = torch.randn(M, Vp, device="cuda").bfloat16()
    dx = torch.empty(M, d, device="cuda").bfloat16()
    t1 = timed(lambda: ops.gemm(x, W, True, torch.bfloat16, out=out16))
    t2 = timed(lambda: ops.gemm(x, W, True, torch.float32, out=out32))
    t3 = timed(lambda: ops.gemm(dl, Wt, True, torch.bfloat16, out=dx))
    print("M=%d: x W^T -> bf16 %.1f us, -> f32 %.1f us; dlogits W (K = 9632) %.1f us" % (M, t1, t2, t3))
