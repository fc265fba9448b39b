"""Where a K-step of gemm_pipe_kernel's 2-buffer loop spends its cycles, as wave 0 of every block sees it (needs the
diagnostic build: make -C mssrnet_style_transfer_amd/csrc clean all STAMPS=1).  usage: python tools/pipe_stamp_bench.py M N K"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssrnet_style_transfer_amd import _lib as L, ops
M, N, K = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda")
ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
lib = L.load()
a = torch.randn(M, K, device=dev).bfloat16()
w = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(4)]
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
stamps = torch.zeros(16384, 8, dtype=torch.int64, device=dev)
for rep in range(3):
    stamps.zero_()
    for i in range(12):
        if i == 8:
            lib.mssr_debug_stamps(stamps.data_ptr())
        ops.gemm(a, w[i % 4], True, torch.bfloat16, out=y)
        if i == 8:
            lib.mssr_debug_stamps(None)
    torch.cuda.synchronize()
s = stamps[stamps[:, 0] > 0].cpu().double()
steps = (K + 63) // 64
t0 = s[:, 0].min()
print("%d x %d x %d: %d blocks, %d K-steps; s_memtime ticks" % (M, N, K, s.shape[0], steps))
print("  block start after the first block's: median %.0f  max %.0f" % (float((s[:, 0] - t0).median()), float((s[:, 0] - t0).max())))
print("  prologue (first stage issued, landed, barrier): median %.0f" % float((s[:, 1] - s[:, 0]).median()))
for i, n in ((2, "issue next stage's DMA"), (3, "fragment reads + MFMA issue"), (4, "s_waitcnt vmcnt(0)"), (5, "block barrier")):
    print("  per K-step %-28s median %7.1f   max %7.1f" % (n, float(s[:, i].median()) / steps, float(s[:, i].max()) / steps))
print("  loop total (start -> last barrier): median %.0f  = %.1f per K-step" % (float((s[:, 6] - s[:, 0]).median()), float((s[:, 6] - s[:, 0]).median()) / steps))
print("  last block's loop end after the first block's start: %.0f" % float((s[:, 6] - t0).max()))
