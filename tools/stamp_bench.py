"""Where a K-split GEMM launch of the decode sweep spends its time: the blocks of ONE launch inside a dependent chain
record the 100 MHz real-time clock at six points (mssr_debug_stamps); printed as microseconds after the first block's
start (median and max over blocks).   usage: python tools/stamp_bench.py M N K [ln]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssrnet_style_transfer_amd import _lib as L, ops
M, N, K = (int(v) for v in sys.argv[1:4])
ln = len(sys.argv) > 4
dev = torch.device("cuda")
ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
lib = L.load()
# a chain x -> [N] -> back to [K] ... : every launch reads the previous one's output; weights differ per link (cold)
w1 = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(8)]
w2 = [(torch.randn(K, N, device=dev) / N ** 0.5).bfloat16() for _ in range(8)]
cs = [w.float().sum(-1).contiguous() for w in w1]
x = torch.randn(M, K, device=dev).bfloat16()
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
stats = torch.empty(2, M, device=dev)
stamps = torch.zeros(8192, 8, dtype=torch.int64, device=dev)


def link(i, stamped=False):
    if stamped:
        lib.mssr_debug_stamps(stamps.data_ptr())
    if ln:
        ops.gemm(x, w1[i % 8], True, torch.bfloat16, out=y, ln=(cs[i % 8], stats[0], stats[1], 1e-6))
    else:
        ops.gemm(x, w1[i % 8], True, torch.bfloat16, out=y)
    if stamped:
        lib.mssr_debug_stamps(None)
    ops.gemm(y, w2[i % 8], True, torch.bfloat16, out=x)


for rep in range(3):
    stamps.zero_()
    for i in range(24):
        link(i, stamped=(i == 16))
    torch.cuda.synchronize()
s = stamps[stamps[:, 0] > 0].cpu().double()
t0 = s[:, 0].min()
names = ["start", "loads issued", "loads landed", "tile in LDS", "after barrier", "end"]
print("%d x %d x %d%s: %d blocks; us after the first block's start (median / max over blocks)" % (M, N, K, " LN" if ln else "", s.shape[0]))
for i, n in enumerate(names):
    c = (s[:, i] - t0) / 100.0
    print("  %-14s median %6.2f   max %6.2f   min %6.2f" % (n, float(c.median()), float(c.max()), float(c.min())))
