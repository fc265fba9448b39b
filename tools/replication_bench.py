"""The two decode products whose A operand every column block re-reads, launched eagerly in
a dependent chain so that rocprofv3 (kernel trace for microseconds, --pmc FETCH_SIZE for bytes) sees them one by one:
    x [M, 512] -> LayerNorm-folded [M, 1536] (N = 1536: QKV)  -> [M, 512] (K = 1536: the backward's dX)
usage: python tools/replication_bench.py [M] [launches]      (MSSR_SKINNY_DMA=0: the round-2 register-fragment kernel)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssrnet_style_transfer_amd import ops  # noqa: E402
M = int(sys.argv[1]) if len(sys.argv) > 1 else 240
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda")
ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
w1 = [(torch.randn(1536, 512, device=dev) / 512 ** 0.5).bfloat16() for _ in range(4)]
w2 = [(torch.randn(512, 1536, device=dev) / 1536 ** 0.5).bfloat16() for _ in range(4)]
x = torch.randn(M, 512, device=dev).bfloat16()
h = torch.empty(M, 1536, device=dev, dtype=torch.bfloat16)
for i in range(N):
    ops.gemm(x, w1[i % 4], True, torch.bfloat16, out=h)
    ops.gemm(h, w2[i % 4], True, torch.bfloat16, out=x)
torch.cuda.synchronize()
print("done: %d launches of %dx1536x512 and %dx512x1536" % (N, M, M))
