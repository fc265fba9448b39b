"""LayerNorm backward (dx + parameter gradients over all rows) and column sums: kernel time by row count.
MSSR_LN_PARTS caps the row slices of the LayerNorm kernels.  python tools/ln_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssrnet_style_transfer_amd import _lib as L  # noqa: E402
from mssrnet_style_transfer_amd.decode_engine import _bsum  # noqa: E402


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


lib = L.load()
print("MSSR_LN_PARTS =", os.environ.get("MSSR_LN_PARTS", "default"))
d = 512
for rows in (1360, 4480, 71680):
    x = torch.randn(rows, d, device="cuda").bfloat16()
    dy = torch.randn(rows, d, device="cuda").bfloat16()
    g = torch.randn(d, device="cuda")
    mean, rstd = torch.randn(rows, device="cuda"), torch.rand(rows, device="cuda") + 0.5
    dx = torch.empty_like(x)
    dg, db = torch.empty(d, device="cuda"), torch.empty(d, device="cuda")
    parts = lib.mssr_layernorm_partial_rows(rows)
    ws = torch.empty(2 * parts * d, device="cuda")

    def f():
        L.check(lib.mssr_layernorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                       dg.data_ptr(), db.data_ptr(), ws.data_ptr(), rows, d, L.BF16, L.stream()), "ln_bwd")
    t = timed(f)
    print("ln_bwd %6d x %d: %.1f us (%.2f TB/s)  parts %d" % (rows, d, t, rows * d * 6 / t * 1e-6, parts))
for rows, n, nl in ((4480, 1536, 12), (4480, 512, 12), (1360, 1024, 12), (71680, 1536, 12)):
    xb = torch.randn(nl, rows, n, device="cuda").bfloat16()
    t = timed(lambda: _bsum(xb))
    print("colsum %d x %6d x %d: %.1f us (%.2f TB/s)" % (nl, rows, n, t, nl * rows * n * 2 / t * 1e-6))
