"""gemm_pipe_kernel at the encoders' medium shapes (3 840 - 5 440 rows): correctness against fp32 torch and microseconds
per launch from the library's own dispatch events.  MSSR_PIPE_SMALL_BELOW=0 disables the 64 x 64 tiles.
usage: python tools/pipe_mid_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssrnet_style_transfer_amd import ops
torch.manual_seed(0)
cases = [("nt", 3840, 512, 512), ("nt", 3840, 1024, 512), ("nt", 3840, 512, 1024), ("nt", 3840, 512, 1536),
         ("nt", 3840, 1536, 512), ("nt", 4320, 256, 256), ("nt", 5440, 256, 1024), ("nt", 5440, 1024, 256),
         ("nt", 4080, 1024, 256), ("nt", 2560, 512, 512), ("nt", 32768, 512, 512), ("nt", 32768, 512, 1024),
         ("ntb", 3840, 1024, 512), ("ntr", 3840, 512, 1024), ("ntf", 3840, 512, 512), ("nt", 3800, 520, 544),
         ("ntr", 2100, 264, 96)]
print("MSSR_PIPE_SMALL_BELOW=%s" % os.environ.get("MSSR_PIPE_SMALL_BELOW", "default"))
for form, m, n, k in cases:
    a = torch.randn(m, k, device="cuda").bfloat16(); b = (torch.randn(n, k, device="cuda") / k ** 0.5).bfloat16()
    bias = torch.randn(n, device="cuda"); res = torch.randn(m, n, device="cuda").bfloat16()
    if form == "nt":
        run = lambda: ops.gemm(a, b, True, torch.bfloat16)
        ref = lambda: a.float() @ b.float().t()
    elif form == "ntb":
        run = lambda: ops.gemm(a, b, True, torch.bfloat16, bias=bias, relu=True)
        ref = lambda: torch.relu(a.float() @ b.float().t() + bias)
    elif form == "ntr":
        run = lambda: ops.gemm(a, b, True, torch.bfloat16, bias=bias, residual=res, gate=res)
        ref = lambda: torch.where(res.float() > 0, a.float() @ b.float().t() + bias + res.float(), torch.zeros((), device="cuda"))
    else:
        run = lambda: ops.gemm(a, b, True, torch.float32, bias=bias, alpha=0.5)
        ref = lambda: 0.5 * (a.float() @ b.float().t()) + bias
    out, r = run(), ref()
    err = float((out.float() - r).abs().max() / r.abs().max())
    assert err < (1e-2 if out.dtype == torch.bfloat16 else 1e-4), (form, m, n, k, err)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ops.start_profile()
    for _ in range(20): run()
    rec = ops.stop_profile()
    us = sorted(x["ms"] for x in rec)[len(rec) // 2] * 1e3
    print("%-4s %6d x %5d x %5d  %-34s %7.1f us  %7.1f TFLOP/s  rel err %.1e" % (form, m, n, k, rec[0]["kernel"], us, 2e-6 * m * n * k / us, err))
