"""Fused attention forward / backward timing: matrix-core kernels (MSSR_ATTN_MFMA=1, default) vs the scalar
kernels (MSSR_ATTN_MFMA=0, L <= 64 only).  python tools/attn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mssrnet_style_transfer_amd import ops  # noqa: E402


def timed(fn, n=50):
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


def main():
    ops.set_compute_dtype(torch.bfloat16)
    print("MSSR_ATTN_MFMA =", os.environ.get("MSSR_ATTN_MFMA", "1"))
    for B, H, L, dh, p in [(240, 8, 18, 64, 0.1), (80, 8, 18, 64, 0.1), (240, 8, 33, 64, 0.1), (256, 8, 64, 64, 0.1),
                           (256, 8, 130, 64, 0.1), (256, 8, 130, 64, 0.0), (512, 8, 160, 64, 0.1)]:
        from mssrnet_style_transfer_amd import _lib
        if not _lib.load().mssr_attention_fits(L, L, dh, _lib.BF16, 1):
            print("B%d H%d L%d: beyond the kernels" % (B, H, L))
            continue
        qkv = torch.randn(B, L, 3 * H * dh, device="cuda").bfloat16()
        q, k, v = (qkv[..., i * H * dh:(i + 1) * H * dh].view(B, L, H, dh).transpose(1, 2).requires_grad_(True) for i in range(3))
        lens = torch.randint(L // 2, L + 1, (B,), device="cuda")
        mask = (torch.arange(L, device="cuda").unsqueeze(0) >= lens.unsqueeze(1)).unsqueeze(1)
        out = [None]

        def fwd():
            out[0] = ops.attention(q, k, v, mask, p, True, want_probs=False)[0]

        w = torch.randn(B, H, L, dh, device="cuda").bfloat16()

        def both():
            fwd()
            torch.autograd.grad(out[0], (q, k, v), w)

        tf = timed(fwd)
        tb = timed(both) - tf
        fl = 4.0 * B * H * L * L * dh
        print("B%d H%d L%d dh%d p%.1f: fwd %.1f us (%.1f TFLOP/s)  bwd %.1f us (%.1f TFLOP/s, incl. host glue)" % (
            B, H, L, dh, p, tf, fl / tf * 1e-6, tb, 2.5 * fl / tb * 1e-6))


if __name__ == "__main__":
    main()
