"""The allocation-free entry point mssr_gemm_noalloc (include/mssr_hip.h) against mssr_gemm, through the C ABI.

It has no counterpart in the reference: it serves callers that cannot call hipMalloc.  Its split-K slabs live in a
static device array; mssr_gemm takes them from a caller workspace.  Bar: bit-identical outputs (same kernels, same
summation order), on two calls in a row, the second with a smaller input (the array keeps the first call's slabs).
Collected last (test_zz_*): the round's end runs pytest -x."""
import ctypes

import pytest
import torch

from mssrnet_style_transfer_amd import _lib as L

pytestmark = pytest.mark.gpu


def desc(a, w, c, bias=None, residual=None, relu=False, workspace=None):
    d = L.GemmDesc()
    m, k = a.shape
    n = w.shape[0]
    d.a, d.b, d.c = a.data_ptr(), w.data_ptr(), c.data_ptr()
    d.bias, d.residual, d.gate = L.ptr(bias), L.ptr(residual), None
    d.m, d.n, d.k = m, n, k
    d.a_rs, d.a_cs, d.b_rs, d.b_cs = a.stride(0), 1, 1, w.stride(0)       # C = A . W^T, both k-contiguous
    d.ldc, d.ldr, d.ldg = c.stride(0), (residual.stride(0) if residual is not None else n), n
    d.batch_outer = d.batch_inner = 1
    d.alpha, d.beta, d.relu = 1.0, 0.0, int(relu)
    d.in_dtype, d.out_dtype = L.dt(a), L.dt(c)
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    return d


def run_pair(lib, m, n, k, seed, out_dtype=torch.bfloat16, epilogue=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(m, k, device="cuda", generator=g).bfloat16()
    w = (torch.randn(n, k, device="cuda", generator=g) / k ** 0.5).bfloat16()
    bias = torch.randn(n, device="cuda", generator=g) if epilogue else None
    res = torch.randn(m, n, device="cuda", generator=g).bfloat16() if epilogue else None
    ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    c_ref = torch.full((m, n), float("nan"), device="cuda", dtype=out_dtype)
    c_new = torch.full((m, n), float("nan"), device="cuda", dtype=out_dtype)
    d_ref = desc(a, w, c_ref, bias, res, epilogue, ws)
    assert lib.mssr_gemm_is_skinny(ctypes.byref(d_ref)) == 1          # the split-K path: the one that uses scratch
    L.check(lib.mssr_gemm(ctypes.byref(d_ref), L.stream()), "gemm")
    d_new = desc(a, w, c_new, bias, res, epilogue, None)
    before = torch.cuda.memory_allocated()
    L.check(lib.mssr_gemm_noalloc(ctypes.byref(d_new), L.stream()), "gemm_noalloc")
    assert torch.cuda.memory_allocated() == before
    torch.cuda.synchronize()
    assert torch.isfinite(c_new.float()).all()
    assert torch.equal(c_ref, c_new), float((c_ref.float() - c_new.float()).abs().max())
    # and both agree with a plain fp32 product of the same bf16 operands
    want = a.float() @ w.float().t()
    if epilogue:
        want = torch.relu(want + bias) + res.float()
    tol = 2e-2 if out_dtype == torch.bfloat16 else 1e-3
    assert float((c_new.float() - want).abs().max()) <= tol * float(want.abs().max())


def test_noalloc_gemm_matches_gemm_on_two_calls_in_a_row():
    lib = L.load()
    assert lib.mssr_gemm_noalloc_scratch_bytes() == 16 << 20
    run_pair(lib, 160, 512, 9632, seed=1)                      # the decode engine's soft-embedding product: 10 slabs
    run_pair(lib, 80, 256, 4096, seed=2, epilogue=True)        # smaller, right after: stale slabs must not leak in
    run_pair(lib, 1440, 256, 9632, seed=3, out_dtype=torch.float32)    # the critics' soft embedding, fp32 result
    run_pair(lib, 16, 32, 2080, seed=4)                        # smallest split-K case


def test_noalloc_gemm_without_slabs_and_limits():
    lib = L.load()
    # k <= 2048: no slabs needed, same kernels as mssr_gemm
    g = torch.Generator(device="cuda").manual_seed(9)
    a = torch.randn(160, 512, device="cuda", generator=g).bfloat16()
    w = torch.randn(1536, 512, device="cuda", generator=g).bfloat16()
    c1, c2 = torch.empty(160, 1536, device="cuda", dtype=torch.bfloat16), torch.empty(160, 1536, device="cuda",
                                                                                     dtype=torch.bfloat16)
    L.check(lib.mssr_gemm(ctypes.byref(desc(a, w, c1)), L.stream()), "gemm")
    L.check(lib.mssr_gemm_noalloc(ctypes.byref(desc(a, w, c2)), L.stream()), "gemm_noalloc")
    assert torch.equal(c1, c2)
    # many rows, long K: mssr_gemm gives this to the LDS-DMA kernel (no scratch at all), so the static array's size
    # is no limit for it -- same dispatch, same bits
    a = torch.randn(4096, 9632, device="cuda", generator=g).bfloat16()
    w = (torch.randn(512, 9632, device="cuda", generator=g) / 98.0).bfloat16()
    c1 = torch.empty(4096, 512, device="cuda", dtype=torch.bfloat16)
    c2 = torch.full((4096, 512), float("nan"), device="cuda", dtype=torch.bfloat16)
    assert lib.mssr_gemm_is_skinny(ctypes.byref(desc(a, w, c1))) == 0
    L.check(lib.mssr_gemm(ctypes.byref(desc(a, w, c1)), L.stream()), "gemm")
    L.check(lib.mssr_gemm_noalloc(ctypes.byref(desc(a, w, c2)), L.stream()), "gemm_noalloc")
    assert torch.equal(c1, c2)
    # a product the K-split kernel WOULD take, whose 10 slabs (20 MB) are beyond the static array: rejected with a
    # message, nothing launched
    a = torch.zeros(1024, 9632, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(512, 9632, device="cuda", dtype=torch.bfloat16)
    c = torch.full((1024, 512), 7.0, device="cuda", dtype=torch.bfloat16)
    ws = torch.empty(32 << 20, dtype=torch.uint8, device="cuda")
    assert lib.mssr_gemm_is_skinny(ctypes.byref(desc(a, w, c, workspace=ws))) == 1
    rc = lib.mssr_gemm_noalloc(ctypes.byref(desc(a, w, c)), L.stream())
    assert rc == 1 and b"static scratch" in lib.mssr_last_error()
    torch.cuda.synchronize()
    assert bool((c == 7.0).all())
