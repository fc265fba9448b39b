"""Differentiable operators over the C-ABI kernels (include/mssr_hip.h).

Every operator is a ``torch.autograd.Function`` whose backward is itself built from these
operators, so gradients of gradients exist wherever the WGAN-GP critics need them
(``autograd.grad(..., create_graph=True)``, onmt/trainer_wgan_wh.py:1247-1250,1315-1318):
matmul and the masked/gated elementwise ops are closed under differentiation; LayerNorm and
softmax carry hand-written second-order kernels.  torch is used for memory, views and
autograd book-keeping only; all arithmetic on activations runs in libmssr_hip.so.
"""
import math
import threading
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L

_state = {"dtype": torch.float32, "seed_counter": 0}


def set_compute_dtype(dtype):
    """Activation / MFMA input type: torch.float32 (parity mode) or torch.bfloat16."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be float32 or bfloat16")
    _state["dtype"] = dtype
    _cast_cache.clear()
    _wt_registry.clear()


def compute_dtype():
    return _state["dtype"]


class compute_as:
    """Context manager: run the ops inside in ``dtype`` (an fp32 island inside a bf16 run, e.g. the small style
    critic whose gradient-penalty double-backward loses too much to 8-bit mantissas).  Every op remembers the
    compute dtype of its forward, and its backward -- first and second order, whenever and on whichever thread
    autograd runs it -- computes in that dtype again (``_in_forward_dtype``).  Unlike ``set_compute_dtype`` nothing
    is flushed: the weight caches are keyed by dtype."""

    def __init__(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be float32 or bfloat16")
        self.dtype = dtype

    def __enter__(self):
        self.prev = _state["dtype"]
        _state["dtype"] = self.dtype
        return self

    def __exit__(self, *exc):
        _state["dtype"] = self.prev


def _in_forward_dtype(backward):
    """Decorator for Function.backward: compute in the dtype the forward ran in (ctx.cd)."""
    def wrapped(ctx, *grads):
        cd = getattr(ctx, "cd", None)
        if cd is None or cd is _state["dtype"]:
            return backward(ctx, *grads)
        with compute_as(cd):
            return backward(ctx, *grads)
    wrapped.__name__ = backward.__name__
    wrapped.__doc__ = backward.__doc__
    return wrapped


_tls = threading.local()        # per Python thread: second_order flag


def next_seed():
    """64-bit dropout seed: torch's initial seed mixed with a per-process call counter."""
    _state["seed_counter"] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _state["seed_counter"] * 0xD1B54A32D192ED03) % (1 << 64)


_cast_cache = {}
_param_epoch = {}


def params_changed(*modules):
    """Tell the weight caches (bf16 shadows here, the engines' packed weights) that the parameters of
    ``modules`` were updated.  In-place updates that bump ``Parameter._version`` (torch's default / foreach
    optimizers, ``load_state_dict``, ``p.add_``) are seen without this call; torch's FUSED optimizers write
    through ``torch._fused_adam_`` and leave ``_version`` untouched, so a trainer that uses them must call
    this after ``optimizer.step()`` (trainer_wgan_wh.Trainer does)."""
    for m in modules:
        for p in m.parameters():
            _param_epoch[id(p)] = _param_epoch.get(id(p), 0) + 1


def params_changed_list(params):
    """params_changed for an explicit list of parameters (the in-tree optimizer calls it after every step)."""
    for p in params:
        _param_epoch[id(p)] = _param_epoch.get(id(p), 0) + 1


def param_epoch(p):
    return _param_epoch.get(id(p), 0)


_wt_registry = {}


def register_transposed(p, wt):
    """``wt`` [in, out] holds the current values of the Linear weight ``p`` [out, in] in the compute dtype, k-contiguous
    for dX = dY W (the engines' packs make these copies anyway).  _Matmul uses it while ``p`` stays unchanged: the NN
    product then runs as an NT product on the K-split MFMA kernel (1440 x 512 x 512: 17.2 -> 9.3 us)."""
    _wt_registry[id(p)] = (weakref.ref(p), (p._version, _param_epoch.get(id(p), 0)), p.data_ptr(), wt)


def transposed(p):
    """The registered transposed copy of ``p`` if it is still current, else None."""
    hit = _wt_registry.get(id(p))
    if (hit is not None and hit[0]() is p and hit[1] == (p._version, _param_epoch.get(id(p), 0))
            and hit[2] == p.data_ptr() and hit[3].dtype == _state["dtype"]):
        return hit[3]
    return None


def cast(t):
    """Tensor in the compute dtype; casts of parameters are cached until the parameter changes."""
    cd = _state["dtype"]
    if t.dtype == cd:
        return t
    if isinstance(t, torch.nn.Parameter):
        key = (id(t), cd)                   # one slot per (parameter, compute dtype): fp32 islands do not evict bf16
        hit = _cast_cache.get(key)
        ver = (t._version, _param_epoch.get(id(t), 0))
        if hit is not None and hit[0]() is t and hit[1] == ver and hit[2] == t.data_ptr():
            return hit[3]
        c = t.detach().to(cd)
        _cast_cache[key] = (weakref.ref(t), ver, t.data_ptr(), c)
        return c
    return t.detach().to(cd) if not torch.is_grad_enabled() else t.to(cd)


def _f32(t):
    return t if t.dtype == torch.float32 else t.float()


def _mat(t):
    """(row stride, col stride) of the last two dims with one of them 1, copying if needed."""
    r, c = t.shape[-2], t.shape[-1]
    sr, sc = t.stride(-2), t.stride(-1)
    if sc == 1 or c == 1:
        return t, (sr if r > 1 else max(c, 1)), 1
    if sr == 1 or r == 1:
        return t, 1, sc
    t = t.contiguous()
    return t, t.stride(-2), 1


def _batch(t):
    """Leading dims as (outer, inner) sizes and strides (at most two batch dims)."""
    nb = t.dim() - 2
    if nb == 0:
        return (1, 1), (0, 0)
    if nb == 1:
        return (1, t.shape[0]), (0, t.stride(0) if t.shape[0] > 1 else 0)
    if nb == 2:
        return (t.shape[0], t.shape[1]), (t.stride(0) if t.shape[0] > 1 else 0,
                                          t.stride(1) if t.shape[1] > 1 else 0)
    raise ValueError("matmul supports at most two batch dimensions")


def _workspace(device):
    """Persistent fp32 scratch for the split-K slabs of the skinny-M GEMM: one per (device, stream), launches
    on one stream are ordered, launches on different streams must not share it."""
    ws = _state.setdefault("ws", {})
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    t = ws.get(key)
    if t is None:
        t = ws[key] = torch.empty(8 << 20, dtype=torch.float32, device=device)
    return t


def gemm(a, b, tb, out_dtype, bias=None, relu=False, residual=None, gate=None, alpha=1.0, out=None, beta=0.0,
         drop_p=0.0, drop_seed=0, no_skinny=False, ln=None, k_seg=None):
    """out[..., M, N] = epilogue(alpha * a[..., M, K] @ (b[..., N, K]^T if tb else b[..., K, N])).
    ln = (colsum [N], mean_out [M] or None, rstd_out [M] or None, eps): LayerNorm prologue of the
    skinny kernel (see mssr_gemm_desc.ln_colsum)."""
    L.require_cuda(a, b)
    a, a_rs, a_cs = _mat(a)
    b, s0, s1 = _mat(b)
    if tb:                      # b is [.., N, K]; as [k, n]: k stride = s1, n stride = s0
        n, k2 = b.shape[-2], b.shape[-1]
        b_rs, b_cs = s1, s0
    else:
        k2, n = b.shape[-2], b.shape[-1]
        b_rs, b_cs = s0, s1
    m, k = a.shape[-2], a.shape[-1]
    if k != k2:
        raise ValueError("matmul: inner sizes differ (%d vs %d)" % (k, k2))
    (bo, bi), (a_bo, a_bi) = _batch(a)
    (bo2, bi2), (b_bo, b_bi) = _batch(b)
    if (bo, bi) != (bo2, bi2):
        raise ValueError("matmul: batch dims differ %s vs %s" % (a.shape, b.shape))
    if a.dtype != b.dtype:
        raise TypeError("matmul: operand dtypes differ (%s, %s)" % (a.dtype, b.dtype))
    if out is None:
        out = torch.empty(a.shape[:-2] + (m, n), dtype=out_dtype, device=a.device)
    d = L.GemmDesc()
    d.a, d.b, d.c = a.data_ptr(), b.data_ptr(), out.data_ptr()
    d.bias = L.ptr(bias)
    d.residual = L.ptr(residual)
    d.gate = L.ptr(gate)
    d.m, d.n, d.k = m, n, k
    d.a_rs, d.a_cs, d.b_rs, d.b_cs = a_rs, a_cs, b_rs, b_cs
    d.ldc = out.stride(-2) if m > 1 else max(n, 1)
    d.ldr = residual.stride(-2) if residual is not None and m > 1 else n
    d.ldg = gate.stride(-2) if gate is not None and m > 1 else n
    d.batch_outer, d.batch_inner = bo, bi
    d.a_bo, d.a_bi, d.b_bo, d.b_bi = a_bo, a_bi, b_bo, b_bi
    (_, _), (d.c_bo, d.c_bi) = _batch(out)
    d.alpha, d.beta, d.relu = alpha, beta, int(relu)
    d.in_dtype, d.out_dtype = L.dt(a), L.dt(out)
    d.drop_p, d.drop_seed, d.no_skinny = drop_p, drop_seed, int(no_skinny)
    if k > 2048 and (m <= 128 or m * n * ((k + 1023) // 1024) <= (8 << 20)):
        ws = _workspace(a.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    if ln is not None:
        d.ln_colsum, d.ln_mean, d.ln_rstd, d.ln_eps = ln[0].data_ptr(), L.ptr(ln[1]), L.ptr(ln[2]), float(ln[3])
    if k_seg is not None:       # (segment length, segment stride, segments): see mssr_gemm_desc.k_seg_len
        seg_len, seg_stride, n_seg = k_seg
        if (n_seg - 1) * seg_stride + seg_len > k:
            raise ValueError("k segmentation reaches beyond the operands")
        d.k, d.k_seg_len, d.k_seg_stride = n_seg * seg_len, seg_len, seg_stride
    L.check(L.load().mssr_gemm(d, L.stream()), "gemm")
    return out


def start_profile(max_records=200000):
    """Time every GEMM launch (from Python or from inside the decode engine) with HIP events on its
    stream (mssr_prof_start, include/mssr_hip.h)."""
    L.check(L.load().mssr_prof_start(int(max_records)), "prof_start")


def stop_profile():
    """-> list of dicts (kernel, shape, algorithmic bytes / flops, ms) of the launches since start_profile."""
    import ctypes
    lib = L.load()
    n = lib.mssr_prof_stop()
    if n < 0:
        L.check(1, "prof_stop")
    out, r = [], L.ProfRecord()
    for i in range(n):
        L.check(lib.mssr_prof_get(i, ctypes.byref(r)), "prof_get")
        isz, osz = (2 if r.in_dtype == L.BF16 else 4), (2 if r.out_dtype == L.BF16 else 4)
        tin, tout = ("bf16" if isz == 2 else "f32"), ("bf16" if osz == 2 else "f32")
        if r.kernel == 5:
            name = "gemm_wide_kernel<%s,%s>" % (tout, "LN" if r.ln else "noLN")
        elif r.kernel == 4:
            name = "gemm_pipe_kernel<%s>" % tout
        elif r.kernel == 3:
            name = "gemm_lnb_kernel<%s>" % tout
        elif r.kernel == 2:
            name = "gemm_tn_kernel<%s>" % tout
        elif r.kernel == 1:
            # r.tile = the launcher's choice: 100 * (operands staged by LDS-DMA) + 10 * FM + FN; 0: the fp32 kernel
            if r.tile == 0:
                name = "gemm_skinny_f32_kernel"
            else:
                name = "gemm_skinny%s_kernel<%s,%s,%dx%d>%s" % (
                    "_dma" if r.tile >= 100 else "", tout, "LN" if r.ln else "noLN", 16 * (r.tile % 100 // 10),
                    16 * (r.tile % 10), "+splitK" if r.ksplit > 1 else "")
        else:
            name = "gemm_kernel<%s,%s,%d>" % (tin, tout, r.tile)
        if r.ms < 0:
            continue
        out.append(dict(kernel=name, in_dtype=tin, shape=(r.batch, r.m, r.n, r.k), ms=r.ms,
                        bytes=r.batch * ((r.m * r.k + r.k * r.n) * isz + r.m * r.n * osz),
                        flops=2 * r.batch * r.m * r.n * r.k))
    return out


_tpad_cache = {}


def _cast_t_pad(p, kp):
    """bf16 copy of the parameter p [K, N] as [N, kp] (k-contiguous, zero columns K..kp), cached like cast()."""
    key = id(p)
    ver = (p._version, _param_epoch.get(key, 0), kp)
    hit = _tpad_cache.get(key)
    if hit is not None and hit[0]() is p and hit[1] == ver and hit[2] == p.data_ptr():
        return hit[3]
    K, N = p.shape
    t = torch.zeros(N, kp, dtype=torch.bfloat16, device=p.device)
    t[:, :K].copy_(p.detach().t())
    _tpad_cache[key] = (weakref.ref(p), ver, p.data_ptr(), t)
    return t


def _long_k_nn(a, w, out_dtype):
    """a [M, K] @ w [K, N] with a vocabulary-sized K (soft embedding of the critics, onmt/modules/embeddings.py:
    278-280: probabilities times the table).  The table is n-contiguous and K = 9606 rows are not 16-byte
    aligned in bf16, which leaves only the general kernel's serial K loop (440 us at 1440 x 256 x 9606); with the
    operand cast into a K-padded buffer (the cast is needed anyway) and a cached transposed table the split-K
    MFMA kernel applies (both operands k-contiguous)."""
    M, K = a.shape
    kp = (K + 31) // 32 * 32
    wt = _cast_t_pad(w, kp)
    ap = torch.empty(M, kp, dtype=torch.bfloat16, device=a.device)
    ap[:, :K].copy_(a.detach())
    if kp > K:
        ap[:, K:].zero_()
    return gemm(ap, wt, True, out_dtype), ap


_zero_arena = {}
_zero_lock = threading.Lock()
_ZERO_ARENA = 1 << 20          # floats per arena (4 MB)


def zeros_f32(n, device):
    """n zeroed floats for a kernel that ADDS into its output (bias-gradient column sums, the pooling's and the
    penalty's accumulators): a slice of a 4 MB arena that one fill zeroed, instead of one fill launch per output --
    the critics' phase made ~45 such launches of a few hundred bytes each.  Slices are handed out once; an arena lives
    while any of its slices does.  Per (device, stream): the fill is ordered before its users on that stream."""
    if n > 65536:
        return torch.zeros(n, dtype=torch.float32, device=device)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    take = (n + 63) // 64 * 64
    with _zero_lock:
        ent = _zero_arena.get(key)
        if ent is None or ent[1] + take > _ZERO_ARENA:
            ent = _zero_arena[key] = [torch.zeros(_ZERO_ARENA, dtype=torch.float32, device=device), 0]
        off = ent[1]
        ent[1] = off + take
    return ent[0][off:off + n]


class _Matmul(Function):
    """C = A @ B (tb=False, B [.., K, N]) or A @ B^T (tb=True, B [.., N, K]); closed under autograd."""

    @staticmethod
    def forward(ctx, a, b, tb, out_dtype, heads_last=False):
        ctx.cd = _state["dtype"]
        ctx.save_for_backward(a, b)
        ctx.tb = tb
        ctx.a_pad = None
        if (not tb and not heads_last and a.dim() == 2 and b.dim() == 2 and isinstance(b, torch.nn.Parameter)
                and b.shape[0] > 2048 and _state["dtype"] == torch.bfloat16 and a.is_cuda):
            out, ap = _long_k_nn(a, b, out_dtype or _state["dtype"])
            if b.requires_grad:
                ctx.a_pad = ap          # the weight gradient a^T dc reads this bf16 copy (rows padded to 16 bytes)
            return out
        if (not tb and not heads_last and a.dim() == 2 and b.dim() == 2 and isinstance(b, torch.nn.Parameter)
                and a.is_cuda):
            wt = transposed(b)
            if wt is not None:
                return gemm(cast(a), wt, True, out_dtype or _state["dtype"])
        ac, bc = cast(a), cast(b)
        out = None
        if ac.dtype == torch.float32 and out_dtype == torch.bfloat16:
            # an fp32 island (compute_as) handing a gradient back to a bf16 producer: fp32 product, one rounding
            return _Matmul._product(ac, bc, tb, torch.float32, heads_last).to(torch.bfloat16)
        return _Matmul._product(ac, bc, tb, out_dtype, heads_last)

    @staticmethod
    def _product(ac, bc, tb, out_dtype, heads_last):
        out = None
        if heads_last:          # [B,H,M,N] result stored as [B,M,H,N] (the "unshape" layout, mha.py:126-129)
            B, H, M = ac.shape[0], ac.shape[1], ac.shape[2]
            N = bc.shape[-2] if tb else bc.shape[-1]
            out = torch.empty(B, M, H, N, dtype=out_dtype or _state["dtype"], device=ac.device).permute(0, 2, 1, 3)
        return gemm(ac, bc, tb, out_dtype or _state["dtype"], out=out)

    @staticmethod
    @_in_forward_dtype
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        da = db = None
        if ctx.needs_input_grad[0]:
            da = _Matmul.apply(dc, b, not ctx.tb, a.dtype)
        if ctx.needs_input_grad[1] and not _skip(b):
            if ctx.tb:
                db = _Matmul.apply(dc.transpose(-1, -2), a, False, b.dtype)
            elif ctx.a_pad is not None and not (torch.is_grad_enabled() and a.requires_grad):
                # vocabulary-wide weight gradient (soft embedding: table gradient = probabilities^T dc) on the padded
                # bf16 copy the forward made: no second cast of the [rows, V] operand, and its 16-byte aligned rows
                # take the LDS-DMA kernel (9606 x 256 x 2880: 144 -> ~50 us)
                ap = ctx.a_pad
                db = gemm(ap[:, :a.shape[1]].t(), cast(dc).contiguous(), False, b.dtype)
            else:
                db = _Matmul.apply(a.transpose(-1, -2), dc, False, b.dtype)
        return da, db, None, None, None


def matmul(a, b, tb=False, out_dtype=None, heads_last=False):
    return _Matmul.apply(a, b, tb, out_dtype, heads_last)


class _Gate(Function):
    """x * (gate > 0)."""

    @staticmethod
    def forward(ctx, x, gate):
        ctx.cd = _state["dtype"]
        x = cast(x).contiguous()
        gate = cast(gate).contiguous()
        ctx.save_for_backward(gate)
        y = torch.empty_like(x)
        L.check(L.load().mssr_gate_gt0(x.data_ptr(), gate.data_ptr(), y.data_ptr(), x.numel(), L.dt(x), L.stream()),
                "gate_gt0")
        return y

    @staticmethod
    @_in_forward_dtype
    def backward(ctx, g):
        gate, = ctx.saved_tensors
        return _Gate.apply(g, gate), None


class _Colsum(Function):
    """fp32 column sums of a [rows, n] matrix (Linear bias gradient)."""

    @staticmethod
    def forward(ctx, x):
        ctx.cd = _state["dtype"]
        x = cast(x)
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        ctx.shape, ctx.dtype = x.shape, x.dtype
        out = zeros_f32(x2.shape[1], x.device)
        L.check(L.load().mssr_colsum(x2.data_ptr(), out.data_ptr(), x2.shape[0], x2.shape[1],
                                     x2.stride(0) if x2.shape[0] > 1 else x2.shape[1], 1, L.dt(x2), L.stream()),
                "colsum")
        return out

    @staticmethod
    @_in_forward_dtype
    def backward(ctx, g):
        return g.to(ctx.dtype).expand(ctx.shape)


class _Linear(Function):
    """y = [relu](x W^T + b) [+ residual] with the bias / ReLU / residual fused into the GEMM
    epilogue (nn.Linear sites of the path; FFN position_ffn.py:35-37)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, residual):
        ctx.cd = _state["dtype"]
        xc = cast(x)
        if xc.stride(-1) != 1:
            xc = xc.contiguous()
        x2 = xc.reshape(-1, xc.shape[-1])
        n = weight.shape[0]
        res2 = None
        if residual is not None:
            res2 = cast(residual).reshape(-1, n)
            if res2.stride(-1) != 1:
                res2 = res2.contiguous()
        y = gemm(x2, cast(weight), True, _state["dtype"], bias=None if bias is None else _f32(bias.detach()),
                 relu=relu, residual=res2)
        y = y.view(xc.shape[:-1] + (n,))
        ctx.relu = relu
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    @_in_forward_dtype
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dz = _Gate.apply(dy, y) if ctx.relu else dy
        dz2 = dz.reshape(-1, dz.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _Matmul.apply(dz2, weight, False, x.dtype).view(x.shape)
        if ctx.needs_input_grad[1] and not _skip(weight):
            dw = _Matmul.apply(dz2.transpose(0, 1), x.reshape(-1, x.shape[-1]), False, weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2] and not _skip(weight):
            db = _Colsum.apply(dz2)
        dres = dy if ctx.has_res and ctx.needs_input_grad[4] else None
        return dx, dw, db, None, dres


def linear(x, weight, bias=None, relu=False, residual=None):
    return _Linear.apply(x, weight, bias, relu, residual)


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ctx.cd = _state["dtype"]
        xc = cast(x).contiguous()
        L.require_cuda(xc, gamma, beta)
        d = xc.shape[-1]
        rows = xc.numel() // d
        g32, b32 = _f32(gamma.detach()).contiguous(), _f32(beta.detach()).contiguous()
        y = torch.empty_like(xc)
        mean = torch.empty(rows, dtype=torch.float32, device=xc.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=xc.device)
        L.check(L.load().mssr_layernorm_fwd(xc.data_ptr(), g32.data_ptr(), b32.data_ptr(), y.data_ptr(),
                                            mean.data_ptr(), rstd.data_ptr(), rows, d, eps, L.dt(xc), L.stream()),
                "layernorm_fwd")
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    @_in_forward_dtype
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dx, dg, db = _LayerNormBwd.apply(dy, x, gamma, mean, rstd)
        if _skip(gamma):
            dg = db = None
        return dx, dg, db, None


class _LayerNormBwd(Function):
    @staticmethod
    def forward(ctx, dy, x, gamma, mean, rstd):
        ctx.cd = _state["dtype"]
        xc, dyc = cast(x).contiguous(), cast(dy).contiguous()
        d = xc.shape[-1]
        rows = xc.numel() // d
        lib = L.load()
        g32 = _f32(gamma.detach()).contiguous()
        dx = torch.empty_like(xc)
        dg = torch.empty(d, dtype=torch.float32, device=xc.device)
        db = torch.empty(d, dtype=torch.float32, device=xc.device)
        parts = lib.mssr_layernorm_partial_rows(rows)
        ws = torch.empty(2 * parts * d, dtype=torch.float32, device=xc.device)
        L.check(lib.mssr_layernorm_bwd(dyc.data_ptr(), xc.data_ptr(), g32.data_ptr(), mean.data_ptr(),
                                       rstd.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                       rows, d, L.dt(xc), L.stream()), "layernorm_bwd")
        ctx.save_for_backward(dy, x, gamma, mean, rstd)
        if dx.dtype != x.dtype:
            dx = dx.to(x.dtype)
        return dx, dg.to(gamma.dtype), db.to(gamma.dtype)

    @staticmethod
    @once_differentiable
    @_in_forward_dtype
    def backward(ctx, ggx, ggg, ggb):
        dy, x, gamma, mean, rstd = ctx.saved_tensors
        xc, dyc = cast(x).contiguous(), cast(dy).contiguous()
        d = xc.shape[-1]
        rows = xc.numel() // d
        lib = L.load()
        g32 = _f32(gamma.detach()).contiguous()
        u = None if ggx is None else cast(ggx).contiguous()
        vg = None if ggg is None else _f32(ggg).contiguous()
        vb = None if ggb is None else _f32(ggb).contiguous()
        g_dy = torch.empty_like(xc)
        g_x = torch.empty_like(xc)
        g_gamma = torch.empty(d, dtype=torch.float32, device=xc.device)
        parts = lib.mssr_layernorm_partial_rows(rows)
        ws = torch.empty(parts * d, dtype=torch.float32, device=xc.device)
        L.check(lib.mssr_layernorm_bwd_bwd(L.ptr(u), L.ptr(vg), L.ptr(vb), dyc.data_ptr(), xc.data_ptr(),
                                           g32.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g_dy.data_ptr(),
                                           g_x.data_ptr(), g_gamma.data_ptr(), ws.data_ptr(), rows, d, L.dt(xc),
                                           L.stream()), "layernorm_bwd_bwd")
        return g_dy.to(dy.dtype), g_x.to(x.dtype), g_gamma.to(gamma.dtype), None, None


def layer_norm(x, gamma, beta, eps=1e-6):
    return _LayerNorm.apply(x, gamma, beta, eps)


def _mask_args(mask, x):
    """uint8 view + (rows_per_batch, q_len, batch stride, query stride) of an attention mask
    [B,1,Lk] / [B,Lq,Lk] applied to scores x [B,H,Lq,Lk]."""
    if mask is None:
        return None, 1, 1, 0, 0
    m = mask if mask.stride(-1) == 1 else mask.contiguous()
    if m.dtype == torch.bool:
        m = m.view(torch.uint8)
    elif m.dtype != torch.uint8:
        m = (m != 0).view(torch.uint8)
    B, H, Lq, Lk = x.shape
    if m.dim() != 3 or m.shape[0] != B or m.shape[2] != Lk or m.shape[1] not in (1, Lq):
        raise ValueError("mask shape %s does not fit scores %s" % (tuple(mask.shape), tuple(x.shape)))
    return m, H * Lq, Lq, (m.stride(0) if B > 1 else 0), (m.stride(1) if m.shape[1] > 1 else 0)


class _Softmax(Function):
    """softmax / log_softmax(scale * x) along the last axis with optional attention mask."""

    @staticmethod
    def forward(ctx, x, mask, scale, log_mode, out_dtype):
        ctx.cd = _state["dtype"]
        xc = cast(x).contiguous()
        L.require_cuda(xc)
        n = xc.shape[-1]
        rows = xc.numel() // n
        m, rpb, ql, mbs, mqs = _mask_args(mask, xc) if mask is not None else (None, 1, 1, 0, 0)
        y = torch.empty(xc.shape, dtype=out_dtype or xc.dtype, device=xc.device)
        L.check(L.load().mssr_softmax_fwd(xc.data_ptr(), y.data_ptr(), L.ptr(m), rows, n, n, n, rpb, ql, mbs, mqs,
                                          scale, int(log_mode), L.dt(xc), L.dt(y), L.stream()), "softmax_fwd")
        ctx.save_for_backward(y)
        ctx.scale, ctx.log_mode, ctx.in_dtype = scale, log_mode, x.dtype
        ctx.mask = None if mask is None else (mask != 0)
        return y

    @staticmethod
    @_in_forward_dtype
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        dx = _SoftmaxBwd.apply(y, dy, ctx.scale, ctx.log_mode, ctx.in_dtype)
        if ctx.mask is not None:
            # masked_fill blocks the gradient of the entries it overwrites (multi_headed_attn.py:197-199).  Where a
            # row keeps some key, masked entries have y = 0 and so dx = 0 already; a row with EVERY key masked is
            # uniform and the softmax formula alone would hand its scores a gradient they never see in the reference
            dx = dx.masked_fill(ctx.mask.unsqueeze(1), 0)
        return dx, None, None, None, None


class _SoftmaxBwd(Function):
    @staticmethod
    def forward(ctx, y, dy, scale, log_mode, dx_dtype):
        ctx.cd = _state["dtype"]
        yc, dyc = y.contiguous(), dy.contiguous()
        n = yc.shape[-1]
        rows = yc.numel() // n
        dx = torch.empty(yc.shape, dtype=dx_dtype, device=yc.device)
        L.check(L.load().mssr_softmax_bwd(yc.data_ptr(), dyc.data_ptr(), dx.data_ptr(), rows, n, n, n, n, scale,
                                          int(log_mode), L.dt(yc), L.dt(dyc), L.dt(dx), L.stream()), "softmax_bwd")
        ctx.save_for_backward(y, dy)
        ctx.scale, ctx.log_mode = scale, log_mode
        return dx

    @staticmethod
    @_in_forward_dtype
    def backward(ctx, g):
        y, dy = ctx.saved_tensors
        if ctx.log_mode:
            raise NotImplementedError("double backward of log_softmax is not on the MSSRNet path")
        g_y = g_dy = None
        if ctx.needs_input_grad[0]:
            gc = g.to(y.dtype).contiguous()
            dyc = dy.to(y.dtype).contiguous()
            yc = y.contiguous()
            n = yc.shape[-1]
            g_y = torch.empty_like(yc)
            L.check(L.load().mssr_softmax_bwd_bwd(gc.data_ptr(), yc.data_ptr(), dyc.data_ptr(), g_y.data_ptr(),
                                                  yc.numel() // n, n, n, ctx.scale, L.dt(yc), L.stream()),
                    "softmax_bwd_bwd")
        if ctx.needs_input_grad[1]:
            g_dy = _SoftmaxBwd.apply(y, g, ctx.scale, ctx.log_mode, dy.dtype)
        return g_y, g_dy, None, None, None


def softmax(x, mask=None, scale=1.0, out_dtype=None):
    return _Softmax.apply(x, mask, scale, False, out_dtype)


def log_softmax(x, scale=1.0, out_dtype=None):
    return _Softmax.apply(x, None, scale, True, out_dtype)


class _MaskScale(Function):
    """y = x * keep * scale: the linear map dropout applies once its keep mask is fixed."""

    @staticmethod
    def forward(ctx, x, keep, scale):
        ctx.cd = _state["dtype"]
        xc = cast(x).contiguous()
        y = torch.empty_like(xc)
        L.check(L.load().mssr_mask_scale(xc.data_ptr(), keep.data_ptr(), None, y.data_ptr(), xc.numel(), scale,
                                         L.dt(xc), L.stream()), "mask_scale")
        ctx.save_for_backward(keep)
        ctx.scale, ctx.dtype = scale, x.dtype
        return y

    @staticmethod
    @_in_forward_dtype
    def backward(ctx, g):
        keep, = ctx.saved_tensors
        return _MaskScale.apply(g, keep, ctx.scale).to(ctx.dtype), None, None


class _Dropout(Function):
    """y = dropout_p(x) [+ residual]; keep mask from the in-kernel counter-based RNG."""

    @staticmethod
    def forward(ctx, x, p, residual):
        ctx.cd = _state["dtype"]
        xc = cast(x).contiguous()
        L.require_cuda(xc)
        rc = None if residual is None else cast(residual).contiguous()
        y = torch.empty_like(xc)
        keep = torch.empty(xc.shape, dtype=torch.uint8, device=xc.device)
        L.check(L.load().mssr_dropout_fwd(xc.data_ptr(), L.ptr(rc), y.data_ptr(), keep.data_ptr(), xc.numel(), p,
                                          next_seed(), L.dt(xc), L.stream()), "dropout_fwd")
        ctx.save_for_backward(keep)
        ctx.scale, ctx.has_res = 1.0 / (1.0 - p), residual is not None
        ctx.dtype = x.dtype
        return y

    @staticmethod
    @_in_forward_dtype
    def backward(ctx, g):
        keep, = ctx.saved_tensors
        dx = _MaskScale.apply(g, keep, ctx.scale)
        if dx.dtype != ctx.dtype:
            dx = dx.to(ctx.dtype)
        return dx, None, (g if ctx.has_res else None)


def dropout(x, p, training, residual=None):
    """dropout(x) (+ residual).  Identity (+ residual) when not training or p == 0."""
    if training and p > 0.0:
        return _Dropout.apply(x, float(p), residual)
    if residual is None:
        return x
    return add(x, residual)


class _Axpby(Function):
    """a*x + b*z."""

    @staticmethod
    def forward(ctx, x, z, a, b):
        ctx.cd = _state["dtype"]
        xc = cast(x).contiguous()
        L.require_cuda(xc)
        zc = None if z is None else cast(z).contiguous()
        y = torch.empty_like(xc)
        L.check(L.load().mssr_axpby(xc.data_ptr(), L.ptr(zc), y.data_ptr(), xc.numel(), a, b, L.dt(xc), L.stream()),
                "axpby")
        ctx.a, ctx.b, ctx.has_z = a, b, z is not None
        ctx.dx, ctx.dz = x.dtype, (None if z is None else z.dtype)
        return y

    @staticmethod
    @_in_forward_dtype
    def backward(ctx, g):
        gx = g if ctx.a == 1.0 else _Axpby.apply(g, None, ctx.a, 0.0)
        gz = None
        if ctx.has_z:
            gz = g if ctx.b == 1.0 else _Axpby.apply(g, None, ctx.b, 0.0)
            if gz.dtype != ctx.dz:
                gz = gz.to(ctx.dz)
        if gx.dtype != ctx.dx:
            gx = gx.to(ctx.dx)
        return gx, gz, None, None


def add(x, z):
    if x.shape != z.shape:
        raise ValueError("add: shapes differ %s vs %s" % (tuple(x.shape), tuple(z.shape)))
    return _Axpby.apply(x, z, 1.0, 1.0)


def scale(x, a):
    return _Axpby.apply(x, None, float(a), 0.0)


class _ScaleAddPe(Function):
    """x*sqrt(d) + pe[pos] (PositionalEncoding.forward, onmt/modules/embeddings.py:48-52)."""

    @staticmethod
    def forward(ctx, x, pe, step, s):
        ctx.cd = _state["dtype"]
        xc = cast(x).contiguous()            # [len, B, d]
        L.require_cuda(xc, pe)
        ln, B, d = xc.shape
        y = torch.empty_like(xc)
        L.check(L.load().mssr_scale_add_pe(xc.data_ptr(), pe.data_ptr(), y.data_ptr(), ln * B, d, B,
                                           -1 if step is None else int(step), s, L.dt(xc), L.stream()),
                "scale_add_pe")
        ctx.s, ctx.dtype = s, x.dtype
        return y

    @staticmethod
    @_in_forward_dtype
    def backward(ctx, g):
        gx = _Axpby.apply(g, None, ctx.s, 0.0)
        return (gx if gx.dtype == ctx.dtype else gx.to(ctx.dtype)), None, None, None


def scale_add_pe(x, pe2d, step, s):
    return _ScaleAddPe.apply(x, pe2d, step, float(s))


class _EmbedPe(Function):
    """weight[ids]*sqrt(d) + pe[pos]; the padding row gets no gradient (nn.Embedding(padding_idx))."""

    @staticmethod
    def forward(ctx, ids, weight, pe, step, s, pad_idx):
        ctx.cd = _state["dtype"]
        L.require_cuda(ids, weight, pe)
        ln, B = ids.shape
        V, d = weight.shape
        idc = ids.contiguous()
        w32 = _f32(weight.detach()).contiguous()
        y = torch.empty(ln, B, d, dtype=_state["dtype"], device=weight.device)
        L.check(L.load().mssr_embedding_fwd(idc.data_ptr(), w32.data_ptr(), pe.data_ptr(), y.data_ptr(), ln * B, d,
                                            V, B, -1 if step is None else int(step), s, L.dt(y), L.stream()),
                "embedding_fwd")
        ctx.save_for_backward(idc)
        ctx.shape, ctx.s, ctx.pad, ctx.wdtype = (V, d), s, pad_idx, weight.dtype
        return y

    @staticmethod
    @once_differentiable
    @_in_forward_dtype
    def backward(ctx, g):
        idc, = ctx.saved_tensors
        V, d = ctx.shape
        gc = g.contiguous()
        dw = torch.zeros(V, d, dtype=torch.float32, device=g.device)
        L.check(L.load().mssr_embedding_bwd(gc.data_ptr(), idc.data_ptr(), dw.data_ptr(), idc.numel(), d, V,
                                            -1 if ctx.pad is None else int(ctx.pad), ctx.s, L.dt(gc), L.stream()),
                "embedding_bwd")
        return None, dw.to(ctx.wdtype), None, None, None, None


def embed_pe(ids, weight, pe2d, step, s, pad_idx):
    return _EmbedPe.apply(ids, weight, pe2d, step, float(s), pad_idx)


class second_order:
    """Context manager: build graphs that can be differentiated twice (the WGAN-GP penalties,
    onmt/trainer_wgan_wh.py:1247-1250,1315-1318).  Inside it the fused attention kernel and the encoder
    engine, whose backward passes are single kernels / C calls, step aside for the compositions of
    matmul / softmax / dropout ops above, every one of which has a differentiable backward."""

    def __enter__(self):
        self.prev = getattr(_tls, "second_order", False)
        _tls.second_order = True          # per thread: only forward-time dispatch reads it

    def __exit__(self, *exc):
        _tls.second_order = self.prev


class input_grads_only:
    """Context manager for a ``torch.autograd.grad(outputs, x)`` call that wants the gradient of an INPUT only (the
    critics' input gradient of the WGAN-GP penalty, onmt/trainer_wgan_wh.py:1247-1250, 1315-1318): the backward
    nodes of the ops above leave out their parameter gradients (weight / bias / LayerNorm gain products), which
    autograd would compute and then drop -- ``needs_input_grad`` only knows that the parameters require gradients
    in general, not that this call does not collect them.  A process-wide switch, not a thread-local one: the
    backward nodes run on autograd's device thread.  Never wrap a ``backward()`` that accumulates into parameters."""

    def __enter__(self):
        self.prev = _state.get("skip_param_grads", False)
        _state["skip_param_grads"] = True

    def __exit__(self, *exc):
        _state["skip_param_grads"] = self.prev


def _skip(p):
    return _state.get("skip_param_grads", False) and isinstance(p, torch.nn.Parameter)


class _GainPickBwd(Function):
    """dx = onehot(labels) * dy * f'(x): the backward of _GainPick as an op of its own, so that graphs built with
    create_graph (the gradient penalty) can differentiate through it."""

    @staticmethod
    def forward(ctx, dy, x, labels, ratio, sig):
        n, C = x.shape
        dyc = dy.to(torch.float32).contiguous()
        dx = torch.empty(n, C, dtype=torch.float32, device=x.device)
        L.check(L.load().mssr_gain_pick_bwd(dyc.data_ptr(), x.data_ptr(), x.stride(0), labels.data_ptr(), dx.data_ptr(), n, C,
                                            ratio, sig, L.stream()), "gain_pick_bwd")
        ctx.save_for_backward(dyc, x, labels)
        ctx.ratio, ctx.sig = ratio, sig
        return dx

    @staticmethod
    @once_differentiable
    def backward(ctx, ddx):
        dy, x, labels = ctx.saved_tensors
        n, C = x.shape
        if ddx.stride(1) != 1 or ddx.dtype != torch.float32:
            ddx = ddx.to(torch.float32).contiguous()
        ddy = torch.empty(n, dtype=torch.float32, device=x.device)
        dx2 = torch.empty(n, C, dtype=torch.float32, device=x.device)
        L.check(L.load().mssr_gain_pick_bwd_bwd(ddx.data_ptr(), ddx.stride(0), dy.data_ptr(), x.data_ptr(), x.stride(0),
                                                labels.data_ptr(), ddy.data_ptr(), dx2.data_ptr(), n, C, ctx.ratio, ctx.sig,
                                                L.stream()), "gain_pick_bwd_bwd")
        return ddy, dx2, None, None, None


class _GainPick(Function):
    @staticmethod
    def forward(ctx, x, labels, ratio, sig):
        n, C = x.shape
        out = torch.empty(n, dtype=torch.float32, device=x.device)
        L.check(L.load().mssr_gain_pick_fwd(x.data_ptr(), x.stride(0), labels.data_ptr(), out.data_ptr(), n, C, ratio, sig,
                                            L.stream()), "gain_pick_fwd")
        ctx.save_for_backward(x, labels)
        ctx.ratio, ctx.sig = ratio, sig
        return out

    @staticmethod
    def backward(ctx, dy):
        x, labels = ctx.saved_tensors
        return _GainPickBwd.apply(dy, x, labels, ctx.ratio, ctx.sig), None, None, None


def gain_pick(logits, labels, ratio=1.0, sigmoid=True):
    """The critic's gain at each row's label: ratio * sigmoid(logits[i, labels[i]]) (or the raw logit), float32 [n].
    One launch for the reference's sigmoid gain + cal_disc_loss pick (onmt/trainer_wgan_wh.py:402-416, :1192-1200);
    twice differentiable."""
    L.require_cuda(logits, labels)
    if logits.dim() != 2 or labels.shape != logits.shape[:1] or labels.dtype != torch.int64:
        raise ValueError("gain_pick: logits [n, C] and int64 labels [n] expected (got %s, %s %s)" % (
            tuple(logits.shape), tuple(labels.shape), labels.dtype))
    x = logits if logits.dtype == torch.float32 and logits.stride(1) == 1 else logits.float().contiguous()
    return _GainPick.apply(x, labels.contiguous(), float(ratio), int(bool(sigmoid)))


class _GpPenalty(Function):
    """lamb * mean over the masked rows of (||g_row|| - k)^2 in one launch (csrc/elementwise.hip, mssr_gp_penalty_*).
    g [n0, n1, d] with any row strides (d contiguous), mask Bool [n0, n1] with any strides.  g carries the graph of
    the critic's input gradient; this node's own backward is first order (the penalty is differentiated once)."""

    @staticmethod
    def forward(ctx, g, mask, k, lamb):
        n0, n1, d = g.shape
        m8 = mask.view(torch.uint8)
        norms = torch.empty(n0 * n1, dtype=torch.float32, device=g.device)
        acc = zeros_f32(4, g.device)
        L.check(L.load().mssr_gp_penalty_fwd(g.data_ptr(), g.stride(0), g.stride(1), n0, n1, d, m8.data_ptr(), m8.stride(0),
                                             m8.stride(1), k, lamb, norms.data_ptr(), acc.data_ptr(), L.dt(g), L.stream()),
                "gp_penalty_fwd")
        ctx.save_for_backward(g, m8, norms, acc)
        ctx.k, ctx.lamb = k, lamb
        return acc[3]

    @staticmethod
    @once_differentiable
    def backward(ctx, dpen):
        g, m8, norms, acc = ctx.saved_tensors
        n0, n1, d = g.shape
        dg = torch.empty(n0, n1, d, dtype=g.dtype, device=g.device)
        dp = dpen.to(torch.float32).reshape(1).contiguous()
        L.check(L.load().mssr_gp_penalty_bwd(g.data_ptr(), g.stride(0), g.stride(1), n0, n1, d, m8.data_ptr(), m8.stride(0),
                                             m8.stride(1), ctx.k, ctx.lamb, norms.data_ptr(), acc.data_ptr(), dp.data_ptr(),
                                             dg.data_ptr(), L.dt(g), L.stream()), "gp_penalty_bwd")
        return dg, None, None, None


def gp_penalty(g, mask, k, lamb):
    """WGAN-GP penalty of per-token gradients g [n0, n1, d] over the rows where mask [n0, n1] (Bool) is set:
    ((||g_row||_2 - k) ** 2).mean() * lamb (onmt/trainer_wgan_wh.py:1253-1261, :1321-1329), float32 scalar."""
    L.require_cuda(g, mask)
    if g.dim() != 3 or mask.shape != g.shape[:2] or mask.dtype != torch.bool:
        raise ValueError("gp_penalty: g [n0, n1, d] and a Bool mask [n0, n1] expected (got %s, %s %s)" % (
            tuple(g.shape), tuple(mask.shape), mask.dtype))
    if g.stride(2) != 1 and g.shape[2] > 1:
        g = g.contiguous()
    return _GpPenalty.apply(g, mask, float(k), float(lamb))


def _blf(t):
    """[B, L, f] tensor, features contiguous -> (tensor, batch stride, time stride): [L, B, f] storage is read in place."""
    if t.stride(2) != 1 and t.shape[2] > 1:
        t = t.contiguous()
    return t, t.stride(0), t.stride(1)


def _empty_blf(B, Lx, f, time_major, dtype, device):
    if time_major:
        return torch.empty(Lx, B, f, dtype=dtype, device=device).transpose(0, 1)
    return torch.empty(B, Lx, f, dtype=dtype, device=device)


class _CtxPoolBwd(Function):
    """(dx, dh, dw2, db2) of _CtxPool from the gradients arriving at its two outputs, as an op of its own: the
    gradient penalty differentiates the critic's input gradient, which runs through here (mssr_ctx_pool_bwd_bwd)."""

    @staticmethod
    def forward(ctx, dp, q, x, h, w2, b2, a, lengths):
        ctx.cd = _state["dtype"]
        ctx.set_materialize_grads(False)
        xc, xbs, xts = _blf(cast(x))
        hc, hbs, hts = _blf(cast(h))
        B, Lx, d = xc.shape
        m = hc.shape[2]
        dpc = None if dp is None else cast(dp).contiguous()
        qc = None if q is None else q.to(torch.float32).contiguous()
        w32 = _f32(w2.detach()).contiguous()
        dx = None if dp is None else _empty_blf(B, Lx, d, xbs < xts, xc.dtype, xc.device)
        dh = _empty_blf(B, Lx, m, hbs < hts, hc.dtype, hc.device)
        acc = zeros_f32(m + 1, xc.device)
        L.check(L.load().mssr_ctx_pool_bwd(L.ptr(dpc), L.ptr(qc), xc.data_ptr(), xbs, xts, hc.data_ptr(), hbs, hts,
                                           w32.data_ptr(), a.data_ptr(), lengths.data_ptr(), L.ptr(dx),
                                           0 if dx is None else dx.stride(0), 0 if dx is None else dx.stride(1),
                                           dh.data_ptr(), dh.stride(0), dh.stride(1), acc.data_ptr(), B, Lx, d, m,
                                           L.dt(xc), L.stream()), "ctx_pool_bwd")
        ctx.save_for_backward(dp, q, x, h, w2, a, lengths)
        if dx is not None and dx.dtype != x.dtype:
            dx = dx.to(x.dtype)
        if dh.dtype != h.dtype:
            dh = dh.to(h.dtype)
        return dx, dh, acc[:m].view(w2.shape).to(w2.dtype), acc[m:].to(b2.dtype)

    @staticmethod
    @once_differentiable
    @_in_forward_dtype
    def backward(ctx, ddx, ddh, ddw2, ddb2):
        dp, q, x, h, w2, a, lengths = ctx.saved_tensors
        if ddx is None and ddh is None and ddw2 is None and ddb2 is None:
            return (None,) * 8
        xc, xbs, xts = _blf(cast(x))
        hc, hbs, hts = _blf(cast(h))
        B, Lx, d = xc.shape
        m = hc.shape[2]
        dev = xc.device
        dpc = None if dp is None else cast(dp).contiguous()
        qc = None if q is None else q.to(torch.float32).contiguous()
        w32 = _f32(w2.detach()).contiguous()
        X, Xbs, Xts = (None, 0, 0) if ddx is None else _blf(cast(ddx))
        H, Hbs, Hts = (None, 0, 0) if ddh is None else _blf(cast(ddh))
        W = None if ddw2 is None else ddw2.to(torch.float32).contiguous()
        beta = None if ddb2 is None else ddb2.to(torch.float32).contiguous()
        g_dp = torch.empty(B, d, dtype=xc.dtype, device=dev)
        g_x = _empty_blf(B, Lx, d, xbs < xts, xc.dtype, dev)
        g_h = None if W is None else _empty_blf(B, Lx, m, hbs < hts, hc.dtype, dev)
        g_w2 = None if H is None else zeros_f32(m, dev)
        g_a = torch.empty(B, Lx, dtype=torch.float32, device=dev)
        L.check(L.load().mssr_ctx_pool_bwd_bwd(
            L.ptr(X), Xbs, Xts, L.ptr(H), Hbs, Hts, L.ptr(W), L.ptr(beta), L.ptr(dpc), L.ptr(qc), xc.data_ptr(), xbs, xts,
            hc.data_ptr(), hbs, hts, w32.data_ptr(), a.data_ptr(), lengths.data_ptr(), g_dp.data_ptr(), g_x.data_ptr(),
            g_x.stride(0), g_x.stride(1), L.ptr(g_h), 0 if g_h is None else g_h.stride(0),
            0 if g_h is None else g_h.stride(1), L.ptr(g_w2), g_a.data_ptr(), B, Lx, d, m, L.dt(xc), L.stream()),
            "ctx_pool_bwd_bwd")
        return (None if dp is None else g_dp.to(dp.dtype), None, g_x.to(x.dtype), None if g_h is None else g_h.to(h.dtype),
                None if g_w2 is None else g_w2.view(w2.shape).to(w2.dtype), None, g_a, None)


class _CtxPool(Function):
    """ContextReps2 behind its first Linear: scores of the second Linear, length mask, softmax over time, weighted sum
    (onmt/style_generator_builder.py:250-256) in one launch; also returns the weights a [B, L] (float32), through which
    the second-order pass hands its gradient back."""

    @staticmethod
    def forward(ctx, x, h, w2, b2, lengths):
        ctx.cd = _state["dtype"]
        ctx.set_materialize_grads(False)
        xc, xbs, xts = _blf(cast(x))
        hc, hbs, hts = _blf(cast(h))
        B, Lx, d = xc.shape
        m = hc.shape[2]
        w32, b32 = _f32(w2.detach()).contiguous(), _f32(b2.detach()).contiguous()
        pooled = torch.empty(B, d, dtype=xc.dtype, device=xc.device)
        a = torch.empty(B, Lx, dtype=torch.float32, device=xc.device)
        L.check(L.load().mssr_ctx_pool_fwd(xc.data_ptr(), xbs, xts, hc.data_ptr(), hbs, hts, w32.data_ptr(), b32.data_ptr(),
                                           lengths.data_ptr(), pooled.data_ptr(), a.data_ptr(), B, Lx, d, m, L.dt(xc),
                                           L.stream()), "ctx_pool_fwd")
        ctx.save_for_backward(x, h, w2, b2, a, lengths)
        return pooled, a

    @staticmethod
    @_in_forward_dtype
    def backward(ctx, dp, da):
        x, h, w2, b2, a, lengths = ctx.saved_tensors
        if dp is None and da is None:
            return None, None, None, None, None
        dx, dh, dw2, db2 = _CtxPoolBwd.apply(dp, da, x, h, w2, b2, a, lengths)
        if _skip(w2):
            dw2 = db2 = None
        return dx, dh, dw2, db2, None


def ctx_pool(x, h, w2, b2, lengths):
    """ContextReps2's pooling of the states x [B, L, d] with h = relu(w_fn[0](x)) [B, L, m]: softmax over the first
    lengths[b] time steps of h . w2 + b2, weighted sum of x -> [B, d].  x / h may be [L, B, .] storage viewed batch first.
    Twice differentiable (onmt/style_generator_builder.py:243-256 inside the WGAN-GP penalties)."""
    L.require_cuda(x, h, w2, b2, lengths)
    if x.dim() != 3 or h.dim() != 3 or h.shape[:2] != x.shape[:2] or w2.numel() != h.shape[2] or b2.numel() != 1 \
            or lengths.shape != x.shape[:1]:
        raise ValueError("ctx_pool: x [B, L, d], h [B, L, m], w2 [1, m], b2 [1], lengths [B] expected (got %s, %s, %s, %s, %s)"
                         % (tuple(x.shape), tuple(h.shape), tuple(w2.shape), tuple(b2.shape), tuple(lengths.shape)))
    if lengths.dtype != torch.int64 or not lengths.is_contiguous():
        lengths = lengths.to(torch.int64).contiguous()
    return _CtxPool.apply(x, h, w2, b2, lengths)[0]


def second_order_active():
    return getattr(_tls, "second_order", False)


def _head_view_ok(t):
    """[B,H,L,dh] view of [B,L,H,dh]-like memory: unit feature stride, heads dh apart, 4-element aligned."""
    dh = t.shape[-1]
    return (t.stride(-1) == 1 and (t.shape[1] == 1 or t.stride(1) == dh) and t.stride(0) % 4 == 0
            and t.stride(2) % 4 == 0 and t.data_ptr() % (4 * t.element_size()) == 0)


def _as_head_view(t):
    t = cast(t)
    if _head_view_ok(t):
        return t
    return t.transpose(1, 2).contiguous().transpose(1, 2)


def _attn_desc(q, k, v, mask, p_drop, seed):
    d = L.AttnDesc()
    B, H, Lq, dh = q.shape
    d.q, d.k, d.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
    d.B, d.H, d.Lq, d.Lk, d.dh = B, H, Lq, k.shape[2], dh
    d.q_bs, d.q_ls, d.k_bs, d.k_ls, d.v_bs, d.v_ls = (q.stride(0), q.stride(2), k.stride(0), k.stride(2), v.stride(0),
                                                     v.stride(2))
    if mask is not None:
        d.mask, d.mask_bs = mask.data_ptr(), (mask.stride(0) if B > 1 else 0)
        d.mask_qs = mask.stride(1) if mask.shape[1] > 1 else 0
    d.p_drop, d.seed, d.dtype = p_drop, seed, L.dt(q)
    return d


class _FusedAttention(Function):
    """softmax(mask(q k^T / sqrt(dh))) v in one kernel per direction (csrc/attention.hip)."""

    @staticmethod
    def forward(ctx, q, k, v, mask, p_drop, want_probs):
        ctx.cd = _state["dtype"]
        qc, kc, vc = _as_head_view(q), _as_head_view(k), _as_head_view(v)
        L.require_cuda(qc, kc, vc)
        B, H, Lq, dh = qc.shape
        Lk = kc.shape[2]
        m = None
        if mask is not None:
            m = mask if mask.stride(-1) == 1 else mask.contiguous()
            m = m.view(torch.uint8) if m.dtype == torch.bool else (m if m.dtype == torch.uint8 else
                                                                  (m != 0).view(torch.uint8))
            if m.dim() != 3 or m.shape[0] != B or m.shape[2] != Lk or m.shape[1] not in (1, Lq):
                raise ValueError("mask shape %s does not fit scores [%d,%d,%d,%d]" % (tuple(mask.shape), B, H, Lq, Lk))
        seed = next_seed() if p_drop > 0.0 else 0
        out = torch.empty(B, Lq, H, dh, dtype=qc.dtype, device=qc.device)
        probs = torch.empty(B, H, Lq, Lk, dtype=qc.dtype, device=qc.device) if want_probs else None
        d = _attn_desc(qc, kc, vc, m, p_drop, seed)
        L.check(L.load().mssr_attention_fwd(d, out.data_ptr(), Lq * H * dh, H * dh, L.ptr(probs), L.stream()),
                "attention_fwd")
        ctx.save_for_backward(qc, kc, vc, m)
        ctx.p_drop, ctx.seed = p_drop, seed
        ctx.in_dtypes = (q.dtype, k.dtype, v.dtype)
        if probs is not None:
            ctx.mark_non_differentiable(probs)
        return out.permute(0, 2, 1, 3), probs

    @staticmethod
    @once_differentiable
    @_in_forward_dtype
    def backward(ctx, dctx, _dprobs):
        qc, kc, vc, m = ctx.saved_tensors
        B, H, Lq, dh = qc.shape
        Lk = kc.shape[2]
        dc = cast(dctx).permute(0, 2, 1, 3)
        if not dc.is_contiguous():
            dc = dc.contiguous()
        dq = torch.empty(B, Lq, H, dh, dtype=qc.dtype, device=qc.device)
        dk = torch.empty(B, Lk, H, dh, dtype=qc.dtype, device=qc.device)
        dv = torch.empty(B, Lk, H, dh, dtype=qc.dtype, device=qc.device)
        d = _attn_desc(qc, kc, vc, m, ctx.p_drop, ctx.seed)
        L.check(L.load().mssr_attention_bwd(d, dc.data_ptr(), Lq * H * dh, H * dh, dq.data_ptr(), Lq * H * dh, H * dh,
                                            dk.data_ptr(), Lk * H * dh, H * dh, dv.data_ptr(), Lk * H * dh, H * dh,
                                            L.stream()), "attention_bwd")
        tq, tk, tv = ctx.in_dtypes
        return (dq.permute(0, 2, 1, 3).to(tq), dk.permute(0, 2, 1, 3).to(tk), dv.permute(0, 2, 1, 3).to(tv), None, None,
                None)


def attention(q, k, v, mask, p_drop, training, want_probs=True):
    """softmax(mask(q k^T / sqrt(dh))) v on [B,H,L,dh] views; returns (context as a [B,H,Lq,dh]
    view of [B,Lq,H,dh] memory, probabilities [B,H,Lq,Lk] before dropout) -- onmt/modules/multi_headed_attn.py:187-205.
    Short sequences run the fused kernel; inside ``second_order()`` or beyond its limits the same
    arithmetic is composed from matmul / softmax / dropout."""
    dh = q.shape[-1]
    p = float(p_drop) if training else 0.0
    if not second_order_active():
        cd = _state["dtype"]
        code = L.F32 if cd == torch.float32 else L.BF16
        needs_bwd = torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)
        if L.load().mssr_attention_fits(q.shape[2], k.shape[2], dh, code, int(needs_bwd)):
            return _FusedAttention.apply(q, k, v, mask, p, want_probs)
    scores = matmul(q, k, tb=True)
    probs = softmax(scores, mask, 1.0 / math.sqrt(dh))
    pd = dropout(probs, p_drop, training)
    return matmul(pd, v, tb=False, heads_last=True), probs
