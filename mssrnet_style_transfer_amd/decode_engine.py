"""Host side of the step-wise soft-decode engine (csrc/decode.hip, include/mssr_hip.h).

``soft_decode(model, cfg, memory_bank, src_lengths, steps)`` is the differentiable decoding of
the reference -- TrainTranslator.decode_batch_ori (onmt/train_and_translate.py:202-231):
one-hot ``<s>`` -> TransformerDecoder.forward(step=s) (onmt/decoders/transformer.py:289-337)
-> generator Linear -> softmax(/temperature) -> fed back as the next soft input -- as ONE
autograd node.  The forward is a single C call that enqueues every kernel of all steps; the
backward is a single C call for the reverse sweep plus one product per parameter that
contracts the per-step gradient slabs over all (step, sentence) rows at once.

The K/V "cache" is the step-major slab of packed projections, so nothing is concatenated as the
sequence grows (the reference re-copies the cache every step, multi_headed_attn.py:139-148).
Packed / transposed copies of the decoder weights in the compute dtype are rebuilt only when
a parameter's version changes (i.e. after an optimizer step).
"""
import ctypes
import math
import os
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L
from . import ops
from .weight_pack import JobTable

N_HEAD_PARAMS = 5
N_LAYER_PARAMS = 26
# indices into a layer's parameter list (see _layer_params)
(_WQ, _BQ, _WK, _BK, _WV, _BV, _WSO, _BSO, _WCK, _BCK, _WCV, _BCV, _WCQ, _BCQ, _WCO, _BCO, _W1, _B1, _W2, _B2,
 _LN1G, _LN1B, _LN2G, _LN2B, _LNFG, _LNFB) = range(N_LAYER_PARAMS)
# slab kinds (csrc/decode.hip)
K_N1, K_CTX1, K_H1, K_N2, K_QC, K_CTX2, K_H2, K_N3 = range(8)
G_DN1, G_DSO, G_DN2, G_DCQ, G_DCO, G_DN3, G_DF2 = range(7)


def _layer_params(layer):
    sa, ca, ff = layer.self_attn, layer.context_attn, layer.feed_forward
    return [sa.linear_query.weight, sa.linear_query.bias, sa.linear_keys.weight, sa.linear_keys.bias,
            sa.linear_values.weight, sa.linear_values.bias, sa.final_linear.weight, sa.final_linear.bias,
            ca.linear_keys.weight, ca.linear_keys.bias, ca.linear_values.weight, ca.linear_values.bias,
            ca.linear_query.weight, ca.linear_query.bias, ca.final_linear.weight, ca.final_linear.bias,
            ff.w_1.weight, ff.w_1.bias, ff.w_2.weight, ff.w_2.bias,
            layer.layer_norm_1.weight, layer.layer_norm_1.bias, layer.layer_norm_2.weight, layer.layer_norm_2.bias,
            ff.layer_norm.weight, ff.layer_norm.bias]


def decoder_params(model):
    """[E, lnF_g, lnF_b, W_gen, b_gen] + 26 parameters per decoder layer, in a fixed order."""
    dec, gen = model.decoder, model.generator[0]
    out = [dec.embeddings.word_lut.weight, dec.layer_norm.weight, dec.layer_norm.bias, gen.weight, gen.bias]
    for layer in dec.transformer_layers:
        out += _layer_params(layer)
    return out


class _Pack:
    """Decoder + generator weights as the engine wants them (compute dtype, packed, transposed, LN-folded).  The
    buffers are allocated once; ``refresh()`` recomputes all of them from the fp32 parameters in one launch
    (weight_pack.JobTable -> csrc/pack.hip)."""

    def __init__(self, params, n_layers, cd, head=True):
        """``head`` = False: params = [lnF_g, lnF_b] + the layers' 26 each -- the decoder alone, without the embedding
        table and the generator (decoder_stack.py, when no model has claimed the decoder yet)."""
        if head:
            E, lnFg, lnFb, Wg, bg = params[:N_HEAD_PARAMS]
            first = N_HEAD_PARAMS
        else:
            (lnFg, lnFb), first = params[:2], 2
            E = Wg = bg = None
        lp = [params[first + i * N_LAYER_PARAMS: first + (i + 1) * N_LAYER_PARAMS] for i in range(n_layers)]
        NL = n_layers
        d = lp[0][_WQ].shape[1]
        V = E.shape[0] if head else 0
        ff = lp[0][_W1].shape[0]
        Vp = (V + 31) // 32 * 32
        dev = lnFg.device
        self.NL, self.V, self.Vp, self.d, self.ff, self.cd, self.head = NL, V, Vp, d, ff, cd, head
        e = lambda *shape, dtype=cd: torch.empty(*shape, dtype=dtype, device=dev)  # noqa: E731
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
        for name, n_out, n_in in (("qkv", 3 * d, d), ("so", d, d), ("cq", d, d), ("co", d, d), ("1", ff, d), ("2", d, ff)):
            setattr(self, "w_" + name, e(NL, n_out, n_in))
            setattr(self, "wt_" + name, e(NL, n_in, n_out))
            setattr(self, "b_" + name, f(NL, n_out))
        self.w_ckv, self.b_ckv = e(NL * 2 * d, d), f(NL * 2 * d)           # all layers' context keys|values: one product
        self.wt_ckv = e(NL, d, 2 * d)
        for name in ("ln1_g", "ln1_b", "ln2_g", "ln2_b", "lnf_g", "lnf_b"):
            setattr(self, name, f(NL, d))
        if head:
            self.emb, self.w_gen = e(V, d), e(V, d)
            self.emb_t = torch.zeros(d, Vp, dtype=cd, device=dev)          # columns V..Vp stay zero
            self.wt_gen = torch.zeros(d, Vp, dtype=cd, device=dev)
            self.b_gen = f(V)
        self.lnF_g, self.lnF_b = f(d), f(d)
        # LayerNorm folded into the Linear behind it (bf16 skinny kernel, mssr_gemm_desc.ln_colsum):
        # W' = gamma * W (rounded to bf16), colsum(W') of the ROUNDED values, bias' = W beta + b
        self.fuse_ln = cd == torch.bfloat16 and d % 32 == 0 and d <= 2048
        names = ["w_qkv", "wt_qkv", "w_so", "wt_so", "w_cq", "wt_cq", "w_co", "wt_co", "w_1", "wt_1", "w_2",
                 "wt_2", "b_qkv", "b_so", "b_cq", "b_co", "b_1", "b_2", "ln1_g", "ln1_b", "ln2_g", "ln2_b",
                 "lnf_g", "lnf_b"]
        if self.fuse_ln:
            for name, n_out in (("qkv", 3 * d), ("cq", d), ("1", ff)):
                setattr(self, "wg_" + name, e(NL, n_out, d))
                setattr(self, "cs_" + name, f(NL, n_out))
                setattr(self, "bf_" + name, f(NL, n_out))
            if head:
                self.wg_gen, self.cs_gen, self.bf_gen = e(V, d), f(V), f(V)
            names += ["wg_qkv", "wg_cq", "wg_1", "cs_qkv", "cs_cq", "cs_1", "bf_qkv", "bf_cq", "bf_1"]

        jt = self.jobs = JobTable(cd, dev)

        def linear(l, name, w_idx, b_idx, off, ln=None):
            """rows [off, off + n) of packed weight `name` of layer l <- parameter w_idx (+ bias b_idx); ln = (gamma
            index, beta index): also the LayerNorm-folded copy"""
            W, b = lp[l][w_idx], lp[l][b_idx]
            n = W.shape[0]
            kw = {}
            if ln is not None and self.fuse_ln:
                kw = dict(dst_g=getattr(self, "wg_" + name)[l, off:off + n], gamma=lp[l][ln[0]], beta=lp[l][ln[1]], bias=b,
                          colsum=getattr(self, "cs_" + name)[l, off:off + n], bias_f=getattr(self, "bf_" + name)[l, off:off + n])
            jt.add(W, dst=getattr(self, "w_" + name)[l, off:off + n], dst_t=getattr(self, "wt_" + name)[l][:, off:off + n], **kw)
            jt.add(b, dst_f=getattr(self, "b_" + name)[l, off:off + n])

        for l in range(NL):
            linear(l, "qkv", _WQ, _BQ, 0, (_LN1G, _LN1B))
            linear(l, "qkv", _WK, _BK, d, (_LN1G, _LN1B))
            linear(l, "qkv", _WV, _BV, 2 * d, (_LN1G, _LN1B))
            linear(l, "so", _WSO, _BSO, 0)
            linear(l, "cq", _WCQ, _BCQ, 0, (_LN2G, _LN2B))
            linear(l, "co", _WCO, _BCO, 0)
            linear(l, "1", _W1, _B1, 0, (_LNFG, _LNFB))
            linear(l, "2", _W2, _B2, 0)
            for i, (w_idx, b_idx) in enumerate(((_WCK, _BCK), (_WCV, _BCV))):
                r0 = (2 * l + i) * d
                jt.add(lp[l][w_idx], dst=self.w_ckv[r0:r0 + d], dst_t=self.wt_ckv[l][:, i * d:(i + 1) * d])
                jt.add(lp[l][b_idx], dst_f=self.b_ckv[r0:r0 + d])
            for name, idx in (("ln1_g", _LN1G), ("ln1_b", _LN1B), ("ln2_g", _LN2G), ("ln2_b", _LN2B), ("lnf_g", _LNFG),
                              ("lnf_b", _LNFB)):
                jt.add(lp[l][idx], dst_f=getattr(self, name)[l])
        if head:
            jt.add(E, dst=self.emb, dst_t=self.emb_t[:, :V])
            kw = {}
            if self.fuse_ln:
                kw = dict(dst_g=self.wg_gen, gamma=lnFg, beta=lnFb, bias=bg, colsum=self.cs_gen, bias_f=self.bf_gen)
            jt.add(Wg, dst=self.w_gen, dst_t=self.wt_gen[:, :V], **kw)
            jt.add(bg, dst_f=self.b_gen)
        jt.add(lnFg, dst_f=self.lnF_g)
        jt.add(lnFb, dst_f=self.lnF_b)
        jt.finish()

        arr = (L.DecLayer * NL)()
        for l in range(NL):
            for name in names:
                setattr(arr[l], name, getattr(self, name)[l].data_ptr())
        self.layers = arr
        # The transposed copies double as the k-contiguous operand of dX = dY W on the MODULE path (the
        # teacher-forced decoder of the same model in the reconstruct phase): ops._Matmul finds them by parameter.
        self._transposed = []
        for l in range(NL):
            for idx, wt, lo in ((_WQ, self.wt_qkv, 0), (_WK, self.wt_qkv, d), (_WV, self.wt_qkv, 2 * d),
                                (_WSO, self.wt_so, 0), (_WCQ, self.wt_cq, 0), (_WCO, self.wt_co, 0),
                                (_W1, self.wt_1, 0), (_W2, self.wt_2, 0), (_WCK, self.wt_ckv, 0), (_WCV, self.wt_ckv, d)):
                w_p = lp[l][idx]
                if isinstance(w_p, torch.nn.Parameter):
                    self._transposed.append((w_p, wt[l][:, lo:lo + w_p.shape[0]]))
        self.refresh()

    def refresh(self):
        """Recompute every buffer from the current parameter values (one launch) and renew the transposed-weight
        registrations, which are valid for one parameter version."""
        self.jobs.run()
        for w_p, wt in self._transposed:
            ops.register_transposed(w_p, wt)


_pack_cache = {}


def _pack_version(cd, params):
    """What a cached pack was built from: compute dtype, every parameter's in-place version, the
    ``ops.params_changed`` epochs (fused optimizers do not bump ``_version``) and EVERY parameter's
    storage address (``module.to`` / ``load_state_dict`` into fresh tensors, or another model that
    happens to sit at a recycled ``id``)."""
    return (cd, tuple(p._version for p in params), sum(ops.param_epoch(p) for p in params),
            tuple(p.data_ptr() for p in params))


def cached_pack(cache, owner, params, build):
    """The pack of ``owner`` for the current parameter values.  Entries are keyed by ``id(owner)`` but only trusted
    while the weak reference stored beside them still resolves to ``owner`` itself, and they are dropped (with the GPU
    memory of the pack) when the owner is garbage collected.  New VALUES in the same tensors (an optimizer step)
    refresh the pack's buffers in place -- one launch -- unless an autograd graph that was built on them is still
    waiting for its backward (``JobTable.pin``): then a fresh pack is built and the old one stays with that graph."""
    cd = ops.compute_dtype()
    ver = _pack_version(cd, params)
    key = id(owner)
    hit = cache.get(key)
    if hit is not None and hit[2]() is owner:
        if hit[0] == ver:
            return hit[1]
        old = hit[1]
        if old.cd == cd and hit[0][3] == ver[3] and not old.jobs.pinned():
            old.refresh()
            cache[key] = (ver, old, hit[2])
            return old
    pack = build()
    ref = weakref.ref(owner)
    cache[key] = (ver, pack, ref)

    def _evict(cache=cache, key=key, ref=ref):
        held = cache.get(key)
        if held is not None and held[2] is ref:
            del cache[key]

    weakref.finalize(owner, _evict)
    return pack


_decoder_owner = weakref.WeakValueDictionary()      # id(decoder) -> the model whose pack covers it


def get_pack(model, params):
    """The packed weights of ``model``'s decoder, rebuilt when any parameter changed in place."""
    cd = ops.compute_dtype()
    _decoder_owner[id(model.decoder)] = model
    return cached_pack(_pack_cache, model, params,
                       lambda: _Pack(params, len(model.decoder.transformer_layers), cd))


def model_of(decoder):
    """The model that ran a soft / greedy decode with this decoder (its pack already holds every layer's weights), or
    None."""
    m = _decoder_owner.get(id(decoder))
    return m if m is not None and getattr(m, "decoder", None) is decoder else None


def clear_pack_cache():
    _pack_cache.clear()
    ops._wt_registry.clear()        # the registered transposed copies are views of the packs


class _Spec:
    """Everything about one decode call that is not a tensor autograd tracks."""
    __slots__ = ("pack", "B", "Ls", "T", "heads", "training", "p_drop", "p_attn", "temperature", "seed", "src_mask",
                 "bos", "pe", "save", "eval_rows", "forced", "pad", "groups")


def _bgemm_t(dy, x, seg=None):
    """sum over rows of dy[.., R, n]^T x[.., R, k] -> float [.., n, k] (a weight gradient).  seg = (T, B, E0): the
    rows are T blocks of B, and only rows E0.. of every block take part (the others carry no gradient)."""
    if seg is not None and seg[2] > 0 and dy.dtype == torch.bfloat16 and seg[1] - seg[2] >= 32:
        T, B, E0 = seg
        return ops.gemm(dy[..., E0:, :].transpose(-1, -2), x[..., E0:, :], False, torch.float32,
                        k_seg=(B - E0, B, T))
    return ops.gemm(dy.transpose(-1, -2), x, False, torch.float32)


class _Zeros:
    """One zero-filled fp32 arena for all the small gradient outputs of an engine backward (bias and LayerNorm
    gradients are accumulated with float atomics into zeroed memory): one fill instead of one per output."""

    def __init__(self, numel, device):
        self.buf = torch.zeros(int(numel), dtype=torch.float32, device=device)
        self.off = 0

    def take(self, *shape):
        n = 1
        for v in shape:
            n *= int(v)
        n4 = (n + 3) // 4 * 4                       # every output starts on a 16-byte boundary
        if self.off + n4 > self.buf.numel():
            raise RuntimeError("gradient arena too small")
        t = self.buf[self.off:self.off + n].view(*shape)
        self.off += n4
        return t


def _bsum(dy, out=None):
    """column sums of dy [NL, R, n] (bias gradients of all layers) -> float [NL, n]; one launch for all layers.
    ``out``: a ZEROED [NL, n] float view (``_Zeros.take``) to add into; None: allocated and zeroed here."""
    NL, R, n = dy.shape
    if dy.stride(-1) != 1 or dy.stride(-2) < n:
        dy = dy.contiguous()
    acc = out is not None
    if out is None:
        out = torch.empty(NL, n, dtype=torch.float32, device=dy.device)
    L.check(L.load().mssr_colsum_batched(dy.data_ptr(), out.data_ptr(), NL, dy.stride(0) if NL > 1 else 0, R, n,
                                         dy.stride(-2), int(acc), L.dt(dy), L.stream()), "colsum_batched")
    return out


def _ln_param_grads(dy, x, mean, rstd, out, zeroed=False):
    """LayerNorm weight / bias gradients of one LayerNorm KIND over all layers in one launch: dy, x [NL, R, d] views
    (rows contiguous, any layer stride), mean / rstd [NL, R] views of the statistics, out [NL, 2, d] float view
    (gamma | beta gradients; ``zeroed``: it holds zeros already, no fill here).  2-D operands are one layer."""
    if dy.dim() == 2:
        dy, x, mean, rstd, out = dy[None], x[None], mean[None], rstd[None], out[None]
    NL, R, d = dy.shape
    ok = (dy.stride(2) == 1 and dy.stride(1) == d and x.stride(2) == 1 and x.stride(1) == d and mean.stride(1) == 1
          and rstd.stride(1) == 1 and out.stride(2) == 1 and out.stride(1) == d and x.dtype == dy.dtype)
    if NL > 1:
        ok = ok and mean.stride(0) == rstd.stride(0)
    if not ok:
        raise ValueError("ln_param_grads: operands must be row-contiguous slabs with a common layer stride")
    L.check(L.load().mssr_layernorm_param_grads(
        dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), out.data_ptr(), NL,
        dy.stride(0) if NL > 1 else 0, x.stride(0) if NL > 1 else 0, mean.stride(0) if NL > 1 else 0,
        out.stride(0) if NL > 1 else 2 * d, R, d, int(zeroed), L.dt(dy), L.stream()), "layernorm_param_grads")


def _colsum(x):
    """float column sums of a [rows, n] matrix (row stride allowed)."""
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    L.check(L.load().mssr_colsum(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], x.stride(0), 0, L.dt(x),
                                 L.stream()), "colsum")
    return out



class _SlabPool:
    """Fixed homes for the slabs of a decode plan.

    mssr_decode_fwd / _bwd replay a recorded hipGraph when a plan's CONTENTS repeat (include/mssr_hip.h), and the
    contents are mostly buffer addresses.  torch's caching allocator returns the large slabs to the same place step
    after step, but not the medium and small ones (measured: stats, nf, logits, src_mask, dX, d_e, dnf moved on every
    step of the WGAN loop), so no plan ever matched.  The pool hands out the SAME tensor for the same (name, shape,
    dtype, device, stream) whenever the previous holder has returned it: a ``lease`` collects what one decode call
    took and gives everything back when it is dropped (end of the backward, or when autograd frees the node).
    Two decodes in flight at once (both saved for backward) simply hold two different tensors of a kind.
    Reuse is ordered by the stream (part of the key): the next holder's kernels are enqueued behind the previous
    holder's.  MSSR_SLAB_POOL_GB (default 96) bounds what the pool keeps while idle; beyond it the free lists are
    dropped (back to the allocator)."""

    def __init__(self):
        self.free = {}
        self.held_bytes = 0
        self.cap = int(float(os.environ.get("MSSR_SLAB_POOL_GB", "96")) * (1 << 30))

    def lease(self, device):
        return _Lease(self, device)

    def clear(self):
        self.free.clear()
        self.held_bytes = 0


class _Lease:
    def __init__(self, pool, device):
        self.pool, self.device, self.taken = pool, device, []
        self.stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0

    def take(self, name, shape, dtype, zero=False):
        key = (name, tuple(int(v) for v in shape), dtype, self.device, self.stream)
        lst = self.pool.free.get(key)
        if lst:
            t = lst.pop()
            self.pool.held_bytes -= t.numel() * t.element_size()
        else:
            t = torch.empty(key[1], dtype=dtype, device=self.device)
        if zero is True:
            t.zero_()
        elif zero is not False:
            _zero_once(t, "all", zero, lambda u: u.zero_())
        self.taken.append((key, t))
        return t

    def release(self):
        pool, taken, self.taken = self.pool, self.taken, []
        for key, t in taken:
            pool.free.setdefault(key, []).append(t)
            pool.held_bytes += t.numel() * t.element_size()
        if pool.held_bytes > pool.cap:
            pool.clear()

    def __del__(self):
        try:
            self.release()
        except Exception:           # interpreter shutdown
            pass


_slabs = _SlabPool()

# tools/sweep_bench.py: (kind, start event, end event) around every sweep call while a list is installed here
sweep_events = None


def _timed_sweep(kind, fn, plan):
    if sweep_events is None:
        return L.check(fn(ctypes.byref(plan), L.stream()), kind)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(fn(ctypes.byref(plan), L.stream()), kind)
    e1.record()
    sweep_events.append((kind, e0, e1))


def clear_slab_pool():
    _slabs.clear()


def _make_plan(spec, mem, greedy=False):
    """Slabs + mssr_decode_plan for one decode over ``mem`` [B, Ls, d] (forward part).  ``greedy``: a forward-only
    plan that hands token ids from step to step -- no probability slab, one input slot (the one-hot <s>)."""
    pk = spec.pack
    cd, dev = pk.cd, mem.device
    B, Ls, T, NL, d, ff, V, Vp = spec.B, spec.Ls, spec.T, pk.NL, pk.d, pk.ff, pk.V, pk.Vp
    L.require_cuda(mem)
    memc = ops.cast(mem).detach().contiguous().view(B * Ls, d)
    lease = _slabs.lease(dev)           # every slab from its fixed home (recorded sweeps replay only on equal plans)
    e = lambda name, *shape, dtype=cd, zero=False: lease.take(name, shape, dtype, zero)  # noqa: E731
    # context keys|values of the memory bank for every layer in one product: [B*Ls, NL*2d]
    mem_kv = ops.gemm(memc, pk.w_ckv, True, cd, bias=pk.b_ckv, out=e("mem_kv", B * Ls, pk.w_ckv.shape[0]))
    Ts = T if spec.save else 1
    src_mask = e("src_mask", B, Ls, dtype=torch.uint8)
    src_mask.copy_(spec.src_mask)
    bufs = dict(
        lease=lease, mem=memc, mem_kv=mem_kv, src_mask=src_mask,
        in_all=e("in_all", 1 if greedy else T + 1, B, Vp, zero=_regime(spec, greedy)),
        # the probabilities leave with the caller: not pooled (large: the allocator keeps their place)
        probs=torch.empty(1 if greedy else T, 1 if greedy else B, V, dtype=torch.float32, device=dev),
        X=e("X", NL + 1, Ts * B, d), qkv=e("qkv", NL, T * B, 3 * d), actD=e("actD", NL, 8, Ts * B, d),
        actF=e("actF", NL, Ts * B, ff), stats=e("stats", 3 * NL + 1, 2, Ts * B, dtype=torch.float32),
        nf=e("nf", Ts * B, d), logits=e("logits", B, Vp), tmp=e("tmp", B, d), ws=ops._workspace(dev))
    bufs["in_all"][0, :, spec.bos] = 1
    forced = None if greedy else getattr(spec, "forced", None)
    if forced is not None:          # teacher-forced sentences [0, F): one-hot target words as the inputs of steps < Tf
        Tf, F = forced.shape
        rows = bufs["in_all"][:Tf, :F]
        rows.zero_()                # the host writes these rows itself, other words on every call
        rows.scatter_(2, forced.unsqueeze(-1), 1)
        smask = e("self_mask", B, T, dtype=torch.uint8, zero=True)
        smask[:F, :Tf] = forced.t().eq(spec.pad)
        bufs["self_mask"] = smask
    plan = L.DecodePlan()
    plan.dtype = L.F32 if cd == torch.float32 else L.BF16
    plan.B, plan.d, plan.heads, plan.ff, plan.n_layers = B, d, spec.heads, ff, NL
    plan.V, plan.Vp, plan.Ls, plan.T = V, Vp, Ls, T
    plan.save, plan.training = int(spec.save), int(spec.training)
    plan.n_chunks = max(1, min(8, int(os.environ.get("MSSR_DECODE_CHUNKS", "1")), B))
    plan.p_drop, plan.p_attn, plan.temperature, plan.seed = spec.p_drop, spec.p_attn, spec.temperature, spec.seed
    plan.layers = pk.layers
    plan.fuse_ln = int(pk.fuse_ln and os.environ.get("MSSR_DECODE_FUSE_LN", "1") != "0")
    plan.eval_rows = plan.bwd_row0 = int(spec.eval_rows)
    if spec.eval_rows:
        plan.n_chunks = 1
    for name in ("emb_t", "emb", "w_gen", "wt_gen", "b_gen", "lnF_g", "lnF_b") + (
            ("wg_gen", "cs_gen", "bf_gen") if pk.fuse_ln else ()):
        setattr(plan, name, getattr(pk, name).data_ptr())
    plan.pe = spec.pe.data_ptr()
    plan.src_mask = bufs["src_mask"].data_ptr()
    for name in ("mem_kv", "in_all", "probs", "X", "qkv", "actD", "actF", "stats", "nf", "logits", "tmp"):
        setattr(plan, name, bufs[name].data_ptr())
    plan.gemm_ws, plan.gemm_ws_bytes = bufs["ws"].data_ptr(), bufs["ws"].numel() * 4
    if forced is not None:
        plan.forced_rows, plan.forced_steps = forced.shape[1], forced.shape[0]
        plan.self_mask = bufs["self_mask"].data_ptr()
        plan.n_chunks = 1
    return plan, bufs


def _late_regime(spec, n_chunks):
    """(late_step, late_row0) of a sweep whose row groups end at different steps: the sentences [0, late_row0) need no
    step >= late_step (their groups end there, or they are eval rows of such a group) and all the others run to the
    end -- the late steps then cover the suffix only (mssr_decode_plan.late_step / late_row0); (0, 0): no such split."""
    groups = getattr(spec, "groups", None)
    if groups is None or n_chunks != 1 or os.environ.get("MSSR_DECODE_LATE", "1") == "0":
        return 0, 0
    B, T = spec.B, spec.T
    need = [0] * B
    for te, b0, b1 in groups:
        for b in range(b0, b1):
            need[b] = max(need[b], te)
    r = 0
    while r < B - 1 and need[r] < T:
        r += 1
    if r > 0 and need[r] == T and all(n == T for n in need[r:]):
        return max(need[:r]), r
    return 0, 0


def pk_d(spec):
    return spec.pack.d


def _zero_once(t, site, sig, fill):
    """Zero the part of the pooled slab ``t`` that a sweep of regime ``sig`` never writes -- unless the slab's last
    use zeroed it for the same regime: a pooled slab keeps its place and its contents, the sweep's kernels write the
    same (step, sentence) rows and columns on every call of one regime, and nothing else writes the slab, so what was
    zero stays zero.  (Before: 21 fills per WGAN step, 0.41 ms, the two vocabulary-wide slabs 134 MB each.)"""
    tags = t.__dict__.setdefault("_mssr_zeroed", {})
    if tags.get(site) != sig:
        tags.clear()                 # another regime may have written where this site's zeros were
        fill(t)
        tags[site] = sig


def _regime(spec, greedy=False):
    """What decides which rows / columns of the slabs a sweep writes (beyond their shapes, which key the pool)."""
    forced = None if greedy else getattr(spec, "forced", None)
    groups = getattr(spec, "groups", None)
    return (bool(greedy), None if forced is None else tuple(forced.shape), int(spec.eval_rows), bool(spec.save), int(spec.bos),
            None if groups is None else tuple(tuple(int(v) for v in g) for g in groups),
            os.environ.get("MSSR_DECODE_LATE", "1"), os.environ.get("MSSR_DECODE_LATE_FWD", "1"))


class _SoftDecode(Function):
    @staticmethod
    def forward(ctx, spec, mem, *params):
        lib = L.load()
        plan, bufs = _make_plan(spec, mem)
        late_step, late_row0 = _late_regime(spec, plan.n_chunks)
        if late_row0 and os.environ.get("MSSR_DECODE_LATE_FWD", "1") != "0":
            plan.late_step, plan.late_row0, plan.late_fwd = late_step, late_row0, 1
            if spec.save:       # rows the late forward steps skip: finite (zero) where the contractions read them
                T, B = spec.T, spec.B
                sig = _regime(spec) + (late_step, late_row0)
                for name in ("X", "actD", "actF", "nf"):
                    _zero_once(bufs[name], "late_fwd", sig,
                               lambda t: t.view(-1, T, B, t.shape[-1])[:, late_step:, :late_row0].zero_())
                _zero_once(bufs["qkv"], "late_fwd", sig,                                          # sentence-major slab
                           lambda t: t.view(-1, B, T, 3 * pk_d(spec))[:, :late_row0, late_step:].zero_())
                _zero_once(bufs["stats"], "late_fwd", sig, lambda t: t.view(-1, T, B)[:, late_step:, :late_row0].zero_())
        _timed_sweep("decode_fwd", lib.mssr_decode_fwd, plan)
        probs = bufs.pop("probs")
        if spec.save:
            ctx.spec, ctx.plan, ctx.bufs = spec, plan, bufs
            ctx.pin = spec.pack.jobs.pin()      # the backward reads the pack: no in-place refresh until then
            ctx.mem_meta = (mem.shape, mem.dtype)
        groups = getattr(spec, "groups", None)
        if groups is None:
            if spec.save:
                ctx.save_for_backward(probs)    # read by the softmax backward of every step
            return probs
        if spec.save:
            ctx.probs = probs                   # not an output itself (the outputs are views of it): no reference cycle
        # the caller's row groups as separate outputs (views of the slab): their gradients arrive one by one and the
        # backward lays them into its slab itself, instead of autograd padding every slice of one big output with
        # zeros to the full [T, B, V] and adding the padded copies up (~2 GB of traffic per WGAN step)
        return tuple(probs[:te, b0:b1] for te, b0, b1 in groups)

    @staticmethod
    @once_differentiable
    def backward(ctx, *douts):
        spec, plan, bufs = ctx.spec, ctx.plan, ctx.bufs
        if bufs is None:
            raise RuntimeError("soft decode: backward called twice (activations were released)")
        probs = ctx.saved_tensors[0] if getattr(spec, "groups", None) is None else ctx.probs
        assert probs.data_ptr() == plan.probs
        pk = spec.pack
        lib = L.load()
        cd, dev = pk.cd, probs.device
        B, Ls, T, NL, d, ff, V, Vp = spec.B, spec.Ls, spec.T, pk.NL, pk.d, pk.ff, pk.V, pk.Vp
        H = spec.heads
        R = T * B
        e = lambda *shape, dtype=cd: torch.empty(*shape, dtype=dtype, device=dev)  # noqa: E731
        lease = bufs["lease"]           # the gradient slabs join the forward's lease: fixed homes, returned together
        pe = lambda name, *shape, dtype=cd, zero=False: lease.take(name, shape, dtype, zero)  # noqa: E731
        E0 = spec.eval_rows         # sentences [0, E0) ran in eval mode and carry no gradient: the sweep skips them
        # the sweep accumulates into dprobs in place (step s adds the input gradient of step s+1): a slab of its own
        groups = getattr(spec, "groups", None)
        dprobs = pe("dprobs", T, B, V, dtype=torch.float32)
        if groups is None:
            dprobs[:, E0:].copy_(douts[0][:, E0:])
        else:
            dprobs[:, E0:].zero_()
            seen = []
            for (te, b0, b1), dg in zip(groups, douts):
                if dg is None or b1 <= E0:
                    continue
                b0 = max(b0, E0)
                dg = dg[:, dg.shape[1] - (b1 - b0):]
                if any(not (b1 <= c0 or c1 <= b0) for _, c0, c1 in seen):
                    dprobs[:te, b0:b1].add_(dg)
                else:
                    dprobs[:te, b0:b1].copy_(dg)
                seen.append((te, b0, b1))
        g = dict(dprobs=dprobs,
                 dlogits=pe("dlogits", R, Vp, zero=_regime(spec)), dX=pe("dX", 2, B, d), dqkv=pe("dqkv", NL, R, 3 * d),
                 dactD=pe("dactD", NL, 7, R, d), dactF=pe("dactF", NL, R, ff), dctx=pe("dctx", NL, 2, R, d),
                 ds_self=pe("ds_self", NL, T, T, B * H, 2, dtype=torch.float32),      # [key][step][sentence, head] pairs
                 ds_ctx=pe("ds_ctx", NL, 2, T, B * H, Ls, dtype=torch.float32),
                 d_e=pe("d_e", R, d), dnf=pe("dnf", R, d))
        for name, t in g.items():
            setattr(plan, name, t.data_ptr())
        # sentences whose groups end before the sweep does carry no gradient at the later steps: when they are a
        # prefix of the differentiable rows, the late backward steps skip them (mssr_decode_plan.late_step / late_row0)
        late_step, late_row0 = _late_regime(spec, plan.n_chunks)
        sig = _regime(spec) + (late_step, late_row0)

        def skipped(name):          # what this regime's sweep leaves unwritten in a [.., T, B, n] gradient slab
            def fill(t):
                if name == "ds_self":       # [layer][key][step][sentence][head, 2]
                    if late_row0:
                        t.view(NL, T, T, B, H * 2)[:, :, late_step:, E0:late_row0].zero_()
                    return
                v = t.view(NL * 2, T, B, -1) if name == "ds_ctx" else t.view(-1, T, B, t.shape[-1])
                if late_row0:
                    v[:, late_step:, E0:late_row0].zero_()
                if E0 and name in ("dqkv", "dactD", "dactF", "d_e", "dnf"):
                    # the slabs that are contracted over ALL (step, sentence) rows hold zeros at the eval-mode sentences
                    v[:, :, :E0].zero_()
            return fill
        if late_row0:
            plan.late_step, plan.late_row0 = late_step, late_row0
        if late_row0 or E0:
            for name in ("dqkv", "dactD", "dactF", "dctx", "d_e", "dnf", "ds_self", "ds_ctx"):
                _zero_once(g[name], "skipped", sig, skipped(name))
        _timed_sweep("decode_bwd", lib.mssr_decode_bwd, plan)

        actD, actF, dactD, dactF, dqkv, X = bufs["actD"], bufs["actF"], g["dactD"], g["dactF"], g["dqkv"], bufs["X"]
        stats = bufs["stats"]
        seg = (T, B, E0)
        if plan.fuse_ln:
            # the forward normalised inside the projections: materialise LN outputs once, over all steps' rows
            sm, sr = e(R, dtype=torch.float32), e(R, dtype=torch.float32)

            def ln_fwd(x, gam, bet, y):
                L.check(lib.mssr_layernorm_fwd(x.data_ptr(), gam.data_ptr(), bet.data_ptr(), y.data_ptr(), sm.data_ptr(),
                                               sr.data_ptr(), R, d, 1e-6, plan.dtype, L.stream()), "layernorm_fwd")

            for l in range(NL):
                ln_fwd(X[l], pk.ln1_g[l], pk.ln1_b[l], actD[l, K_N1])
                ln_fwd(actD[l, K_H1], pk.ln2_g[l], pk.ln2_b[l], actD[l, K_N2])
                ln_fwd(actD[l, K_H2], pk.lnf_g[l], pk.lnf_b[l], actD[l, K_N3])
            ln_fwd(X[NL], pk.lnF_g, pk.lnF_b, bufs["nf"])
        # bias and LayerNorm gradients are float atomics into zeroed memory: ONE fill for all of them
        zs = _Zeros(NL * (3 * d + 4 * d + ff) + (3 * NL + 1) * 2 * d + 64, dev)
        gW_qkv, gb_qkv = _bgemm_t(dqkv, actD[:, K_N1], seg), _bsum(dqkv, zs.take(NL, 3 * d))
        gW_so, gb_so = _bgemm_t(dactD[:, G_DSO], actD[:, K_CTX1], seg), _bsum(dactD[:, G_DSO], zs.take(NL, d))
        gW_cq, gb_cq = _bgemm_t(dactD[:, G_DCQ], actD[:, K_N2], seg), _bsum(dactD[:, G_DCQ], zs.take(NL, d))
        gW_co, gb_co = _bgemm_t(dactD[:, G_DCO], actD[:, K_CTX2], seg), _bsum(dactD[:, G_DCO], zs.take(NL, d))
        gW_1, gb_1 = _bgemm_t(dactF, actD[:, K_N3], seg), _bsum(dactF, zs.take(NL, ff))
        gW_2, gb_2 = _bgemm_t(dactD[:, G_DF2], actF, seg), _bsum(dactD[:, G_DF2], zs.take(NL, d))
        # gradient of the memory keys / values: contract the per-step score gradients over the steps,
        #   dK_mem[b,h,l,:] = sum_s ds_s[b,h,l] qc_s[b,h,:] ,  dV_mem[b,h,l,:] = sum_s pd_s[b,h,l] dctx2_s[b,h,:]
        # (one batched product per layer and operand), then the K|V projections in one product over all layers
        dh = d // H
        dmem_kv = e(B, Ls, NL, 2, H, dh)
        if dh <= 64:        # one launch for all layers, keys and values (csrc/decode.hip: ctx_kv_grad_kernel)
            qc_all, dctx2_all = actD[:, K_QC], g["dctx"][:, 1]
            L.check(lib.mssr_ctx_kv_grad(g["ds_ctx"].data_ptr(), qc_all.data_ptr(), qc_all.stride(0), dctx2_all.data_ptr(),
                                         dctx2_all.stride(0), dmem_kv.data_ptr(), NL, T, B, E0, H, dh, Ls, plan.dtype,
                                         L.stream()), "ctx_kv_grad")
        else:
            dsc = g["ds_ctx"].to(cd).view(NL, 2, T, B, H, Ls)
            for l in range(NL):
                for which, rhs in ((0, actD[l, K_QC]), (1, g["dctx"][l, 1])):
                    ops.gemm(dsc[l, which].permute(1, 2, 3, 0), rhs.view(T, B, H, dh).permute(1, 2, 0, 3), False, cd,
                             out=dmem_kv[:, :, l, which].permute(0, 2, 1, 3))
        dmem_kv = dmem_kv.view(B * Ls, NL * 2 * d)[E0 * Ls:]            # rows of the sentences that carry gradient
        gW_ckv = _bgemm_t(dmem_kv, bufs["mem"][E0 * Ls:]).view(NL, 2 * d, d)
        gb_ckv = _colsum(dmem_kv).view(NL, 2 * d)
        g_mem = None
        if ctx.needs_input_grad[1]:
            shape, dtype = ctx.mem_meta
            g_mem = torch.zeros(shape, dtype=dtype, device=dev) if E0 else None
            gm = ops.gemm(dmem_kv, pk.w_ckv, False, cd).view((B - E0,) + tuple(shape[1:])).to(dtype)
            if E0:
                g_mem[E0:] = gm
            else:
                g_mem = gm
        # LayerNorm gains / biases: sums over all (step, sentence) rows, one launch per LayerNorm kind over all layers
        # (+ the final one)
        ln_g = zs.take(3 * NL + 1, 2, d)
        st = stats[:3 * NL].view(NL, 3, 2, -1)
        lg = ln_g[:3 * NL].view(NL, 3, 2, d)
        _ln_param_grads(dactD[:, G_DN1], X[:NL], st[:, 0, 0], st[:, 0, 1], lg[:, 0], zeroed=True)
        _ln_param_grads(dactD[:, G_DN2], actD[:, K_H1], st[:, 1, 0], st[:, 1, 1], lg[:, 1], zeroed=True)
        _ln_param_grads(dactD[:, G_DN3], actD[:, K_H2], st[:, 2, 0], st[:, 2, 1], lg[:, 2], zeroed=True)
        _ln_param_grads(g["dnf"], X[NL], stats[3 * NL, 0], stats[3 * NL, 1], ln_g[3 * NL], zeroed=True)
        # generator and (soft) embedding table
        dlg = g["dlogits"][:, :V]
        gW_gen = _bgemm_t(dlg, bufs["nf"], seg)
        gb_gen = _colsum(dlg)
        d_e = g["d_e"]
        if getattr(spec, "forced", None) is not None:
            # a forced <blank> input is an nn.Embedding lookup at padding_idx in the reference: no gradient to that row
            Tf, F = spec.forced.shape
            d_e.view(T, B, d)[:Tf, :F].masked_fill_(spec.forced.eq(spec.pad).unsqueeze(-1), 0)
        gE = _bgemm_t(bufs["in_all"][:T].view(R, Vp)[:, :V], d_e, seg)

        out = [None, g_mem, gE, ln_g[3 * NL, 0], ln_g[3 * NL, 1], gW_gen, gb_gen]
        for l in range(NL):
            out += [gW_qkv[l, :d], gb_qkv[l, :d], gW_qkv[l, d:2 * d], gb_qkv[l, d:2 * d], gW_qkv[l, 2 * d:],
                    gb_qkv[l, 2 * d:], gW_so[l], gb_so[l],
                    gW_ckv[l, :d], gb_ckv[l, :d], gW_ckv[l, d:], gb_ckv[l, d:], gW_cq[l], gb_cq[l], gW_co[l], gb_co[l],
                    gW_1[l], gb_1[l], gW_2[l], gb_2[l],
                    ln_g[3 * l, 0], ln_g[3 * l, 1], ln_g[3 * l + 1, 0], ln_g[3 * l + 1, 1], ln_g[3 * l + 2, 0],
                    ln_g[3 * l + 2, 1]]
        ctx.bufs = ctx.plan = ctx.pin = ctx.probs = None
        return tuple(out)


def _make_spec(model, cfg, memory_bank, src_lengths, steps, params):
    dec = model.decoder
    Ls, B, d = memory_bank.shape
    layer0 = dec.transformer_layers[0]
    spec = _Spec()
    spec.pack, spec.B, spec.Ls, spec.T = get_pack(model, params), B, Ls, int(steps)
    spec.heads = layer0.self_attn.head_count
    spec.training = bool(dec.training)
    spec.p_drop, spec.p_attn = float(layer0.drop.p), float(layer0.self_attn.dropout.p)
    emb_p = float(dec.embeddings.make_embedding[1].dropout.p)
    if spec.training and abs(emb_p - spec.p_drop) > 1e-12:
        raise NotImplementedError("decode engine: embedding dropout (%g) must equal the layer dropout (%g)"
                                  % (emb_p, spec.p_drop))
    spec.temperature = float(cfg.dec_temperature)
    spec.seed = ops.next_seed()
    pos = torch.arange(Ls, device=memory_bank.device).unsqueeze(0)
    spec.src_mask = (pos >= src_lengths.unsqueeze(1)).to(torch.uint8).contiguous()        # [B, Ls], 1 = pad
    spec.bos = cfg.bos_idx
    spec.pe = dec.embeddings.make_embedding[1].table()
    if spec.T > spec.pe.shape[0]:
        raise ValueError("decode engine: %d steps exceed the positional table" % spec.T)
    return spec


def soft_decode(model, cfg, memory_bank, src_lengths, steps, eval_rows=0, forced_ids=None, groups=None):
    """vocab_probs [steps, B, V] (float) of the step-wise soft decode of ``model`` over
    ``memory_bank`` [Ls, B, d] (the bonded encoder states).

    ``forced_ids`` Long [Tf, F]: sentences [0, F) are TEACHER-FORCED for the steps below Tf -- step s reads the word
    forced_ids[s] instead of the previous step's distribution and <blank> target positions are masked as keys of the
    self attention, i.e. probs[:Tf, :F] is softmax(generator(TransformerDecoder.forward(forced_ids, step=None)))
    (the reconstruction pass, onmt/trainer_wgan_wh.py:632-664) computed in the sweep the soft decodes run anyway;
    see mssr_decode_plan.forced_rows.  Their steps from Tf on are unspecified.

    ``groups`` = [(t_end, b0, b1), ...]: return the tuple of probs[:t_end, b0:b1] (views of one slab) instead of the
    whole tensor -- what a caller that batched several decodes into the sweep reads; each group gets its gradient
    directly (no zero-padded full-size gradient per slice).

    ``eval_rows`` > 0: two decodes in one sweep -- sentences [0, eval_rows) are decoded in EVAL mode (no dropout) and
    carry no gradient (the critics' fake text, onmt/trainer_wgan_wh.py:530-535), the rest follow ``model.training``
    and are differentiable (the generator step's decodes, :917-920); see mssr_decode_plan.eval_rows."""
    params = decoder_params(model)
    spec = _make_spec(model, cfg, memory_bank, src_lengths, steps, params)
    spec.eval_rows = int(eval_rows)
    if not 0 <= spec.eval_rows < memory_bank.shape[1]:
        raise ValueError("eval_rows must leave at least one differentiable sentence")
    spec.forced, spec.pad = None, int(cfg.pad_idx)
    if forced_ids is not None:
        if (forced_ids.dim() != 2 or forced_ids.dtype != torch.int64 or forced_ids.shape[0] > spec.T
                or forced_ids.shape[1] > memory_bank.shape[1] or spec.eval_rows):
            raise ValueError("forced_ids: Long [Tf <= steps, F <= B] expected, and no eval_rows")
        spec.forced = forced_ids.contiguous()
    spec.groups = None
    if groups is not None:
        spec.groups = [(int(te), int(b0), int(b1)) for te, b0, b1 in groups]
        for te, b0, b1 in spec.groups:
            if not (0 < te <= spec.T and 0 <= b0 < b1 <= memory_bank.shape[1]):
                raise ValueError("soft_decode: group (%d, %d, %d) outside [steps=%d, B=%d]" % (te, b0, b1, spec.T,
                                                                                             memory_bank.shape[1]))
    needs = torch.is_grad_enabled() and (memory_bank.requires_grad or any(p.requires_grad for p in params))
    spec.save = bool(needs)
    mem = memory_bank.transpose(0, 1).contiguous()          # [B, Ls, d]
    return _SoftDecode.apply(spec, mem, *params)


def greedy_decode_ids(model, cfg, memory_bank, src_lengths, max_length, check_every=4, compact_below=0.75):
    """Greedy search with beam 1 on the decode engine: token ids [steps, B] (int32; steps <= max_length,
    the loop stops once every sentence has produced </s>) -- the step loop of
    onmt/translate/translator_wh.py:709-746 with GreedySearch.advance (greedy_search.py:135-165).

    Finished sentences leave the batch as in the reference (update_finished + map_state(index_select),
    translator_wh.py:744-746, decoders/transformer.py:274-287), but only at the host's check points (every
    ``check_every`` steps, where the loop reads the done flags anyway) and only once fewer than ``compact_below``
    of the current rows are still running: the survivors' memory bank, self-attention cache rows and last tokens are
    gathered into a smaller plan and the sweep continues on it.  What a row holds after its sentence's first </s> is
    unspecified (tokens of the steps it kept running, then </s>); the caller cuts each sentence at its first </s>."""
    lib = L.load()
    params = decoder_params(model)
    dev = memory_bank.device
    with torch.no_grad():
        spec = _make_spec(model, cfg, memory_bank, src_lengths, max_length, params)
        spec.training, spec.save, spec.temperature, spec.eval_rows = False, False, 1.0, 0
        T, B0 = spec.T, spec.B
        out = torch.full((T, B0), cfg.eos_idx, dtype=torch.int32, device=dev)
        rows = torch.arange(B0, device=dev)               # original sentence of every row of the current plan
        mem_cur, len_cur = memory_bank, src_lengths

        def new_plan(sp, mem):
            plan, bufs = _make_plan(sp, mem.transpose(0, 1).contiguous(), greedy=True)
            ids = torch.zeros(sp.T, sp.B, dtype=torch.int32, device=dev)
            plan.ids, plan.greedy = ids.data_ptr(), 1
            return plan, bufs, ids

        plan, bufs, ids = new_plan(spec, mem_cur)
        done = torch.zeros(B0, dtype=torch.bool, device=dev)
        steps = flushed = 0
        for step in range(T):
            L.check(lib.mssr_decode_step_fwd(ctypes.byref(plan), step, L.stream()), "decode_step_fwd")
            steps = step + 1
            if steps % check_every and steps != T:
                continue
            out[flushed:steps, rows] = ids[flushed:steps]
            done[rows] |= (ids[flushed:steps] == cfg.eos_idx).any(0)
            flushed = steps
            live = ~done[rows]
            n_live = int(live.sum())                      # the one host sync of this check point
            if n_live == 0 or steps == T:
                break
            if n_live < compact_below * rows.numel() and rows.numel() - n_live >= 8:
                keep = live.nonzero().squeeze(1)
                mem_cur, len_cur = mem_cur.index_select(1, keep), len_cur.index_select(0, keep)
                sp = _make_spec(model, cfg, mem_cur, len_cur, max_length, params)
                sp.training, sp.save, sp.temperature, sp.eval_rows, sp.seed = False, False, 1.0, 0, spec.seed
                plan2, bufs2, ids2 = new_plan(sp, mem_cur)
                NL, Bc, d3 = bufs["qkv"].shape[0], rows.numel(), bufs["qkv"].shape[2]
                # the survivors' self-attention keys / values of the steps so far, and their last tokens
                bufs2["qkv"].view(NL, n_live, T, d3)[:, :, :steps] = bufs["qkv"].view(NL, Bc, T, d3)[:, :, :steps].index_select(1, keep)
                ids2[:steps] = ids[:steps].index_select(1, keep)
                rows = rows.index_select(0, keep)
                plan, bufs, ids, spec = plan2, bufs2, ids2, sp
    return out[:steps]
