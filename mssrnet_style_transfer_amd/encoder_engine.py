"""Host side of the encoder-stack engine (csrc/encoder.hip, include/mssr_hip.h).

``run_stack(encoder, x, pad_mask)`` runs ``encoder.transformer`` (all TransformerEncoderLayers) and
``encoder.layer_norm`` on already embedded rows x [B, L, d] as ONE autograd node: the layer loop of
TransformerEncoder.forward (onmt/encoders/transformer.py:127-132) and of StyleGenerator2.forward
(onmt/style_generator.py:40-47).  Forward and backward are one C call each; parameter gradients are
contracted afterwards with one batched product per parameter kind over all layers.

The node is differentiable once.  Graphs that are differentiated twice (the WGAN-GP penalties) are
built under ``ops.second_order()``, where ``run_stack`` declines and the caller loops over the
modules (whose ops are closed under differentiation).
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L
from . import ops
from .decode_engine import _Zeros, _bgemm_t, _bsum, _ln_param_grads, cached_pack
from .weight_pack import JobTable

N_ENC_LAYER_PARAMS = 16
(_WQ, _BQ, _WK, _BK, _WV, _BV, _WO, _BO, _W1, _B1, _W2, _B2, _LNFG, _LNFB, _LN1G, _LN1B) = range(N_ENC_LAYER_PARAMS)
E_N1, E_CTX, E_H, E_N2 = range(4)
EG_DN1, EG_DO, EG_DN2, EG_DF2 = range(4)


def _layer_params(layer):
    sa, ff = layer.self_attn, layer.feed_forward
    return [sa.linear_query.weight, sa.linear_query.bias, sa.linear_keys.weight, sa.linear_keys.bias,
            sa.linear_values.weight, sa.linear_values.bias, sa.final_linear.weight, sa.final_linear.bias,
            ff.w_1.weight, ff.w_1.bias, ff.w_2.weight, ff.w_2.bias, ff.layer_norm.weight, ff.layer_norm.bias,
            layer.layer_norm.weight, layer.layer_norm.bias]


def encoder_params(enc):
    """[lnF_g, lnF_b] + 16 parameters per encoder layer, in a fixed order."""
    out = [enc.layer_norm.weight, enc.layer_norm.bias]
    for layer in enc.transformer:
        out += _layer_params(layer)
    return out


class _Pack:
    """Encoder-stack weights as the engine wants them (compute dtype, packed, transposed, LN-folded): buffers
    allocated once, refreshed from the fp32 parameters in one launch (weight_pack.JobTable -> csrc/pack.hip)."""

    def __init__(self, params, n_layers, cd):
        lnFg, lnFb = params[:2]
        lp = [params[2 + i * N_ENC_LAYER_PARAMS: 2 + (i + 1) * N_ENC_LAYER_PARAMS] for i in range(n_layers)]
        NL = n_layers
        d = lp[0][_WQ].shape[1]
        ff = lp[0][_W1].shape[0]
        dev = lnFg.device
        self.NL, self.d, self.ff, self.cd = NL, d, ff, cd
        e = lambda *shape, dtype=cd: torch.empty(*shape, dtype=dtype, device=dev)  # noqa: E731
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
        for name, n_out, n_in in (("qkv", 3 * d, d), ("o", d, d), ("1", ff, d), ("2", d, ff)):
            setattr(self, "w_" + name, e(NL, n_out, n_in))
            setattr(self, "wt_" + name, e(NL, n_in, n_out))
            setattr(self, "b_" + name, f(NL, n_out))
        for name in ("ln1_g", "ln1_b", "lnf_g", "lnf_b"):
            setattr(self, name, f(NL, d))
        self.lnF_g, self.lnF_b = f(d), f(d)
        names = ["w_qkv", "wt_qkv", "w_o", "wt_o", "w_1", "wt_1", "w_2", "wt_2", "b_qkv", "b_o", "b_1", "b_2",
                 "ln1_g", "ln1_b", "lnf_g", "lnf_b"]
        self.fuse_ln = cd == torch.bfloat16 and d % 32 == 0 and d <= 2048
        if self.fuse_ln:
            for name, n_out in (("qkv", 3 * d), ("1", ff)):
                setattr(self, "wg_" + name, e(NL, n_out, d))
                setattr(self, "cs_" + name, f(NL, n_out))
                setattr(self, "bf_" + name, f(NL, n_out))
            names += ["wg_qkv", "wg_1", "cs_qkv", "cs_1", "bf_qkv", "bf_1"]

        jt = self.jobs = JobTable(cd, dev)

        def linear(l, name, w_idx, b_idx, off, ln=None):
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
            linear(l, "o", _WO, _BO, 0)
            linear(l, "1", _W1, _B1, 0, (_LNFG, _LNFB))
            linear(l, "2", _W2, _B2, 0)
            for name, idx in (("ln1_g", _LN1G), ("ln1_b", _LN1B), ("lnf_g", _LNFG), ("lnf_b", _LNFB)):
                jt.add(lp[l][idx], dst_f=getattr(self, name)[l])
        jt.add(lnFg, dst_f=self.lnF_g)
        jt.add(lnFb, dst_f=self.lnF_b)
        jt.finish()
        arr = (L.EncLayer * NL)()
        for l in range(NL):
            for name in names:
                setattr(arr[l], name, getattr(self, name)[l].data_ptr())
        self.layers = arr
        self.refresh()

    def refresh(self):
        self.jobs.run()


_pack_cache = {}


def get_pack(enc, params):
    cd = ops.compute_dtype()
    return cached_pack(_pack_cache, enc, params, lambda: _Pack(params, len(enc.transformer), cd))


class _Spec:
    __slots__ = ("pack", "B", "L", "heads", "training", "p_drop", "p_attn", "seed", "mask", "save", "eval_rows")


class _EncoderStack(Function):
    @staticmethod
    def forward(ctx, spec, x, *params):
        pk = spec.pack
        lib = L.load()
        cd, dev = pk.cd, x.device
        B, Ln, NL, d, ff = spec.B, spec.L, pk.NL, pk.d, pk.ff
        R = B * Ln
        Ls = NL if spec.save else 1
        e = lambda *shape, dtype=cd: torch.empty(*shape, dtype=dtype, device=dev)  # noqa: E731
        bufs = dict(X=e(NL + 1, R, d), out=e(R, d), qkv=e(Ls, R, 3 * d), actD=e(Ls, 4, R, d), actF=e(Ls, R, ff),
                    stats=e(2 * NL + 1, 2, R, dtype=torch.float32))
        bufs["X"][0].view(B, Ln, d).copy_(x.detach())
        plan = L.EncoderPlan()
        plan.dtype = L.F32 if cd == torch.float32 else L.BF16
        plan.B, plan.L, plan.d, plan.heads, plan.ff, plan.n_layers = B, Ln, d, spec.heads, ff, NL
        plan.save, plan.training = int(spec.save), int(spec.training)
        plan.fuse_ln = int(pk.fuse_ln and not spec.save and R <= 8192)      # the K-split GEMM (LN prologue) takes <= 8192 rows
        plan.p_drop, plan.p_attn, plan.seed = spec.p_drop, spec.p_attn, spec.seed
        plan.eval_rows = spec.eval_rows
        plan.layers = pk.layers
        plan.lnF_g, plan.lnF_b = pk.lnF_g.data_ptr(), pk.lnF_b.data_ptr()
        plan.mask = L.ptr(spec.mask)
        for name, t in bufs.items():
            setattr(plan, name, t.data_ptr())
        ws = ops._workspace(dev)
        plan.gemm_ws, plan.gemm_ws_bytes = ws.data_ptr(), ws.numel() * 4
        L.check(lib.mssr_encoder_fwd(ctypes.byref(plan), L.stream()), "encoder_fwd")
        out = bufs.pop("out")
        if spec.save:
            ctx.spec, ctx.plan, ctx.bufs = spec, plan, bufs
            ctx.pin = pk.jobs.pin()             # the backward reads the pack: no in-place refresh until then
            ctx.x_dtype = x.dtype
        return out.view(B, Ln, d)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        spec, plan, bufs = ctx.spec, ctx.plan, ctx.bufs
        if bufs is None:
            raise RuntimeError("encoder stack: backward called twice (activations were released)")
        pk = spec.pack
        lib = L.load()
        cd, dev = pk.cd, dout.device
        B, Ln, NL, d, ff = spec.B, spec.L, pk.NL, pk.d, pk.ff
        R = B * Ln
        e = lambda *shape, dtype=cd: torch.empty(*shape, dtype=dtype, device=dev)  # noqa: E731
        doutc = dout.to(cd).contiguous().view(R, d)
        g = dict(dX=e(R, d), dqkv=e(NL, R, 3 * d), dactD=e(NL, 4, R, d), dactF=e(NL, R, ff), dctx=e(R, d))
        plan.dout = doutc.data_ptr()
        for name, t in g.items():
            setattr(plan, name, t.data_ptr())
        L.check(lib.mssr_encoder_bwd(ctypes.byref(plan), L.stream()), "encoder_bwd")
        actD, actF, dactD, dactF, dqkv, X = bufs["actD"], bufs["actF"], g["dactD"], g["dactF"], g["dqkv"], bufs["X"]
        # bias and LayerNorm gradients are float atomics into zeroed memory: ONE fill for all of them
        zs = _Zeros(NL * (3 * d + 2 * d + ff) + (2 * NL + 1) * 2 * d + 64, dev)
        gW_qkv, gb_qkv = _bgemm_t(dqkv, actD[:, E_N1]), _bsum(dqkv, zs.take(NL, 3 * d))
        gW_o, gb_o = _bgemm_t(dactD[:, EG_DO], actD[:, E_CTX]), _bsum(dactD[:, EG_DO], zs.take(NL, d))
        gW_1, gb_1 = _bgemm_t(dactF, actD[:, E_N2]), _bsum(dactF, zs.take(NL, ff))
        gW_2, gb_2 = _bgemm_t(dactD[:, EG_DF2], actF), _bsum(dactD[:, EG_DF2], zs.take(NL, d))
        # LayerNorm gains / biases: one launch per LayerNorm kind over all layers (+ the final one)
        stats = bufs["stats"]
        ln_g = zs.take(2 * NL + 1, 2, d)
        st = stats[:2 * NL].view(NL, 2, 2, -1)
        lg = ln_g[:2 * NL].view(NL, 2, 2, d)
        _ln_param_grads(dactD[:, EG_DN1], X[:NL], st[:, 0, 0], st[:, 0, 1], lg[:, 0], zeroed=True)
        _ln_param_grads(dactD[:, EG_DN2], actD[:, E_H], st[:, 1, 0], st[:, 1, 1], lg[:, 1], zeroed=True)
        _ln_param_grads(doutc, X[NL], stats[2 * NL, 0], stats[2 * NL, 1], ln_g[2 * NL], zeroed=True)
        out = [None, g["dX"].view(B, Ln, d).to(ctx.x_dtype), ln_g[2 * NL, 0], ln_g[2 * NL, 1]]
        for l in range(NL):
            out += [gW_qkv[l, :d], gb_qkv[l, :d], gW_qkv[l, d:2 * d], gb_qkv[l, d:2 * d], gW_qkv[l, 2 * d:],
                    gb_qkv[l, 2 * d:], gW_o[l], gb_o[l], gW_1[l], gb_1[l], gW_2[l], gb_2[l],
                    ln_g[2 * l + 1, 0], ln_g[2 * l + 1, 1], ln_g[2 * l, 0], ln_g[2 * l, 1]]
        ctx.bufs = ctx.plan = ctx.pin = None
        return tuple(out)


def run_stack(enc, x, pad_mask, eval_rows=0):
    """LayerNorm(layers(x)) for x [B, L, d] (any strides) and pad_mask Bool/uint8 [B, L] (True = padding key),
    or None when the engine does not apply (second-order graph, shapes beyond the fused attention
    kernel, unequal dropout rates): the caller then loops over the modules.  eval_rows: sentences [0, eval_rows)
    run without dropout even in train mode (mssr_encoder_plan.eval_rows)."""
    if not 0 <= eval_rows <= x.shape[0]:
        raise ValueError("run_stack: eval_rows %d outside [0, %d]" % (eval_rows, x.shape[0]))
    if ops.second_order_active() or len(enc.transformer) == 0:
        return None
    layer0 = enc.transformer[0]
    B, Ln, d = x.shape
    heads = layer0.self_attn.head_count
    ffn = layer0.feed_forward
    p_drop = float(layer0.dropout.p)
    if abs(float(ffn.dropout_1.p) - p_drop) > 1e-12 or abs(float(ffn.dropout_2.p) - p_drop) > 1e-12:
        return None
    cd = ops.compute_dtype()
    code = L.F32 if cd == torch.float32 else L.BF16
    params = encoder_params(enc)
    needs = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    if d % heads or d % 8 or ffn.w_1.weight.shape[0] % 8:
        return None
    if not L.load().mssr_attention_fits(Ln, Ln, d // heads, code, int(needs)):
        return None
    spec = _Spec()
    spec.pack = get_pack(enc, params)
    spec.B, spec.L, spec.heads = B, Ln, heads
    spec.training = bool(layer0.training)
    spec.p_drop, spec.p_attn = p_drop, float(layer0.self_attn.dropout.p)
    spec.seed = ops.next_seed()
    spec.eval_rows = int(eval_rows)
    m = pad_mask
    if m is not None:
        m = m.contiguous()
        m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
    spec.mask = m
    spec.save = bool(needs)
    return _EncoderStack.apply(spec, ops.cast(x), *params)
