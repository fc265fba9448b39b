"""Host side of the teacher-forced decoder-stack engine (csrc/decoder_stack.hip, include/mssr_hip.h).

``run_stack(decoder, x, memory, self_mask, src_pad)`` runs ``decoder.transformer_layers`` and ``decoder.layer_norm``
on the embedded target rows x [B, T, d] against the memory bank [B, Ls, d] as ONE autograd node: the layer loop of
TransformerDecoder.forward with step = None (onmt/decoders/transformer.py:289-337), i.e. the decoder pass of the
reconstruction phase (onmt/trainer_wgan_wh.py:632-664).  Forward and backward are one C call each; parameter
gradients are contracted afterwards with one batched product per parameter kind over all layers, the memory bank's
keys | values of all layers are one product, and the weights come from the same pack the step-wise decode engine
reads (decode_engine._Pack).

The node is differentiable once; under ``ops.second_order()``, in fp32 beyond 64 positions, or when the layers'
dropout rates differ, ``run_stack`` declines and the caller loops over the modules.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L
from . import decode_engine as DE
from . import ops
from .decode_engine import _Zeros, _bgemm_t, _bsum, _colsum, _ln_param_grads, cached_pack

S_N1, S_CTXS, S_H1, S_N2, S_CQ, S_CTXC, S_H2, S_N3 = range(8)
SG_DN1, SG_DOS, SG_DN2, SG_DCQ, SG_DOC, SG_DN3, SG_DF2 = range(7)

_pack_cache = {}


def stack_params(dec):
    """[lnF_g, lnF_b] + 26 parameters per decoder layer (decode_engine._layer_params order)."""
    out = [dec.layer_norm.weight, dec.layer_norm.bias]
    for layer in dec.transformer_layers:
        out += DE._layer_params(layer)
    return out


def _pack_for(dec, params):
    """The model's pack when a soft / greedy decode has claimed this decoder (one refresh per optimizer step serves
    both engines), else a pack of the decoder alone."""
    model = DE.model_of(dec)
    if model is not None:
        return DE.get_pack(model, DE.decoder_params(model))
    cd = ops.compute_dtype()
    return cached_pack(_pack_cache, dec, params, lambda: DE._Pack(params, len(dec.transformer_layers), cd, head=False))


class _Spec:
    __slots__ = ("pack", "B", "T", "Ls", "heads", "training", "p_drop", "p_attn", "seed", "self_mask", "src_mask", "save",
                 "want_probs")


class _DecoderStack(Function):
    @staticmethod
    def forward(ctx, spec, x, mem, *params):
        pk = spec.pack
        lib = L.load()
        cd, dev = pk.cd, x.device
        B, T, Ls, NL, d, ff, H = spec.B, spec.T, spec.Ls, pk.NL, pk.d, pk.ff, spec.heads
        R = B * T
        Lslots = NL if spec.save else 1
        e = lambda *shape, dtype=cd: torch.empty(*shape, dtype=dtype, device=dev)  # noqa: E731
        memc = ops.cast(mem).detach().contiguous().view(B * Ls, d)
        # context keys | values of the memory bank for every layer in one product: [B*Ls, NL*2d]
        mem_kv = ops.gemm(memc, pk.w_ckv, True, cd, bias=pk.b_ckv)
        bufs = dict(X=e(NL + 1, R, d), out=e(R, d), qkv=e(Lslots, R, 3 * d), actD=e(Lslots, 8, R, d), actF=e(Lslots, R, ff),
                    stats=e(3 * NL + 1, 2, R, dtype=torch.float32), mem_kv=mem_kv)
        probs = e(B, H, T, Ls) if spec.want_probs else None
        bufs["X"][0].view(B, T, d).copy_(x.detach())
        plan = L.DecStackPlan()
        plan.dtype = L.F32 if cd == torch.float32 else L.BF16
        plan.B, plan.T, plan.Ls, plan.d, plan.heads, plan.ff, plan.n_layers = B, T, Ls, d, H, ff, NL
        plan.save, plan.training = int(spec.save), int(spec.training)
        plan.p_drop, plan.p_attn, plan.seed = spec.p_drop, spec.p_attn, spec.seed
        plan.layers = pk.layers
        plan.lnF_g, plan.lnF_b = pk.lnF_g.data_ptr(), pk.lnF_b.data_ptr()
        plan.self_mask, plan.src_mask = spec.self_mask.data_ptr(), L.ptr(spec.src_mask)
        plan.probs = L.ptr(probs)
        for name, t in bufs.items():
            setattr(plan, name, t.data_ptr())
        ws = ops._workspace(dev)
        plan.gemm_ws, plan.gemm_ws_bytes = ws.data_ptr(), ws.numel() * 4
        L.check(lib.mssr_decstack_fwd(ctypes.byref(plan), L.stream()), "decstack_fwd")
        out = bufs.pop("out")
        if spec.save:
            bufs["memc"] = memc
            ctx.spec, ctx.plan, ctx.bufs = spec, plan, bufs
            ctx.pin = pk.jobs.pin()             # the backward reads the pack: no in-place refresh until then
            ctx.dtypes = (x.dtype, mem.dtype)
        top = probs[:, 0].contiguous() if probs is not None else torch.empty(0, device=dev)
        ctx.mark_non_differentiable(top)
        return out.view(B, T, d), top

    @staticmethod
    @once_differentiable
    def backward(ctx, dout, _dtop):
        spec, plan, bufs = ctx.spec, ctx.plan, ctx.bufs
        if bufs is None:
            raise RuntimeError("decoder stack: backward called twice (activations were released)")
        pk = spec.pack
        lib = L.load()
        cd, dev = pk.cd, dout.device
        B, T, Ls, NL, d, ff = spec.B, spec.T, spec.Ls, pk.NL, pk.d, pk.ff
        R = B * T
        e = lambda *shape, dtype=cd: torch.empty(*shape, dtype=dtype, device=dev)  # noqa: E731
        doutc = dout.to(cd).contiguous().view(R, d)
        g = dict(dX=e(R, d), dqkv=e(NL, R, 3 * d), dactD=e(NL, 7, R, d), dactF=e(NL, R, ff), dctx=e(R, d),
                 dmem_kv=e(B * Ls, NL * 2 * d))
        plan.dout = doutc.data_ptr()
        for name, t in g.items():
            setattr(plan, name, t.data_ptr())
        L.check(lib.mssr_decstack_bwd(ctypes.byref(plan), L.stream()), "decstack_bwd")
        actD, actF, dactD, dactF, dqkv, X = bufs["actD"], bufs["actF"], g["dactD"], g["dactF"], g["dqkv"], bufs["X"]
        memc, dmem_kv = bufs["memc"], g["dmem_kv"]
        # bias and LayerNorm gradients are float atomics into zeroed memory: ONE fill for all of them
        zs = _Zeros(NL * (3 * d + 4 * d + ff) + (3 * NL + 1) * 2 * d + 64, dev)
        gW_qkv, gb_qkv = _bgemm_t(dqkv, actD[:, S_N1]), _bsum(dqkv, zs.take(NL, 3 * d))
        gW_so, gb_so = _bgemm_t(dactD[:, SG_DOS], actD[:, S_CTXS]), _bsum(dactD[:, SG_DOS], zs.take(NL, d))
        gW_cq, gb_cq = _bgemm_t(dactD[:, SG_DCQ], actD[:, S_N2]), _bsum(dactD[:, SG_DCQ], zs.take(NL, d))
        gW_co, gb_co = _bgemm_t(dactD[:, SG_DOC], actD[:, S_CTXC]), _bsum(dactD[:, SG_DOC], zs.take(NL, d))
        gW_1, gb_1 = _bgemm_t(dactF, actD[:, S_N3]), _bsum(dactF, zs.take(NL, ff))
        gW_2, gb_2 = _bgemm_t(dactD[:, SG_DF2], actF), _bsum(dactD[:, SG_DF2], zs.take(NL, d))
        # memory-bank projections of all layers: weight gradient [NL*2d, d], bias gradient, input gradient -- one product each
        gW_ckv = ops.gemm(dmem_kv.transpose(0, 1), memc, False, torch.float32).view(NL, 2 * d, d)
        gb_ckv = _colsum(dmem_kv).view(NL, 2 * d)
        dmem = None
        if ctx.needs_input_grad[2]:
            dmem = ops.gemm(dmem_kv, pk.w_ckv, False, cd).view(B, Ls, d).to(ctx.dtypes[1])
        # LayerNorm gains / biases: one launch per LayerNorm kind over all layers (+ the final one)
        stats = bufs["stats"]
        ln_g = zs.take(3 * NL + 1, 2, d)
        st = stats[:3 * NL].view(NL, 3, 2, -1)
        lg = ln_g[:3 * NL].view(NL, 3, 2, d)
        _ln_param_grads(dactD[:, SG_DN1], X[:NL], st[:, 0, 0], st[:, 0, 1], lg[:, 0], zeroed=True)
        _ln_param_grads(dactD[:, SG_DN2], actD[:, S_H1], st[:, 1, 0], st[:, 1, 1], lg[:, 1], zeroed=True)
        _ln_param_grads(dactD[:, SG_DN3], actD[:, S_H2], st[:, 2, 0], st[:, 2, 1], lg[:, 2], zeroed=True)
        _ln_param_grads(doutc, X[NL], stats[3 * NL, 0], stats[3 * NL, 1], ln_g[3 * NL], zeroed=True)
        dx = g["dX"].view(B, T, d).to(ctx.dtypes[0]) if ctx.needs_input_grad[1] else None
        out = [None, dx, dmem, ln_g[3 * NL, 0], ln_g[3 * NL, 1]]
        for l in range(NL):
            out += [gW_qkv[l, :d], gb_qkv[l, :d], gW_qkv[l, d:2 * d], gb_qkv[l, d:2 * d], gW_qkv[l, 2 * d:], gb_qkv[l, 2 * d:],
                    gW_so[l], gb_so[l], gW_ckv[l, :d], gb_ckv[l, :d], gW_ckv[l, d:], gb_ckv[l, d:],
                    gW_cq[l], gb_cq[l], gW_co[l], gb_co[l], gW_1[l], gb_1[l], gW_2[l], gb_2[l],
                    ln_g[3 * l, 0], ln_g[3 * l, 1], ln_g[3 * l + 1, 0], ln_g[3 * l + 1, 1], ln_g[3 * l + 2, 0], ln_g[3 * l + 2, 1]]
        ctx.bufs = ctx.plan = ctx.pin = None
        return tuple(out)


def run_stack(dec, x, memory, self_mask, src_pad, want_probs=True):
    """(LayerNorm(layers(x)) [B, T, d], head 0 of the last layer's context attention [B, T, Ls]) for x [B, T, d],
    memory [B, Ls, d], self_mask Bool/uint8 [B, T, T] (True = masked), src_pad Bool/uint8 [B, Ls] or None -- or None
    when the engine does not apply: the caller then loops over the modules."""
    layers = dec.transformer_layers
    if ops.second_order_active() or len(layers) == 0:
        return None
    layer0 = layers[0]
    B, T, d = x.shape
    Ls = memory.shape[1]
    heads = layer0.self_attn.head_count
    ffn = layer0.feed_forward
    p_drop = float(layer0.drop.p)
    p_attn = float(layer0.self_attn.dropout.p)
    for layer in layers:
        f = layer.feed_forward
        if (abs(float(layer.drop.p) - p_drop) > 1e-12 or abs(float(f.dropout_1.p) - p_drop) > 1e-12
                or abs(float(f.dropout_2.p) - p_drop) > 1e-12 or abs(float(layer.self_attn.dropout.p) - p_attn) > 1e-12
                or abs(float(layer.context_attn.dropout.p) - p_attn) > 1e-12 or layer.training != layer0.training):
            return None
    cd = ops.compute_dtype()
    code = L.F32 if cd == torch.float32 else L.BF16
    params = stack_params(dec)
    if any(p.dtype != torch.float32 or not p.is_cuda for p in params):
        return None
    needs = torch.is_grad_enabled() and (x.requires_grad or memory.requires_grad or any(p.requires_grad for p in params))
    if d % heads or d % 8 or ffn.w_1.weight.shape[0] % 8:
        return None
    lib = L.load()
    if not (lib.mssr_attention_fits(T, T, d // heads, code, int(needs)) and lib.mssr_attention_fits(T, Ls, d // heads, code,
                                                                                                   int(needs))):
        return None
    spec = _Spec()
    spec.pack = _pack_for(dec, params)
    spec.B, spec.T, spec.Ls, spec.heads = B, T, Ls, heads
    spec.training = bool(layer0.training)
    spec.p_drop, spec.p_attn = p_drop, p_attn
    spec.seed = ops.next_seed()

    def u8(m):
        m = m.contiguous()
        return m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)

    spec.self_mask = u8(self_mask.expand(B, T, T))
    spec.src_mask = None if src_pad is None else u8(src_pad.expand(B, Ls))
    spec.save = bool(needs)
    spec.want_probs = bool(want_probs)
    out, top = _DecoderStack.apply(spec, ops.cast(x), memory, *params)
    return out, (top if want_probs else None)
