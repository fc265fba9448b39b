// Decoder-stack engine: the TEACHER-FORCED decoder (whole target sequence at once) as one C call per direction.
//
// Replaces the Python layer loop of TransformerDecoder.forward with step = None (onmt/decoders/transformer.py:289-337)
// -> n_layers x TransformerDecoderLayer._forward (:110-171: LN1 -> masked self attention -> dropout + residual ->
// LN2 -> context attention over the memory bank -> dropout + residual -> PositionwiseFeedForward) -> final LayerNorm,
// and its autograd backward: the reconstruction phase's decoder pass (onmt/trainer_wgan_wh.py:632-664), 25 of the 31
// phase calls of the README schedule.  The step-wise soft decode of the same layers is csrc/decode.hip.
//
// Per layer forward: LN1 -> packed QKV projection -> fused attention (causal | target-padding mask [B,T,T]) ->
// out-proj (+bias, dropout, residual) -> LN2 -> query projection -> fused attention over this layer's keys|values of
// the memory bank (ONE product for all layers, done by the caller: mem_kv [B*Ls, n_layers*2d]) -> out-proj (+bias,
// dropout, residual) -> LN -> FFN1 (+bias, ReLU, dropout) -> FFN2 (+bias, dropout, residual).
// Weight / bias / LayerNorm gradients are not formed here: backward leaves the gradient of every GEMM output in a
// per-layer slab (and of mem_kv in one slab) and the caller contracts each kind once over all layers.
#include "engine_common.hpp"

namespace mssr {

enum { S_N1 = 0, S_CTXS, S_H1, S_N2, S_CQ, S_CTXC, S_H2, S_N3, S_NKINDS };
enum { SG_DN1 = 0, SG_DOS, SG_DN2, SG_DCQ, SG_DOC, SG_DN3, SG_DF2, SG_NKINDS };

static inline uint64_t ds_seed(const mssr_decstack_plan* p, int layer, int site) {
    return p->seed + (uint64_t)(layer + 1) * 0x9E3779B1ull + (uint64_t)site * 7919ull;
}

static int ds_check(const mssr_decstack_plan* p, bool backward, const char* who) {
    MSSR_REQUIRE(p, "%s: null plan", who);
    MSSR_REQUIRE(valid_dtype(p->dtype), "%s: bad dtype", who);
    MSSR_REQUIRE(p->B > 0 && p->T > 0 && p->Ls > 0 && p->d > 0 && p->heads > 0 && p->d % p->heads == 0 && p->d % 8 == 0 &&
                     p->ff % 8 == 0 && p->n_layers > 0 && p->d <= 4096,
                 "%s: bad dimensions", who);
    MSSR_REQUIRE(p->layers && p->lnF_g && p->lnF_b && p->X && p->out && p->mem_kv && p->qkv && p->actD && p->actF &&
                     p->stats && p->self_mask,
                 "%s: null buffer in plan", who);
    MSSR_REQUIRE(p->p_drop >= 0.f && p->p_drop < 1.f && p->p_attn >= 0.f && p->p_attn < 1.f, "%s: bad dropout", who);
    const int dh = p->d / p->heads;
    MSSR_REQUIRE(mssr_attention_fits(p->T, p->T, dh, p->dtype, backward ? 1 : 0) &&
                     mssr_attention_fits(p->T, p->Ls, dh, p->dtype, backward ? 1 : 0),
                 "%s: T=%d, Ls=%d, dh=%d is beyond the fused attention kernels", who, p->T, p->Ls, dh);
    if (backward) {
        MSSR_REQUIRE(p->save, "%s: the forward did not save activations (save = 0)", who);
        MSSR_REQUIRE(p->dout && p->dX && p->dqkv && p->dactD && p->dactF && p->dctx && p->dmem_kv,
                     "%s: null gradient buffer in plan", who);
    }
    return MSSR_OK;
}

struct StackGeo {
    const mssr_decstack_plan* p;
    int64_t R;
    int d, ff, NL;
    size_t es;
    StackGeo(const mssr_decstack_plan* p_) : p(p_), R((int64_t)p_->B * p_->T), d(p_->d), ff(p_->ff), NL(p_->n_layers),
                                             es(dsize(p_->dtype)) {}
    int slot(int l) const { return p->save ? l : 0; }
    char* X(int l) const { return at(p->X, (int64_t)l * R * d, es); }
    char* qkv(int l) const { return at(p->qkv, (int64_t)slot(l) * R * 3 * d, es); }
    char* act(int l, int kind) const { return at(p->actD, ((int64_t)slot(l) * S_NKINDS + kind) * R * d, es); }
    char* inter(int l) const { return at(p->actF, (int64_t)slot(l) * R * ff, es); }
    float* stat(int idx, int which) const { return p->stats + ((int64_t)idx * 2 + which) * R; }
    char* dqkv(int l) const { return at(p->dqkv, (int64_t)l * R * 3 * d, es); }
    char* dact(int l, int kind) const { return at(p->dactD, ((int64_t)l * SG_NKINDS + kind) * R * d, es); }
    char* dinter(int l) const { return at(p->dactF, (int64_t)l * R * ff, es); }
};

// masked self attention over the packed projections of layer l
static void ds_self_desc(const mssr_decstack_plan* p, const StackGeo& G, int l, float p_attn, mssr_attn_desc& a) {
    const int d = p->d;
    char* qkv = G.qkv(l);
    a.q = qkv; a.k = qkv + (int64_t)d * G.es; a.v = qkv + 2LL * d * G.es;
    a.mask = p->self_mask;
    a.B = p->B; a.H = p->heads; a.Lq = p->T; a.Lk = p->T; a.dh = d / p->heads;
    a.q_bs = a.k_bs = a.v_bs = (int64_t)p->T * 3 * d;
    a.q_ls = a.k_ls = a.v_ls = 3 * d;
    a.mask_bs = (int64_t)p->T * p->T; a.mask_qs = p->T;
    a.p_drop = p_attn; a.seed = ds_seed(p, l, 0);
    a.dtype = p->dtype;
    a.drop_b0 = 0;
}

// context attention: queries of layer l against its keys | values inside mem_kv (columns [l*2d, l*2d + 2d))
static void ds_ctx_desc(const mssr_decstack_plan* p, const StackGeo& G, int l, float p_attn, const void* mem_kv,
                        mssr_attn_desc& a) {
    const int d = p->d;
    const int64_t row = (int64_t)G.NL * 2 * d;
    a.q = G.act(l, S_CQ);
    a.k = (const char*)mem_kv + (int64_t)l * 2 * d * G.es;
    a.v = (const char*)mem_kv + ((int64_t)l * 2 * d + d) * G.es;
    a.mask = p->src_mask;
    a.B = p->B; a.H = p->heads; a.Lq = p->T; a.Lk = p->Ls; a.dh = d / p->heads;
    a.q_bs = (int64_t)p->T * d; a.q_ls = d;
    a.k_bs = a.v_bs = (int64_t)p->Ls * row; a.k_ls = a.v_ls = row;
    a.mask_bs = p->Ls; a.mask_qs = 0;
    a.p_drop = p_attn; a.seed = ds_seed(p, l, 2);
    a.dtype = p->dtype;
    a.drop_b0 = 0;
}

}  // namespace mssr

using namespace mssr;

extern "C" int mssr_decstack_fwd(const mssr_decstack_plan* p, void* stream) {
    if (int e = ds_check(p, false, "decstack_fwd")) return e;
    const StackGeo G(p);
    GemmCtx c{(hipStream_t)stream, p->dtype, G.es, p->gemm_ws, p->gemm_ws_bytes};
    const int64_t R = G.R;
    const int d = G.d, ff = G.ff, NL = G.NL;
    const float pd = p->training ? p->p_drop : 0.f, pa = p->training ? p->p_attn : 0.f;
    int rc;
    for (int l = 0; l < NL; ++l) {
        const mssr_dec_layer& L = p->layers[l];
        char *x = G.X(l), *n1 = G.act(l, S_N1), *ctxs = G.act(l, S_CTXS), *h1 = G.act(l, S_H1), *n2 = G.act(l, S_N2);
        char *cq = G.act(l, S_CQ), *ctxc = G.act(l, S_CTXC), *h2 = G.act(l, S_H2), *n3 = G.act(l, S_N3);
        char *qkv = G.qkv(l), *inter = G.inter(l);
        // --- masked self attention: h1 = x + drop(Wso attn(LN1 x) + bso)
        if ((rc = mssr_layernorm_fwd(x, L.ln1_g, L.ln1_b, n1, G.stat(3 * l, 0), G.stat(3 * l, 1), R, d, 1e-6f, c.dt, c.s)))
            return rc;
        if ((rc = gemm_nt(c, n1, d, L.w_qkv, d, qkv, 3 * d, R, 3 * d, d, L.b_qkv, 0, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt,
                          0.f, 0)))
            return rc;
        mssr_attn_desc a;
        ds_self_desc(p, G, l, pa, a);
        if ((rc = mssr_attention_fwd(&a, ctxs, (int64_t)p->T * d, d, nullptr, c.s))) return rc;
        if ((rc = gemm_nt(c, ctxs, d, L.w_so, d, h1, d, R, d, d, L.b_so, 0, x, d, nullptr, 0, 1.f, 0.f, c.dt, pd,
                          ds_seed(p, l, 1))))
            return rc;
        // --- context attention: h2 = h1 + drop(Wco attn(Wcq LN2 h1; memory) + bco)
        if ((rc = mssr_layernorm_fwd(h1, L.ln2_g, L.ln2_b, n2, G.stat(3 * l + 1, 0), G.stat(3 * l + 1, 1), R, d, 1e-6f,
                                     c.dt, c.s)))
            return rc;
        if ((rc = gemm_nt(c, n2, d, L.w_cq, d, cq, d, R, d, d, L.b_cq, 0, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt, 0.f, 0)))
            return rc;
        ds_ctx_desc(p, G, l, pa, p->mem_kv, a);
        if ((rc = mssr_attention_fwd(&a, ctxc, (int64_t)p->T * d, d, l == NL - 1 ? p->probs : nullptr, c.s))) return rc;
        if ((rc = gemm_nt(c, ctxc, d, L.w_co, d, h2, d, R, d, d, L.b_co, 0, h1, d, nullptr, 0, 1.f, 0.f, c.dt, pd,
                          ds_seed(p, l, 3))))
            return rc;
        // --- feed forward: x' = h2 + drop(W2 drop(relu(W1 LN h2 + b1)) + b2)
        if ((rc = mssr_layernorm_fwd(h2, L.lnf_g, L.lnf_b, n3, G.stat(3 * l + 2, 0), G.stat(3 * l + 2, 1), R, d, 1e-6f,
                                     c.dt, c.s)))
            return rc;
        if ((rc = gemm_nt(c, n3, d, L.w_1, d, inter, ff, R, ff, d, L.b_1, 1, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt, pd,
                          ds_seed(p, l, 4))))
            return rc;
        if ((rc = gemm_nt(c, inter, ff, L.w_2, ff, G.X(l + 1), d, R, d, ff, L.b_2, 0, h2, d, nullptr, 0, 1.f, 0.f, c.dt, pd,
                          ds_seed(p, l, 5))))
            return rc;
    }
    return mssr_layernorm_fwd(G.X(NL), p->lnF_g, p->lnF_b, p->out, G.stat(3 * NL, 0), G.stat(3 * NL, 1), R, d, 1e-6f,
                              c.dt, c.s);
}

extern "C" int mssr_decstack_bwd(const mssr_decstack_plan* p, void* stream) {
    if (int e = ds_check(p, true, "decstack_bwd")) return e;
    const StackGeo G(p);
    GemmCtx c{(hipStream_t)stream, p->dtype, G.es, p->gemm_ws, p->gemm_ws_bytes};
    const int64_t R = G.R;
    const int d = G.d, ff = G.ff, NL = G.NL;
    const float pd = p->training ? p->p_drop : 0.f, pa = p->training ? p->p_attn : 0.f;
    const float keep = pd > 0.f ? 1.0f / (1.0f - pd) : 1.0f;
    int rc;
    auto ln_bwd = [&](const void* dy, const void* x, const float* g, int idx, const void* dres, void* dx, void* dx2,
                      uint64_t seed) {
        if (c.dt == MSSR_F32)
            launch_ln_bwd_dx<float>(c.s, dy, x, g, G.stat(idx, 0), G.stat(idx, 1), dres, dx, dx2, (int)R, d, pd, seed, 1.f);
        else
            launch_ln_bwd_dx<bf16>(c.s, dy, x, g, G.stat(idx, 0), G.stat(idx, 1), dres, dx, dx2, (int)R, d, pd, seed, 1.f);
    };
    const int64_t kv_row = (int64_t)NL * 2 * d;
    char* dx = (char*)p->dX;        // running gradient of the residual stream
    // final LayerNorm; below it sits the last layer's W2 output behind its dropout (site 5)
    ln_bwd(p->dout, G.X(NL), p->lnF_g, 3 * NL, nullptr, dx, G.dact(NL - 1, SG_DF2), ds_seed(p, NL - 1, 5));
    for (int l = NL - 1; l >= 0; --l) {
        const mssr_dec_layer& L = p->layers[l];
        char *h1 = G.act(l, S_H1), *h2 = G.act(l, S_H2), *inter = G.inter(l);
        char *dn1 = G.dact(l, SG_DN1), *d_os = G.dact(l, SG_DOS), *dn2 = G.dact(l, SG_DN2), *d_cq = G.dact(l, SG_DCQ);
        char *d_oc = G.dact(l, SG_DOC), *dn3 = G.dact(l, SG_DN3), *d_f2 = G.dact(l, SG_DF2);
        char *d_f1 = G.dinter(l), *dqkv = G.dqkv(l);
        // --- feed forward
        if ((rc = gemm_nt(c, d_f2, d, L.wt_2, d, d_f1, ff, R, ff, d, nullptr, 0, nullptr, 0, inter, ff, keep, 0.f, c.dt, 0.f,
                          0)))
            return rc;
        if ((rc = gemm_nt(c, d_f1, ff, L.wt_1, ff, dn3, d, R, d, ff, nullptr, 0, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt, 0.f,
                          0)))
            return rc;
        ln_bwd(dn3, h2, L.lnf_g, 3 * l + 2, dx, dx, d_oc, ds_seed(p, l, 3));         // dx <- grad wrt h2
        // --- context attention
        if ((rc = gemm_nt(c, d_oc, d, L.wt_co, d, p->dctx, d, R, d, d, nullptr, 0, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt, 0.f,
                          0)))
            return rc;
        mssr_attn_desc a;
        ds_ctx_desc(p, G, l, pa, p->mem_kv, a);
        char* dk = (char*)p->dmem_kv + (int64_t)l * 2 * d * G.es;
        if ((rc = mssr_attention_bwd(&a, p->dctx, (int64_t)p->T * d, d, d_cq, (int64_t)p->T * d, d, dk,
                                     (int64_t)p->Ls * kv_row, kv_row, dk + (int64_t)d * G.es, (int64_t)p->Ls * kv_row,
                                     kv_row, c.s)))
            return rc;
        if ((rc = gemm_nt(c, d_cq, d, L.wt_cq, d, dn2, d, R, d, d, nullptr, 0, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt, 0.f,
                          0)))
            return rc;
        ln_bwd(dn2, h1, L.ln2_g, 3 * l + 1, dx, dx, d_os, ds_seed(p, l, 1));         // dx <- grad wrt h1
        // --- masked self attention
        if ((rc = gemm_nt(c, d_os, d, L.wt_so, d, p->dctx, d, R, d, d, nullptr, 0, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt, 0.f,
                          0)))
            return rc;
        ds_self_desc(p, G, l, pa, a);
        const int64_t bs = (int64_t)p->T * 3 * d, ls = 3 * d;
        if ((rc = mssr_attention_bwd(&a, p->dctx, (int64_t)p->T * d, d, dqkv, bs, ls, dqkv + (int64_t)d * G.es, bs, ls,
                                     dqkv + 2LL * d * G.es, bs, ls, c.s)))
            return rc;
        if ((rc = gemm_nt(c, dqkv, 3 * d, L.wt_qkv, 3 * d, dn1, d, R, d, 3 * d, nullptr, 0, nullptr, 0, nullptr, 0, 1.f, 0.f,
                          c.dt, 0.f, 0)))
            return rc;
        // dx <- grad wrt the layer input; below it: layer l-1's W2 output behind its dropout, or the stack input
        if (l > 0) ln_bwd(dn1, G.X(l), L.ln1_g, 3 * l, dx, dx, G.dact(l - 1, SG_DF2), ds_seed(p, l - 1, 5));
        else ln_bwd(dn1, G.X(l), L.ln1_g, 3 * l, dx, dx, nullptr, 0);
    }
    MSSR_LAUNCH_OK("decstack_bwd");
}
