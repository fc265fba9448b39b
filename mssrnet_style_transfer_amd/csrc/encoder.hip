// Encoder-stack engine: one C call runs all layers of a TransformerEncoder forward, another its
// exact reverse, every kernel enqueued back to back on the stream.
//
// Replaces the Python layer loop of the reference's encoder path -- TransformerEncoder.forward
// (onmt/encoders/transformer.py:117-136) -> n_layers x TransformerEncoderLayer.forward (:37-52:
// LN -> MultiHeadedAttention (onmt/modules/multi_headed_attn.py:161-227) -> dropout + residual ->
// PositionwiseFeedForward (onmt/modules/position_ffn.py:25-37)) -> final LayerNorm -- and the same
// loop inside StyleGenerator2.forward (onmt/style_generator.py:40-47), plus their autograd backward.
// Per layer forward: [LN] -> packed QKV projection -> fused attention -> out-proj (+bias, dropout,
// residual in the epilogue) -> [LN] -> FFN1 (+bias, ReLU, dropout) -> FFN2 (+bias, dropout, residual).
// Forward-only plans (save = 0, bf16) fold both LayerNorms into the projection behind them.
// Weight / bias / LayerNorm gradients are NOT formed here: backward leaves the gradient of every
// GEMM output in a per-layer slab and the caller contracts each kind once over all layers.
#include "engine_common.hpp"

namespace mssr {

enum { E_N1 = 0, E_CTX, E_H, E_N2, E_NKINDS };
enum { EG_DN1 = 0, EG_DO, EG_DN2, EG_DF2, EG_NKINDS };

static inline uint64_t enc_seed(const mssr_encoder_plan* p, int layer, int site) {
    return p->seed + (uint64_t)(layer + 1) * 0x9E3779B1ull + (uint64_t)site * 7919ull;
}

static int enc_check(const mssr_encoder_plan* p, bool backward, const char* who) {
    MSSR_REQUIRE(p, "%s: null plan", who);
    MSSR_REQUIRE(valid_dtype(p->dtype), "%s: bad dtype", who);
    MSSR_REQUIRE(p->B > 0 && p->L > 0 && p->d > 0 && p->heads > 0 && p->d % p->heads == 0 && p->d % 8 == 0 &&
                     p->ff % 8 == 0 && p->n_layers > 0 && p->d <= 4096,
                 "%s: bad dimensions", who);
    MSSR_REQUIRE(p->layers && p->lnF_g && p->lnF_b && p->X && p->out && p->qkv && p->actD && p->actF && p->stats,
                 "%s: null buffer in plan", who);
    MSSR_REQUIRE(p->p_drop >= 0.f && p->p_drop < 1.f && p->p_attn >= 0.f && p->p_attn < 1.f, "%s: bad dropout", who);
    MSSR_REQUIRE(p->eval_rows >= 0 && p->eval_rows <= p->B, "%s: eval_rows outside [0, B]", who);
    const int dh = p->d / p->heads;
    MSSR_REQUIRE(mssr_attention_fits(p->L, p->L, dh, p->dtype, backward ? 1 : 0),
                 "%s: L=%d, dh=%d is beyond the fused attention kernel", who, p->L, dh);
    if (p->fuse_ln) {
        MSSR_REQUIRE(!p->save && p->dtype == MSSR_BF16 && p->d % 32 == 0 && p->d <= 2048 && (int64_t)p->B * p->L <= 8192,
                     "%s: fuse_ln needs a forward-only bf16 plan, a model dim that is a multiple of 32, <= 2048, and B*L <= 8192",
                     who);
        for (int l = 0; l < p->n_layers; ++l) {
            const mssr_enc_layer& L = p->layers[l];
            MSSR_REQUIRE(L.wg_qkv && L.wg_1 && L.cs_qkv && L.cs_1 && L.bf_qkv && L.bf_1,
                         "%s: fuse_ln needs the folded weights of layer %d", who, l);
        }
    }
    if (backward) {
        MSSR_REQUIRE(p->save, "%s: the forward did not save activations (save = 0)", who);
        MSSR_REQUIRE(p->dout && p->dX && p->dqkv && p->dactD && p->dactF && p->dctx,
                     "%s: null gradient buffer in plan", who);
    }
    return MSSR_OK;
}

struct EncGeo {
    const mssr_encoder_plan* p;
    int64_t R;
    int d, ff, NL;
    size_t es;
    EncGeo(const mssr_encoder_plan* p_) : p(p_), R((int64_t)p_->B * p_->L), d(p_->d), ff(p_->ff), NL(p_->n_layers),
                                         es(dsize(p_->dtype)) {}
    int slot(int l) const { return p->save ? l : 0; }
    char* X(int l) const { return at(p->X, (int64_t)l * R * d, es); }
    char* qkv(int l) const { return at(p->qkv, (int64_t)slot(l) * R * 3 * d, es); }
    char* act(int l, int kind) const { return at(p->actD, ((int64_t)slot(l) * E_NKINDS + kind) * R * d, es); }
    char* inter(int l) const { return at(p->actF, (int64_t)slot(l) * R * ff, es); }
    float* stat(int idx, int which) const { return p->stats + ((int64_t)idx * 2 + which) * R; }
    char* dqkv(int l) const { return at(p->dqkv, (int64_t)l * R * 3 * d, es); }
    char* dact(int l, int kind) const { return at(p->dactD, ((int64_t)l * EG_NKINDS + kind) * R * d, es); }
    char* dinter(int l) const { return at(p->dactF, (int64_t)l * R * ff, es); }
};

static void enc_attn_desc(const mssr_encoder_plan* p, const EncGeo& G, int l, float p_attn, mssr_attn_desc& a) {
    const int d = p->d;
    char* qkv = G.qkv(l);
    a.q = qkv; a.k = qkv + (int64_t)d * G.es; a.v = qkv + 2LL * d * G.es;
    a.mask = p->mask;
    a.B = p->B; a.H = p->heads; a.Lq = p->L; a.Lk = p->L; a.dh = d / p->heads;
    a.q_bs = a.k_bs = a.v_bs = (int64_t)p->L * 3 * d;
    a.q_ls = a.k_ls = a.v_ls = 3 * d;
    a.mask_bs = p->L; a.mask_qs = 0;
    a.p_drop = p_attn; a.seed = enc_seed(p, l, 0);
    a.dtype = p->dtype;
    a.drop_b0 = p->eval_rows;
}

}  // namespace mssr

using namespace mssr;

extern "C" int mssr_encoder_fwd(const mssr_encoder_plan* p, void* stream) {
    if (int e = enc_check(p, false, "encoder_fwd")) return e;
    const EncGeo G(p);
    GemmCtx c{(hipStream_t)stream, p->dtype, G.es, p->gemm_ws, p->gemm_ws_bytes};
    const int64_t R = G.R;
    const int d = G.d, ff = G.ff, NL = G.NL;
    const float pd = p->training ? p->p_drop : 0.f, pa = p->training ? p->p_attn : 0.f;
    const bool fuse = p->fuse_ln != 0;
    const int64_t row0 = (int64_t)p->eval_rows * p->L;        // rows below it (eval-mode sentences) skip dropout
    int rc;
    for (int l = 0; l < NL; ++l) {
        const mssr_enc_layer& L = p->layers[l];
        char *x = G.X(l), *n1 = G.act(l, E_N1), *ctx = G.act(l, E_CTX), *h = G.act(l, E_H), *n2 = G.act(l, E_N2);
        char *qkv = G.qkv(l), *inter = G.inter(l);
        // --- self attention block
        if (fuse) {
            if ((rc = gemm_nt(c, x, d, L.wg_qkv, d, qkv, 3 * d, R, 3 * d, d, L.bf_qkv, 0, nullptr, 0, nullptr, 0, 1.f, 0.f,
                              c.dt, 0.f, 0, L.cs_qkv, G.stat(2 * l, 0), G.stat(2 * l, 1))))
                return rc;
        } else {
            if ((rc = mssr_layernorm_fwd(x, L.ln1_g, L.ln1_b, n1, G.stat(2 * l, 0), G.stat(2 * l, 1), R, d, 1e-6f, c.dt,
                                         c.s)))
                return rc;
            if ((rc = gemm_nt(c, n1, d, L.w_qkv, d, qkv, 3 * d, R, 3 * d, d, L.b_qkv, 0, nullptr, 0, nullptr, 0, 1.f, 0.f,
                              c.dt, 0.f, 0)))
                return rc;
        }
        mssr_attn_desc a;
        enc_attn_desc(p, G, l, pa, a);
        if ((rc = mssr_attention_fwd(&a, ctx, (int64_t)p->L * d, d, nullptr, c.s))) return rc;
        if ((rc = gemm_nt(c, ctx, d, L.w_o, d, h, d, R, d, d, L.b_o, 0, x, d, nullptr, 0, 1.f, 0.f, c.dt, pd,
                          enc_seed(p, l, 1), nullptr, nullptr, nullptr, row0)))
            return rc;
        // --- feed forward block
        if (fuse) {
            if ((rc = gemm_nt(c, h, d, L.wg_1, d, inter, ff, R, ff, d, L.bf_1, 1, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt, pd,
                              enc_seed(p, l, 2), L.cs_1, G.stat(2 * l + 1, 0), G.stat(2 * l + 1, 1), row0)))
                return rc;
        } else {
            if ((rc = mssr_layernorm_fwd(h, L.lnf_g, L.lnf_b, n2, G.stat(2 * l + 1, 0), G.stat(2 * l + 1, 1), R, d, 1e-6f,
                                         c.dt, c.s)))
                return rc;
            if ((rc = gemm_nt(c, n2, d, L.w_1, d, inter, ff, R, ff, d, L.b_1, 1, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt, pd,
                              enc_seed(p, l, 2), nullptr, nullptr, nullptr, row0)))
                return rc;
        }
        if ((rc = gemm_nt(c, inter, ff, L.w_2, ff, G.X(l + 1), d, R, d, ff, L.b_2, 0, h, d, nullptr, 0, 1.f, 0.f, c.dt, pd,
                          enc_seed(p, l, 3), nullptr, nullptr, nullptr, row0)))
            return rc;
    }
    return mssr_layernorm_fwd(G.X(NL), p->lnF_g, p->lnF_b, p->out, G.stat(2 * NL, 0), G.stat(2 * NL, 1), R, d, 1e-6f,
                              c.dt, c.s);
}

extern "C" int mssr_encoder_bwd(const mssr_encoder_plan* p, void* stream) {
    if (int e = enc_check(p, true, "encoder_bwd")) return e;
    const EncGeo G(p);
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
    char* dx = (char*)p->dX;        // running gradient of the residual stream
    // final LayerNorm; below it sits the last layer's W2 output behind its dropout (site 3)
    ln_bwd(p->dout, G.X(NL), p->lnF_g, 2 * NL, nullptr, dx, G.dact(NL - 1, EG_DF2), enc_seed(p, NL - 1, 3));
    for (int l = NL - 1; l >= 0; --l) {
        const mssr_enc_layer& L = p->layers[l];
        char *h = G.act(l, E_H), *inter = G.inter(l);
        char *dn1 = G.dact(l, EG_DN1), *d_o = G.dact(l, EG_DO), *dn2 = G.dact(l, EG_DN2), *d_f2 = G.dact(l, EG_DF2);
        char *d_f1 = G.dinter(l), *dqkv = G.dqkv(l);
        // --- feed forward block: y = h + drop(W2 inter + b2), inter = drop(relu(W1 n2 + b1))
        if ((rc = gemm_nt(c, d_f2, d, L.wt_2, d, d_f1, ff, R, ff, d, nullptr, 0, nullptr, 0, inter, ff, keep, 0.f, c.dt, 0.f,
                          0)))
            return rc;
        if ((rc = gemm_nt(c, d_f1, ff, L.wt_1, ff, dn2, d, R, d, ff, nullptr, 0, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt, 0.f,
                          0)))
            return rc;
        ln_bwd(dn2, h, L.lnf_g, 2 * l + 1, dx, dx, d_o, enc_seed(p, l, 1));        // dx <- grad wrt h
        // --- self attention block: h = x + drop(Wo ctx + bo)
        if ((rc = gemm_nt(c, d_o, d, L.wt_o, d, p->dctx, d, R, d, d, nullptr, 0, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt, 0.f,
                          0)))
            return rc;
        mssr_attn_desc a;
        enc_attn_desc(p, G, l, pa, a);
        const int64_t bs = (int64_t)p->L * 3 * d, ls = 3 * d;
        if ((rc = mssr_attention_bwd(&a, p->dctx, (int64_t)p->L * d, d, dqkv, bs, ls, dqkv + (int64_t)d * G.es, bs, ls,
                                     dqkv + 2LL * d * G.es, bs, ls, c.s)))
            return rc;
        if ((rc = gemm_nt(c, dqkv, 3 * d, L.wt_qkv, 3 * d, dn1, d, R, d, 3 * d, nullptr, 0, nullptr, 0, nullptr, 0, 1.f, 0.f,
                          c.dt, 0.f, 0)))
            return rc;
        // dx <- grad wrt the layer input; below it: layer l-1's W2 output behind its dropout, or the stack input
        if (l > 0) ln_bwd(dn1, G.X(l), L.ln1_g, 2 * l, dx, dx, G.dact(l - 1, EG_DF2), enc_seed(p, l - 1, 3));
        else ln_bwd(dn1, G.X(l), L.ln1_g, 2 * l, dx, dx, nullptr, 0);
    }
    MSSR_LAUNCH_OK("encoder_bwd");
}
