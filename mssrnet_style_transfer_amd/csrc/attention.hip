// Fused softmax(mask(Q K^T / sqrt(dh))) V for the short sequences of the MSSRNet path (Yelp-length
// sentences: L <= 64 keys), forward and first-order backward.
//
// Replaces the op chain of onmt/modules/multi_headed_attn.py:187-205 (scale, matmul, masked_fill(-1e18),
// fp32 softmax, dropout, matmul) and its autograd backward with ONE kernel per direction: one block per
// (sentence, head), the head's whole K and V tiles staged in LDS (rows padded so that a lane reading
// its own key row and a wave reading one row across lanes are both bank-conflict free), one wave per
// query row: lane <-> key for the scores and the softmax (wave reductions), lane <-> feature for the
// context product.  The [B,H,Lq,Lk] score / probability tensors never reach HBM unless the caller
// asks for the probabilities (the module returns them).  Backward recomputes the probabilities from
// Q and K, keeps dS and the dropped probabilities of the head in LDS, and restages Q and dO over K
// and V for the key / value gradients -- no atomics, no accumulators in memory.
#include "attention_common.hpp"
#include <stdlib.h>

namespace mssr {

constexpr int FA_MAXL = 64;      // keys (and query rows in the backward) per head
constexpr int FA_MAXD = 64;      // features per head
constexpr int FA_LDS_LIMIT = 64 * 1024;

template <typename T> __device__ __host__ constexpr int fa_pad() { return sizeof(T) == 4 ? 1 : 2; }   // odd dword stride

static inline size_t fa_fwd_lds(int Lk, int dh, size_t es) {
    const int pad = es == 4 ? 1 : 2;
    return 2 * (size_t)Lk * (dh + pad) * es + 4 * (size_t)(FA_MAXL + FA_MAXD) * sizeof(float);
}
static inline size_t fa_bwd_lds(int Lq, int Lk, int dh, size_t es) {
    const int pad = es == 4 ? 1 : 2;
    const int Lm = Lq > Lk ? Lq : Lk;
    return 2 * (size_t)Lm * (dh + pad) * es + 2 * (size_t)Lq * Lk * sizeof(float) +
           4 * (size_t)(2 * FA_MAXL + 2 * FA_MAXD) * sizeof(float);
}

// stage rows [0, n) of a [n, dh] head slice (row stride ls) into an LDS tile with row stride dh + pad
template <typename T>
__device__ __forceinline__ void fa_stage(T* tile, const T* src, int64_t ls, int n, int dh) {
    const int ld = dh + fa_pad<T>();
    const int per_row = dh / 4;
    for (int c = threadIdx.x; c < n * per_row; c += blockDim.x) {
        const int r = c / per_row, j = (c % per_row) * 4;
        float v[4];
        V4<T>::load(src + (int64_t)r * ls + j, v);
#pragma unroll
        for (int u = 0; u < 4; ++u) tile[r * ld + j + u] = from_f<T>(v[u]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a, T* __restrict__ ctx, int64_t ctx_bs,
                                                       int64_t ctx_ls, T* __restrict__ probs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int ld = a.dh + fa_pad<T>();
    T* Ks = reinterpret_cast<T*>(smem);
    T* Vs = Ks + a.Lk * ld;
    float* pw = reinterpret_cast<float*>(Vs + a.Lk * ld);      // [4][FA_MAXL] dropped probabilities of the wave's row
    float* qw = pw + 4 * FA_MAXL;                              // [4][FA_MAXD] the wave's query row
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
    fa_stage<T>(Ks, reinterpret_cast<const T*>(a.k) + (int64_t)b * a.k_bs + h * a.dh, a.k_ls, a.Lk, a.dh);
    fa_stage<T>(Vs, reinterpret_cast<const T*>(a.v) + (int64_t)b * a.v_bs + h * a.dh, a.v_ls, a.Lk, a.dh);
    __syncthreads();
    const T* qb = reinterpret_cast<const T*>(a.q) + (int64_t)b * a.q_bs + h * a.dh;
    float* p_row = pw + wave * FA_MAXL;
    float* q_row = qw + wave * FA_MAXD;
    for (int i = wave; i < a.Lq; i += 4) {          // uniform trip count per wave; no block barrier inside
        if (lane < a.dh) q_row[lane] = to_f(qb[(int64_t)i * a.q_ls + lane]);
        __builtin_amdgcn_wave_barrier();
        float s = -INFINITY;
        if (lane < a.Lk) {
            float acc = 0.f;
            const T* kr = Ks + lane * ld;
            for (int j = 0; j < a.dh; ++j) acc += q_row[j] * to_f(kr[j]);
            s = acc * a.scale;
            if (a.mask && a.mask[(int64_t)b * a.mask_bs + (int64_t)i * a.mask_qs + lane]) s = -1e18f;
        }
        const float mx = wave_max(s);
        const float e = lane < a.Lk ? expf(s - mx) : 0.f;
        const float p = e / wave_sum(e);
        if (lane < a.Lk) {
            if (probs) probs[(((int64_t)bh * a.Lq) + i) * a.Lk + lane] = from_f<T>(p);
            // the product uses the probability as the reference stores it (compute dtype), then dropout
            p_row[lane] = to_f(from_f<T>(p)) * fa_keep(a, bh, i, lane);
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < a.dh) {
            float acc = 0.f;
            for (int t = 0; t < a.Lk; ++t) acc += p_row[t] * to_f(Vs[t * ld + lane]);
            ctx[(int64_t)b * ctx_bs + (int64_t)i * ctx_ls + h * a.dh + lane] = from_f<T>(acc);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnArgs a, const T* __restrict__ dctx, int64_t dc_bs,
                                                       int64_t dc_ls, T* __restrict__ dq, int64_t dq_bs,
                                                       int64_t dq_ls, T* __restrict__ dk, int64_t dk_bs,
                                                       int64_t dk_ls, T* __restrict__ dv, int64_t dv_bs,
                                                       int64_t dv_ls) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int ld = a.dh + fa_pad<T>();
    const int Lm = a.Lq > a.Lk ? a.Lq : a.Lk;
    T* T0 = reinterpret_cast<T*>(smem);                        // K, later Q
    T* T1 = T0 + Lm * ld;                                      // V, later dO
    float* dS = reinterpret_cast<float*>(T1 + Lm * ld);        // [Lq][Lk]
    float* Pd = dS + a.Lq * a.Lk;                              // [Lq][Lk] dropped probabilities
    float* qw = Pd + a.Lq * a.Lk;                              // [4][FA_MAXD]
    float* ow = qw + 4 * FA_MAXD;                              // [4][FA_MAXD] the wave's dO row
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
    const T* qb = reinterpret_cast<const T*>(a.q) + (int64_t)b * a.q_bs + h * a.dh;
    const T* ob = dctx + (int64_t)b * dc_bs + h * a.dh;
    fa_stage<T>(T0, reinterpret_cast<const T*>(a.k) + (int64_t)b * a.k_bs + h * a.dh, a.k_ls, a.Lk, a.dh);
    fa_stage<T>(T1, reinterpret_cast<const T*>(a.v) + (int64_t)b * a.v_bs + h * a.dh, a.v_ls, a.Lk, a.dh);
    __syncthreads();
    float* q_row = qw + wave * FA_MAXD;
    float* o_row = ow + wave * FA_MAXD;
    // phase 1: one wave per query row -> dS, Pd rows in LDS, dQ row to memory
    for (int i = wave; i < a.Lq; i += 4) {
        if (lane < a.dh) {
            q_row[lane] = to_f(qb[(int64_t)i * a.q_ls + lane]);
            o_row[lane] = to_f(ob[(int64_t)i * dc_ls + lane]);
        }
        __builtin_amdgcn_wave_barrier();
        float s = -INFINITY, dp = 0.f, keep = 0.f;
        bool masked = false;
        if (lane < a.Lk) {
            float acc = 0.f, acc2 = 0.f;
            const T* kr = T0 + lane * ld;
            const T* vr = T1 + lane * ld;
            for (int j = 0; j < a.dh; ++j) {
                acc += q_row[j] * to_f(kr[j]);
                acc2 += o_row[j] * to_f(vr[j]);
            }
            s = acc * a.scale;
            masked = a.mask && a.mask[(int64_t)b * a.mask_bs + (int64_t)i * a.mask_qs + lane];
            if (masked) s = -1e18f;
            keep = fa_keep(a, bh, i, lane);
            dp = acc2 * keep;
        }
        const float mx = wave_max(s);
        const float e = lane < a.Lk ? expf(s - mx) : 0.f;
        float p = e / wave_sum(e);
        p = to_f(from_f<T>(p));                      // as stored by the forward
        const float r = wave_sum(lane < a.Lk ? p * dp : 0.f);
        if (lane < a.Lk) {
            dS[i * a.Lk + lane] = masked ? 0.f : p * (dp - r) * a.scale;
            Pd[i * a.Lk + lane] = p * keep;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < a.dh) {
            float acc = 0.f;
            for (int t = 0; t < a.Lk; ++t) acc += dS[i * a.Lk + t] * to_f(T0[t * ld + lane]);
            dq[(int64_t)b * dq_bs + (int64_t)i * dq_ls + h * a.dh + lane] = from_f<T>(acc);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // phase 2: Q and dO take the place of K and V; one wave per key row -> dK, dV
    fa_stage<T>(T0, qb, a.q_ls, a.Lq, a.dh);
    fa_stage<T>(T1, ob, dc_ls, a.Lq, a.dh);
    __syncthreads();
    for (int t = wave; t < a.Lk; t += 4) {
        if (lane < a.dh) {
            float gk = 0.f, gv = 0.f;
            for (int i = 0; i < a.Lq; ++i) {
                gk += dS[i * a.Lk + t] * to_f(T0[i * ld + lane]);
                gv += Pd[i * a.Lk + t] * to_f(T1[i * ld + lane]);
            }
            dk[(int64_t)b * dk_bs + (int64_t)t * dk_ls + h * a.dh + lane] = from_f<T>(gk);
            dv[(int64_t)b * dv_bs + (int64_t)t * dv_ls + h * a.dh + lane] = from_f<T>(gv);
        }
    }
}

static int fa_check(const char* who, const mssr_attn_desc* d) {
    MSSR_REQUIRE(d, "%s: null descriptor", who);
    MSSR_REQUIRE(valid_dtype(d->dtype), "%s: bad dtype", who);
    MSSR_REQUIRE(d->B > 0 && d->H > 0 && d->Lq > 0 && d->Lk > 0 && d->dh > 0 && d->dh % 4 == 0, "%s: bad sizes", who);
    MSSR_REQUIRE((int64_t)d->B * d->H <= 0x7fffffff, "%s: too many (sentence, head) pairs", who);
    MSSR_REQUIRE(d->q && d->k && d->v, "%s: null operand", who);
    MSSR_REQUIRE(d->p_drop >= 0.f && d->p_drop < 1.f, "%s: p_drop out of [0,1)", who);
    MSSR_REQUIRE(d->drop_b0 >= 0 && d->drop_b0 <= d->B, "%s: drop_b0 outside [0, B]", who);
    const size_t es = dsize(d->dtype);
    const int64_t strides[] = {d->q_bs, d->q_ls, d->k_bs, d->k_ls, d->v_bs, d->v_ls};
    for (int64_t st : strides) MSSR_REQUIRE(st % 4 == 0, "%s: strides must be multiples of 4 elements", who);
    MSSR_REQUIRE(aligned(d->q, 4 * es) && aligned(d->k, 4 * es) && aligned(d->v, 4 * es),
                 "%s: operands must be aligned to 4 elements", who);
    return MSSR_OK;
}

static void fa_args(const mssr_attn_desc* d, AttnArgs& a) {
    a.q = d->q; a.k = d->k; a.v = d->v; a.mask = d->mask;
    a.B = d->B; a.H = d->H; a.Lq = d->Lq; a.Lk = d->Lk; a.dh = d->dh;
    a.q_bs = d->q_bs; a.q_ls = d->q_ls; a.k_bs = d->k_bs; a.k_ls = d->k_ls; a.v_bs = d->v_bs; a.v_ls = d->v_ls;
    a.mask_bs = d->mask_bs; a.mask_qs = d->mask_qs;
    a.scale = 1.0f / sqrtf((float)d->dh);
    a.p_drop = d->p_drop; a.seed = d->seed;
    a.drop_b0 = d->drop_b0;
}

}  // namespace mssr

using namespace mssr;

// 1: prefer the matrix-core kernels wherever they fit (MSSR_ATTN_MFMA=0: only beyond the scalar kernels' reach)
static bool prefer_mfma() {
    static const bool on = [] {
        const char* e = getenv("MSSR_ATTN_MFMA");
        return !e || atoi(e) != 0;
    }();
    return on;
}
static int scalar_fits(int Lq, int Lk, int dh, int dtype, int backward);

extern "C" int mssr_attention_fits(int Lq, int Lk, int dh, int dtype, int backward) {
    if (valid_dtype(dtype) && attn_mfma_fits(Lq, Lk, dh, dtype, backward)) return 1;
    return scalar_fits(Lq, Lk, dh, dtype, backward);
}

static int scalar_fits(int Lq, int Lk, int dh, int dtype, int backward) {
    if (!valid_dtype(dtype) || Lq < 1 || Lk < 1 || Lk > FA_MAXL || dh < 4 || dh > FA_MAXD || dh % 4 != 0) return 0;
    if (backward && Lq > FA_MAXL) return 0;
    const size_t need = backward ? fa_bwd_lds(Lq, Lk, dh, dsize(dtype)) : fa_fwd_lds(Lk, dh, dsize(dtype));
    return need <= (size_t)FA_LDS_LIMIT ? 1 : 0;
}

extern "C" int mssr_attention_fwd(const mssr_attn_desc* d, void* ctx, int64_t ctx_bs, int64_t ctx_ls, void* probs,
                                  void* stream) {
    if (int e = fa_check("attention_fwd", d)) return e;
    MSSR_REQUIRE(ctx, "attention_fwd: null output");
    {
        AttnArgs a;
        fa_args(d, a);
        const bool scalar_ok = scalar_fits(d->Lq, d->Lk, d->dh, d->dtype, 0);
        if ((prefer_mfma() || !scalar_ok) && attn_mfma_fwd_try(a, d->dtype, ctx, ctx_bs, ctx_ls, probs, (hipStream_t)stream))
            MSSR_LAUNCH_OK("attention_fwd(mfma)");
    }
    MSSR_REQUIRE(scalar_fits(d->Lq, d->Lk, d->dh, d->dtype, 0),
                 "attention_fwd: Lk=%d dh=%d does not fit the fused kernel (<= %d keys, <= %d features, 64 KB LDS)",
                 d->Lk, d->dh, FA_MAXL, FA_MAXD);
    AttnArgs a;
    fa_args(d, a);
    const size_t lds = fa_fwd_lds(d->Lk, d->dh, dsize(d->dtype));
    const dim3 grid((unsigned)(d->B * d->H));
    if (d->dtype == MSSR_F32)
        attn_fwd_kernel<float><<<grid, 256, lds, (hipStream_t)stream>>>(a, (float*)ctx, ctx_bs, ctx_ls, (float*)probs);
    else
        attn_fwd_kernel<bf16><<<grid, 256, lds, (hipStream_t)stream>>>(a, (bf16*)ctx, ctx_bs, ctx_ls, (bf16*)probs);
    MSSR_LAUNCH_OK("attention_fwd");
}

extern "C" int mssr_attention_bwd(const mssr_attn_desc* d, const void* dctx, int64_t dc_bs, int64_t dc_ls, void* dq,
                                  int64_t dq_bs, int64_t dq_ls, void* dk, int64_t dk_bs, int64_t dk_ls, void* dv,
                                  int64_t dv_bs, int64_t dv_ls, void* stream) {
    if (int e = fa_check("attention_bwd", d)) return e;
    MSSR_REQUIRE(dctx && dq && dk && dv, "attention_bwd: null pointer");
    MSSR_REQUIRE(dc_bs % 4 == 0 && dc_ls % 4 == 0 && aligned(dctx, 4 * dsize(d->dtype)),
                 "attention_bwd: dctx strides / alignment must be multiples of 4 elements");
    {
        AttnArgs a;
        fa_args(d, a);
        const bool scalar_ok = scalar_fits(d->Lq, d->Lk, d->dh, d->dtype, 1);
        if ((prefer_mfma() || !scalar_ok) &&
            attn_mfma_bwd_try(a, d->dtype, dctx, dc_bs, dc_ls, dq, dq_bs, dq_ls, dk, dk_bs, dk_ls, dv, dv_bs, dv_ls, (hipStream_t)stream))
            MSSR_LAUNCH_OK("attention_bwd(mfma)");
    }
    MSSR_REQUIRE(scalar_fits(d->Lq, d->Lk, d->dh, d->dtype, 1),
                 "attention_bwd: Lq=%d Lk=%d dh=%d does not fit the fused kernel", d->Lq, d->Lk, d->dh);
    AttnArgs a;
    fa_args(d, a);
    const size_t lds = fa_bwd_lds(d->Lq, d->Lk, d->dh, dsize(d->dtype));
    const dim3 grid((unsigned)(d->B * d->H));
    if (d->dtype == MSSR_F32)
        attn_bwd_kernel<float><<<grid, 256, lds, (hipStream_t)stream>>>(a, (const float*)dctx, dc_bs, dc_ls, (float*)dq,
                                                                       dq_bs, dq_ls, (float*)dk, dk_bs, dk_ls,
                                                                       (float*)dv, dv_bs, dv_ls);
    else
        attn_bwd_kernel<bf16><<<grid, 256, lds, (hipStream_t)stream>>>(a, (const bf16*)dctx, dc_bs, dc_ls, (bf16*)dq,
                                                                      dq_bs, dq_ls, (bf16*)dk, dk_bs, dk_ls, (bf16*)dv,
                                                                      dv_bs, dv_ls);
    MSSR_LAUNCH_OK("attention_bwd");
}
