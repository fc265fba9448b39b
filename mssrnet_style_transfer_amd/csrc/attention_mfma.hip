// Fused attention on the matrix cores (bf16, 32 or 64 features per head, up to 160 keys / query rows):
// ctx = dropout(softmax(mask(Q K^T / sqrt(dh)))) V and its backward, one block per (sentence, head).
// Same op chain as attention.hip (onmt/modules/multi_headed_attn.py:187-205); this file carries it past 64 keys
// (config 5: 64-128 words per sentence) and replaces that file's scalar FMA loops wherever it fits.
//
// Everything is computed TRANSPOSED so that no tile ever changes hands between lanes:
//   S^T = K Q^T      A = 16 key rows of the K tile in LDS (ds_read_b128), B = 16 query rows straight from memory
//                    -> C layout: lane (j = lane & 15, g = lane >> 4) holds keys 16*tile + 4g + {0..3} of query j
//   softmax over keys = in-lane over the tiles + two cross-group exchanges (lanes j, j+16, j+32, j+48)
//   O^T = V^T P^T    B = P^T taken from the S^T accumulators AS THEY LIE: k-slots 0-3 <- tile 2c rows 4g..4g+3,
//                    slots 4-7 <- tile 2c+1 rows 4g..4g+3; A = V^T read from the natural [key][feature] tile with
//                    the transposing LDS read (ds_read_b64_tr_b16), its two 4-row blocks aimed at those same keys
//                    -> lane holds 4 consecutive features of query j: one 8-byte store.
// Backward, phase 1 (a wave per 16 query rows): S^T and dP^T = V dO^T in the same layout -> dS^T -> dQ^T = K^T dS^T
// (K^T through the transposing read); row statistics (max, 1/sum, sum p dP) go to LDS.  Phase 2 (a wave per 16 keys,
// Q and dO restaged over K and V): S = Q K^T and dP = dO V^T recomputed with QUERIES on the accumulator rows, so
// that P and dS come out as B operands over the query index: dV^T = dO^T P, dK^T = Q^T dS.  No atomics, no
// [B,H,Lq,Lk] tensor in memory, LDS = two natural tiles (46 KB at 160 x 64: three blocks per CU).
#include "attention_common.hpp"

namespace mssr {

constexpr int FM_MAXL = 160;
constexpr int FM_PAD = 8;          // elements: rows of dh + 8 -> 16-B fragment reads of 16 consecutive rows hit 64 distinct banks

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// rows [0, n) of a [n, DH] head slice -> LDS tile [np][DH + 8]; rows n..np zero
template <int DH>
__device__ __forceinline__ void fm_stage(short* tile, const bf16* __restrict__ src, int64_t ls, int n, int np) {
    constexpr int LD = DH + FM_PAD, PER = DH / 8;
    for (int c = threadIdx.x; c < np * PER; c += blockDim.x) {
        const int r = c / PER, j = (c % PER) * 8;
        s16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r < n) v = *reinterpret_cast<const s16x8*>(src + (int64_t)r * ls + j);
        *reinterpret_cast<s16x8*>(tile + r * LD + j) = v;
    }
}

// operand fragment over 16 consecutive tile rows: row r0 + (lane & 15), k = 32 kk + 8g .. + 7
__device__ __forceinline__ bf16x8 fm_rows(const short* tile, int ld, int r0, int kk, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + (r0 + (lane & 15)) * ld + 32 * kk + 8 * (lane >> 4));
}
// the same from memory (row clamped to the last valid one; the caller discards what clamped rows produce)
__device__ __forceinline__ bf16x8 fm_rows_g(const bf16* __restrict__ base, int64_t ls, int row, int kk, int lane) {
    return *reinterpret_cast<const bf16x8*>(base + (int64_t)row * ls + 32 * kk + 8 * (lane >> 4));
}
// transposed operand over a 32-row chunk of a natural [row][col] tile: i = column c0 + (lane & 15);
// k-slots 0-3 = rows r0 + 4g .. +3, slots 4-7 = rows r0 + 16 + 4g .. +3 (the order P^T / dS^T leave the accumulators in)
__device__ __forceinline__ bf16x8 fm_tr(const short* tile, int ld, int r0, int c0, int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const short* base = tile + (r0 + 4 * g + q) * ld + c0 + 4 * p;
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + 16 * ld));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ float fm_xsum(float v) {      // over the four lane groups (lanes j, j+16, j+32, j+48)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float fm_xmax(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// keep(b,h,i,t) * 1/(1-p) for 4 consecutive keys t0..t0+3 of query i (t0 % 4 == 0: two hash pairs)
__device__ __forceinline__ void fm_keep4(const AttnArgs& a, uint32_t thr, float inv, int bh, int i, int t0, float (&k)[4]) {
    const uint64_t idx2 = (((uint64_t)bh * a.Lq + i) * FA_KSTRIDE + t0) >> 1;
    const uint32_t m0 = keep16_pair(a.seed, idx2, thr), m1 = keep16_pair(a.seed, idx2 + 1, thr);
    k[0] = (m0 & 1) ? inv : 0.f; k[1] = (m0 & 2) ? inv : 0.f;
    k[2] = (m1 & 1) ? inv : 0.f; k[3] = (m1 & 2) ? inv : 0.f;
}

// scores of one query (lane & 15) against the keys this lane holds -> probabilities (fp32, in place), returns max and 1/sum
// mkeys: key-padding mask of the sentence staged in LDS (one byte per key, 0 beyond Lk), or null; mrow: the query's
// own mask row in memory (per-query masks), or null
template <int NC>
__device__ __forceinline__ void fm_softmax_T(f32x4 (&s)[2 * NC], const AttnArgs& a, const uint8_t* mkeys,
                                             const uint8_t* __restrict__ mrow, int g, uint32_t (&mbits)[NC], float& mx_out,
                                             float& linv_out) {
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NC; ++c) mbits[c] = 0;
#pragma unroll
    for (int t = 0; t < 2 * NC; ++t) {
        const uint32_t m4 = mkeys ? *reinterpret_cast<const uint32_t*>(mkeys + 16 * t + 4 * g) : 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * t + 4 * g + r;
            float v = s[t][r] * a.scale;
            if (key < a.Lk) {
                if (((m4 >> (8 * r)) & 0xffu) || (mrow && mrow[key])) {
                    v = -1e18f;
                    mbits[t >> 1] |= 1u << ((t & 1) * 4 + r);
                }
            } else {
                v = -INFINITY;
            }
            s[t][r] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = fm_xmax(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 2 * NC; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __expf(s[t][r] - mx);
            s[t][r] = e;
            sum += e;
        }
    sum = fm_xsum(sum);
    const float linv = 1.0f / sum;
#pragma unroll
    for (int t = 0; t < 2 * NC; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[t][r] *= linv;
    mx_out = mx;
    linv_out = linv;
}

__device__ __forceinline__ bf16x8 fm_pack(const float (&lo)[4], const float (&hi)[4]) {
    bf16x8 r;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        r[u] = (bf16)lo[u];
        r[4 + u] = (bf16)hi[u];
    }
    return r;
}

__device__ __forceinline__ void fm_store4(bf16* p, const f32x4& v) {
    bf16x4 o;
    o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}

#define FM_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)

template <int DH, int NC>      // NC = 32-key chunks (keys padded to 32 NC)
__global__ __launch_bounds__(256) void attn_mfma_fwd_kernel(const AttnArgs a, bf16* __restrict__ ctx, int64_t ctx_bs,
                                                            int64_t ctx_ls, bf16* __restrict__ probs) {
    constexpr int LD = DH + FM_PAD, KK = DH / 32, DT = DH / 16, LKP = 32 * NC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    short* Ks = reinterpret_cast<short*>(smem);
    short* Vs = Ks + LKP * LD;
    uint8_t* Mk = reinterpret_cast<uint8_t*>(Vs + LKP * LD);      // [LKP] key-padding mask bytes
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
    fm_stage<DH>(Ks, reinterpret_cast<const bf16*>(a.k) + (int64_t)b * a.k_bs + h * DH, a.k_ls, a.Lk, LKP);
    fm_stage<DH>(Vs, reinterpret_cast<const bf16*>(a.v) + (int64_t)b * a.v_bs + h * DH, a.v_ls, a.Lk, LKP);
    const bool key_mask = a.mask && a.mask_qs == 0;
    if (key_mask)
        for (int t = threadIdx.x; t < LKP; t += blockDim.x) Mk[t] = t < a.Lk ? a.mask[(int64_t)b * a.mask_bs + t] : 0;
    __syncthreads();
    const bf16* qb = reinterpret_cast<const bf16*>(a.q) + (int64_t)b * a.q_bs + h * DH;
    const uint32_t thr = drop_thr16(a.p_drop);
    const float inv = 1.0f / (1.0f - a.p_drop);
    const int nqt = (a.Lq + 15) >> 4;
    for (int qt = wave; qt < nqt; qt += nwaves) {
        const int i = 16 * qt + li, ic = i < a.Lq ? i : a.Lq - 1;
        bf16x8 qf[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) qf[kk] = fm_rows_g(qb, a.q_ls, ic, kk, lane);
        f32x4 s[2 * NC];
#pragma unroll
        for (int t = 0; t < 2 * NC; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) s[t] = FM_MFMA(fm_rows(Ks, LD, 16 * t, kk, lane), qf[kk], s[t]);
        }
        const uint8_t* mrow = a.mask && !key_mask ? a.mask + (int64_t)b * a.mask_bs + (int64_t)ic * a.mask_qs : nullptr;
        uint32_t mbits[NC];
        float mx, linv;
        fm_softmax_T<NC>(s, a, key_mask ? Mk : nullptr, mrow, g, mbits, mx, linv);
        bf16x8 pf[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float pl[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t0 = 32 * c + 16 * u + 4 * g;
                float keep[4] = {1.f, 1.f, 1.f, 1.f};
                if (a.p_drop > 0.f && b >= a.drop_b0) fm_keep4(a, thr, inv, bh, ic, t0, keep);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bf16 pb = (bf16)s[2 * c + u][r];            // the probability as the reference stores it
                    if (probs && i < a.Lq && t0 + r < a.Lk) probs[((int64_t)bh * a.Lq + i) * a.Lk + t0 + r] = pb;
                    pl[u][r] = (float)pb * keep[r];
                }
            }
            pf[c] = fm_pack(pl[0], pl[1]);
        }
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c) o[dt] = FM_MFMA(fm_tr(Vs, LD, 32 * c, 16 * dt, lane), pf[c], o[dt]);
        }
        if (i < a.Lq) {
            bf16* dst = ctx + (int64_t)b * ctx_bs + (int64_t)i * ctx_ls + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) fm_store4(dst + 16 * dt, o[dt]);
        }
    }
}

template <int DH, int NC>      // NC = 32-key chunks; query rows are walked in 32-row chunks at run time
__global__ __launch_bounds__(256) void attn_mfma_bwd_kernel(const AttnArgs a, const bf16* __restrict__ dctx, int64_t dc_bs,
                                                            int64_t dc_ls, bf16* __restrict__ dq, int64_t dq_bs,
                                                            int64_t dq_ls, bf16* __restrict__ dk, int64_t dk_bs,
                                                            int64_t dk_ls, bf16* __restrict__ dv, int64_t dv_bs,
                                                            int64_t dv_ls) {
    constexpr int LD = DH + FM_PAD, KK = DH / 32, DT = DH / 16, LKP = 32 * NC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lqp = (a.Lq + 31) & ~31;
    const int lmp = lqp > LKP ? lqp : LKP;
    short* T0 = reinterpret_cast<short*>(smem);        // K, later Q
    short* T1 = T0 + lmp * LD;                         // V, later dO
    float* st_m = reinterpret_cast<float*>(T1 + lmp * LD);   // [lqp] row max
    float* st_l = st_m + lqp;                                // [lqp] 1 / row sum
    float* st_r = st_l + lqp;                                // [lqp] sum_t p dP
    uint8_t* Mk = reinterpret_cast<uint8_t*>(st_r + lqp);    // [LKP] key-padding mask bytes
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
    const bf16* qb = reinterpret_cast<const bf16*>(a.q) + (int64_t)b * a.q_bs + h * DH;
    const bf16* kb = reinterpret_cast<const bf16*>(a.k) + (int64_t)b * a.k_bs + h * DH;
    const bf16* vb = reinterpret_cast<const bf16*>(a.v) + (int64_t)b * a.v_bs + h * DH;
    const bf16* ob = dctx + (int64_t)b * dc_bs + h * DH;
    fm_stage<DH>(T0, kb, a.k_ls, a.Lk, LKP);
    fm_stage<DH>(T1, vb, a.v_ls, a.Lk, LKP);
    const bool key_mask = a.mask && a.mask_qs == 0;
    if (key_mask)
        for (int t = threadIdx.x; t < LKP; t += blockDim.x) Mk[t] = t < a.Lk ? a.mask[(int64_t)b * a.mask_bs + t] : 0;
    for (int r = threadIdx.x; r < lqp; r += blockDim.x) {
        st_m[r] = 0.f;
        st_l[r] = 0.f;
        st_r[r] = 0.f;
    }
    __syncthreads();
    const uint32_t thr = drop_thr16(a.p_drop);
    const float inv = 1.0f / (1.0f - a.p_drop);
    // ---- phase 1: a wave per 16 query rows -> dQ, row statistics
    const int nqt = (a.Lq + 15) >> 4;
    for (int qt = wave; qt < nqt; qt += nwaves) {
        const int i = 16 * qt + li, ic = i < a.Lq ? i : a.Lq - 1;
        bf16x8 qf[KK], of[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            qf[kk] = fm_rows_g(qb, a.q_ls, ic, kk, lane);
            of[kk] = fm_rows_g(ob, dc_ls, ic, kk, lane);
        }
        f32x4 s[2 * NC], dp[2 * NC];
#pragma unroll
        for (int t = 0; t < 2 * NC; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                s[t] = FM_MFMA(fm_rows(T0, LD, 16 * t, kk, lane), qf[kk], s[t]);
                dp[t] = FM_MFMA(fm_rows(T1, LD, 16 * t, kk, lane), of[kk], dp[t]);
            }
        }
        const uint8_t* mrow = a.mask && !key_mask ? a.mask + (int64_t)b * a.mask_bs + (int64_t)ic * a.mask_qs : nullptr;
        uint32_t mbits[NC];
        float mx, linv;
        fm_softmax_T<NC>(s, a, key_mask ? Mk : nullptr, mrow, g, mbits, mx, linv);
        float rsum = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * NC; ++t) {
            float keep[4] = {1.f, 1.f, 1.f, 1.f};
            if (a.p_drop > 0.f && b >= a.drop_b0) fm_keep4(a, thr, inv, bh, ic, 16 * t + 4 * g, keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = (float)(bf16)s[t][r];
                s[t][r] = p;
                dp[t][r] *= keep[r];
                rsum += p * dp[t][r];
            }
        }
        rsum = fm_xsum(rsum);
        if (g == 0 && i < a.Lq) {
            st_m[i] = mx;
            st_l[i] = linv;
            st_r[i] = rsum;
        }
        bf16x8 dsf[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float dl[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool masked = (mbits[c] >> (u * 4 + r)) & 1u;
                    dl[u][r] = masked ? 0.f : s[2 * c + u][r] * (dp[2 * c + u][r] - rsum) * a.scale;
                }
            dsf[c] = fm_pack(dl[0], dl[1]);
        }
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c) o[dt] = FM_MFMA(fm_tr(T0, LD, 32 * c, 16 * dt, lane), dsf[c], o[dt]);
        }
        if (i < a.Lq) {
            bf16* dst = dq + (int64_t)b * dq_bs + (int64_t)i * dq_ls + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) fm_store4(dst + 16 * dt, o[dt]);
        }
    }
    __syncthreads();
    // ---- phase 2: Q and dO take the place of K and V; a wave per 16 keys -> dK, dV
    fm_stage<DH>(T0, qb, a.q_ls, a.Lq, lqp);
    fm_stage<DH>(T1, ob, dc_ls, a.Lq, lqp);
    __syncthreads();
    const int nkt = (a.Lk + 15) >> 4;
    const int nqc = lqp >> 5;
    for (int kt = wave; kt < nkt; kt += nwaves) {
        const int t = 16 * kt + li, tc = t < a.Lk ? t : a.Lk - 1;
        bf16x8 kf[KK], vf[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            kf[kk] = fm_rows_g(kb, a.k_ls, tc, kk, lane);
            vf[kk] = fm_rows_g(vb, a.v_ls, tc, kk, lane);
        }
        const bool key_masked_all = key_mask && Mk[tc];
        f32x4 gk[DT], gv[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            gk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            gv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int c = 0; c < nqc; ++c) {
            float pl[2][4], dl[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r0 = 32 * c + 16 * u;
                f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, da = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    sa = FM_MFMA(fm_rows(T0, LD, r0, kk, lane), kf[kk], sa);
                    da = FM_MFMA(fm_rows(T1, LD, r0, kk, lane), vf[kk], da);
                }
                const int i0 = r0 + 4 * g;
                const f32x4 m4 = *reinterpret_cast<const f32x4*>(st_m + i0);
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(st_l + i0);
                const f32x4 r4 = *reinterpret_cast<const f32x4*>(st_r + i0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + r;
                    const bool live = i < a.Lq && t < a.Lk;
                    bool masked = key_masked_all;
                    if (a.mask && a.mask_qs != 0 && live) masked = a.mask[(int64_t)b * a.mask_bs + (int64_t)i * a.mask_qs + t];
                    const float v = masked ? -1e18f : sa[r] * a.scale;
                    float p = live ? (float)(bf16)(__expf(v - m4[r]) * l4[r]) : 0.f;
                    float keep = 1.f;
                    if (a.p_drop > 0.f && b >= a.drop_b0)
                        keep = keep16(a.seed, ((uint64_t)bh * a.Lq + (live ? i : 0)) * FA_KSTRIDE + tc, thr) ? inv : 0.f;
                    pl[u][r] = p * keep;
                    dl[u][r] = masked ? 0.f : p * (da[r] * keep - r4[r]) * a.scale;
                }
            }
            const bf16x8 pf = fm_pack(pl[0], pl[1]), df = fm_pack(dl[0], dl[1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                gv[dt] = FM_MFMA(fm_tr(T1, LD, 32 * c, 16 * dt, lane), pf, gv[dt]);
                gk[dt] = FM_MFMA(fm_tr(T0, LD, 32 * c, 16 * dt, lane), df, gk[dt]);
            }
        }
        if (t < a.Lk) {
            bf16* dkp = dk + (int64_t)b * dk_bs + (int64_t)t * dk_ls + h * DH + 4 * g;
            bf16* dvp = dv + (int64_t)b * dv_bs + (int64_t)t * dv_ls + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                fm_store4(dkp + 16 * dt, gk[dt]);
                fm_store4(dvp + 16 * dt, gv[dt]);
            }
        }
    }
}

static size_t fm_fwd_lds(int Lk, int dh) {
    const int lkp = (Lk + 31) & ~31;
    return 2 * (size_t)lkp * (dh + FM_PAD) * 2 + lkp;
}
static size_t fm_bwd_lds(int Lq, int Lk, int dh) {
    const int lqp = (Lq + 31) & ~31, lkp = (Lk + 31) & ~31, lmp = lqp > lkp ? lqp : lkp;
    return 2 * (size_t)lmp * (dh + FM_PAD) * 2 + 3 * (size_t)lqp * sizeof(float) + lkp;
}

bool attn_mfma_fits(int Lq, int Lk, int dh, int dtype, int backward) {
    if (dtype != MSSR_BF16 || (dh != 32 && dh != 64)) return false;
    if (Lq < 1 || Lk < 1 || Lk > FM_MAXL) return false;
    if (backward && Lq > FM_MAXL) return false;
    return true;
}

static bool fm_vec_ok(const void* p, int64_t bs, int64_t ls, size_t bytes) {
    return aligned(p, bytes) && (bs * 2) % (int64_t)bytes == 0 && (ls * 2) % (int64_t)bytes == 0;
}

template <int DH>
static bool fm_fwd_launch(const AttnArgs& a, bf16* ctx, int64_t ctx_bs, int64_t ctx_ls, bf16* probs, hipStream_t s) {
    const int nc = (a.Lk + 31) / 32;
    const int nqt = (a.Lq + 15) / 16;
    const dim3 grid((unsigned)(a.B * a.H)), block(64u * (nqt < 4 ? nqt : 4));
    const unsigned lds = (unsigned)fm_fwd_lds(a.Lk, DH);
    switch (nc) {
    case 1: klaunch(attn_mfma_fwd_kernel<DH, 1>, grid, block, lds, s, a, ctx, ctx_bs, ctx_ls, probs); break;
    case 2: klaunch(attn_mfma_fwd_kernel<DH, 2>, grid, block, lds, s, a, ctx, ctx_bs, ctx_ls, probs); break;
    case 3: klaunch(attn_mfma_fwd_kernel<DH, 3>, grid, block, lds, s, a, ctx, ctx_bs, ctx_ls, probs); break;
    case 4: klaunch(attn_mfma_fwd_kernel<DH, 4>, grid, block, lds, s, a, ctx, ctx_bs, ctx_ls, probs); break;
    case 5: klaunch(attn_mfma_fwd_kernel<DH, 5>, grid, block, lds, s, a, ctx, ctx_bs, ctx_ls, probs); break;
    default: return false;
    }
    return true;
}

bool attn_mfma_fwd_try(const AttnArgs& a, int dtype, void* ctx, int64_t ctx_bs, int64_t ctx_ls, void* probs, hipStream_t s) {
    if (!attn_mfma_fits(a.Lq, a.Lk, a.dh, dtype, 0)) return false;
    if (!fm_vec_ok(a.q, a.q_bs, a.q_ls, 16) || !fm_vec_ok(a.k, a.k_bs, a.k_ls, 16) || !fm_vec_ok(a.v, a.v_bs, a.v_ls, 16) ||
        !fm_vec_ok(ctx, ctx_bs, ctx_ls, 8))
        return false;
    return a.dh == 64 ? fm_fwd_launch<64>(a, (bf16*)ctx, ctx_bs, ctx_ls, (bf16*)probs, s)
                      : fm_fwd_launch<32>(a, (bf16*)ctx, ctx_bs, ctx_ls, (bf16*)probs, s);
}

template <int DH>
static bool fm_bwd_launch(const AttnArgs& a, const bf16* dctx, int64_t dc_bs, int64_t dc_ls, bf16* dq, int64_t dq_bs,
                          int64_t dq_ls, bf16* dk, int64_t dk_bs, int64_t dk_ls, bf16* dv, int64_t dv_bs, int64_t dv_ls,
                          hipStream_t s) {
    const int nc = (a.Lk + 31) / 32;
    const int lm = a.Lq > a.Lk ? a.Lq : a.Lk;
    const int nt = (lm + 15) / 16;
    const dim3 grid((unsigned)(a.B * a.H)), block(64u * (nt < 4 ? nt : 4));
    const unsigned lds = (unsigned)fm_bwd_lds(a.Lq, a.Lk, DH);
#define FM_BWD(NCV)                                                                                                        \
    klaunch(attn_mfma_bwd_kernel<DH, NCV>, grid, block, lds, s, a, dctx, dc_bs, dc_ls, dq, dq_bs, dq_ls, dk, dk_bs, dk_ls, \
            dv, dv_bs, dv_ls)
    switch (nc) {
    case 1: FM_BWD(1); break;
    case 2: FM_BWD(2); break;
    case 3: FM_BWD(3); break;
    case 4: FM_BWD(4); break;
    case 5: FM_BWD(5); break;
    default: return false;
    }
#undef FM_BWD
    return true;
}

bool attn_mfma_bwd_try(const AttnArgs& a, int dtype, const void* dctx, int64_t dc_bs, int64_t dc_ls, void* dq, int64_t dq_bs,
                       int64_t dq_ls, void* dk, int64_t dk_bs, int64_t dk_ls, void* dv, int64_t dv_bs, int64_t dv_ls,
                       hipStream_t s) {
    if (!attn_mfma_fits(a.Lq, a.Lk, a.dh, dtype, 1)) return false;
    if (!fm_vec_ok(a.q, a.q_bs, a.q_ls, 16) || !fm_vec_ok(a.k, a.k_bs, a.k_ls, 16) || !fm_vec_ok(a.v, a.v_bs, a.v_ls, 16) ||
        !fm_vec_ok(dctx, dc_bs, dc_ls, 16) || !fm_vec_ok(dq, dq_bs, dq_ls, 8) || !fm_vec_ok(dk, dk_bs, dk_ls, 8) ||
        !fm_vec_ok(dv, dv_bs, dv_ls, 8))
        return false;
    return a.dh == 64 ? fm_bwd_launch<64>(a, (const bf16*)dctx, dc_bs, dc_ls, (bf16*)dq, dq_bs, dq_ls, (bf16*)dk, dk_bs, dk_ls,
                                          (bf16*)dv, dv_bs, dv_ls, s)
                      : fm_bwd_launch<32>(a, (const bf16*)dctx, dc_bs, dc_ls, (bf16*)dq, dq_bs, dq_ls, (bf16*)dk, dk_bs, dk_ls,
                                          (bf16*)dv, dv_bs, dv_ls, s);
}

}  // namespace mssr
