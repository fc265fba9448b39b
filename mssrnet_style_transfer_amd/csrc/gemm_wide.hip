// Vocabulary-wide products with few rows: C[M, N] = A[M, K] . W[N, K]^T, N in the thousands, M <= a few hundred,
// K = 256 or 512 -- the generator logits of every decode step (M x 9606 x 512, LayerNorm folded in:
// onmt/train_and_translate.py:95-116 after onmt/decoders/transformer.py:331) and the input gradient of the soft
// embedding, dprobs += d_e E^T (onmt/modules/embeddings.py:278-280 backward).
//
// The K-split kernel's 32x32 tiles re-read both operands through L2 once per tile (240 x 9606 x 512: 2408 blocks x
// 64 KB = 154 MB for a 14 MB problem, 26 us: L2-bandwidth bound).  Here a block owns 48 output columns for ALL rows
// (up to 256 per block): its W slice (48 x K, 50 KB) is staged in LDS once, every wave streams whole 16-row fragments
// of A straight from memory into MFMA registers (all k-steps of a fragment in flight at once, the next fragment's
// loads issued before the current one's MFMAs), so A crosses L2 once per block and W once in total: 201 blocks x
// 295 KB = 59 MB.  A wave owns whole rows, so the LayerNorm statistics of the folded form are its own lane sums.
#include "gemm_common.hpp"
#include <stdlib.h>

namespace mssr {

constexpr int GW_NF = 3;                   // 16-column fragments per block: 48 columns (9606 -> 201 blocks on 256 CUs)
constexpr int GW_COLS = 16 * GW_NF;
constexpr int GW_ROWS = 256;               // rows per block (blockIdx.y walks taller problems)
constexpr int GW_PAD = 8;                  // LDS row padding (elements): 16-byte fragment reads of 16 rows hit distinct banks

template <typename TO, int KS, bool LN>    // KS = k-steps of 32 (K = 32 KS)
__global__ __launch_bounds__(256) void gemm_wide_kernel(const GemmArgs p) {
    constexpr int K = 32 * KS, LD = K + GW_PAD;
    __shared__ __attribute__((aligned(16))) short Ws[GW_COLS * LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int64_t n0 = (int64_t)blockIdx.x * GW_COLS, m0 = (int64_t)blockIdx.y * GW_ROWS;
    const bf16* A = reinterpret_cast<const bf16*>(p.a);
    const bf16* W = reinterpret_cast<const bf16*>(p.b);
    // W slice -> LDS (rows past N: zeros)
    for (int c = threadIdx.x; c < GW_COLS * (K / 8); c += 256) {
        const int r = c / (K / 8), j = (c % (K / 8)) * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (n0 + r < p.n) v = *reinterpret_cast<const uint4*>(W + (n0 + r) * p.b_cs + j);
        *reinterpret_cast<uint4*>(Ws + r * LD + j) = v;
    }
    int64_t rows_here = p.m - m0;
    if (rows_here > GW_ROWS) rows_here = GW_ROWS;
    const int nfrag = (int)((rows_here + 15) >> 4);
    // epilogue column operands of this lane: columns n0 + 16 j + li
    float bias[GW_NF], csum[GW_NF];
    bool cok[GW_NF];
#pragma unroll
    for (int j = 0; j < GW_NF; ++j) {
        const int64_t col = n0 + 16 * j + li;
        cok[j] = col < p.n;
        bias[j] = cok[j] && p.bias ? p.bias[col] : 0.f;
        csum[j] = LN && cok[j] ? p.ln_colsum[col] : 0.f;
    }
    auto load_frag = [&](int f, bf16x8 (&fa)[KS]) {
        int64_t r = m0 + 16 * f + li;
        r = r < p.m ? r : p.m - 1;
        const bf16* src = A + r * p.a_rs + 8 * g;
#pragma unroll
        for (int s = 0; s < KS; ++s) fa[s] = *reinterpret_cast<const bf16x8*>(src + 32 * s);
    };
    bf16x8 cur[KS], nxt[KS];
    int f = wave;
    if (f < nfrag) load_frag(f, cur);
    __syncthreads();                       // W slice staged
    TO* C = reinterpret_cast<TO*>(p.c);
    for (; f < nfrag; f += 4) {
        const bool more = f + 4 < nfrag;
        if (more) load_frag(f + 4, nxt);   // in flight under this fragment's MFMAs
        f32x4 acc[GW_NF];
#pragma unroll
        for (int j = 0; j < GW_NF; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        float sx = 0.f, sxx = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int j = 0; j < GW_NF; ++j) {
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(Ws + (16 * j + li) * LD + 32 * s + 8 * g);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[s], fb, acc[j], 0, 0, 0);
            }
            if (LN) {           // row sums on the packed dot instruction: 8 issues per fragment instead of 24
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bf16x2 x2 = {cur[s][2 * u], cur[s][2 * u + 1]};
                    sx = __builtin_amdgcn_fdot2_f32_bf16(x2, ones, sx, false);
                    sxx = __builtin_amdgcn_fdot2_f32_bf16(x2, x2, sxx, false);
                }
            }
        }
        float mean = 0.f, rstd = 1.f;
        if (LN) {                          // row li of the fragment: its four lane groups hold the k-slices
            sx += __shfl_xor(sx, 16, 64);
            sx += __shfl_xor(sx, 32, 64);
            sxx += __shfl_xor(sxx, 16, 64);
            sxx += __shfl_xor(sxx, 32, 64);
            mean = sx / (float)K;
            rstd = rsqrtf(fmaxf(sxx / (float)K - mean * mean, 0.f) + p.ln_eps);
            const int64_t row = m0 + 16 * f + li;
            if (blockIdx.x == 0 && g == 0 && row < p.m) {
                if (p.ln_mean) p.ln_mean[row] = mean;
                if (p.ln_rstd) p.ln_rstd[row] = rstd;
            }
        }
        // MFMA C layout: this lane holds rows 4g + r of the fragment, column li of every column fragment
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = m0 + 16 * f + 4 * g + r;
            const float mr = LN ? __shfl(mean, 4 * g + r, 64) : 0.f;
            const float rr = LN ? __shfl(rstd, 4 * g + r, 64) : 1.f;
            if (row >= p.m) continue;
#pragma unroll
            for (int j = 0; j < GW_NF; ++j) {
                if (!cok[j]) continue;
                const int64_t idx = row * p.ldc + n0 + 16 * j + li;
                float v = acc[j][r];
                if (LN) v = rr * (v - mr * csum[j]);
                v = p.alpha * v + bias[j];
                if constexpr (sizeof(TO) == 4) {
                    if (p.beta == 1.f) {       // accumulate: one add per address and launch, no read latency in the wave
                        atomicAdd(reinterpret_cast<float*>(C) + idx, v);
                        continue;
                    }
                }
                if (p.beta != 0.f) v += p.beta * to_f(C[idx]);
                C[idx] = from_f<TO>(v);
            }
        }
        if (more) {
#pragma unroll
            for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
        }
    }
}

bool gemm_wide_try(const GemmArgs& a, int in_dtype, int out_dtype, int64_t batch, hipStream_t s) {
    static const bool enabled = [] {
        const char* e = getenv("MSSR_GEMM_WIDE");
        return !e || atoi(e) != 0;
    }();
    if (!enabled || in_dtype != MSSR_BF16 || batch != 1) return false;
    if (a.k != 256 && a.k != 512) return false;
    if (a.n < 2048 || a.m < 64 || a.m > 4096) return false;
    if (a.a_cs != 1 || a.b_rs != 1 || a.a_rs % 8 != 0 || a.b_cs % 8 != 0 || !aligned(a.a, 16) || !aligned(a.b, 16)) return false;
    if (a.residual || a.gate || a.relu || a.drop_p > 0.f || a.lnb_x) return false;
    const dim3 grid((unsigned)((a.n + GW_COLS - 1) / GW_COLS), (unsigned)((a.m + GW_ROWS - 1) / GW_ROWS));
#define GW_LAUNCH(TO, KS)                                                                            \
    do {                                                                                             \
        if (a.ln_colsum) klaunch(gemm_wide_kernel<TO, KS, true>, grid, dim3(256), 0, s, a);          \
        else klaunch(gemm_wide_kernel<TO, KS, false>, grid, dim3(256), 0, s, a);                     \
    } while (0)
    if (out_dtype == MSSR_BF16) {
        if (a.k == 512) GW_LAUNCH(bf16, 16); else GW_LAUNCH(bf16, 8);
    } else {
        if (a.k == 512) GW_LAUNCH(float, 16); else GW_LAUNCH(float, 8);
    }
#undef GW_LAUNCH
    return true;
}

}  // namespace mssr
