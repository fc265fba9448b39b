// LayerNorm backward fused with the Linear backward behind it, for the decode engine's reverse sweep:
//     dx  = dres + rstd (g - mean(g) - xhat mean(g xhat)),   g = gamma * dy          (onmt LayerNorm, eps 1e-6)
//     dx2 = dropout_replay(dx)                                (the residual-dropout site below the LayerNorm)
//     C   = epilogue(dx2 . W^T)                               (gradient w.r.t. the input of that site's Linear)
// One launch instead of ln_bwd_dx_kernel + the K-split GEMM (1 008 pairs per WGAN step).
//
// Every block needs whole rows of dx2, i.e. repeats the LayerNorm backward and the dropout replay of its 16 rows
// (~40 VALU instructions per element, all on one CU), and the bytes a block pulls grow with its tile width while
// one CU's load bandwidth does not.  Measured per launch (bs160, d = 512, dropout 0.1), against 5.4 + 4.4 us for the
// two separate kernels plus ~1.3 us of kernel boundary: 16x16 tiles 11.8 us, 16x32 9.3, 16x64 9.8, 16x128 12.5.
// The 16x32 tile is kept for n = 512 (~1 % of the WGAN step); outputs of 1024 columns and more take 16x64 tiles (half
// as many repeats of the row work).  All loads (dy, x, dres, gamma, W) are issued before the first use.
// Waves per block (same tile, K split over the waves): 4 -> 8.9 ms per WGAN step in this kernel, 8 -> 8.3 (two waves
// per SIMD overlap their load -> reduce -> MFMA chains), 16 -> 9.3; 8 waves with 16x64 tiles 8.4.
#include "gemm_common.hpp"
#include <initializer_list>
#include <stdlib.h>

namespace mssr {

typedef unsigned lnb_u32x4 __attribute__((ext_vector_type(4)));
constexpr int LNB_NW = 8;                  // waves per block (two per SIMD: one wave's load / reduce chain hides the other's)
constexpr int LNB_THREADS = 64 * LNB_NW;
// 16-column fragments per block tile: template parameter LNB_FN (2, or 4 for wide outputs; see the measurements above)
constexpr int LNB_G = 2;                   // k-steps per wave (K <= LNB_NW * LNB_G * 32 = 512)

template <typename TO, int LNB_FN>
__global__ __launch_bounds__(LNB_THREADS) void gemm_lnb_kernel(const GemmArgs p) {
    constexpr int LNB_TSN = 16 * LNB_FN;       // tile columns
    __shared__ __attribute__((aligned(16))) float red[LNB_NW][16][LNB_TSN];
    __shared__ float stat[LNB_NW][16][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * LNB_TSN, m0 = (int64_t)blockIdx.y * 16;
    const int64_t steps = p.k / 32;
    const bf16* DY = reinterpret_cast<const bf16*>(p.a);
    const bf16* W = reinterpret_cast<const bf16*>(p.b);
    const bf16* X = reinterpret_cast<const bf16*>(p.lnb_x);
    const bool has_res = p.lnb_dres != nullptr;
    const bf16* DR = has_res ? reinterpret_cast<const bf16*>(p.lnb_dres) : X;          // unconditional loads
    const int koff = 8 * (lane >> 4);
    const int64_t arl = m0 + (lane & 15);
    const bool row_ok = arl < p.m;
    const int64_t ar = row_ok ? arl : p.m - 1;
    const int64_t arow = ar * p.a_rs;
    int64_t wrow[LNB_FN];
#pragma unroll
    for (int j = 0; j < LNB_FN; ++j) {
        const int64_t wn = n0 + j * 16 + (lane & 15);
        wrow[j] = (wn < p.n ? wn : p.n - 1) * p.b_cs;
    }
    // epilogue operands of the elements this thread finishes (e = tid + i*256: row e / LNB_TSN, col e % LNB_TSN)
    constexpr int EPT = (16 * LNB_TSN + LNB_THREADS - 1) / LNB_THREADS;
    float gate[EPT];
    bool eok[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = threadIdx.x + i * LNB_THREADS;
        const int64_t erow = m0 + e / LNB_TSN, ecol = n0 + e % LNB_TSN;
        eok[i] = e < 16 * LNB_TSN && erow < p.m && ecol < p.n;
        gate[i] = 1.f;
        if (eok[i] && p.gate) gate[i] = to_f(reinterpret_cast<const bf16*>(p.gate)[erow * p.ldg + ecol]);
    }
    const float mu = p.lnb_mean[ar], rs = p.lnb_rstd[ar];
    const uint64_t lnb_seed = eff_seed(p, p.lnb_seed);
    bf16x8 fb[LNB_G][LNB_FN], fdy[LNB_G], fx[LNB_G], fr[LNB_G];
    f32x4 g0[LNB_G], g1[LNB_G];
#pragma unroll
    for (int g = 0; g < LNB_G; ++g) {
        int64_t s = wave + LNB_NW * g;
        s = s < steps ? s : steps - 1;
        const int64_t k = s * 32 + koff;
        fdy[g] = *reinterpret_cast<const bf16x8*>(DY + arow + k);
        fx[g] = *reinterpret_cast<const bf16x8*>(X + arow + k);
        fr[g] = *reinterpret_cast<const bf16x8*>(DR + arow + k);
        g0[g] = *reinterpret_cast<const f32x4*>(p.lnb_gamma + k);
        g1[g] = *reinterpret_cast<const f32x4*>(p.lnb_gamma + k + 4);
#pragma unroll
        for (int j = 0; j < LNB_FN; ++j) fb[g][j] = *reinterpret_cast<const bf16x8*>(W + wrow[j] + k);
    }
    // the dropout decisions depend on indices and the seed only: taken here, in the shadow of the loads above
    // (the whole sweep's backward: -0.3 ms)
    const uint32_t thr = drop_thr16(p.lnb_p);
    uint32_t kp[LNB_G][4];
#pragma unroll
    for (int g = 0; g < LNB_G; ++g) {
        int64_t s = wave + LNB_NW * g;
        s = s < steps ? s : steps - 1;
        const uint64_t i2 = (uint64_t)((ar + p.lnb_row_off) * p.k + s * 32 + koff) >> 1;   // a multiple of 4 pairs
#pragma unroll
        for (int q = 0; q < 4; ++q) kp[g][q] = p.lnb_p > 0.f ? keep16_pair(lnb_seed, i2 + q, thr) : 3u;
    }
    __builtin_amdgcn_sched_barrier(0);
    // row statistics of g = gamma * dy over the whole row: lane partials -> the 4 lanes of a row -> the 4 waves
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int g = 0; g < LNB_G; ++g) {
        if (wave + LNB_NW * g < steps) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = (float)fdy[g][j] * (j < 4 ? g0[g][j] : g1[g][j - 4]);
                s1 += a;
                s2 += a * (((float)fx[g][j] - mu) * rs);
            }
        }
    }
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lane < 16) {
        stat[wave][lane][0] = s1;
        stat[wave][lane][1] = s2;
    }
    __syncthreads();
    const int rl = lane & 15;
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int w = 0; w < LNB_NW; ++w) {
        c1 += stat[w][rl][0];
        c2 += stat[w][rl][1];
    }
    c1 /= (float)p.k;
    c2 /= (float)p.k;
    const float sc = p.lnb_extra / (1.0f - p.lnb_p);
    const bool writer = blockIdx.x == 0 && row_ok;
    f32x4 acc[LNB_FN];
#pragma unroll
    for (int j = 0; j < LNB_FN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < LNB_G; ++g) {
        const int64_t sg = wave + LNB_NW * g;
        const bool live = sg < steps;
        const int64_t k = (live ? sg : steps - 1) * 32 + koff;
        bf16x8 dxv, dx2v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = (float)fdy[g][j] * (j < 4 ? g0[g][j] : g1[g][j - 4]);
            const float xh = ((float)fx[g][j] - mu) * rs;
            const float o = (has_res ? (float)fr[g][j] : 0.f) + rs * (a - c1 - xh * c2);
            const bf16 ob = (bf16)o;
            const float ov = (float)ob;               // dx IS stored in bf16: replay from the stored value
            const float o2 = p.lnb_p > 0.f ? (((kp[g][j >> 1] >> (j & 1)) & 1u) ? ov * sc : 0.f) : ov * p.lnb_extra;
            dxv[j] = ob;
            dx2v[j] = (bf16)o2;
        }
        if (writer && live) {
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.lnb_dx) + arow + k) = dxv;
            if (p.lnb_dx2) *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.lnb_dx2) + arow + k) = dx2v;
        }
        const unsigned msk = live ? 0xffffffffu : 0u;
        lnb_u32x4 ab = __builtin_bit_cast(lnb_u32x4, dx2v);
        ab &= msk;                                    // a clamped (dead) k-step contributes nothing
        const bf16x8 am = __builtin_bit_cast(bf16x8, ab);
#pragma unroll
        for (int j = 0; j < LNB_FN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, fb[g][j], acc[j], 0, 0, 0);
    }
    // partial tiles of the waves -> LDS (MFMA C layout: col = lane & 15, row = (lane >> 4)*4 + r)
#pragma unroll
    for (int j = 0; j < LNB_FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(lane >> 4) * 4 + r][j * 16 + (lane & 15)] = acc[j][r];
    __syncthreads();
    TO* C = reinterpret_cast<TO*>(p.c);
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        if (!eok[i]) continue;
        const int e = threadIdx.x + i * LNB_THREADS;
        const int rr = e / LNB_TSN, cc = e % LNB_TSN;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < LNB_NW; ++w) v += red[w][rr][cc];
        v *= p.alpha;
        if (p.gate) v = gate[i] > 0.f ? v : 0.f;
        C[(m0 + rr) * p.ldc + n0 + cc] = from_f<TO>(v);
    }
}

bool gemm_lnb_try(const GemmArgs& a, int out_dtype, hipStream_t s) {
    if (!a.lnb_x || !a.lnb_gamma || !a.lnb_mean || !a.lnb_rstd || !a.lnb_dx) return false;
    if (a.bias || a.residual || a.relu || a.drop_p > 0.f || a.beta != 0.f || a.ln_colsum) return false;   // plain epilogue (+ gate)
    if (a.m < 1 || a.n < 1 || a.k < 32 || a.k % 32 != 0 || a.k > LNB_NW * LNB_G * 32) return false;
    if (a.a_cs != 1 || a.b_rs != 1 || a.a_rs % 8 != 0 || a.b_cs % 8 != 0) return false;
    if ((a.m + 15) / 16 > 65535) return false;
    if (a.lnb_dres == a.lnb_dx) return false;
    for (const void* q : {a.a, a.b, a.lnb_x, a.lnb_dres, (const void*)a.lnb_dx, (const void*)a.lnb_dx2,
                          (const void*)a.lnb_gamma})
        if (q && !aligned(q, 16)) return false;
    if (a.lnb_p < 0.f || a.lnb_p >= 1.f) return false;
    static const int64_t wide_n = [] {
        const char* e = getenv("MSSR_LNB_WIDE_N");
        return e ? (int64_t)atoll(e) : (int64_t)1024;      // WGAN step 54.56 ms with 16x32 everywhere, 54.07 with 16x64
                                                            // from n = 1024 (the FFN inner gradient), 54.69 from n = 512
    }();
    if (a.n >= wide_n) {
        const dim3 grid((unsigned)((a.n + 63) / 64), (unsigned)((a.m + 15) / 16), 1);
        if (out_dtype == MSSR_BF16) klaunch(gemm_lnb_kernel<bf16, 4>, grid, dim3(LNB_THREADS), 0, s, a);
        else klaunch(gemm_lnb_kernel<float, 4>, grid, dim3(LNB_THREADS), 0, s, a);
        return true;
    }
    const dim3 grid((unsigned)((a.n + 31) / 32), (unsigned)((a.m + 15) / 16), 1);
    if (out_dtype == MSSR_BF16) klaunch(gemm_lnb_kernel<bf16, 2>, grid, dim3(LNB_THREADS), 0, s, a);
    else klaunch(gemm_lnb_kernel<float, 2>, grid, dim3(LNB_THREADS), 0, s, a);
    return true;
}

}  // namespace mssr
