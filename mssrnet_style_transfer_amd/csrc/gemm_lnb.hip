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
// Wave w owns the 64-wide k-block w (k-steps 2w, 2w + 1): K <= LNB_NW * 64 = 512.

__device__ __forceinline__ int lnb_swz(int r) { return (r >> 1) & 7; }

// Round 3: the operands come in by LDS-DMA, as in gemm_skinny_dma_kernel (gemm_skinny.hip): the fragment-shaped loads
// of the first version (16 rows x 64 bytes per instruction, 14 per wave) spent more time being ISSUED than in flight.
// Every wave copies its 64-wide k-block of dy, x, dres (16 rows x 128 bytes each) and of its W columns into its own
// LDS region with whole 128-byte lines (10 / 14 DMA instructions), reads the fragments back by ds_read_b128 from a
// 16-byte-chunk XOR-swizzled image, and later reuses the region for its partial tile.  One LDS object only.
template <typename TO, int LNB_FN, bool DROP>
__global__ __launch_bounds__(LNB_THREADS) void gemm_lnb_kernel(const GemmArgs p, const int region_bytes) {
    constexpr int LNB_TSN = 16 * LNB_FN;       // tile columns
    extern __shared__ __attribute__((aligned(1024))) unsigned char lnb_smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: region / k-range arithmetic on the SALU
    // 32-bit index arithmetic throughout (the host checked that every operand spans < 4 GB and m, n, k < 2^31): lanes
    // carry byte offsets from the scalar operand bases
    const int n0 = (int)blockIdx.x * LNB_TSN, m0 = (int)blockIdx.y * 16;
    const int mi = (int)p.m, ni = (int)p.n, ki = (int)p.k;
    const int steps = ki >> 5;
    const bf16* DY = reinterpret_cast<const bf16*>(p.a);
    const bf16* W = reinterpret_cast<const bf16*>(p.b);
    const bf16* X = reinterpret_cast<const bf16*>(p.lnb_x);
    const bool has_res = p.lnb_dres != nullptr;
    const bf16* DR = has_res ? reinterpret_cast<const bf16*>(p.lnb_dres) : X;          // unconditional loads
    unsigned char* region = lnb_smem + (size_t)wave * region_bytes;
    unsigned char* rDY = region;                 // [16][128 B]
    unsigned char* rX = region + 2048;
    unsigned char* rDR = region + 4096;
    unsigned char* rW = region + 6144;           // [LNB_TSN][128 B]
    float* rstat = reinterpret_cast<float*>(region + 6144 + LNB_TSN * 128);     // [16][2]
    const bool wave_live = 2 * wave < steps;     // K < 512: the upper waves own nothing
    const int k0 = wave_live ? wave * 64 : 0;
    const int k_last = ki - 8;
    const uint32_t a_rsb = (uint32_t)p.a_rs * 2u, b_csb = (uint32_t)p.b_cs * 2u;
    {
        const int drow = lane >> 3, dpc = lane & 7;
        const char* DYb = reinterpret_cast<const char*>(DY);
        const char* Xb = reinterpret_cast<const char*>(X);
        const char* DRb = reinterpret_cast<const char*>(DR);
        const char* Wb = reinterpret_cast<const char*>(W);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int r = g * 8 + drow;
            int k = k0 + (dpc ^ lnb_swz(r)) * 8;
            k = k <= k_last ? k : k_last;            // second half past the end (odd count of k-steps): never used
            const uint32_t off = (uint32_t)min(m0 + r, mi - 1) * a_rsb + (uint32_t)k * 2u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(DYb + off),
                                             (__attribute__((address_space(3))) void*)(rDY + g * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xb + off),
                                             (__attribute__((address_space(3))) void*)(rX + g * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(DRb + off),
                                             (__attribute__((address_space(3))) void*)(rDR + g * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < LNB_TSN / 8; ++g) {
            const int r = g * 8 + drow;
            int k = k0 + (dpc ^ lnb_swz(r)) * 8;
            k = k <= k_last ? k : k_last;
            const uint32_t off = (uint32_t)min(n0 + r, ni - 1) * b_csb + (uint32_t)k * 2u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wb + off),
                                             (__attribute__((address_space(3))) void*)(rW + g * 1024), 16, 0, 0);
        }
    }
    // The small operands right BEHIND the DMA (ahead of it their ~7 scattered loads delay its issue), all of them
    // unconditional and kept raw until after the wait: vector-memory loads return in order, and the first version
    // converted the gate values here (an `s_waitcnt vmcnt(0)` of the compiler's), so the statistics / gamma loads were
    // not even issued, and no dropout decision hashed, before the operand tiles had landed.
    const int fr = lane & 15, fc = lane >> 4;
    const int arl = m0 + fr;
    const bool row_ok = arl < mi;
    const int ar = row_ok ? arl : mi - 1;
    const int64_t arow = (int64_t)ar * p.a_rs;
    constexpr int EPT = (16 * LNB_TSN + LNB_THREADS - 1) / LNB_THREADS;
    bf16 graw[EPT];
    bool eok[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = threadIdx.x + i * LNB_THREADS;
        const int erow = m0 + e / LNB_TSN, ecol = n0 + e % LNB_TSN;
        eok[i] = e < 16 * LNB_TSN && erow < mi && ecol < ni;
        // an UNCONDITIONAL load from a clamped address (no gate: an element of dy, never used): a select between a
        // constant and the loaded value would make the compiler wait for the load right here
        const int gr = erow < mi ? erow : mi - 1, gc = ecol < ni ? ecol : ni - 1;
        const bf16* gp = p.gate ? reinterpret_cast<const bf16*>(p.gate) + (int64_t)gr * p.ldg + gc : DY + (int64_t)gr * p.a_rs;
        graw[i] = *gp;
    }
    const float mu = p.lnb_mean[ar], rs = p.lnb_rstd[ar];
    // gamma of this lane's k-chunks (half h: k = k0 + 32 h + 8 fc .. + 8)
    f32x4 g0[2], g1[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int k = k0 + 32 * h + 8 * fc;
        k = k <= k_last ? k : k_last;
        g0[h] = *reinterpret_cast<const f32x4*>(p.lnb_gamma + k);
        g1[h] = *reinterpret_cast<const f32x4*>(p.lnb_gamma + k + 4);
    }
    const uint64_t lnb_seed = eff_seed(p, p.lnb_seed);
    // the dropout decisions depend on indices and the seed only: taken here, in the shadow of the loads
    const uint32_t thr = drop_thr16(p.lnb_p);
    uint32_t kp[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint64_t i2 = (uint64_t)((ar + p.lnb_row_off) * p.k + k0 + 32 * h + 8 * fc) >> 1;   // a multiple of 4 pairs
#pragma unroll
        for (int q = 0; q < 4; ++q) kp[h][q] = DROP ? keep16_pair(lnb_seed, i2 + q, thr) : 3u;
    }
    // pin the decisions ahead of the wait: nothing orders plain arithmetic against the wait, and the scheduler had
    // sunk most of the hashes behind the barrier, onto the critical path between the statistics and the MFMAs -- and
    // with them the two divisions below (a ~12-instruction dependent chain each)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" ::"v"(kp[h][q]));
    float inv_k = 1.0f / (float)ki;
    float sc = p.lnb_extra / (1.0f - p.lnb_p);
    asm volatile("" : "+v"(inv_k), "+v"(sc));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bf16x8 fb[2][LNB_FN], fdy[2], fx[2], fr_[2];
    bool live[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        live[h] = 2 * wave + h < steps;
        const int co = ((h * 4 + fc) ^ lnb_swz(fr)) << 4;
        fdy[h] = *reinterpret_cast<const bf16x8*>(rDY + fr * 128 + co);
        fx[h] = *reinterpret_cast<const bf16x8*>(rX + fr * 128 + co);
        fr_[h] = *reinterpret_cast<const bf16x8*>(rDR + fr * 128 + co);
#pragma unroll
        for (int j = 0; j < LNB_FN; ++j) {
            const int r = j * 16 + fr;
            fb[h][j] = *reinterpret_cast<const bf16x8*>(rW + r * 128 + (((h * 4 + fc) ^ lnb_swz(r)) << 4));
        }
    }
    // row statistics of g = gamma * dy over the whole row: lane partials -> the 4 lanes of a row -> the 8 waves
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (live[h]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = (float)fdy[h][j] * (j < 4 ? g0[h][j] : g1[h][j - 4]);
                s1 += a;
                s2 += a * (((float)fx[h][j] - mu) * rs);
            }
        }
    }
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lane < 16) {
        rstat[lane * 2] = s1;
        rstat[lane * 2 + 1] = s2;
    }
    __syncthreads();
    const int rstride = region_bytes / 4;
    const float* stat0 = reinterpret_cast<const float*>(lnb_smem + 6144 + LNB_TSN * 128);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int w = 0; w < LNB_NW; ++w) {
        c1 += stat0[w * rstride + fr * 2];
        c2 += stat0[w * rstride + fr * 2 + 1];
    }
    c1 *= inv_k;
    c2 *= inv_k;
    const bool writer = blockIdx.x == 0 && row_ok;
    f32x4 acc[LNB_FN];
#pragma unroll
    for (int j = 0; j < LNB_FN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = k0 + 32 * h + 8 * fc;
        bf16x8 dxv, dx2v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = (float)fdy[h][j] * (j < 4 ? g0[h][j] : g1[h][j - 4]);
            const float xh = ((float)fx[h][j] - mu) * rs;
            const float o = (has_res ? (float)fr_[h][j] : 0.f) + rs * (a - c1 - xh * c2);
            const bf16 ob = (bf16)o;
            const float ov = (float)ob;               // dx IS stored in bf16: replay from the stored value
            const float o2 = DROP ? (((kp[h][j >> 1] >> (j & 1)) & 1u) ? ov * sc : 0.f) : ov * p.lnb_extra;
            dxv[j] = ob;
            dx2v[j] = (bf16)o2;
        }
        if (writer && live[h]) {
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.lnb_dx) + arow + k) = dxv;
            if (p.lnb_dx2) *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.lnb_dx2) + arow + k) = dx2v;
        }
        const unsigned msk = live[h] ? 0xffffffffu : 0u;
        lnb_u32x4 ab = __builtin_bit_cast(lnb_u32x4, dx2v);
        ab &= msk;                                    // an absent k-step contributes nothing
        const bf16x8 am = __builtin_bit_cast(bf16x8, ab);
#pragma unroll
        for (int j = 0; j < LNB_FN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, fb[h][j], acc[j], 0, 0, 0);
    }
    // partial tiles of the waves -> their own regions (MFMA C layout: col = lane & 15, row = (lane >> 4)*4 + r); every
    // fragment read of the region was consumed above, and the other waves read only the statistics words behind it
    float* red = reinterpret_cast<float*>(region);
#pragma unroll
    for (int j = 0; j < LNB_FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((lane >> 4) * 4 + r) * LNB_TSN + j * 16 + (lane & 15)] = acc[j][r];
    __syncthreads();
    const float* red0 = reinterpret_cast<const float*>(lnb_smem);
    TO* C = reinterpret_cast<TO*>(p.c);
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        if (!eok[i]) continue;
        const int e = threadIdx.x + i * LNB_THREADS;
        const int rr = e / LNB_TSN, cc = e % LNB_TSN;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < LNB_NW; ++w) v += red0[w * rstride + e];
        v *= p.alpha;
        if (p.gate) v = to_f(graw[i]) > 0.f ? v : 0.f;
        C[(int64_t)(m0 + rr) * p.ldc + n0 + cc] = from_f<TO>(v);
    }
}

template <int LNB_FN>
static int lnb_region_bytes() { return (6144 + 16 * LNB_FN * 128 + 128 + 1023) & ~1023; }

template <typename TO, int LNB_FN>
static bool lnb_launch(const GemmArgs& a, const dim3 grid, hipStream_t s) {
    const int region = lnb_region_bytes<LNB_FN>();
    const int lds = LNB_NW * region;
    static const bool attr_ok =
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_lnb_kernel<TO, LNB_FN, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_lnb_kernel<TO, LNB_FN, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    if (lds > 64 * 1024 && !attr_ok) return false;
    // the dropout replay as a template parameter: no per-element branch on p in the row arithmetic
    if (a.lnb_p > 0.f) klaunch(gemm_lnb_kernel<TO, LNB_FN, true>, grid, dim3(LNB_THREADS), (unsigned)lds, s, a, region);
    else klaunch(gemm_lnb_kernel<TO, LNB_FN, false>, grid, dim3(LNB_THREADS), (unsigned)lds, s, a, region);
    return true;
}

bool gemm_lnb_try(const GemmArgs& a, int out_dtype, hipStream_t s) {
    if (!a.lnb_x || !a.lnb_gamma || !a.lnb_mean || !a.lnb_rstd || !a.lnb_dx) return false;
    if (a.bias || a.residual || a.relu || a.drop_p > 0.f || a.beta != 0.f || a.ln_colsum) return false;   // plain epilogue (+ gate)
    if (a.m < 1 || a.n < 1 || a.k < 32 || a.k % 32 != 0 || a.k > LNB_NW * 64) return false;
    if (a.a_cs != 1 || a.b_rs != 1 || a.a_rs % 8 != 0 || a.b_cs % 8 != 0) return false;
    if ((a.m + 15) / 16 > 65535) return false;
    // the kernel's lanes carry 32-bit byte offsets from the operand bases (dy, x and dres share dy's row stride)
    if (a.m * a.a_rs * 2 >= ((int64_t)1 << 32) || a.n * a.b_cs * 2 >= ((int64_t)1 << 32)) return false;
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
    // ... and from the row count at which the wide tiles still fill the chip (256 blocks: 512 rows at n = 512): half as
    // many repeats of the row work, half the operand traffic (--workload long, 512-768 rows: 518 -> 507 ms per step)
    if (a.n >= wide_n || ((a.m + 15) / 16) * ((a.n + 63) / 64) >= 256) {
        const dim3 grid((unsigned)((a.n + 63) / 64), (unsigned)((a.m + 15) / 16), 1);
        return out_dtype == MSSR_BF16 ? lnb_launch<bf16, 4>(a, grid, s) : lnb_launch<float, 4>(a, grid, s);
    }
    const dim3 grid((unsigned)((a.n + 31) / 32), (unsigned)((a.m + 15) / 16), 1);
    return out_dtype == MSSR_BF16 ? lnb_launch<bf16, 2>(a, grid, s) : lnb_launch<float, 2>(a, grid, s);
}

}  // namespace mssr
