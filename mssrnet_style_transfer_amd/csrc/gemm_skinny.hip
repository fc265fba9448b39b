// Skinny-M GEMM for the step-wise decode: C[M<=128, N] = A[M,K] . W[N,K]^T with both operands
// k-contiguous (bf16).  Latency-bound shape: M = 80 rows, N,K = 512..9632, 137+ of them chained per
// decode step, so what counts is the length of ONE launch's dependency chain, not bandwidth.
//
// Mapping (measured on MI355X, scratch microbenchmark: 3.9 us per chained 80x512x512 launch against
// 7.4 us for a 16-column x all-rows block and 2.5 us for an empty kernel): one block per 16x16 output
// tile, the block's 4 waves split K, every wave issues ALL its 16-B fragment loads (2 per k-step)
// before the first MFMA, i.e. the kernel exposes a single memory round trip; a block moves 32 KB
// through its CU's L1 instead of 96 KB, W is re-read by the M/16 row tiles out of L2 / Infinity Cache.
// The four partial tiles are summed through LDS and the epilogue runs one element per thread with
// its bias / residual / gate / C operands fetched at kernel start, under the main loads.
//
// LN mode (LayerNorm folded into the product: onmt/decoders/transformer.py:150-157 `LN -> Linear`):
// A holds the raw rows x, W holds gamma-scaled weights, and
//     LN(x) W^T + b = rstd (x W'^T - mean * colsum(W')) + (beta W^T + b)
// with the row statistics taken from the fragments the block has loaded anyway.
//
// K > 2048 is split across blockIdx.z into fp32 slabs; a second tiny kernel reduces them and applies
// the epilogue (deterministic, no atomics).
#include "gemm_common.hpp"
#include <stdlib.h>

namespace mssr {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int SK_THREADS = 256;
// k-steps (of 32) a wave keeps in flight: template parameter G = 4, 8 or 12, the smallest that covers the wave's
// share of K in ONE batch of loads (K = 512 -> 4, 1024 -> 8, 1536 -> 12; one memory round trip instead of 2-3)
constexpr int SK_MAX_STEPS = 64;     // k-steps per block without split-K (K <= 2048)
constexpr int SK_MAX_M = 8192;       // above this the general LDS-tiled kernel (operand reuse) takes over

// FM x FN fragments per tile: 1 x 1 -> 16x16 tile (few rows), 2 x 2 -> 32x32 tile (larger M), 1 x 2 -> 16 rows x 32
// columns (long-K decode products: a wave's A fragment feeds two column fragments, 25 % fewer bytes through the CU's
// load path than two 16x16 tiles, and half as many blocks: one round on the chip)
template <typename TO, bool LN, int FM, int FN, int SK_GROUP>
__global__ __launch_bounds__(SK_THREADS) void gemm_skinny_kernel(const GemmArgs p) {
    constexpr int TSM = 16 * FM, TSN = 16 * FN;    // tile rows, columns
    constexpr int TE = TSM * TSN;                  // tile elements
    constexpr int EPT = TE / SK_THREADS;           // epilogue elements per thread (1, 2 or 4)
    __shared__ __attribute__((aligned(16))) float red[4 * TE];
    __shared__ float stat[4][TSM][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    dbg_stamp(p, 0);
    const int64_t n0 = (int64_t)blockIdx.x * TSN, m0 = (int64_t)blockIdx.y * TSM;
    const int split = blockIdx.z;
    // 32-bit arithmetic, and no division at all without K slabs: the 64-bit `/ p.ksplit` was a ~130-instruction software
    // division on the scalar unit at the head of every launch (stamps: 0.5 us of index arithmetic before the first load)
    const int steps_total = (int)(p.k >> 5);
    const int per = p.ksplit == 1 ? steps_total : (int)((unsigned)(steps_total + p.ksplit - 1) / (unsigned)p.ksplit);
    const int s_begin = split * per;
    const int s_end = (s_begin + per < steps_total) ? s_begin + per : steps_total;

    const bf16* A = reinterpret_cast<const bf16*>(p.a);
    const bf16* W = reinterpret_cast<const bf16*>(p.b);
    const int koff = 8 * (lane >> 4);
    int64_t arow[FM], wrow[FN];
#pragma unroll
    for (int f = 0; f < FM; ++f) {
        const int64_t ar = m0 + f * 16 + (lane & 15);
        arow[f] = (ar < p.m ? ar : p.m - 1) * p.a_rs;
    }
#pragma unroll
    for (int f = 0; f < FN; ++f) {
        const int64_t wn = n0 + f * 16 + (lane & 15);
        wrow[f] = (wn < p.n ? wn : p.n - 1) * p.b_cs;            // B viewed [k,n]: n stride = b_cs
    }

    // the elements this thread finishes (tile-local index e = threadIdx.x + i*256: row e / TSN, col e % TSN),
    // and their epilogue operands, issued before the main loads
    float bias[EPT], res[EPT], gate[EPT], cold[EPT], lns[EPT];
    bool eok[EPT], keepm[EPT];      // keepm: the element's dropout decision (index + seed only), taken under the loads
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = threadIdx.x + i * SK_THREADS;
        const int64_t erow = m0 + e / TSN, ecol = n0 + e % TSN;
        eok[i] = erow < p.m && ecol < p.n;
        bias[i] = res[i] = cold[i] = lns[i] = 0.f;
        gate[i] = 1.f;
        keepm[i] = true;
        if (eok[i] && p.ksplit == 1) {
            if (p.drop_p > 0.f && erow >= p.drop_row0)
                keepm[i] = drop_keep(eff_seed(p, p.drop_seed), (uint64_t)(erow * p.n + ecol), p.drop_p);
            if (p.bias) bias[i] = p.bias[ecol];
            if (p.residual) res[i] = to_f(reinterpret_cast<const bf16*>(p.residual)[erow * p.ldr + ecol]);
            if (p.gate) gate[i] = to_f(reinterpret_cast<const bf16*>(p.gate)[erow * p.ldg + ecol]);
            if (p.beta != 0.f) cold[i] = to_f(reinterpret_cast<const TO*>(p.c)[erow * p.ldc + ecol]);
            if (LN) lns[i] = p.ln_colsum[ecol];
        }
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float sx[FM], sxx[FM];
#pragma unroll
    for (int f = 0; f < FM; ++f) sx[f] = sxx[f] = 0.f;
    // wave w owns k-steps s_begin + w, + 4, + 8 ... ; SK_GROUP of them in flight at a time  (contiguous quarters per
    // wave, so that one wave requests both halves of a 128-byte line back to back: WGAN step 51.0-51.6 ms against 50.8)
    for (int s0 = s_begin + wave; s0 < s_end; s0 += 4 * SK_GROUP) {
        bf16x8 fa[SK_GROUP][FM], fb[SK_GROUP][FN];
#pragma unroll
        for (int g = 0; g < SK_GROUP; ++g) {
            // unconditional loads (clamped to the last valid k-step): a branch around a load makes
            // hipcc wait vmcnt(0) per load and serialises the whole group
            int64_t s = s0 + 4 * g;
            s = s < s_end ? s : s_end - 1;
            const int64_t k = s * 32 + koff;
#pragma unroll
            for (int f = 0; f < FN; ++f) fb[g][f] = *reinterpret_cast<const bf16x8*>(W + wrow[f] + k);
#pragma unroll
            for (int f = 0; f < FM; ++f) fa[g][f] = *reinterpret_cast<const bf16x8*>(A + arow[f] + k);
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the whole batch of loads ahead of the first MFMA
        dbg_stamp(p, 1);
        if (MSSR_GEMM_STAMPS && p.dbg_stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_stamp(p, 2); }
#pragma unroll
        for (int g = 0; g < SK_GROUP; ++g) {
            // k-steps past the end were loaded from a clamped address: zero the weight fragments with a
            // bit mask (no branch, so the loads above stay one batch in flight)
            const bool live = s0 + 4 * g < s_end;
            const unsigned msk = live ? 0xffffffffu : 0u;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                u32x4 wb = __builtin_bit_cast(u32x4, fb[g][j]);
                wb &= msk;
                const bf16x8 fbm = __builtin_bit_cast(bf16x8, wb);
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[g][i], fbm, acc[i][j], 0, 0, 0);
            }
            if (LN && live) {
#pragma unroll
                for (int f = 0; f < FM; ++f)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float x = (float)fa[g][f][j];
                        sx[f] += x;
                        sxx[f] += x * x;
                    }
            }
        }
    }
    // partial tiles -> LDS as [wave][row*TSN + col]  (MFMA C layout: col = lane & 15, row = (lane >> 4)*4 + r)
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[wave * TE + (i * 16 + (lane >> 4) * 4 + r) * TSN + j * 16 + (lane & 15)] = acc[i][j][r];
    if (LN) {       // row sums: the 4 lanes (lane & 15, lane >> 4 = 0..3) of a wave hold one row's k-chunks
#pragma unroll
        for (int f = 0; f < FM; ++f) {
            sx[f] += __shfl_xor(sx[f], 16, 64);
            sx[f] += __shfl_xor(sx[f], 32, 64);
            sxx[f] += __shfl_xor(sxx[f], 16, 64);
            sxx[f] += __shfl_xor(sxx[f], 32, 64);
            if (lane < 16) {
                stat[wave][f * 16 + lane][0] = sx[f];
                stat[wave][f * 16 + lane][1] = sxx[f];
            }
        }
    }
    dbg_stamp(p, 3);
    __syncthreads();
    dbg_stamp(p, 4);
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        if (!eok[i]) continue;
        const int e = threadIdx.x + i * SK_THREADS;
        const int rl = e / TSN;
        const int64_t erow = m0 + rl, ecol = n0 + e % TSN;
        float v = (red[e] + red[TE + e]) + (red[2 * TE + e] + red[3 * TE + e]);
        if (p.ksplit > 1) {
            p.partial[((int64_t)split * p.m + erow) * p.n + ecol] = v;
            continue;
        }
        if (LN) {
            const float s1 = (stat[0][rl][0] + stat[1][rl][0]) + (stat[2][rl][0] + stat[3][rl][0]);
            const float s2 = (stat[0][rl][1] + stat[1][rl][1]) + (stat[2][rl][1] + stat[3][rl][1]);
            const float mean = s1 / (float)p.k;
            const float var = fmaxf(s2 / (float)p.k - mean * mean, 0.f);
            const float rstd = rsqrtf(var + p.ln_eps);
            v = rstd * (v - mean * lns[i]);
            if (blockIdx.x == 0 && (e % TSN) == 0) {
                if (p.ln_mean) p.ln_mean[erow] = mean;
                if (p.ln_rstd) p.ln_rstd[erow] = rstd;
            }
        }
        v = p.alpha * v + bias[i];
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.drop_p > 0.f && erow >= p.drop_row0) v = keepm[i] ? v * (1.0f / (1.0f - p.drop_p)) : 0.f;
        v += res[i];
        if (p.gate) v = gate[i] > 0.f ? v : 0.f;
        if (p.beta != 0.f) v += p.beta * cold[i];
        reinterpret_cast<TO*>(p.c)[erow * p.ldc + ecol] = from_f<TO>(v);
    }
    dbg_stamp(p, 5);
}

// ---- the same product with the operands staged through LDS by DMA (global_load_lds_dwordx4)
// In-kernel stamps (tools/stamp_bench.py) showed where the register form above spends a launch: not in waiting for
// memory but in ISSUING its loads.  An MFMA-fragment-shaped load (16 rows x 64 bytes per wave instruction) touches 16
// cache lines and uses half of each; the CU's address path takes ~245 cycles per such instruction with 8 waves on the
// CU, so the 8 / 24 / 36 loads a wave needs for K = 512 / 1024 / 1536 took 0.9 / 2.3 / 4.2 us of a 4.8 / 6.4 / 7.9 us
// launch before the first byte was waited for.  Here every wave copies ITS quarter of K of both operand tiles into its
// own LDS region with whole 128-byte lines per row (8 rows x 128 bytes per DMA instruction, ~60 cycles each, no
// registers), waits once, and reads the fragments back with ds_read_b128 (16-byte chunks XOR-swizzled by the row, as in
// gemm_pipe.hip: conflict-free for the MFMA lane mapping).  Regions are private to a wave: no block barrier before the
// MFMAs; afterwards the wave's partial tile overwrites its own region and the four tiles are summed as before.
// Wave w owns the 64-wide k-blocks [w * kbw, (w + 1) * kbw) of the block's k-range (contiguous quarters).
__device__ __forceinline__ int sk_swz(int r) { return (r >> 1) & 7; }
typedef float sk_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sk_bf16x2 __attribute__((ext_vector_type(2)));

// KBW = 64-wide k-blocks per wave (compile time: the fragment reads of all blocks are scheduled ahead of the MFMAs that
// use them); VEC: the EPT consecutive output columns a thread finishes are read / written as one vector (alignment and
// n % EPT == 0 checked on the host), else element by element.
template <typename TO, bool LN, int FM, int FN, int KBW, bool VEC>
__global__ __launch_bounds__(SK_THREADS) void gemm_skinny_dma_kernel(const GemmArgs p, const int region_bytes) {
    constexpr int TSM = 16 * FM, TSN = 16 * FN;
    constexpr int TE = TSM * TSN;
    constexpr int EPT = TE / SK_THREADS;
    extern __shared__ __attribute__((aligned(1024))) unsigned char sk_smem[];     // the kernel's ONLY LDS object
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: region / k-range arithmetic on the SALU
    dbg_stamp(p, 0);
    const int64_t n0 = (int64_t)blockIdx.x * TSN, m0 = (int64_t)blockIdx.y * TSM;
    const int split = blockIdx.z;
    // 32-bit arithmetic, and no division at all without K slabs: the 64-bit `/ p.ksplit` was a ~130-instruction software
    // division on the scalar unit at the head of every launch (stamps: 0.5 us of index arithmetic before the first load)
    const int steps_total = (int)(p.k >> 5);
    const int per = p.ksplit == 1 ? steps_total : (int)((unsigned)(steps_total + p.ksplit - 1) / (unsigned)p.ksplit);
    const int s_begin = split * per;
    const int s_end = (s_begin + per < steps_total) ? s_begin + per : steps_total;
    const int nb = (int)((s_end - s_begin + 1) / 2);            // 64-wide k-blocks of this block's range
    const int b_lo = wave * KBW;

    const bf16* A = reinterpret_cast<const bf16*>(p.a);
    const bf16* W = reinterpret_cast<const bf16*>(p.b);
    unsigned char* region = sk_smem + (size_t)wave * region_bytes;
    unsigned char* regA = region;
    unsigned char* regW = region + KBW * TSM * 128;
    // DMA: lane L of an instruction fills slot L of an 8-row group: row L >> 3, physical 16-byte chunk L & 7, which holds
    // the logical chunk (L & 7) ^ swz(row)
    const int drow = lane >> 3, dpc = lane & 7;
    const int k_last = (int)p.k - 8;
    {
        // 32-bit byte offsets from the (scalar) operand bases: the host checked that both operands span < 4 GB
        uint32_t aoff[TSM / 8], woff[TSN / 8];
        const char* Ab = reinterpret_cast<const char*>(A);
        const char* Wb = reinterpret_cast<const char*>(W);
        const int mi = (int)p.m, ni = (int)p.n, m0i = (int)m0, n0i = (int)n0;
        const uint32_t a_rsb = (uint32_t)p.a_rs * 2u, b_csb = (uint32_t)p.b_cs * 2u;
#pragma unroll
        for (int g = 0; g < TSM / 8; ++g) {
            const int r = g * 8 + drow;
            const int gr = min(m0i + r, mi - 1);
            aoff[g] = (uint32_t)gr * a_rsb + (uint32_t)((dpc ^ sk_swz(r)) * 16);
        }
#pragma unroll
        for (int g = 0; g < TSN / 8; ++g) {
            const int r = g * 8 + drow;
            const int gn = min(n0i + r, ni - 1);
            woff[g] = (uint32_t)gn * b_csb + (uint32_t)((dpc ^ sk_swz(r)) * 16);
        }
        // a chunk past the end of K (second half of an odd count of k-steps) is never multiplied: clamp the lanes whose
        // 16 bytes would start beyond k - 8 (only the last block of the range can hold such chunks)
        const uint32_t kb_last = (uint32_t)k_last * 2u;
#pragma unroll
        for (int bl = 0; bl < KBW; ++bl) {
            if (b_lo + bl >= nb) break;                          // wave-uniform: the last wave's range may be short
            const uint32_t kb0 = (uint32_t)((s_begin + 2 * (b_lo + bl)) * 32) * 2u;      // scalar
            const bool tail = kb0 + 112u > kb_last;              // scalar: some chunk of this block may pass k - 8
#pragma unroll
            for (int g = 0; g < TSN / 8; ++g) {
                uint32_t o = woff[g] + kb0;
                if (tail) {
                    const uint32_t rowb = woff[g] - (uint32_t)((dpc ^ sk_swz(g * 8 + drow)) * 16);
                    const uint32_t kb = kb0 + (uint32_t)((dpc ^ sk_swz(g * 8 + drow)) * 16);
                    o = rowb + (kb <= kb_last ? kb : kb_last);
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wb + o),
                                                 (__attribute__((address_space(3))) void*)(regW + (bl * TSN + g * 8) * 128), 16, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < TSM / 8; ++g) {
                uint32_t o = aoff[g] + kb0;
                if (tail) {
                    const uint32_t rowb = aoff[g] - (uint32_t)((dpc ^ sk_swz(g * 8 + drow)) * 16);
                    const uint32_t kb = kb0 + (uint32_t)((dpc ^ sk_swz(g * 8 + drow)) * 16);
                    o = rowb + (kb <= kb_last ? kb : kb_last);
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Ab + o),
                                                 (__attribute__((address_space(3))) void*)(regA + (bl * TSM + g * 8) * 128), 16, 0, 0);
            }
        }
    }
    dbg_stamp(p, 1);
    // epilogue operands of the EPT consecutive elements this thread finishes, requested behind the DMA
    const int e0 = threadIdx.x * EPT;
    const int rl = e0 / TSN;
    const int64_t erow = m0 + rl, ecol0 = n0 + e0 % TSN;
    float bias[EPT], res[EPT], gate[EPT], cold[EPT], lns[EPT];
    bool eok[EPT], keepm[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        eok[i] = erow < p.m && ecol0 + i < p.n;
        bias[i] = res[i] = cold[i] = lns[i] = 0.f;
        gate[i] = 1.f;
        keepm[i] = true;
    }
    if (eok[0] && p.ksplit == 1) {
        if (p.drop_p > 0.f && erow >= p.drop_row0) {
#pragma unroll
            for (int i = 0; i < EPT; ++i)
                if (eok[i]) keepm[i] = drop_keep(eff_seed(p, p.drop_seed), (uint64_t)(erow * p.n + ecol0 + i), p.drop_p);
        }
        if (VEC && EPT == 4) {
            if (p.bias) { const f32x4 t = *reinterpret_cast<const f32x4*>(p.bias + ecol0); bias[0] = t[0]; bias[1] = t[1]; bias[2] = t[2]; bias[3] = t[3]; }
            if (LN) { const f32x4 t = *reinterpret_cast<const f32x4*>(p.ln_colsum + ecol0); lns[0] = t[0]; lns[1] = t[1]; lns[2] = t[2]; lns[3] = t[3]; }
            if (p.residual) {
                const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p.residual) + erow * p.ldr + ecol0);
#pragma unroll
                for (int i = 0; i < 4; ++i) res[i] = (float)t[i];
            }
            if (p.gate) {
                const bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p.gate) + erow * p.ldg + ecol0);
#pragma unroll
                for (int i = 0; i < 4; ++i) gate[i] = (float)t[i];
            }
        } else if (VEC && EPT == 2) {
            if (p.bias) { const sk_f32x2 t = *reinterpret_cast<const sk_f32x2*>(p.bias + ecol0); bias[0] = t[0]; bias[1 % EPT] = t[1]; }
            if (LN) { const sk_f32x2 t = *reinterpret_cast<const sk_f32x2*>(p.ln_colsum + ecol0); lns[0] = t[0]; lns[1 % EPT] = t[1]; }
            if (p.residual) {
                const sk_bf16x2 t = *reinterpret_cast<const sk_bf16x2*>(reinterpret_cast<const bf16*>(p.residual) + erow * p.ldr + ecol0);
                res[0] = (float)t[0]; res[1 % EPT] = (float)t[1];
            }
            if (p.gate) {
                const sk_bf16x2 t = *reinterpret_cast<const sk_bf16x2*>(reinterpret_cast<const bf16*>(p.gate) + erow * p.ldg + ecol0);
                gate[0] = (float)t[0]; gate[1 % EPT] = (float)t[1];
            }
        } else {
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                if (!eok[i]) continue;
                if (p.bias) bias[i] = p.bias[ecol0 + i];
                if (LN) lns[i] = p.ln_colsum[ecol0 + i];
                if (p.residual) res[i] = to_f(reinterpret_cast<const bf16*>(p.residual)[erow * p.ldr + ecol0 + i]);
                if (p.gate) gate[i] = to_f(reinterpret_cast<const bf16*>(p.gate)[erow * p.ldg + ecol0 + i]);
            }
        }
        if (p.beta != 0.f) {
#pragma unroll
            for (int i = 0; i < EPT; ++i)
                if (eok[i]) cold[i] = to_f(reinterpret_cast<const TO*>(p.c)[erow * p.ldc + ecol0 + i]);
        }
    }
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float sx[FM], sxx[FM];
#pragma unroll
    for (int f = 0; f < FM; ++f) sx[f] = sxx[f] = 0.f;
    // uniform reciprocals of the epilogue, taken here (pinned: nothing else orders arithmetic against the wait) instead of as
    // three ~12-instruction division chains behind the last barrier
    float inv_k = 1.0f / (float)p.k, keep_scale = 1.0f / (1.0f - p.drop_p);
    asm volatile("" : "+v"(inv_k), "+v"(keep_scale));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's region is complete (and only it reads it)
    dbg_stamp(p, 2);
    const int fr = lane & 15, fc = lane >> 4;
#pragma unroll
    for (int bl = 0; bl < KBW; ++bl) {
        if (b_lo + bl >= nb) break;                              // wave-uniform
        const unsigned char* tA = regA + bl * TSM * 128;
        const unsigned char* tW = regW + bl * TSN * 128;
        bf16x8 fa[2][FM], fb[2][FN];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int r = i * 16 + fr;
                fa[h][i] = *reinterpret_cast<const bf16x8*>(tA + r * 128 + (((h * 4 + fc) ^ sk_swz(r)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int r = j * 16 + fr;
                fb[h][j] = *reinterpret_cast<const bf16x8*>(tW + r * 128 + (((h * 4 + fc) ^ sk_swz(r)) << 4));
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // odd count of k-steps: the last block's second half is absent -- its weight fragments are zeroed (uniform)
            const bool live = s_begin + 2 * (int64_t)(b_lo + bl) + h < s_end;
            const unsigned msk = live ? 0xffffffffu : 0u;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                u32x4 wb = __builtin_bit_cast(u32x4, fb[h][j]);
                wb &= msk;
                const bf16x8 fbm = __builtin_bit_cast(bf16x8, wb);
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[h][i], fbm, acc[i][j], 0, 0, 0);
            }
            if (LN && live) {
                // row sums of x and x^2 on the packed-bf16 dot instruction (exact products, fp32 accumulation): 8
                // instructions per fragment instead of 8 conversions + 8 adds + 8 fmas
                const sk_bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
                for (int f = 0; f < FM; ++f) {
#define SK_LN_PAIR(J)                                                                                   \
                    {                                                                                   \
                        const sk_bf16x2 xp = __builtin_shufflevector(fa[h][f], fa[h][f], J, J + 1);     \
                        sx[f] = __builtin_amdgcn_fdot2_f32_bf16(xp, ones, sx[f], false);                \
                        sxx[f] = __builtin_amdgcn_fdot2_f32_bf16(xp, xp, sxx[f], false);                \
                    }
                    SK_LN_PAIR(0)
                    SK_LN_PAIR(2)
                    SK_LN_PAIR(4)
                    SK_LN_PAIR(6)
#undef SK_LN_PAIR
                }
            }
        }
    }
    // partial tile -> the wave's own region (its fragment reads are complete: the MFMAs consumed them), then the row
    // statistics behind it
    float* red = reinterpret_cast<float*>(region);
    float* stat = red + TE;                                      // [TSM][2]
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(i * 16 + (lane >> 4) * 4 + r) * TSN + j * 16 + (lane & 15)] = acc[i][j][r];
    if (LN) {
#pragma unroll
        for (int f = 0; f < FM; ++f) {
            sx[f] += __shfl_xor(sx[f], 16, 64);
            sx[f] += __shfl_xor(sx[f], 32, 64);
            sxx[f] += __shfl_xor(sxx[f], 16, 64);
            sxx[f] += __shfl_xor(sxx[f], 32, 64);
            if (lane < 16) {
                stat[(f * 16 + lane) * 2] = sx[f];
                stat[(f * 16 + lane) * 2 + 1] = sxx[f];
            }
        }
    }
    dbg_stamp(p, 3);
    __syncthreads();
    dbg_stamp(p, 4);
    const int rstride = region_bytes / 4;                        // floats between the waves' regions
    const float* red0 = reinterpret_cast<const float*>(sk_smem);
    if (!eok[0]) return;
    float v[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) v[i] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (EPT == 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(red0 + w * rstride + e0);
            v[0] += t[0]; v[1 % EPT] += t[1]; v[2 % EPT] += t[2]; v[3 % EPT] += t[3];
        } else if (EPT == 2) {
            const sk_f32x2 t = *reinterpret_cast<const sk_f32x2*>(red0 + w * rstride + e0);
            v[0] += t[0]; v[1 % EPT] += t[1];
        } else {
#pragma unroll
            for (int i = 0; i < EPT; ++i) v[i] += red0[w * rstride + e0 + i];
        }
    }
    if (p.ksplit > 1) {
#pragma unroll
        for (int i = 0; i < EPT; ++i)
            if (eok[i]) p.partial[((int64_t)split * p.m + erow) * p.n + ecol0 + i] = v[i];
        return;
    }
    float mean = 0.f, rstd = 1.f;
    if (LN) {
        const float* st = red0 + TE + rl * 2;
        const float s1 = (st[0] + st[rstride]) + (st[2 * rstride] + st[3 * rstride]);
        const float s2 = (st[1] + st[rstride + 1]) + (st[2 * rstride + 1] + st[3 * rstride + 1]);
        mean = s1 * inv_k;
        const float var = fmaxf(s2 * inv_k - mean * mean, 0.f);
        rstd = rsqrtf(var + p.ln_eps);
        if (blockIdx.x == 0 && (e0 % TSN) == 0) {
            if (p.ln_mean) p.ln_mean[erow] = mean;
            if (p.ln_rstd) p.ln_rstd[erow] = rstd;
        }
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        float x = v[i];
        if (LN) x = rstd * (x - mean * lns[i]);
        x = p.alpha * x + bias[i];
        if (p.relu) x = fmaxf(x, 0.f);
        if (p.drop_p > 0.f && erow >= p.drop_row0) x = keepm[i] ? x * keep_scale : 0.f;
        x += res[i];
        if (p.gate) x = gate[i] > 0.f ? x : 0.f;
        if (p.beta != 0.f) x += p.beta * cold[i];
        v[i] = x;
    }
    TO* crow = reinterpret_cast<TO*>(p.c) + erow * p.ldc + ecol0;
    if (VEC && EPT == 4) {
        if constexpr (sizeof(TO) == 2) {
            bf16x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i % EPT];
            *reinterpret_cast<bf16x4*>(crow) = o;
        } else {
            *reinterpret_cast<f32x4*>(crow) = f32x4{v[0], v[1 % EPT], v[2 % EPT], v[3 % EPT]};
        }
    } else if (VEC && EPT == 2) {
        if constexpr (sizeof(TO) == 2) {
            sk_bf16x2 o;
            o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1 % EPT];
            *reinterpret_cast<sk_bf16x2*>(crow) = o;
        } else {
            *reinterpret_cast<sk_f32x2*>(crow) = sk_f32x2{v[0], v[1 % EPT]};
        }
    } else {
#pragma unroll
        for (int i = 0; i < EPT; ++i)
            if (eok[i]) crow[i] = from_f<TO>(v[i]);
    }
    dbg_stamp(p, 5);
}

// LDS bytes of one wave's region: its k-blocks of both operand tiles (and, later, its partial tile + row statistics)
template <int FM, int FN>
static int sk_dma_region(int kbw) {
    const int stage = kbw * (16 * FM + 16 * FN) * 128;
    const int tail = 16 * FM * 16 * FN * 4 + 16 * FM * 2 * 4;
    return ((stage > tail ? stage : tail) + 1023) & ~1023;
}

// true when the DMA form was launched (false: the LDS regions do not fit / an unusual k-block count -- the register
// form takes the problem)
template <typename TO, bool LN, int FM, int FN, int KBW, bool VEC>
static bool skinny_dma_go(const GemmArgs& a, const dim3 grid, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const int region = sk_dma_region<FM, FN>(KBW);
    const int lds = 4 * region;
    if (lds > 160 * 1024) return false;
    static const bool attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_dma_kernel<TO, LN, FM, FN, KBW, VEC>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    if (lds > 64 * 1024 && !attr_ok) return false;
    klaunch_events(gemm_skinny_dma_kernel<TO, LN, FM, FN, KBW, VEC>, grid, dim3(SK_THREADS), (unsigned)lds, s, e0, e1, a, region);
    return true;
}

template <typename TO, bool LN, int FM, int FN>
static bool skinny_dma_launch(const GemmArgs& a, const dim3 grid, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    static const bool enabled = [] { const char* e = getenv("MSSR_SKINNY_DMA"); return !e || atoi(e) != 0; }();
    static const bool splitk_too = [] { const char* e = getenv("MSSR_SKINNY_DMA_SPLITK"); return e && atoi(e) != 0; }();
    if (!enabled || (a.ksplit > 1 && !splitk_too)) return false;      // split-K slabs (K = 9632): 25.8 -> 35.3 us with DMA
    // the kernel's lanes carry 32-bit byte offsets from the operand bases
    if (a.m * a.a_rs * 2 >= ((int64_t)1 << 32) || a.n * a.b_cs * 2 >= ((int64_t)1 << 32)) return false;
    const int64_t steps = (a.k / 32 + a.ksplit - 1) / a.ksplit;
    const int nb = (int)((steps + 1) / 2), kbw = (nb + 3) / 4;
    // vector epilogue (the 4 / 2 consecutive columns of a thread as one access): 32 x 32 and 16 x 32 tiles
    constexpr int EPT = 16 * FM * 16 * FN / SK_THREADS;
    bool vec = EPT >= 2 && a.n % 4 == 0 && a.ldc % 4 == 0 && aligned(a.c, 16) && (!a.bias || aligned(a.bias, 16)) &&
               (!a.ln_colsum || aligned(a.ln_colsum, 16)) && (!a.residual || (aligned(a.residual, 8) && a.ldr % 4 == 0)) &&
               (!a.gate || (aligned(a.gate, 8) && a.ldg % 4 == 0));
#define SK_DMA_CASE(KB)                                                                                     \
    case KB:                                                                                                \
        return vec ? skinny_dma_go<TO, LN, FM, FN, KB, true>(a, grid, s, e0, e1)                            \
                   : skinny_dma_go<TO, LN, FM, FN, KB, false>(a, grid, s, e0, e1)
    switch (kbw) {
        SK_DMA_CASE(1);
        SK_DMA_CASE(2);
        SK_DMA_CASE(3);
        SK_DMA_CASE(4);
        SK_DMA_CASE(6);
        SK_DMA_CASE(8);
        default: return false;
    }
#undef SK_DMA_CASE
}

template <typename TO>
__global__ __launch_bounds__(256) void gemm_skinny_reduce_kernel(const GemmArgs p) {
    const int64_t total = p.m * p.n;
    TO* C = reinterpret_cast<TO*>(p.c);
    const bf16* R = reinterpret_cast<const bf16*>(p.residual);
    const bf16* G = reinterpret_cast<const bf16*>(p.gate);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < p.ksplit; ++s) v += p.partial[(int64_t)s * total + i];
        const int64_t row = (int64_t)((uint32_t)i / (uint32_t)p.n), col = i - row * p.n;      // m * n < 2^31 (m <= 1024 here)
        epilogue_store<bf16, TO>(p, v, row, col, p.bias ? p.bias[col] : 0.f, C, R, G, 0);
    }
}

// ---- fp32 operands (exact f32 MFMA, v_mfma_f32_16x16x4_f32): the style critic's fp32 island inside bf16 runs
// (cfg.style_critic_fp32) and the fp32 parity runs -- rows in the hundreds, K <= 1024.  Same mapping as above: one block
// per 16 x 16 tile, the 4 waves split K in chunks of 16 (wave w owns chunks w, w+4, ...), every 16-byte load of a wave
// issued before its first MFMA.  A lane holds row (lane & 15) and the k-quad 4 (lane >> 4) .. +3 of a chunk; MFMA j of a
// chunk multiplies element j of every quad, so the four quads of a chunk are summed by the instruction's own k index.
// (The general kernel's 32 x 32 LDS tiles with a serial K loop: 13-16 us for 240 x 512 x 512; this one ~5 us.)
constexpr int SKF_MAX_CHUNKS = 16;      // 16-k chunks per wave held in registers (K <= 4 * 16 * 16 = 1024)

template <int NCH>
__global__ __launch_bounds__(SK_THREADS) void gemm_skinny_f32_kernel(const GemmArgs p) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * 16, m0 = (int64_t)blockIdx.y * 16;
    const float* A = reinterpret_cast<const float*>(p.a);
    const float* W = reinterpret_cast<const float*>(p.b);
    const int64_t ar = m0 + (lane & 15), wn = n0 + (lane & 15);
    const float* ap = A + (ar < p.m ? ar : p.m - 1) * p.a_rs + 4 * (lane >> 4);
    const float* wp = W + (wn < p.n ? wn : p.n - 1) * p.b_cs + 4 * (lane >> 4);
    const int chunks = (int)(p.k / 16);
    // this thread's output element and its epilogue operands, requested before the main loads
    const int64_t erow = m0 + (threadIdx.x >> 4), ecol = n0 + (threadIdx.x & 15);
    const bool eok = erow < p.m && ecol < p.n;
    float bias = 0.f;
    if (eok && p.bias) bias = p.bias[ecol];
    f32x4 fa[NCH], fb[NCH];
#pragma unroll
    for (int g = 0; g < NCH; ++g) {
        int c = wave + 4 * g;
        c = c < chunks ? c : chunks - 1;                  // clamped: unconditional loads stay one batch in flight
        fa[g] = *reinterpret_cast<const f32x4*>(ap + 16 * c);
        fb[g] = *reinterpret_cast<const f32x4*>(wp + 16 * c);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NCH; ++g) {
        const bool live = wave + 4 * g < chunks;          // a clamped (repeated) chunk contributes nothing
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[g][j], live ? fb[g][j] : 0.f, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
    __syncthreads();
    if (!eok) return;
    const int e = threadIdx.x;
    const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    epilogue_store<float, float>(p, v, erow, ecol, bias, reinterpret_cast<float*>(p.c),
                                 reinterpret_cast<const float*>(p.residual), reinterpret_cast<const float*>(p.gate), 0);
}

// true when the fp32 K-split kernel took the problem
static bool skinny_f32_try(const GemmArgs& a, int out_dtype, int64_t batch, hipStream_t s) {
    static const bool enabled = [] {
        const char* e = getenv("MSSR_SKINNY_F32");
        return !e || atoi(e) != 0;
    }();
    if (!enabled || out_dtype != MSSR_F32 || batch != 1 || a.ln_colsum || a.lnb_x) return false;
    if (a.m < 1 || a.m > 1024 || a.n < 1 || a.k < 16 || a.k % 16 != 0 || a.k > 4 * SKF_MAX_CHUNKS * 16) return false;
    if (a.a_cs != 1 || a.b_rs != 1 || a.a_rs % 4 != 0 || a.b_cs % 4 != 0 || !aligned(a.a, 16) || !aligned(a.b, 16)) return false;
    const int64_t blocks = ((a.m + 15) / 16) * ((a.n + 15) / 16);
    if (blocks > 4096 || (a.m + 15) / 16 > 65535) return false;      // large problems: the LDS-tiled kernel reuses operands
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)prof_take(&e0, &e1);
    const dim3 grid((unsigned)((a.n + 15) / 16), (unsigned)((a.m + 15) / 16));
    const int per_wave = (int)((a.k / 16 + 3) / 4);
    if (per_wave <= 4) klaunch_events(gemm_skinny_f32_kernel<4>, grid, dim3(SK_THREADS), 0, s, e0, e1, a);
    else if (per_wave <= 8) klaunch_events(gemm_skinny_f32_kernel<8>, grid, dim3(SK_THREADS), 0, s, e0, e1, a);
    else klaunch_events(gemm_skinny_f32_kernel<16>, grid, dim3(SK_THREADS), 0, s, e0, e1, a);
    return true;
}

// 16x16 tiles (more blocks, 32 KB of loads each) while the row count is small; 32x32 tiles halve the
// L2 traffic once there are plenty of blocks anyway (measured: bs160 decode 104.8 ms/step with 16x16
// vs 107.9 with 32x32; M = 1280 projections 11.6 us with 32x32 vs 14.2 with 16x16)
// (32x32 tiles for the long-K decode products, K >= 1024 or K = 1536 at N = 512, measured in the WGAN step: 52.6 / 51.4 ms
// against 50.7 ms with 16x16 -- 128 blocks are too few even though they halve the L2 traffic.)
int gemm_skinny_tile(int64_t m, int64_t n, int64_t k) {
    static const int64_t f1_max_m = [] {
        const char* e = getenv("MSSR_SKINNY_F1_MAXM");
        return e ? (int64_t)atoll(e) : (int64_t)512;
    }();
    static const int64_t f1_max_blocks = [] {
        const char* e = getenv("MSSR_SKINNY_F1_MAXBLOCKS");
        return e ? (int64_t)atoll(e) : (int64_t)900;     // WGAN step, 240-row sweep: 54.06 ms at 1500 (every decode product
                                                          // on 16x16 tiles), 53.93 at 1400 (N = 1536 on 32x32), 53.86 at 900
                                                          // (N = 1024 too) and at 600
    }();
    if (m <= 16) return 16;
    // vocabulary-sized products (generator, soft embedding): thousands of 16x16 blocks re-read the operands
    // through L2; 32x32 tiles halve that traffic
    const int64_t blocks16 = ((m + 15) / 16) * ((n + 15) / 16) * gemm_skinny_ksplit(k);
    return (m <= f1_max_m && blocks16 <= f1_max_blocks) ? 16 : 32;
}

// in-flight group for a problem: the wave's k-steps, rounded up to 4 / 8 / 12 (32x32 tiles hold 4x the fragments
// per step, so they stay at 4)
static int skinny_group(const GemmArgs& a, int tile) {
    if (tile != 16) return 4;
    const int64_t per_wave = ((a.k / 32 + a.ksplit - 1) / a.ksplit + 3) / 4;
    return per_wave <= 4 ? 4 : per_wave <= 8 ? 8 : per_wave <= 12 ? 12 : 4;
}

template <typename TO, bool LN, int FN>
static void skinny_launch_g(const GemmArgs& a, const dim3 grid, int group, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    tl_skinny_choice = 110 + FN;
    if (skinny_dma_launch<TO, LN, 1, FN>(a, grid, s, e0, e1)) return;
    tl_skinny_choice = 10 + FN;
    if (group == 8) klaunch_events(gemm_skinny_kernel<TO, LN, 1, FN, 8>, grid, dim3(SK_THREADS), 0, s, e0, e1, a);
    else if (group == 12) klaunch_events(gemm_skinny_kernel<TO, LN, 1, FN, 12>, grid, dim3(SK_THREADS), 0, s, e0, e1, a);
    else klaunch_events(gemm_skinny_kernel<TO, LN, 1, FN, 4>, grid, dim3(SK_THREADS), 0, s, e0, e1, a);
}

template <typename TO>
static void skinny_launch(const GemmArgs& a, hipStream_t s) {
    const int tile = gemm_skinny_tile(a.m, a.n, a.k);
    // profiling: the record spans the whole product -- with split-K, from the main kernel's begin to the reduction's end
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)prof_take(&e0, &e1);
    hipEvent_t e1_main = a.ksplit > 1 ? nullptr : e1;
    // 16 x 32 tiles for the long-K products without a LayerNorm prologue (see the kernel's header comment)
    static const int64_t wide_min_k = [] {
        const char* e = getenv("MSSR_SKINNY_16X32_MINK");
        return e ? (int64_t)atoll(e) : (int64_t)1024;
    }();
    static const bool ln_wide = [] { const char* e = getenv("MSSR_SKINNY_LN_16X32"); return !e || atoi(e) != 0; }();
    if (tile == 16 && !a.ln_colsum && a.ksplit == 1 && a.k >= wide_min_k && a.n % 32 == 0 && a.m >= 64) {
        const dim3 grid((unsigned)(a.n / 32), (unsigned)((a.m + 15) / 16), 1);
        skinny_launch_g<TO, false, 2>(a, grid, skinny_group(a, tile), s, e0, e1_main);
    } else if (ln_wide && a.ln_colsum && a.ksplit == 1 && a.n >= 1024 && a.n % 32 == 0 && a.m <= 512) {
        // LayerNorm-folded wide products (QKV, FFN inner) of the decode sweep on 16 x 32 tiles instead of 32 x 32: forward
        // sweep 14.66 -> 14.51 ms
        const dim3 grid((unsigned)(a.n / 32), (unsigned)((a.m + 15) / 16), 1);
        skinny_launch_g<TO, true, 2>(a, grid, skinny_group(a, 16), s, e0, e1_main);
    } else if (tile == 16) {
        const dim3 grid((unsigned)((a.n + 15) / 16), (unsigned)((a.m + 15) / 16), (unsigned)a.ksplit);
        const int g = skinny_group(a, tile);
        if (a.ln_colsum) skinny_launch_g<TO, true, 1>(a, grid, g, s, e0, e1_main);
        else skinny_launch_g<TO, false, 1>(a, grid, g, s, e0, e1_main);
    } else {
        const dim3 grid((unsigned)((a.n + 31) / 32), (unsigned)((a.m + 31) / 32), (unsigned)a.ksplit);
        tl_skinny_choice = 122;
        if (a.ln_colsum) {
            if (!skinny_dma_launch<TO, true, 2, 2>(a, grid, s, e0, e1_main)) {
                tl_skinny_choice = 22;
                klaunch_events(gemm_skinny_kernel<TO, true, 2, 2, 4>, grid, dim3(SK_THREADS), 0, s, e0, e1_main, a);
            }
        } else if (!skinny_dma_launch<TO, false, 2, 2>(a, grid, s, e0, e1_main)) {
            tl_skinny_choice = 22;
            klaunch_events(gemm_skinny_kernel<TO, false, 2, 2, 4>, grid, dim3(SK_THREADS), 0, s, e0, e1_main, a);
        }
    }
    if (a.ksplit > 1) {
        int64_t blocks = (a.m * a.n + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        klaunch_events(gemm_skinny_reduce_kernel<TO>, dim3((unsigned)blocks), dim3(256), 0, s, (hipEvent_t) nullptr, e1, a);
    }
}

int gemm_skinny_ksplit(int64_t k) {
    // K <= 2048 is one slab (LayerNorm prologue, no workspace); beyond that, slabs of `steps` k-steps
    static const int64_t steps = [] {
        const char* e = getenv("MSSR_SKINNY_SLAB_STEPS");
        return e ? (int64_t)atoll(e) : (int64_t)32;      // 160 x 512 x 9632: 23.4 us with 64-step slabs, 17.1 with 32
    }();
    if (k / 32 <= SK_MAX_STEPS) return 1;
    return (int)((k / 32 + steps - 1) / steps);
}

bool gemm_skinny_eligible(const GemmArgs& a, int in_dtype, int64_t batch) {
    if (in_dtype != MSSR_BF16 || batch != 1) return false;
    static const int64_t max_m = [] {
        const char* e = getenv("MSSR_SKINNY_MAXM");
        return e ? (int64_t)atoll(e) : (int64_t)SK_MAX_M;
    }();
    if (a.m < 1 || a.m > (a.ln_colsum ? (int64_t)SK_MAX_M : max_m) || (a.m + 31) / 32 > 65535 || a.k < 32 || a.k % 32 != 0) return false;
    if (a.a_cs != 1 || a.b_rs != 1) return false;                       // both operands k-contiguous
    // many rows AND many columns: the LDS-tiled kernel reuses operands, the K-split tiles re-read them through L2
    // (2560 x 1024 x 512: 22 vs 28 us, 4080 x 1024 x 256: 18 vs 41 us, 1440 x 9606 x 256: 56 vs 127 us); with
    // n <= 512 the K-split kernel still wins (2560 x 512 x 1024: 23 vs 36 us)
    if (!a.ln_colsum && a.m > 1024 && a.n >= 1024) return false;
    if (a.a_rs % 8 != 0 || a.b_cs % 8 != 0 || !aligned(a.a, 16) || !aligned(a.b, 16)) return false;
    const int ks = gemm_skinny_ksplit(a.k);
    if (ks > 1 && (!a.partial || a.ln_colsum)) return false;             // no workspace / LN needs whole rows
    return true;
}

bool gemm_skinny_try(const GemmArgs& a0, int in_dtype, int out_dtype, int64_t batch, hipStream_t s) {
    tl_skinny_choice = 0;
    if (in_dtype == MSSR_F32) return skinny_f32_try(a0, out_dtype, batch, s);
    if (!gemm_skinny_eligible(a0, in_dtype, batch)) return false;
    GemmArgs a = a0;
    a.ksplit = gemm_skinny_ksplit(a.k);
    if (out_dtype == MSSR_BF16) skinny_launch<bf16>(a, s);
    else skinny_launch<float>(a, s);
    return true;
}

}  // namespace mssr
