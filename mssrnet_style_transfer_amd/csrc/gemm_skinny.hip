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

namespace mssr {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int SK_THREADS = 256;
constexpr int SK_GROUP = 4;          // k-steps (of 32) in flight per wave
constexpr int SK_MAX_STEPS = 64;     // k-steps per block without split-K (K <= 2048)

template <typename TO, bool LN>
__global__ __launch_bounds__(SK_THREADS) void gemm_skinny_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float red[4 * 256];
    __shared__ float stat[4][16][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * 16, m0 = (int64_t)blockIdx.y * 16;
    const int split = blockIdx.z;
    const int64_t steps_total = p.k / 32;
    const int64_t per = (steps_total + p.ksplit - 1) / p.ksplit;
    const int64_t s_begin = split * per;
    const int64_t s_end = (s_begin + per < steps_total) ? s_begin + per : steps_total;

    const bf16* A = reinterpret_cast<const bf16*>(p.a);
    const bf16* W = reinterpret_cast<const bf16*>(p.b);
    const int koff = 8 * (lane >> 4);
    const int64_t ar = m0 + (lane & 15), wn = n0 + (lane & 15);
    const int64_t arow = (ar < p.m ? ar : p.m - 1) * p.a_rs;
    const int64_t wrow = (wn < p.n ? wn : p.n - 1) * p.b_cs;     // B viewed [k,n]: n stride = b_cs

    // the element this thread finishes, and its epilogue operands (issued before the main loads)
    const int64_t erow = m0 + (threadIdx.x >> 4), ecol = n0 + (threadIdx.x & 15);
    const bool eok = erow < p.m && ecol < p.n;
    float bias = 0.f, res = 0.f, gate = 1.f, cold = 0.f, lns = 0.f;
    if (eok && p.ksplit == 1) {
        if (p.bias) bias = p.bias[ecol];
        if (p.residual) res = to_f(reinterpret_cast<const bf16*>(p.residual)[erow * p.ldr + ecol]);
        if (p.gate) gate = to_f(reinterpret_cast<const bf16*>(p.gate)[erow * p.ldg + ecol]);
        if (p.beta != 0.f) cold = to_f(reinterpret_cast<const TO*>(p.c)[erow * p.ldc + ecol]);
        if (LN) lns = p.ln_colsum[ecol];
    }

    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    float sx = 0.f, sxx = 0.f;
    // wave w owns k-steps s_begin + w, + 4, + 8 ... ; SK_GROUP of them in flight at a time
    for (int64_t s0 = s_begin + wave; s0 < s_end; s0 += 4 * SK_GROUP) {
        bf16x8 fa[SK_GROUP], fb[SK_GROUP];
#pragma unroll
        for (int g = 0; g < SK_GROUP; ++g) {
            // unconditional loads (clamped to the last valid k-step): a branch around a load makes
            // hipcc wait vmcnt(0) per load and serialises the whole group
            int64_t s = s0 + 4 * g;
            s = s < s_end ? s : s_end - 1;
            const int64_t k = s * 32 + koff;
            fb[g] = *reinterpret_cast<const bf16x8*>(W + wrow + k);
            fa[g] = *reinterpret_cast<const bf16x8*>(A + arow + k);
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the whole batch of loads ahead of the first MFMA
#pragma unroll
        for (int g = 0; g < SK_GROUP; ++g) {
            // k-steps past the end were loaded from a clamped address: zero the weight fragment with a
            // bit mask (no branch, so the loads above stay one batch in flight)
            const bool live = s0 + 4 * g < s_end;
            const unsigned msk = live ? 0xffffffffu : 0u;
            u32x4 wb = __builtin_bit_cast(u32x4, fb[g]);
            wb &= msk;
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[g], __builtin_bit_cast(bf16x8, wb), acc, 0, 0, 0);
            if (LN && live) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = (float)fa[g][j];
                    sx += x;
                    sxx += x * x;
                }
            }
        }
    }
    // partial tiles -> LDS as [wave][row*16 + col]  (MFMA C layout: col = lane & 15, row = (lane >> 4)*4 + r)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
    if (LN) {       // row sums: the 4 lanes (lane & 15, lane >> 4 = 0..3) of a wave hold one row's k-chunks
        sx += __shfl_xor(sx, 16, 64);
        sx += __shfl_xor(sx, 32, 64);
        sxx += __shfl_xor(sxx, 16, 64);
        sxx += __shfl_xor(sxx, 32, 64);
        if (lane < 16) {
            stat[wave][lane][0] = sx;
            stat[wave][lane][1] = sxx;
        }
    }
    __syncthreads();
    if (!eok) return;
    const int t = threadIdx.x;
    float v = (red[t] + red[256 + t]) + (red[512 + t] + red[768 + t]);
    if (p.ksplit > 1) {
        p.partial[((int64_t)split * p.m + erow) * p.n + ecol] = v;
        return;
    }
    if (LN) {
        const int rl = t >> 4;
        const float s1 = (stat[0][rl][0] + stat[1][rl][0]) + (stat[2][rl][0] + stat[3][rl][0]);
        const float s2 = (stat[0][rl][1] + stat[1][rl][1]) + (stat[2][rl][1] + stat[3][rl][1]);
        const float mean = s1 / (float)p.k;
        const float var = fmaxf(s2 / (float)p.k - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.ln_eps);
        v = rstd * (v - mean * lns);
        if (blockIdx.x == 0 && (t & 15) == 0) {
            if (p.ln_mean) p.ln_mean[erow] = mean;
            if (p.ln_rstd) p.ln_rstd[erow] = rstd;
        }
    }
    v = p.alpha * v + bias;
    if (p.relu) v = fmaxf(v, 0.f);
    if (p.drop_p > 0.f)
        v = drop_keep(p.drop_seed, (uint64_t)(erow * p.n + ecol), p.drop_p) ? v * (1.0f / (1.0f - p.drop_p)) : 0.f;
    v += res;
    if (p.gate) v = gate > 0.f ? v : 0.f;
    if (p.beta != 0.f) v += p.beta * cold;
    reinterpret_cast<TO*>(p.c)[erow * p.ldc + ecol] = from_f<TO>(v);
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
        const int64_t row = i / p.n, col = i - row * p.n;
        epilogue_store<bf16, TO>(p, v, row, col, p.bias ? p.bias[col] : 0.f, C, R, G, 0);
    }
}

template <typename TO>
static void skinny_launch(const GemmArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)((a.n + 15) / 16), (unsigned)((a.m + 15) / 16), (unsigned)a.ksplit);
    if (a.ln_colsum) gemm_skinny_kernel<TO, true><<<grid, SK_THREADS, 0, s>>>(a);
    else gemm_skinny_kernel<TO, false><<<grid, SK_THREADS, 0, s>>>(a);
    if (a.ksplit > 1) {
        int64_t blocks = (a.m * a.n + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        gemm_skinny_reduce_kernel<TO><<<dim3((unsigned)blocks), 256, 0, s>>>(a);
    }
}

int gemm_skinny_ksplit(int64_t k) { return (int)((k / 32 + SK_MAX_STEPS - 1) / SK_MAX_STEPS); }

bool gemm_skinny_eligible(const GemmArgs& a, int in_dtype, int64_t batch) {
    if (in_dtype != MSSR_BF16 || batch != 1) return false;
    if (a.m < 1 || a.m > 128 || a.k < 32 || a.k % 32 != 0) return false;
    if (a.a_cs != 1 || a.b_rs != 1) return false;                       // both operands k-contiguous
    if (a.a_rs % 8 != 0 || a.b_cs % 8 != 0 || !aligned(a.a, 16) || !aligned(a.b, 16)) return false;
    const int ks = gemm_skinny_ksplit(a.k);
    if (ks > 1 && (!a.partial || a.ln_colsum)) return false;             // no workspace / LN needs whole rows
    return true;
}

bool gemm_skinny_try(const GemmArgs& a0, int in_dtype, int out_dtype, int64_t batch, hipStream_t s) {
    if (!gemm_skinny_eligible(a0, in_dtype, batch)) return false;
    GemmArgs a = a0;
    a.ksplit = gemm_skinny_ksplit(a.k);
    if (out_dtype == MSSR_BF16) skinny_launch<bf16>(a, s);
    else skinny_launch<float>(a, s);
    return true;
}

}  // namespace mssr
