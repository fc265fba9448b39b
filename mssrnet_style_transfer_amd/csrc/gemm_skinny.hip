// Skinny-M GEMM for the step-wise decode: C[M<=128, N] = A[M,K] . W[N,K]^T with both operands
// k-contiguous (bf16).  Weight-streaming / latency-bound shape: M = 80 rows, N,K = 512..9632.
//
// Mapping: one block per 16 output columns (x ksplit K-chunks); the block's 4 waves split the
// K range between them, so every wave issues ALL the 16-B loads of its K sub-range for a group
// of 4 k-steps straight into MFMA fragment registers (no LDS staging: A is L2-resident, W is
// read exactly once), i.e. a K = 512 product exposes a single memory latency instead of 16
// dependent tile iterations.  Partial accumulators of the 4 waves are summed through LDS; when
// K is split across blocks the fp32 slabs go to a workspace and a second tiny kernel applies
// the epilogue (deterministic, no atomics).
#include "gemm_common.hpp"

namespace mssr {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int SK_THREADS = 256;
constexpr int SK_GROUP = 4;          // k-steps (of 32) in flight per wave
constexpr int SK_MAX_STEPS = 32;     // k-steps per block without split-K (K <= 1024)

template <int MF, typename TO>
__global__ __launch_bounds__(SK_THREADS) void gemm_skinny_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float red[4 * MF * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * 16;
    const int split = blockIdx.y;
    const int64_t steps_total = p.k / 32;
    const int64_t per = (steps_total + p.ksplit - 1) / p.ksplit;
    const int64_t s_begin = split * per;
    const int64_t s_end = (s_begin + per < steps_total) ? s_begin + per : steps_total;

    const bf16* A = reinterpret_cast<const bf16*>(p.a);
    const bf16* W = reinterpret_cast<const bf16*>(p.b);
    const int koff = 8 * (lane >> 4);
    int64_t arow[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        int64_t r = i * 16 + (lane & 15);
        arow[i] = (r < p.m ? r : p.m - 1) * p.a_rs;
    }
    int64_t wn = n0 + (lane & 15);
    const int64_t wrow = (wn < p.n ? wn : p.n - 1) * p.b_cs;     // B viewed [k,n]: n stride = b_cs

    f32x4 acc[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // wave w owns k-steps s_begin + w, + 4, + 8 ... ; processed SK_GROUP at a time
    for (int64_t s0 = s_begin + wave; s0 < s_end; s0 += 4 * SK_GROUP) {
        bf16x8 fa[SK_GROUP][MF], fb[SK_GROUP];
#pragma unroll
        for (int g = 0; g < SK_GROUP; ++g) {
            // unconditional loads (clamped to the last valid k-step): a branch around a load makes
            // hipcc wait vmcnt(0) per load and serialises the whole group
            int64_t s = s0 + 4 * g;
            s = s < s_end ? s : s_end - 1;
            const int64_t k = s * 32 + koff;
            fb[g] = *reinterpret_cast<const bf16x8*>(W + wrow + k);
#pragma unroll
            for (int i = 0; i < MF; ++i) fa[g][i] = *reinterpret_cast<const bf16x8*>(A + arow[i] + k);
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the whole batch of loads ahead of the first MFMA
#pragma unroll
        for (int g = 0; g < SK_GROUP; ++g) {
            // k-steps past the end were loaded from a clamped address: zero the weight fragment with a
            // bit mask (no branch, so the loads above stay one batch in flight)
            const unsigned msk = (s0 + 4 * g < s_end) ? 0xffffffffu : 0u;
            u32x4 wb = __builtin_bit_cast(u32x4, fb[g]);
            wb &= msk;
            const bf16x8 fbm = __builtin_bit_cast(bf16x8, wb);
#pragma unroll
            for (int i = 0; i < MF; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[g][i], fbm, acc[i], 0, 0, 0);
        }
    }
    // cross-wave reduction through LDS: red[wave][frag][lane*4 + r]
#pragma unroll
    for (int i = 0; i < MF; ++i)
        *reinterpret_cast<f32x4*>(&red[(wave * MF + i) * 256 + lane * 4]) = acc[i];
    __syncthreads();
    TO* C = reinterpret_cast<TO*>(p.c);
    const bf16* R = reinterpret_cast<const bf16*>(p.residual);
    const bf16* G = reinterpret_cast<const bf16*>(p.gate);
    const int64_t col = n0 + (lane & 15);
    const float bias = (p.bias && col < p.n) ? p.bias[col] : 0.f;
    for (int i = wave; i < MF; i += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&red[(0 * MF + i) * 256 + lane * 4]);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(&red[(w * MF + i) * 256 + lane * 4]);
            v += t;
        }
        if (col >= p.n) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = i * 16 + (lane >> 4) * 4 + r;
            if (row >= p.m) continue;
            if (p.ksplit > 1) p.partial[((int64_t)split * p.m + row) * p.n + col] = v[r];
            else epilogue_store<bf16, TO>(p, v[r], row, col, bias, C, R, G, 0);
        }
    }
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
static void skinny_launch(GemmArgs a, hipStream_t s) {
    const int mf = (int)((a.m + 15) / 16);
    const dim3 grid((unsigned)((a.n + 15) / 16), (unsigned)a.ksplit);
    switch (mf) {
        case 1: gemm_skinny_kernel<1, TO><<<grid, SK_THREADS, 0, s>>>(a); break;
        case 2: gemm_skinny_kernel<2, TO><<<grid, SK_THREADS, 0, s>>>(a); break;
        case 3: gemm_skinny_kernel<3, TO><<<grid, SK_THREADS, 0, s>>>(a); break;
        case 4: gemm_skinny_kernel<4, TO><<<grid, SK_THREADS, 0, s>>>(a); break;
        case 5: gemm_skinny_kernel<5, TO><<<grid, SK_THREADS, 0, s>>>(a); break;
        case 6: gemm_skinny_kernel<6, TO><<<grid, SK_THREADS, 0, s>>>(a); break;
        case 7: gemm_skinny_kernel<7, TO><<<grid, SK_THREADS, 0, s>>>(a); break;
        default: gemm_skinny_kernel<8, TO><<<grid, SK_THREADS, 0, s>>>(a); break;
    }
    if (a.ksplit > 1) {
        int64_t blocks = (a.m * a.n + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        gemm_skinny_reduce_kernel<TO><<<dim3((unsigned)blocks), 256, 0, s>>>(a);
    }
}

bool gemm_skinny_try(const GemmArgs& a0, int in_dtype, int out_dtype, int64_t batch, hipStream_t s) {
    if (in_dtype != MSSR_BF16 || batch != 1) return false;
    if (a0.m < 1 || a0.m > 128 || a0.k < 32 || a0.k % 32 != 0 || a0.beta != 0.f) return false;
    if (a0.a_cs != 1 || a0.b_rs != 1) return false;                       // both operands k-contiguous
    if (a0.a_rs % 8 != 0 || a0.b_cs % 8 != 0 || !aligned(a0.a, 16) || !aligned(a0.b, 16)) return false;
    GemmArgs a = a0;
    const int64_t steps = a.k / 32;
    a.ksplit = (int)((steps + SK_MAX_STEPS - 1) / SK_MAX_STEPS);
    if (a.ksplit > 1) {
        if (!a.partial) return false;      // caller gave no workspace: use the general kernel
    }
    if (out_dtype == MSSR_BF16) skinny_launch<bf16>(a, s);
    else skinny_launch<float>(a, s);
    return true;
}

}  // namespace mssr
