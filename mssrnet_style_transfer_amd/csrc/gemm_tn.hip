// Weight-gradient GEMM: C[M,N] = alpha * A^T B (+ beta*C) with BOTH operands stored k-major --
// A as [K, M] rows (m contiguous), B as [K, N] rows (n contiguous), bf16 -- i.e. dW = dY^T X over
// K = all (step, sentence) rows, the one place on this path where the MFMA roof, not latency, counts
// (decode backward at bs160: 12 layers x 1536 x 512 x 4480 = 85 GFLOP in one launch).
//
// The MFMA fragments want 8 consecutive k per lane, the memory has k as the slow dimension.  Tiles go
// to LDS exactly as they lie in memory ([k][m] / [k][n] rows: coalesced 16-B global loads, 16-B LDS
// writes) and come back through gfx950's transposing LDS read ds_read_b64_tr_b16, which hands each
// lane one column of a 4-row block: two of them are one 16x16x32 operand.  No scalar LDS traffic, no
// transposed copy of the activations in HBM.
#include "gemm_common.hpp"
#include <stdlib.h>

namespace mssr {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int TN_THREADS = 256;
constexpr int TN_BK = 32;

struct TnArgs {
    const bf16* a;      // [K, M] rows, row stride lda
    const bf16* b;      // [K, N] rows, row stride ldb
    void* c;
    int64_t m, n, k, lda, ldb, ldc;
    int64_t batch_inner, a_bo, a_bi, b_bo, b_bi, c_bo, c_bi;
    float alpha, beta;
    int a_vec, b_vec;
    int ksplit;         // > 1: blockIdx.z = batch * ksplit + slice, fp32 C pre-set (zero or beta*C), partial sums by atomics
    int64_t seg_len, seg_stride;    // > 0: logical k = s*seg_len + j lives at physical row s*seg_stride + j (mssr_gemm_desc.k_seg_len)
};

// one [TN_BK][W] tile: every thread owns W*TN_BK/8/256 chunks of 8 elements
template <int W>
struct TnLoader {
    static constexpr int CHUNKS = W * TN_BK / 8;
    static constexpr int NCH = CHUNKS / TN_THREADS > 0 ? CHUNKS / TN_THREADS : 1;
    static constexpr int LD = W + 8;                 // +16 B per row
    s16x8 reg[NCH];

    // krow0 = physical row of logical row k0; rows kk >= rem of the tile lie in the next segment, `jump` rows further on
    __device__ __forceinline__ void fetch(const bf16* __restrict__ src, int64_t ld, int64_t k0, int64_t c0, int64_t K,
                                          int64_t cols, bool vec, int64_t krow0, int rem, int64_t jump) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = i * TN_THREADS + threadIdx.x;
            s16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (idx < CHUNKS) {
                const int kk = idx / (W / 8), cc = (idx % (W / 8)) * 8;
                const int64_t gk = k0 + kk, gc = c0 + cc;
                if (gk < K && gc < cols) {
                    const int64_t row = krow0 + kk + (kk >= rem ? jump : 0);
                    const bf16* p = src + row * ld + gc;
                    if (vec && gc + 8 <= cols) {
                        v = *reinterpret_cast<const s16x8*>(p);
                    } else {
                        const short* ps = reinterpret_cast<const short*>(p);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = (gc + j < cols) ? ps[j] : (short)0;
                    }
                }
            }
            reg[i] = v;
        }
    }
    __device__ __forceinline__ void commit(short* __restrict__ lds) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idx = i * TN_THREADS + threadIdx.x;
            if (idx < CHUNKS) {
                const int kk = idx / (W / 8), cc = (idx % (W / 8)) * 8;
                *reinterpret_cast<s16x8*>(lds + kk * LD + cc) = reg[i];
            }
        }
    }
};

// operand fragment (16 columns c0.., 32 k) of a [k][col] LDS tile through the transposing read:
// group g = lane >> 4 takes k = 8g .. 8g+7 as two 4-row blocks; lane 4q+p of the group addresses
// row q, columns 4p .. 4p+3 of the block and receives column (lane & 15) of its 4 rows
__device__ __forceinline__ bf16x8 tn_frag(const short* tile, int ld, int c0, int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const short* base = tile + (8 * g + q) * ld + c0 + 4 * p;
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + 4 * ld));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
}

template <typename TO, int BT>      // BT x BT block tile, 4 waves as 2 x 2
__global__ __launch_bounds__(TN_THREADS) void gemm_tn_kernel(const TnArgs p) {
    constexpr int LD = TnLoader<BT>::LD;
    constexpr int WT = BT / 2;          // wave tile edge
    constexpr int F = WT / 16;          // fragments per wave-tile edge
    __shared__ __attribute__((aligned(16))) short As[TN_BK * LD];
    __shared__ __attribute__((aligned(16))) short Bs[TN_BK * LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int zbi = (int)blockIdx.z / p.ksplit;                       // 32-bit divisions (gridDim.z <= 65535)
    const int64_t zs = (int)blockIdx.z - zbi * p.ksplit;
    const int zoi = zbi / (int)p.batch_inner;
    const int64_t zo = zoi, zi = zbi - zoi * (int)p.batch_inner;
    const bf16* A = p.a + zo * p.a_bo + zi * p.a_bi;
    const bf16* B = p.b + zo * p.b_bo + zi * p.b_bi;
    const int64_t m0 = (int64_t)blockIdx.y * BT, n0 = (int64_t)blockIdx.x * BT;

    f32x4 acc[F][F];
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
        for (int j = 0; j < F; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    TnLoader<BT> la, lb;
    const int64_t nk_all = (p.k + TN_BK - 1) / TN_BK;
    const int64_t per = p.ksplit == 1 ? nk_all                        // k-tiles of this slice
                                      : (int64_t)(((unsigned)nk_all + (unsigned)p.ksplit - 1u) / (unsigned)p.ksplit);
    const int64_t t_begin = zs * per, nk = t_begin + per < nk_all ? t_begin + per : nk_all;
    // segmented k (seg_len >= TN_BK, so a tile crosses at most one segment boundary): uniform 32-bit arithmetic per tile
    const int64_t jump = p.seg_stride - p.seg_len;
    int64_t krow0;
    int rem;
    auto locate = [&](int64_t k0) {
        if (p.seg_len > 0) {
            const uint32_t sg = (uint32_t)k0 / (uint32_t)p.seg_len, j = (uint32_t)k0 - sg * (uint32_t)p.seg_len;
            krow0 = (int64_t)sg * p.seg_stride + j;
            rem = (int)((uint32_t)p.seg_len - j);
        } else {
            krow0 = k0;
            rem = 1 << 30;
        }
    };
    locate(t_begin * TN_BK);
    la.fetch(A, p.lda, t_begin * TN_BK, m0, p.k, p.m, p.a_vec, krow0, rem, jump);
    lb.fetch(B, p.ldb, t_begin * TN_BK, n0, p.k, p.n, p.b_vec, krow0, rem, jump);
    for (int64_t t = t_begin; t < nk; ++t) {
        la.commit(As);
        lb.commit(Bs);
        __syncthreads();
        if (t + 1 < nk) {
            locate((t + 1) * TN_BK);
            la.fetch(A, p.lda, (t + 1) * TN_BK, m0, p.k, p.m, p.a_vec, krow0, rem, jump);
            lb.fetch(B, p.ldb, (t + 1) * TN_BK, n0, p.k, p.n, p.b_vec, krow0, rem, jump);
        }
        bf16x8 fa[F], fb[F];
#pragma unroll
        for (int i = 0; i < F; ++i) fa[i] = tn_frag(As, LD, wm * WT + i * 16, lane);
#pragma unroll
        for (int j = 0; j < F; ++j) fb[j] = tn_frag(Bs, LD, wn * WT + j * 16, lane);
#pragma unroll
        for (int i = 0; i < F; ++i)
#pragma unroll
            for (int j = 0; j < F; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        __syncthreads();
    }
    TO* C = reinterpret_cast<TO*>(p.c) + zo * p.c_bo + zi * p.c_bi;
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const int64_t col = n0 + wn * WT + j * 16 + (lane & 15);
            if (col >= p.n) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + wm * WT + i * 16 + (lane >> 4) * 4 + r;
                if (row >= p.m) continue;
                float v = p.alpha * acc[i][j][r];
                if (p.ksplit > 1) {
                    if constexpr (sizeof(TO) == 4) atomicAdd(reinterpret_cast<float*>(C) + row * p.ldc + col, v);
                    continue;
                }
                if (p.beta != 0.f) v += p.beta * to_f(C[row * p.ldc + col]);
                C[row * p.ldc + col] = from_f<TO>(v);
            }
        }
}

// true when this TN problem was launched here (bf16, A m-contiguous, B n-contiguous, plain epilogue)
bool gemm_tn_try(const GemmArgs& g, int in_dtype, int out_dtype, int64_t batch, hipStream_t s) {
    if (in_dtype != MSSR_BF16) return false;
    if (g.a_rs != 1 || g.b_cs != 1 || g.a_cs < 1 || g.b_rs < 1) return false;      // A[m,k] at m + k*a_cs ; B[k,n] at k*b_rs + n
    if (g.bias || g.residual || g.gate || g.relu || g.drop_p > 0.f || g.ln_colsum) return false;
    if (g.m < 16 || g.n < 16 || g.k < 64) return false;                             // tiny products: the general kernel
    if (!aligned(g.a, 2) || !aligned(g.b, 2)) return false;
    TnArgs t;
    t.a = reinterpret_cast<const bf16*>(g.a); t.b = reinterpret_cast<const bf16*>(g.b); t.c = g.c;
    t.m = g.m; t.n = g.n; t.k = g.k; t.lda = g.a_cs; t.ldb = g.b_rs; t.ldc = g.ldc;
    t.batch_inner = g.batch_inner;
    t.a_bo = g.a_bo; t.a_bi = g.a_bi; t.b_bo = g.b_bo; t.b_bi = g.b_bi; t.c_bo = g.c_bo; t.c_bi = g.c_bi;
    t.alpha = g.alpha; t.beta = g.beta;
    if (g.k_seg_len > 0 && (g.k_seg_len < TN_BK || g.k >= (int64_t)1 << 31 || g.k_seg_stride >= (int64_t)1 << 31)) return false;
    t.seg_len = g.k_seg_len; t.seg_stride = g.k_seg_stride;
    t.a_vec = aligned(g.a, 16) && g.a_cs % 8 == 0 && g.a_bo % 8 == 0 && g.a_bi % 8 == 0;
    t.b_vec = aligned(g.b, 16) && g.b_rs % 8 == 0 && g.b_bo % 8 == 0 && g.b_bi % 8 == 0;
    const int64_t tiles64 = ((g.m + 63) / 64) * ((g.n + 63) / 64) * batch;
    if ((g.m + 31) / 32 > 65535 || batch > 65535) return false;
    // Few output tiles and a long reduction (a critic's 256 x 256 weight over 33 280 rows: 64 blocks, 1 040 k-tiles
    // each, 900 us): slice K over blockIdx.z and add the partial tiles with fp32 atomics into the pre-set output.
    t.ksplit = 1;
    {
        const int64_t blocks = tiles64 >= 128 ? tiles64 : ((g.m + 31) / 32) * ((g.n + 31) / 32) * batch;
        const bool plain_c = out_dtype == MSSR_F32 && g.ldc == g.n && (batch == 1 || (g.c_bi == g.m * g.n && g.batch_inner == batch));
        if (blocks < 512 && g.k >= 4096 && plain_c && (g.beta == 0.f || g.beta == 1.f)) {
            int64_t ks = (1024 + blocks - 1) / blocks;
            if (ks > g.k / 1024) ks = g.k / 1024;
            if (ks > 32) ks = 32;
            if (ks * batch > 65535) ks = 65535 / batch;
            if (ks > 1) {
                t.ksplit = (int)ks;
                // (profiling: the memset before the accumulating kernel is ~2 us and not part of the record)
                if (g.beta == 0.f && hipMemsetAsync(g.c, 0, sizeof(float) * (size_t)(g.m * g.n * batch), s) != hipSuccess)
                    return false;
            }
        }
    }
    const unsigned gz = (unsigned)(batch * t.ksplit);
    // 128x128 tiles (16 MFMAs per wave and LDS barrier instead of 4) once there are >= 1.5 tiles per CU: 12 x 1536 x
    // 512 x 4480 236 -> 161 us (524 TFLOP/s); with 192-304 tiles the 64x64 tile's finer grain wins (84 vs 113 us)
    static const int64_t t128_min = [] {
        const char* e = getenv("MSSR_TN_T128_MIN");
        return e ? (int64_t)atoll(e) : (int64_t)384;
    }();
    const int64_t tiles128 = ((g.m + 127) / 128) * ((g.n + 127) / 128) * batch;
    if (g.m >= 128 && g.n >= 128 && tiles128 >= t128_min) {
        const dim3 grid((unsigned)((g.n + 127) / 128), (unsigned)((g.m + 127) / 128), gz);
        if (out_dtype == MSSR_F32) klaunch(gemm_tn_kernel<float, 128>, grid, dim3(TN_THREADS), 0, s, t);
        else klaunch(gemm_tn_kernel<bf16, 128>, grid, dim3(TN_THREADS), 0, s, t);
    } else if (tiles64 >= 128) {
        const dim3 grid((unsigned)((g.n + 63) / 64), (unsigned)((g.m + 63) / 64), gz);
        if (out_dtype == MSSR_F32) klaunch(gemm_tn_kernel<float, 64>, grid, dim3(TN_THREADS), 0, s, t);
        else klaunch(gemm_tn_kernel<bf16, 64>, grid, dim3(TN_THREADS), 0, s, t);
    } else {
        const dim3 grid((unsigned)((g.n + 31) / 32), (unsigned)((g.m + 31) / 32), gz);
        if (out_dtype == MSSR_F32) klaunch(gemm_tn_kernel<float, 32>, grid, dim3(TN_THREADS), 0, s, t);
        else klaunch(gemm_tn_kernel<bf16, 32>, grid, dim3(TN_THREADS), 0, s, t);
    }
    return true;
}

}  // namespace mssr
