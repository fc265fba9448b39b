// Strided-batched GEMM with fused epilogue on the gfx950 matrix cores.
//   bf16 inputs: v_mfma_f32_16x16x32_bf16     f32 inputs: v_mfma_f32_16x16x4_f32 (exact f32)
// Block = 256 threads = 4 waves (2 x 2); block tile BM x BN (64x64 or 32x32), BK = 32.
// Both operands are staged through LDS as [row][k] images (k contiguous, rows padded by 16 B)
// so one fragment is one 16-B LDS read (bf16) regardless of the operand's layout in HBM;
// the next K-tile is fetched into registers while the current one feeds the MFMAs.
// Operands may be k-contiguous ("NT" weights, activations) or row-contiguous (transposed
// views: dW = dY^T X, P V, ...): the loader picks the coalesced direction per operand.
#include "gemm_common.hpp"
#include <mutex>
#include <stdlib.h>

namespace mssr {

constexpr int GEMM_THREADS = 256;
constexpr int BK = 32;

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static constexpr int KSTEP = 32;
    static constexpr int PAD = 8;   // elements: +16 B per row
    typedef bf16x8 Frag;
    static __device__ __forceinline__ Frag load(const bf16* row, int kk, int lane) {
        return *reinterpret_cast<const bf16x8*>(row + kk * KSTEP + 8 * (lane >> 4));
    }
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int KSTEP = 4;
    static constexpr int PAD = 4;   // +16 B per row
    typedef float Frag;
    static __device__ __forceinline__ Frag load(const float* row, int kk, int lane) {
        return row[kk * KSTEP + (lane >> 4)];
    }
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};

// One operand tile [ROWS][BK] of a matrix addressed as elem(r, k) = src[r*rs + k*cs].
// Each thread owns NCH chunks of CH = 16 B / sizeof(T) elements.
template <typename T, int ROWS>
struct TileLoader {
    static constexpr int CH = 16 / sizeof(T);
    static constexpr int CHUNKS = ROWS * BK / CH;
    static constexpr int NCH = (CHUNKS + GEMM_THREADS - 1) / GEMM_THREADS;
    static constexpr int LD = BK + Mma<T>::PAD;
    T reg[NCH][CH];

    // kmajor: k is the contiguous direction (cs == 1); else rows are contiguous (rs == 1)
    __device__ __forceinline__ void fetch(const T* __restrict__ src, int64_t row0, int64_t k0, int64_t nrows,
                                          int64_t K, int64_t rs, int64_t cs, bool kmajor, bool vec) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int idx = c * GEMM_THREADS + threadIdx.x;
            if (CHUNKS % GEMM_THREADS != 0 && idx >= CHUNKS) break;
            int r, kk;
            if (kmajor) {
                r = idx / (BK / CH);
                kk = (idx % (BK / CH)) * CH;
            } else {
                kk = idx / (ROWS / CH);
                r = (idx % (ROWS / CH)) * CH;
            }
            const int64_t gr = row0 + r, gk = k0 + kk;
            const T* p = src + gr * rs + gk * cs;
            const bool full = kmajor ? (gr < nrows && gk + CH <= K) : (gk < K && gr + CH <= nrows);
            if (vec && full) {
                *reinterpret_cast<uint4*>(&reg[c][0]) = *reinterpret_cast<const uint4*>(p);
            } else {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const bool ok = kmajor ? (gr < nrows && gk + j < K) : (gk < K && gr + j < nrows);
                    reg[c][j] = ok ? p[kmajor ? j * cs : j * rs] : from_f<T>(0.f);
                }
            }
        }
    }
    __device__ __forceinline__ void commit(T* __restrict__ lds, bool kmajor) const {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int idx = c * GEMM_THREADS + threadIdx.x;
            if (CHUNKS % GEMM_THREADS != 0 && idx >= CHUNKS) break;
            if (kmajor) {
                const int r = idx / (BK / CH), kk = (idx % (BK / CH)) * CH;
                *reinterpret_cast<uint4*>(lds + r * LD + kk) = *reinterpret_cast<const uint4*>(&reg[c][0]);
            } else {
                const int kk = idx / (ROWS / CH), r = (idx % (ROWS / CH)) * CH;
#pragma unroll
                for (int j = 0; j < CH; ++j) lds[(r + j) * LD + kk] = reg[c][j];
            }
        }
    }
};

template <typename TI, typename TO, int BM, int BN>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_kernel(const GemmArgs p) {
    constexpr int LD = BK + Mma<TI>::PAD;
    constexpr int WM = BM / 2, WN = BN / 2;        // wave tile
    constexpr int FM = WM / 16, FN = WN / 16;      // 16x16 fragments per wave
    __shared__ __attribute__((aligned(16))) TI As[BM * LD];
    __shared__ __attribute__((aligned(16))) TI Bs[BN * LD];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // p.ksplit > 1: blockIdx.z = batch * ksplit + K slice; the slices add into the pre-set fp32 output with atomics
    // 32-bit divisions (gridDim.z <= 65535; a 64-bit one is a ~130-instruction software routine at the head of the launch)
    const int zbi = (int)blockIdx.z / p.ksplit;
    const int64_t zb = zbi, zs = (int)blockIdx.z - zbi * p.ksplit;
    const int zoi = zbi / (int)p.batch_inner;
    const int64_t zo = zoi, zi = zbi - zoi * (int)p.batch_inner;
    const TI* A = reinterpret_cast<const TI*>(p.a) + zo * p.a_bo + zi * p.a_bi;
    const TI* B = reinterpret_cast<const TI*>(p.b) + zo * p.b_bo + zi * p.b_bi;
    const int64_t coff = zo * p.c_bo + zi * p.c_bi;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    const bool a_km = p.a_cs == 1, b_km = p.b_rs == 1;   // B viewed [k,n]: k contiguous when b_rs == 1

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    TileLoader<TI, BM> la;
    TileLoader<TI, BN> lb;
    const int64_t nk_all = (p.k + BK - 1) / BK;
    const int64_t per = p.ksplit == 1 ? nk_all : (int64_t)(((unsigned)nk_all + (unsigned)p.ksplit - 1u) / (unsigned)p.ksplit);
    const int64_t t0 = zs * per, nk = t0 + per < nk_all ? t0 + per : nk_all;
    la.fetch(A, m0, t0 * BK, p.m, p.k, p.a_rs, p.a_cs, a_km, p.a_vec);
    lb.fetch(B, n0, t0 * BK, p.n, p.k, p.b_cs, p.b_rs, b_km, p.b_vec);   // rows of the B image are n
    for (int64_t t = t0; t < nk; ++t) {
        la.commit(As, a_km);
        lb.commit(Bs, b_km);
        __syncthreads();
        if (t + 1 < nk) {
            la.fetch(A, m0, (t + 1) * BK, p.m, p.k, p.a_rs, p.a_cs, a_km, p.a_vec);
            lb.fetch(B, n0, (t + 1) * BK, p.n, p.k, p.b_cs, p.b_rs, b_km, p.b_vec);
        }
#pragma unroll
        for (int kk = 0; kk < BK / Mma<TI>::KSTEP; ++kk) {
            typename Mma<TI>::Frag fa[FM], fb[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = Mma<TI>::load(As + (wm * WM + i * 16 + (lane & 15)) * LD, kk, lane);
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[j] = Mma<TI>::load(Bs + (wn * WN + j * 16 + (lane & 15)) * LD, kk, lane);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = Mma<TI>::mma(fa[i], fb[j], acc[i][j]);
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4)*4 + reg
    TO* C = reinterpret_cast<TO*>(p.c) + coff;
    const TI* R = p.residual ? reinterpret_cast<const TI*>(p.residual) + coff : nullptr;
    const TI* G = p.gate ? reinterpret_cast<const TI*>(p.gate) + coff : nullptr;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int64_t col = n0 + wn * WN + j * 16 + (lane & 15);
            if (col >= p.n) continue;
            const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
                if (row >= p.m) continue;
                if (p.ksplit > 1) {
                    if constexpr (sizeof(TO) == 4) atomicAdd(reinterpret_cast<float*>(C) + row * p.ldc + col, p.alpha * acc[i][j][r]);
                    continue;
                }
                epilogue_store<TI, TO>(p, acc[i][j][r], row, col, bias, C, R, G, (int64_t)blockIdx.z * p.m * p.n);
            }
        }
    }
}

static inline bool use_tile64(int64_t m, int64_t n, int64_t batch) {
    const int64_t tiles64 = ((m + 63) / 64) * ((n + 63) / 64) * batch;
    return tiles64 >= 192 && m > 32 && n > 32;
}
// block tile for a problem: 128x128 (4x the MFMA work per LDS barrier of the 64x64 tile) once there are plenty of
// tiles for the 256 CUs, else 64x64, else 32x32
static inline int pick_tile(int64_t m, int64_t n, int64_t batch) {
    static const int forced = [] {
        const char* e = getenv("MSSR_GEMM_TILE");
        return e ? atoi(e) : 0;
    }();
    if (forced) return forced;
    const auto tiles = [&](int bm, int bn) { return ((m + bm - 1) / bm) * ((n + bn - 1) / bn) * batch; };
    // measured (scratch/gemm_bench.py, bf16 NT): 128x128 wins from ~1000 tiles on (32768 x 512 x 512: 93 vs 105 us),
    // 64x128 never beat 64x64; 64x64 is best in between
    if (m >= 128 && n >= 128 && tiles(128, 128) >= 1024) return 128128;
    return use_tile64(m, n, batch) ? 6464 : 3232;
}

// K slices for the general kernel: few output tiles and a long reduction with a plain fp32 output (the critics' score
// layers: dW of a Linear(d -> 1) is a 1 x d x rows product, one block looping over 32 768 rows took 0.9-1.5 ms)
static int general_ksplit(const GemmArgs& a, int64_t batch, int out_dtype, int tile) {
    if (out_dtype != MSSR_F32 || a.k < 2048) return 1;
    if (a.bias || a.residual || a.gate || a.relu || a.drop_p > 0.f || (a.beta != 0.f && a.beta != 1.f)) return 1;
    if (a.ldc != a.n || !(batch == 1 || (a.c_bi == a.m * a.n && a.batch_inner == batch))) return 1;
    const int bm = tile == 3232 ? 32 : tile == 6464 ? 64 : 128, bn = tile == 3232 ? 32 : tile == 6464 ? 64 : 128;
    const int64_t blocks = ((a.m + bm - 1) / bm) * ((a.n + bn - 1) / bn) * batch;
    if (blocks >= 256) return 1;
    int64_t ks = (512 + blocks - 1) / blocks;
    if (ks > a.k / 512) ks = a.k / 512;
    if (ks > 128) ks = 128;
    if (ks * batch > 65535) ks = 65535 / batch;
    return ks > 1 ? (int)ks : 1;
}

template <typename TI, typename TO>
static void launch(const GemmArgs& a0, int64_t batch0, hipStream_t s) {
    const int t = pick_tile(a0.m, a0.n, batch0);
    GemmArgs a = a0;
    a.ksplit = general_ksplit(a0, batch0, sizeof(TO) == 4 ? MSSR_F32 : MSSR_BF16, t);
    if (a.ksplit > 1 && a.beta == 0.f &&
        hipMemsetAsync(a.c, 0, sizeof(float) * (size_t)(a.m * a.n * batch0), s) != hipSuccess) a.ksplit = 1;
    const int64_t batch = batch0 * a.ksplit;
    // MSSR_GEMM_LOG=1: one stderr line per launch of this (general) kernel -- which problems miss the K-split kernel
    static const bool log_shapes = getenv("MSSR_GEMM_LOG") != nullptr;
    if (log_shapes)
        fprintf(stderr, "mssr_gemm general tile=%d m=%lld n=%lld k=%lld batch=%lld a_cs=%lld b_rs=%lld in=%d out=%d\n", t,
                (long long)a.m, (long long)a.n, (long long)a.k, (long long)batch, (long long)a.a_cs, (long long)a.b_rs,
                (int)sizeof(TI), (int)sizeof(TO));
    if (t == 128128) {
        dim3 grid((unsigned)((a.n + 127) / 128), (unsigned)((a.m + 127) / 128), (unsigned)batch);
        klaunch(gemm_kernel<TI, TO, 128, 128>, grid, dim3(GEMM_THREADS), 0, s, a);
    } else if (t == 64128) {
        dim3 grid((unsigned)((a.n + 127) / 128), (unsigned)((a.m + 63) / 64), (unsigned)batch);
        klaunch(gemm_kernel<TI, TO, 64, 128>, grid, dim3(GEMM_THREADS), 0, s, a);
    } else if (t == 6464) {
        dim3 grid((unsigned)((a.n + 63) / 64), (unsigned)((a.m + 63) / 64), (unsigned)batch);
        klaunch(gemm_kernel<TI, TO, 64, 64>, grid, dim3(GEMM_THREADS), 0, s, a);
    } else {
        dim3 grid((unsigned)((a.n + 31) / 32), (unsigned)((a.m + 31) / 32), (unsigned)batch);
        klaunch(gemm_kernel<TI, TO, 32, 32>, grid, dim3(GEMM_THREADS), 0, s, a);
    }
}

}  // namespace mssr

thread_local const uint64_t* mssr::tl_seed_base = nullptr;
thread_local int mssr::tl_skinny_choice = 0;
uint64_t* mssr::g_dbg_stamps = nullptr;
extern "C" void mssr_debug_stamps(void* buf) { mssr::g_dbg_stamps = (uint64_t*)buf; }

using namespace mssr;

extern "C" int mssr_gemm_tile(int64_t m, int64_t n, int64_t batch) { return use_tile64(m, n, batch) ? 64 : 32; }

void mssr::gemm_fill_args(const mssr_gemm_desc* d, GemmArgs& a) {
    a.lnb_x = nullptr; a.lnb_gamma = a.lnb_mean = a.lnb_rstd = nullptr; a.lnb_dres = nullptr; a.lnb_dx = a.lnb_dx2 = nullptr;
    a.lnb_p = 0.f; a.lnb_extra = 1.f; a.lnb_seed = 0; a.lnb_row_off = 0;
    a.a = d->a; a.b = d->b; a.c = d->c; a.bias = d->bias; a.residual = d->residual; a.gate = d->gate;
    a.m = d->m; a.n = d->n; a.k = d->k;
    a.a_rs = d->a_rs; a.a_cs = d->a_cs; a.b_rs = d->b_rs; a.b_cs = d->b_cs;
    a.ldc = d->ldc; a.ldr = d->ldr; a.ldg = d->ldg;
    a.batch_inner = d->batch_inner;
    a.a_bo = d->a_bo; a.a_bi = d->a_bi; a.b_bo = d->b_bo; a.b_bi = d->b_bi; a.c_bo = d->c_bo; a.c_bi = d->c_bi;
    a.alpha = d->alpha; a.beta = d->beta; a.relu = d->relu;
    a.drop_p = d->drop_p; a.drop_seed = d->drop_seed; a.drop_row0 = d->drop_row0;
    a.seed_base = tl_seed_base;
    a.dbg_stamps = g_dbg_stamps;
    a.k_seg_len = d->k_seg_len; a.k_seg_stride = d->k_seg_stride;
    a.ksplit = 1;
    a.partial = nullptr;
    if (d->workspace && d->k > 32 * 64) {
        const int64_t need = gemm_skinny_ksplit(d->k) * d->m * d->n * (int64_t)sizeof(float);
        if (d->workspace_bytes >= need && aligned(d->workspace, 16)) a.partial = (float*)d->workspace;
    }
    a.ln_colsum = d->ln_colsum; a.ln_mean = d->ln_mean; a.ln_rstd = d->ln_rstd; a.ln_eps = d->ln_eps;
    const int ch = 16 / (int)dsize(d->in_dtype);
    // 16-B vector loads need: base aligned, the non-unit stride and batch strides multiples of the chunk
    const int64_t a_ld = d->a_cs == 1 ? d->a_rs : d->a_cs;
    const int64_t b_ld = d->b_rs == 1 ? d->b_cs : d->b_rs;
    a.a_vec = aligned(d->a, 16) && a_ld % ch == 0 && d->a_bo % ch == 0 && d->a_bi % ch == 0;
    a.b_vec = aligned(d->b, 16) && b_ld % ch == 0 && d->b_bo % ch == 0 && d->b_bi % ch == 0;
}

// 1 when mssr_gemm routes this problem to the skinny-M kernel (same eligibility test, no launch)
extern "C" int mssr_gemm_is_skinny(const mssr_gemm_desc* d) {
    if (!d || d->no_skinny || !valid_dtype(d->in_dtype)) return 0;
    GemmArgs a;
    gemm_fill_args(d, a);
    return gemm_skinny_eligible(a, d->in_dtype, d->batch_outer * d->batch_inner) ? 1 : 0;
}

// measured (scratch/pipe_bench.py): from this many rows on the LDS-tiled DMA kernel beats the K-split one
static int64_t pipe_first_min_m() {
    static const int64_t v = [] {
        const char* e = getenv("MSSR_PIPE_FIRST_MINM");
        return e ? (int64_t)atoll(e) : (int64_t)2048;
    }();
    return v;
}

static int gemm_run(const mssr_gemm_desc* d, void* stream, const char* who) {
    MSSR_REQUIRE(d, "%s: null descriptor", who);
    MSSR_REQUIRE(valid_dtype(d->in_dtype) && valid_dtype(d->out_dtype), "%s: bad dtype", who);
    MSSR_REQUIRE(!(d->in_dtype == MSSR_F32 && d->out_dtype == MSSR_BF16), "%s: f32 -> bf16 is not supported", who);
    MSSR_REQUIRE(d->m >= 0 && d->n >= 0 && d->k >= 0, "%s: negative size", who);
    MSSR_REQUIRE(d->batch_outer >= 1 && d->batch_inner >= 1, "%s: batch counts must be >= 1", who);
    const int64_t batch = d->batch_outer * d->batch_inner;
    MSSR_REQUIRE(batch <= 65535, "%s: batch %lld exceeds 65535", who, (long long)batch);
    if (d->m == 0 || d->n == 0) return MSSR_OK;
    MSSR_REQUIRE(d->a && d->b && d->c, "%s: null pointer", who);
    MSSR_REQUIRE(d->a_rs == 1 || d->a_cs == 1, "%s: A needs a unit stride (a_rs=%lld a_cs=%lld)", who,
                 (long long)d->a_rs, (long long)d->a_cs);
    MSSR_REQUIRE(d->b_rs == 1 || d->b_cs == 1, "%s: B needs a unit stride (b_rs=%lld b_cs=%lld)", who,
                 (long long)d->b_rs, (long long)d->b_cs);
    MSSR_REQUIRE(d->ldc >= d->n, "%s: ldc < n", who);
    MSSR_REQUIRE(!d->residual || d->ldr >= d->n, "%s: ldr < n", who);
    MSSR_REQUIRE(!d->gate || d->ldg >= d->n, "%s: ldg < n", who);
    MSSR_REQUIRE((d->m + 31) / 32 <= 65535, "%s: m too large", who);

    MSSR_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "%s: drop_p=%f out of [0,1)", who, d->drop_p);
    MSSR_REQUIRE((d->k_seg_len == 0 && d->k_seg_stride == 0) || (d->k_seg_len > 0 && d->k_seg_stride >= d->k_seg_len),
                 "%s: bad k segmentation (len %lld, stride %lld)", who, (long long)d->k_seg_len, (long long)d->k_seg_stride);
    GemmArgs a;
    gemm_fill_args(d, a);
    hipStream_t s = (hipStream_t)stream;
    const int pi = prof_begin(s);
    if (d->k_seg_len > 0) {         // only the weight-gradient kernels walk a segmented k
        if (!d->no_skinny && gemm_pipe_try(a, d->in_dtype, d->out_dtype, batch, s)) {
            prof_end(pi, s, 4, d->in_dtype, d->out_dtype, 128, batch, d->m, d->n, d->k);
            MSSR_LAUNCH_OK("gemm(pipe)");
        }
        if (!d->no_skinny && gemm_tn_try(a, d->in_dtype, d->out_dtype, batch, s)) {
            prof_end(pi, s, 2, d->in_dtype, d->out_dtype, 64, batch, d->m, d->n, d->k);
            MSSR_LAUNCH_OK("gemm(tn)");
        }
        set_error("%s: a segmented k needs the weight-gradient form (bf16, A m-contiguous, B n-contiguous, plain epilogue)", who);
        return MSSR_ERR_UNSUPPORTED;
    }
    if (!d->no_skinny && d->m >= pipe_first_min_m() && gemm_pipe_try(a, d->in_dtype, d->out_dtype, batch, s)) {
        prof_end(pi, s, 4, d->in_dtype, d->out_dtype, 128, batch, d->m, d->n, d->k);
        MSSR_LAUNCH_OK("gemm(pipe)");
    }
    if (!d->no_skinny && gemm_wide_try(a, d->in_dtype, d->out_dtype, batch, s)) {
        prof_end(pi, s, 5, d->in_dtype, d->out_dtype, 48, batch, d->m, d->n, d->k, 1, d->ln_colsum ? 1 : 0);
        MSSR_LAUNCH_OK("gemm(wide)");
    }
    if (!d->no_skinny && gemm_skinny_try(a, d->in_dtype, d->out_dtype, batch, s)) {
        prof_end(pi, s, 1, d->in_dtype, d->out_dtype, tl_skinny_choice, batch, d->m, d->n, d->k,
                 gemm_skinny_ksplit(d->k), d->ln_colsum ? 1 : 0);
        MSSR_LAUNCH_OK("gemm(skinny)");
    }
    if (!d->no_skinny && gemm_pipe_try(a, d->in_dtype, d->out_dtype, batch, s)) {
        prof_end(pi, s, 4, d->in_dtype, d->out_dtype, 128, batch, d->m, d->n, d->k);
        MSSR_LAUNCH_OK("gemm(pipe)");
    }
    if (!d->no_skinny && gemm_tn_try(a, d->in_dtype, d->out_dtype, batch, s)) {
        prof_end(pi, s, 2, d->in_dtype, d->out_dtype, 64, batch, d->m, d->n, d->k);
        MSSR_LAUNCH_OK("gemm(tn)");
    }
    if (d->ln_colsum) {
        set_error("%s: the LayerNorm prologue needs the K-split kernel (bf16, m <= 8192, k <= 2048, k %% 32 == 0, "
                  "k-contiguous 16-byte aligned operands)", who);
        return MSSR_ERR_UNSUPPORTED;
    }
    if (d->in_dtype == MSSR_F32) launch<float, float>(a, batch, s);
    else if (d->out_dtype == MSSR_BF16) launch<bf16, bf16>(a, batch, s);
    else launch<bf16, float>(a, batch, s);
    prof_end(pi, s, 0, d->in_dtype, d->out_dtype, use_tile64(d->m, d->n, batch) ? 64 : 32, batch, d->m, d->n, d->k);
    MSSR_LAUNCH_OK("gemm");
}

extern "C" int mssr_gemm(const mssr_gemm_desc* d, void* stream) { return gemm_run(d, stream, "gemm"); }

// ------------------------------------------------------------------ the allocation-free variant
// mssr_gemm for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
// forbid hipMalloc): the split-K slabs of the K-split kernel -- the library's one scratch buffer, a caller-supplied
// workspace in mssr_gemm -- live in a file-scope device array instead.  Each GPU the library runs on holds its own
// copy of the array (MSSR_NOALLOC_SCRATCH_BYTES) at its own address until the process ends; the address is looked
// up once per device with hipGetSymbolAddress and only that pointer is ever handed to kernels (the array's NAME in
// host code is a host-side stand-in, not a device address).
static __device__ __align__(256) unsigned char scratch[MSSR_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
struct ScratchSlot {
    std::mutex mu;
    void* ptr = nullptr;
};
ScratchSlot g_scratch[kMaxDevices];
}  // namespace

// the calling thread's current device's copy of `scratch`; nullptr + error message on failure
static void* scratch_of_current_device(const char* who) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        set_error("%s: cannot identify the current device", who);
        return nullptr;
    }
    ScratchSlot& slot = g_scratch[dev];
    std::lock_guard<std::mutex> lock(slot.mu);
    if (!slot.ptr) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            set_error("%s: hipGetSymbolAddress(scratch) failed on device %d: %s", who, dev, hipGetErrorString(e));
            return nullptr;
        }
        slot.ptr = p;
    }
    return slot.ptr;
}

extern "C" int64_t mssr_gemm_noalloc_scratch_bytes(void) { return (int64_t)MSSR_NOALLOC_SCRATCH_BYTES; }

extern "C" int mssr_gemm_noalloc(const mssr_gemm_desc* d, void* stream) {
    MSSR_REQUIRE(d, "gemm_noalloc: null descriptor");
    mssr_gemm_desc local = *d;
    local.workspace = nullptr;                 // the caller's workspace, if any, is not used
    local.workspace_bytes = 0;
    if (valid_dtype(d->in_dtype) && d->m > 0 && d->n > 0 && d->k > 32 * 64) {
        // would the K-split kernel take this problem if it had its slabs?  (probe with a non-null placeholder that
        // is never dereferenced: eligibility only tests it against nullptr)
        GemmArgs probe;
        gemm_fill_args(&local, probe);
        probe.partial = reinterpret_cast<float*>(uintptr_t(256));
        // mssr_gemm's own order (gemm_run): the pipelined kernel comes first from pipe_first_min_m rows on and needs
        // no scratch (gemm_wide never sees k > 2048); only a problem the K-split kernel would really get is held to
        // the size of the static array
        const int64_t batch = d->batch_outer * d->batch_inner;
        const bool pipe_first = d->m >= pipe_first_min_m() && gemm_pipe_eligible(probe, d->in_dtype, d->out_dtype, batch);
        if (!d->no_skinny && !d->ln_colsum && !pipe_first && gemm_skinny_eligible(probe, d->in_dtype, batch)) {
            const int64_t need = gemm_skinny_ksplit(d->k) * d->m * d->n * (int64_t)sizeof(float);
            MSSR_REQUIRE(need <= (int64_t)MSSR_NOALLOC_SCRATCH_BYTES,
                         "gemm_noalloc: split-K slabs of a %lld x %lld x %lld product need %lld bytes, beyond the "
                         "%lld-byte static scratch (use mssr_gemm with a workspace)",
                         (long long)d->m, (long long)d->n, (long long)d->k, (long long)need,
                         (long long)MSSR_NOALLOC_SCRATCH_BYTES);
            void* ptr = scratch_of_current_device("gemm_noalloc");
            if (!ptr) return MSSR_ERR_LAUNCH;
            local.workspace = ptr;             // every slab element is written before the reduce kernel reads it:
            local.workspace_bytes = (int64_t)MSSR_NOALLOC_SCRATCH_BYTES;   // nothing has to start at zero
        }
    }
    return gemm_run(&local, stream, "gemm_noalloc");
}
