// Large-M GEMM on the matrix cores: 128 x 128 block tile, BK = 64, operands DMA'd global -> LDS
// (global_load_lds_dwordx4, no staging registers) into two LDS buffers, one barrier per K-step.
//
//   C[M,N] = epilogue(alpha * A B)     bf16 operands, fp32 accumulation, bf16 or fp32 output
//
// Operand layouts (per operand, template flags):
//   KM  k contiguous  (activations [rows][k], "NT" weights [n][k]):   LDS image [128 rows][64 k], 128-B rows,
//       fragments by ds_read_b128;
//   RM  rows of k     (a [k][n] operand with n contiguous: dX = dY W, and both operands of dW = dY^T X):
//       LDS image [64 k][128 cols], 256-B rows, fragments by the transposing read ds_read_b64_tr_b16.
// The LDS-DMA writes wave-uniform base + lane * 16 B, so the images cannot be padded; bank conflicts are removed by
// an XOR swizzle of the 16-B chunk index applied on the SOURCE side (each lane fetches the chunk that belongs at its
// slot) and again when the fragments are read (LDS serves a ds_read_b128 in the lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31}, +32 and a tr read in halves of 32 lanes -- MI355X_MICROARCH.md, LDS):
//   KM: chunk c of row r sits at c ^ ((r >> 1) & 7)      (every lane group -> 16 distinct 16-B slots of the bank row)
//   RM: chunk c of row k sits at c ^ 2*((k & 3) | ((k >> 3) & 1) << 2)   (32 lanes of a tr read = 8 rows x 32 B)
// PMC (scratch/pmc_pipe.sh): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.00 in the main loop of both layouts.
//
// Schedule per K-step t:   issue DMA of tile t+1 -> buf[(t+1)&1] ; 32 MFMAs per wave from buf[t&1] ;
//                          s_waitcnt vmcnt(0) ; barrier.
// The barrier orders both hazards: tile t+1 is complete and visible before anyone reads it (RAW), and every wave
// has finished reading buf[t&1] before the DMA of tile t+2 overwrites it (WAR).  64 KB of LDS -> 2 blocks per CU,
// whose K-steps overlap each other's barrier waits.  Measured alternatives (32768 x 512 x 512, bf16): this form
// 34.4 us; a ring of four BK = 32 buffers with three K-steps in flight behind counted vmcnt + raw s_barrier 38.4 us
// (twice the barriers per MFMA); the register-staged BK = 32 kernel of gemm.hip 96.6 us.  For grids of <= 256 blocks (one
// block per CU, nothing to overlap a K-step's DMA with) a ring of four BK = 64 stages, two K-steps in flight behind
// counted vmcnt and one raw barrier per step, was SLOWER: 2560 x 1024 x 512 13.5 -> 15.5 us, 2560 x 512 x 1024 18.0 ->
// 22.2 us (scratch/pipe_bench.py) -- those launches are not waiting for their DMA.  A 128 x 128 tile needs
// 64 B of operand per 4096 flop, i.e. the CU's whole 64 B/clk vector-memory path at the MFMA peak: ~600 TFLOP/s is
// what this tile shape reaches (measured 616 on a 9632 x 512 x 71680 weight gradient); beyond that needs 256-wide tiles.
//
// Blocks are numbered so that the tiles of one 128-row block of A run back to back on ONE XCD (A is then fetched
// from HBM once; the weights, <= 1.5 MB, live in every XCD's L2).
//
// The weight-gradient form (both operands RM) also walks a segmented k (mssr_gemm_desc.k_seg_len) and can slice K
// over blockIdx.y, adding fp32 partial tiles with atomics into a pre-set output.
#include "gemm_common.hpp"
#include <stdlib.h>

namespace mssr {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int PP_THREADS = 256;
constexpr int PP_BM = 128, PP_BN = 128, PP_BK = 64;
constexpr int PP_TILE_BYTES = 128 * PP_BK * 2;      // one operand tile, either layout: 16 KB

struct PipeArgs {
    GemmArgs g;
    int64_t a_ld, b_ld;         // the non-unit stride of each operand
    int tiles_m, tiles_n;
    int ksplit;                 // > 1: gridDim.y = ksplit, fp32 atomics into the pre-set output
    int64_t seg_len, seg_stride;
    int vec_ok;                 // 16-byte epilogue accesses allowed (alignment of C / residual / gate / bias)
};

__device__ __forceinline__ int km_swz(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int rm_swz(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)) << 1; }

// physical row of logical reduction index k (segmented k: logical s*seg_len + j lives at row s*seg_stride + j)
__device__ __forceinline__ int64_t seg_row(int64_t k, int64_t seg_len, int64_t seg_stride) {
    if (seg_len <= 0) return k;
    const uint32_t s = (uint32_t)k / (uint32_t)seg_len;
    return (int64_t)s * seg_stride + ((uint32_t)k - s * (uint32_t)seg_len);
}

// Issues the DMA of one operand tile.  ext = number of rows (KM) / columns (RM) of the operand, x0 = first of this tile.
// X = rows (KM) / columns (RM) of the tile: 128, or 64 for the small-tile KM form
template <bool KM, int X = 128>
__device__ __forceinline__ void pp_stage(const bf16* __restrict__ src, int64_t ld, int64_t x0, int64_t ext, int64_t k0,
                                         int64_t K, unsigned char* tile, int64_t seg_len, int64_t seg_stride) {
    static_assert(KM || X == 128, "the k-major image is 128 columns wide");
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < X * PP_BK * 2 / 16 / PP_THREADS; ++i) {
        const int L = i * PP_THREADS + tid;          // 16-B slot of the image this lane fills
        const bf16* gp;
        if (KM) {
            const int r = L >> 3, pc = L & 7, c = pc ^ km_swz(r);
            int64_t gr = x0 + r, gk = k0 + c * 8;
            gr = gr < ext ? gr : ext - 1;            // rows past the edge: re-read the last row (never stored)
            gk = gk < K ? gk : K - 8;                // k past the end (K % 64 == 32): that half-step is skipped
            gp = src + gr * ld + gk;
        } else {
            const int k = L >> 4, pc = L & 15, c = pc ^ rm_swz(k);
            int64_t gk = k0 + k, gc = x0 + c * 8;
            gk = gk < K ? gk : K - 1;
            gc = gc < ext ? gc : ext - 8;            // ext % 8 == 0
            gp = src + seg_row(gk, seg_len, seg_stride) * ld + gc;
        }
        unsigned char* dst = tile + (size_t)(i * PP_THREADS + (tid & ~63)) * 16;     // wave-uniform; + lane*16 by the DMA
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

// 16 (rows / cols x0..x0+15) x 32 (k = 32*kk ..) operand fragment of a staged tile
template <bool KM>
__device__ __forceinline__ bf16x8 pp_frag(const unsigned char* tile, int x0, int kk, int lane) {
    if (KM) {
        const int r = x0 + (lane & 15), c = kk * 4 + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ km_swz(r)) << 4));
    } else {
        const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
        const int kr = kk * 32 + 8 * g + q, col = x0 + 4 * p;
        const unsigned char* base = tile + kr * 256 + ((((col >> 3) ^ rm_swz(kr)) << 3) + (col & 7)) * 2;
        typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + 4 * 256));   // rows + 4: same swizzle
        s16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return __builtin_bit_cast(bf16x8, r);
    }
}

// BM x BN: block tile.  128 x 128 everywhere but for the k-contiguous ("NT") products with few 128 x 128 tiles (the
// encoders' projections at 3 840 - 5 440 rows: 120 - 170 tiles on 256 CUs, each CU waiting out every K-step's DMA round
// trip alone): 64 x 64 tiles put four times as many blocks on the chip, several per CU, whose K-steps overlap.
template <typename TO, bool A_KM, bool B_KM, int BM = 128, int BN = 128>
__global__ __launch_bounds__(PP_THREADS, 2) void gemm_pipe_kernel(const PipeArgs P) {
    static_assert((BM == 128 && BN == 128) || (A_KM && B_KM), "small tiles: k-contiguous operands only");
    constexpr int A_BYTES = BM * PP_BK * 2, B_BYTES = BN * PP_BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;      // a wave's sub-tile and its fragments
    // ONE shared array (a second LDS object makes hipcc drain the DMA queue before every fragment read)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
    const GemmArgs& p = P.g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // tile of this block: ids congruent mod 8 share an XCD; give each XCD a contiguous run of tiles (bijective remap)
    const int nt = P.tiles_m * P.tiles_n;
    const int id = blockIdx.x, xcd = id & 7, q8 = nt >> 3, r8 = nt & 7;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
    const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    // 32-bit divisions (batch <= 65535, K-steps < 2^31): each 64-bit one is a ~130-instruction software division
    const int bi = (int)p.batch_inner;
    const int64_t zo = (int)blockIdx.z / bi, zi = (int)blockIdx.z % bi;
    const bf16* A = reinterpret_cast<const bf16*>(p.a) + zo * p.a_bo + zi * p.a_bi;
    const bf16* B = reinterpret_cast<const bf16*>(p.b) + zo * p.b_bo + zi * p.b_bi;

    // K-steps of this slice
    const int64_t nk_all = (p.k + PP_BK - 1) / PP_BK;
    const int64_t per = P.ksplit == 1 ? nk_all : (int64_t)(((unsigned)nk_all + (unsigned)P.ksplit - 1u) / (unsigned)P.ksplit);
    const int64_t t_begin = (int64_t)blockIdx.y * per;
    const int64_t t_end = t_begin + per < nk_all ? t_begin + per : nk_all;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // RM operands are fetched in 8-column chunks: a ragged width reads on into the row's padding (ld >= width rounded up)
    const int64_t a_ext = A_KM ? p.m : (p.m + 7) & ~(int64_t)7, b_ext = B_KM ? p.n : (p.n + 7) & ~(int64_t)7;
    if (t_begin < t_end) {
        pp_stage<A_KM, BM>(A, P.a_ld, m0, a_ext, t_begin * PP_BK, p.k, smem, P.seg_len, P.seg_stride);
        pp_stage<B_KM, BN>(B, P.b_ld, n0, b_ext, t_begin * PP_BK, p.k, smem + A_BYTES, P.seg_len, P.seg_stride);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int64_t t = t_begin; t < t_end; ++t) {
        const int cur = (int)((t - t_begin) & 1);
        const unsigned char* At = smem + cur * STAGE;
        const unsigned char* Bt = At + A_BYTES;
        if (t + 1 < t_end) {
            unsigned char* An = smem + (cur ^ 1) * STAGE;
            pp_stage<A_KM, BM>(A, P.a_ld, m0, a_ext, (t + 1) * PP_BK, p.k, An, P.seg_len, P.seg_stride);
            pp_stage<B_KM, BN>(B, P.b_ld, n0, b_ext, (t + 1) * PP_BK, p.k, An + A_BYTES, P.seg_len, P.seg_stride);
        }
        const int ksteps = (p.k - t * PP_BK) >= PP_BK ? 2 : 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk < ksteps) {
                bf16x8 fa[FM], fb[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i) fa[i] = pp_frag<A_KM>(At, wm * WM + i * 16, kk, lane);
#pragma unroll
                for (int j = 0; j < FN; ++j) fb[j] = pp_frag<B_KM>(Bt, wn * WN + j * 16, kk, lane);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // epilogue.  The accumulators (MFMA C layout: col = lane & 15, row = (lane >> 4)*4 + reg) go through the wave's own
    // 16 KB of the now idle LDS as a [64][64] fp32 image (16-column groups XOR-swizzled by the row quad, so the four
    // row groups of one store instruction hit different banks) and come back as whole rows: every lane finishes 16
    // bytes of one output row, with its bias / residual / gate / C operands fetched as 16-byte vectors too.
    // (a wave's [WM][WN] fp32 image: WM * WN * 4 bytes x 4 waves = BM * BN * 4 <= 2 * STAGE for both tile shapes)
    static_assert(BM * BN * 4 <= 2 * STAGE, "epilogue image does not fit the staging buffers");
    float* Wt = reinterpret_cast<float*>(smem) + wave * (WM * WN);
    constexpr int SWZ = FN - 1;                   // 16-column groups of a row, minus one (a power of two)
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i * 16 + (lane >> 4) * 4 + r, col = j * 16 + (lane & 15);
                Wt[row * WN + (col ^ (((row >> 2) & SWZ) << 4))] = acc[i][j][r];
            }
    constexpr int V = 16 / (int)sizeof(TO);      // output elements per lane and instruction: 8 (bf16) or 4 (fp32)
    constexpr int LPR = WN / V, RPI = 64 / LPR;
    const int64_t coff = zo * p.c_bo + zi * p.c_bi;
    TO* C = reinterpret_cast<TO*>(p.c) + coff;
    const bf16* R = p.residual ? reinterpret_cast<const bf16*>(p.residual) + coff : nullptr;
    const bf16* G = p.gate ? reinterpret_cast<const bf16*>(p.gate) + coff : nullptr;
    const int64_t zoff = (int64_t)blockIdx.z * p.m * p.n;
    const float keep_scale = 1.0f / (1.0f - p.drop_p);
    if (P.ksplit > 1) {
        // K slices: one 256-byte row segment per wave instruction (the shape the memory-side fp32 atomics run at full rate)
        if constexpr (sizeof(TO) == 4 && BM == 128 && BN == 128) {
            const int64_t gcol = n0 + wn * 64 + lane;
            if (gcol < p.n) {
#pragma unroll 8
                for (int row = 0; row < 64; ++row) {
                    const int64_t grow = m0 + wm * 64 + row;
                    if (grow >= p.m) break;
                    atomicAdd(reinterpret_cast<float*>(C) + grow * p.ldc + gcol,
                              p.alpha * Wt[row * 64 + (lane ^ (((row >> 2) & 3) << 4))]);
                }
            }
        }
        return;
    }
#pragma unroll 4
    for (int it = 0; it < WM / RPI; ++it) {
        const int row = it * RPI + lane / LPR, c = (lane % LPR) * V;
        const int64_t grow = m0 + wm * WM + row, gcol = n0 + wn * WN + c;
        if (grow >= p.m || gcol >= p.n) continue;
        float v[V];
        const float* src = Wt + row * WN + (c ^ (((row >> 2) & SWZ) << 4));
#pragma unroll
        for (int e = 0; e < V; e += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(src + e);
            v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
        }
        if (P.vec_ok && gcol + V <= p.n) {
            float bias[V], res[V], gt[V], old[V];
#pragma unroll
            for (int e = 0; e < V; ++e) { bias[e] = 0.f; res[e] = 0.f; gt[e] = 1.f; old[e] = 0.f; }
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < V; e += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(p.bias + gcol + e);
                    bias[e] = t[0]; bias[e + 1] = t[1]; bias[e + 2] = t[2]; bias[e + 3] = t[3];
                }
            }
            if (R) {
                const bf16x8 t = *reinterpret_cast<const bf16x8*>(R + grow * p.ldr + (gcol & ~(int64_t)7));
#pragma unroll
                for (int e = 0; e < V; ++e) res[e] = (float)t[(gcol & 7) + e];
            }
            if (G) {
                const bf16x8 t = *reinterpret_cast<const bf16x8*>(G + grow * p.ldg + (gcol & ~(int64_t)7));
#pragma unroll
                for (int e = 0; e < V; ++e) gt[e] = (float)t[(gcol & 7) + e];
            }
            if (p.beta != 0.f) {
                if constexpr (sizeof(TO) == 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(C + grow * p.ldc + gcol);
                    old[0] = t[0]; old[1] = t[1]; old[2] = t[2]; old[3] = t[3];
                } else {
                    const bf16x8 t = *reinterpret_cast<const bf16x8*>(C + grow * p.ldc + gcol);
#pragma unroll
                    for (int e = 0; e < V; ++e) old[e] = (float)t[e];
                }
            }
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float x = p.alpha * v[e] + bias[e];
                if (p.relu) x = fmaxf(x, 0.f);
                if (p.drop_p > 0.f && grow >= p.drop_row0)
                    x = drop_keep(eff_seed(p, p.drop_seed), (uint64_t)(zoff + grow * p.n + gcol + e), p.drop_p) ? x * keep_scale : 0.f;
                x += res[e];
                if (G) x = gt[e] > 0.f ? x : 0.f;
                if (p.beta != 0.f) x += p.beta * old[e];
                v[e] = x;
            }
            if constexpr (sizeof(TO) == 4) {
                *reinterpret_cast<f32x4*>(C + grow * p.ldc + gcol) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = (__bf16)v[e];
                *reinterpret_cast<bf16x8*>(C + grow * p.ldc + gcol) = o;
            }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e)
                if (gcol + e < p.n)
                    epilogue_store<bf16, TO>(p, v[e], grow, gcol + e, p.bias ? p.bias[gcol + e] : 0.f, C, R, G, zoff);
        }
    }
}

template <typename TO>
static void pipe_launch(const PipeArgs& P, bool a_km, bool b_km, dim3 grid, hipStream_t s, bool small_tiles) {
    if (small_tiles) klaunch(gemm_pipe_kernel<TO, true, true, 64, 64>, grid, dim3(PP_THREADS), 0, s, P);
    else if (a_km && b_km) klaunch(gemm_pipe_kernel<TO, true, true>, grid, dim3(PP_THREADS), 0, s, P);
    else if (a_km) klaunch(gemm_pipe_kernel<TO, true, false>, grid, dim3(PP_THREADS), 0, s, P);
    else klaunch(gemm_pipe_kernel<TO, false, false>, grid, dim3(PP_THREADS), 0, s, P);
}

// Eligibility + launch plan of the pipelined kernel; nothing is launched.  false: outside its reach.
static bool pipe_plan(const GemmArgs& g, int in_dtype, int out_dtype, int64_t batch, PipeArgs& P, bool& a_km, bool& b_km) {
    static const bool enabled = [] {
        const char* e = getenv("MSSR_GEMM_PIPE");
        return !e || atoi(e) != 0;
    }();
    static const int64_t min_tiles = [] {
        const char* e = getenv("MSSR_GEMM_PIPE_MIN_TILES");
        return e ? (int64_t)atoll(e) : (int64_t)64;
    }();
    if (!enabled || in_dtype != MSSR_BF16 || g.ln_colsum || g.lnb_x) return false;
    if (g.k < 64 || g.k % 32 != 0 || g.m < 128 || g.n < 64) return false;
    a_km = g.a_cs == 1; b_km = g.b_rs == 1;
    if (!a_km && b_km) return false;                         // A^T with k-contiguous B: not a form this path produces
    const int64_t a_ld = a_km ? g.a_rs : g.a_cs, b_ld = b_km ? g.b_cs : g.b_rs;
    if (!a_km && g.a_rs != 1) return false;
    if (!b_km && g.b_cs != 1) return false;
    if (a_ld % 8 || b_ld % 8 || a_ld < 8 || b_ld < 8 || !aligned(g.a, 16) || !aligned(g.b, 16)) return false;
    if (g.a_bo % 8 || g.a_bi % 8 || g.b_bo % 8 || g.b_bi % 8) return false;
    // RM operands are fetched in 8-column chunks: a ragged width needs rows padded to the next multiple of 8
    if (!a_km && a_ld < ((g.m + 7) & ~(int64_t)7)) return false;
    if (!b_km && b_ld < ((g.n + 7) & ~(int64_t)7)) return false;
    if (g.k_seg_len > 0 && (a_km || g.k_seg_len < 1 || g.k >= (int64_t)1 << 31 || g.k_seg_stride >= (int64_t)1 << 31)) return false;
    if (!a_km && (g.bias || g.residual || g.gate || g.relu || g.drop_p > 0.f)) return false;
    const int64_t tm = (g.m + PP_BM - 1) / PP_BM, tn = (g.n + PP_BN - 1) / PP_BN;
    if (tm * tn > (int64_t)1 << 24 || batch > 65535) return false;
    P.g = g;
    P.a_ld = a_ld; P.b_ld = b_ld;
    P.tiles_m = (int)tm; P.tiles_n = (int)tn;
    P.seg_len = g.k_seg_len; P.seg_stride = g.k_seg_stride;
    P.ksplit = 1;
    const int64_t tiles = tm * tn * batch;
    {
        const int64_t v = out_dtype == MSSR_F32 ? 4 : 8;
        bool ok = aligned(g.c, 16) && g.ldc % v == 0 && g.c_bo % v == 0 && g.c_bi % v == 0;
        if (g.bias) ok = ok && aligned(g.bias, 16);
        if (g.residual) ok = ok && aligned(g.residual, 16) && g.ldr % 8 == 0 && g.c_bo % 8 == 0 && g.c_bi % 8 == 0;
        if (g.gate) ok = ok && aligned(g.gate, 16) && g.ldg % 8 == 0 && g.c_bo % 8 == 0 && g.c_bi % 8 == 0;
        P.vec_ok = ok ? 1 : 0;
    }
    if (!a_km) {
        // weight gradients: few output tiles, long reduction -> slice K over blockIdx.y, fp32 atomics into the pre-set
        // output.  Slice count from a cost model fitted to scratch/pipe_bench.py sweeps (MSSR_PIPE_KS = 1..16 over 21
        // shapes): a K-step takes ~1.14 us with one block on a CU and ~1.45 us with two, 512 blocks run at once, and the
        // memory-side fp32 atomics absorb ~1.6 TB/s of added bytes.
        const bool plain_c = out_dtype == MSSR_F32 && g.ldc == g.n && (batch == 1 || (g.c_bi == g.m * g.n && g.batch_inner == batch));
        if (plain_c && (g.beta == 0.f || g.beta == 1.f)) {
            static const int64_t forced_ks = [] {
                const char* e = getenv("MSSR_PIPE_KS");
                return e ? (int64_t)atoll(e) : (int64_t)0;
            }();
            const int64_t nk = (g.k + PP_BK - 1) / PP_BK;
            int64_t kmax = nk / 4;
            if (kmax > 64) kmax = 64;
            if (kmax * batch > 65535) kmax = 65535 / batch;
            const double out_mb = (double)(g.m * g.n * batch) * 4e-6;
            double best = 1e30;
            for (int64_t ks = 1; ks <= kmax; ++ks) {
                const int64_t blocks = tiles * ks, rounds = (blocks + 511) / 512, per = (nk + ks - 1) / ks;
                const double in_round = (double)blocks / (double)rounds;
                const double t = (double)(rounds * per) * (in_round <= 256.0 ? 1.14 : 1.45) +
                                 (ks > 1 ? (double)ks * out_mb / 1.6 + 2.0 : 0.0);
                if (t < best) { best = t; P.ksplit = (int)ks; }
            }
            if (forced_ks > 0 && kmax >= 1) P.ksplit = (int)(forced_ks < kmax ? forced_ks : kmax);
        }
        if (tiles * P.ksplit < 16) return false;
    } else if (tiles < min_tiles) {
        return false;
    }
    return true;
}

// true when mssr_gemm's pipe-first rule would hand this problem to the pipelined kernel (no launch)
bool gemm_pipe_eligible(const GemmArgs& g, int in_dtype, int out_dtype, int64_t batch) {
    PipeArgs P;
    bool a_km, b_km;
    return pipe_plan(g, in_dtype, out_dtype, batch, P, a_km, b_km);
}

// true when the pipelined kernel took the problem
bool gemm_pipe_try(const GemmArgs& g, int in_dtype, int out_dtype, int64_t batch, hipStream_t s) {
    PipeArgs P;
    bool a_km, b_km;
    if (!pipe_plan(g, in_dtype, out_dtype, batch, P, a_km, b_km)) return false;
    if (!a_km && P.ksplit > 1 && g.beta == 0.f &&
        hipMemsetAsync(g.c, 0, sizeof(float) * (size_t)(g.m * g.n * batch), s) != hipSuccess) return false;
    // few 128 x 128 tiles (under one per CU): 64 x 64 tiles, several blocks per CU (see the kernel's comment)
    static const int64_t small_below = [] {
        const char* e = getenv("MSSR_PIPE_SMALL_BELOW");
        // tools/pipe_mid_bench.py, 128 x 128 -> 64 x 64 tiles: 3840 x 512 x 512 (120 tiles) 13.1 -> 8.3 us, x 1024 (K)
        // 18.1 -> 11.7, 3840 x 1024 x 512 (240) 14.2 -> 12.1, 3840 x 1536 x 512 (360) 17.8 -> 16.6; 32768 x 512 x 512
        // (1024 tiles) 33.2 -> 37.7: the small tiles lose once the big ones fill the chip twice over
        return e ? (int64_t)atoll(e) : (int64_t)400;
    }();
    const bool small = a_km && b_km && (int64_t)P.tiles_m * P.tiles_n * batch < small_below;
    if (small) {
        P.tiles_m = (int)((g.m + 63) / 64);
        P.tiles_n = (int)((g.n + 63) / 64);
    }
    const dim3 grid((unsigned)(P.tiles_m * P.tiles_n), (unsigned)P.ksplit, (unsigned)batch);
    if (out_dtype == MSSR_F32) pipe_launch<float>(P, a_km, b_km, grid, s, small);
    else pipe_launch<bf16>(P, a_km, b_km, grid, s, small);
    return true;
}

}  // namespace mssr
