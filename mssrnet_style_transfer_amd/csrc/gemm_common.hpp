// Shared GEMM argument block and epilogue (gemm.hip: general tiled kernel; gemm_skinny.hip: M <= 128).
#pragma once
#include "common.hpp"

namespace mssr {

struct GemmArgs {
    const void* a;
    const void* b;
    void* c;
    const float* bias;
    const void* residual;
    const void* gate;
    int64_t m, n, k;
    int64_t a_rs, a_cs, b_rs, b_cs, ldc, ldr, ldg;
    int64_t batch_inner;
    int64_t a_bo, a_bi, b_bo, b_bi, c_bo, c_bi;
    float alpha, beta;
    int relu;
    int a_vec, b_vec;   // 16-B vector loads allowed for this operand
    float drop_p;       // > 0: dropout in the epilogue, keep = hash(seed, z*m*n + row*n + col) >= p*2^32
    uint64_t drop_seed;
    // Optional device word ADDED to drop_seed and lnb_seed by the kernels (null: nothing added).  A decode sweep that
    // is replayed as a hipGraph keeps the per-call seed there, so the captured kernel arguments never change
    // (decode.hip: the by-value seeds then hold the site offsets only).
    const uint64_t* seed_base;
    uint64_t* dbg_stamps;   // diagnostics (tools/stamp_bench.py, MSSR_GEMM_STAMPS): [block][8] s_memtime stamps; null otherwise
    int64_t drop_row0;  // rows below this one are NOT dropped (and not rescaled): eval-mode rows sharing a launch
    int64_t k_seg_len, k_seg_stride;    // segmented reduction index (weight-gradient kernel only), 0 = plain
    float* partial;     // skinny split-K: fp32 slabs [ksplit][m][n]
    int ksplit;
    // LayerNorm prologue (skinny kernel only): non-null ln_colsum switches it on
    const float* ln_colsum;
    float* ln_mean;
    float* ln_rstd;
    float ln_eps;
    // LayerNorm-BACKWARD prologue (gemm_lnb.hip): `a` holds dy, the gradient of a LayerNorm output, and the A operand
    // of the product is  dx2 = dropout_replay(dx),  dx = dres + rstd (g - mean(g) - xhat mean(g xhat)),  g = gamma * dy
    // (ln_bwd_dx_kernel's math, engine_common.hpp); the blocks of tile column 0 also store dx and dx2.  Every row
    // operand has the row stride a_rs.  dres must not alias dx (other blocks still read it).
    const void* lnb_x;
    const float* lnb_gamma;
    const float* lnb_mean;
    const float* lnb_rstd;
    const void* lnb_dres;
    void* lnb_dx;
    void* lnb_dx2;
    float lnb_p, lnb_extra;
    uint64_t lnb_seed;
    int64_t lnb_row_off;    // added to the row index in the dropout hash: the rows are a slice of the forward's rows
};

__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, float p) {
    return keep16(seed, idx, drop_thr16(p));
}
// by-value seed + the word at seed_base (uniform: a scalar load)
__device__ __forceinline__ uint64_t eff_seed(const GemmArgs& p, uint64_t s) { return p.seed_base ? s + seed_word(p.seed_base) : s; }
// the seed_base every GemmArgs filled on this thread receives (gemm_fill_args); set by the decode engine while it
// records a sweep, null otherwise
extern thread_local const uint64_t* tl_seed_base;
// what the K-split launcher chose for the calling thread's last product (bench records): 100 * (operands staged by
// LDS-DMA) + 10 * FM + FN for an (16 FM) x (16 FN) tile; 0: the fp32 kernel
extern thread_local int tl_skinny_choice;
extern uint64_t* g_dbg_stamps;
// diagnostics: lane 0 of wave 0 records the real-time clock at point `i` of its block.  Compiled in only by
// `make STAMPS=1` (-DMSSR_GEMM_STAMPS=1): even never-taken stamp branches cost the decode sweep 5 % (they pin the
// instruction schedule around them).
#ifndef MSSR_GEMM_STAMPS
#define MSSR_GEMM_STAMPS 0
#endif
__device__ __forceinline__ void dbg_stamp(const GemmArgs& p, int i) {
    if (MSSR_GEMM_STAMPS && p.dbg_stamps && threadIdx.x == 0) {
        const int64_t b = blockIdx.x + (int64_t)gridDim.x * (blockIdx.y + (int64_t)gridDim.y * blockIdx.z);
        p.dbg_stamps[b * 8 + i] = __builtin_amdgcn_s_memrealtime();
    }
}

// v = alpha*acc (+bias) ; relu ; dropout ; (+residual) ; gate ; (+beta*C_old)
template <typename TI, typename TO>
__device__ __forceinline__ void epilogue_store(const GemmArgs& p, float acc, int64_t row, int64_t col, float bias,
                                               TO* C, const TI* R, const TI* G, int64_t zoff) {
    float v = p.alpha * acc + bias;
    if (p.relu) v = fmaxf(v, 0.f);
    if (p.drop_p > 0.f && row >= p.drop_row0)
        v = drop_keep(eff_seed(p, p.drop_seed), (uint64_t)(zoff + row * p.n + col), p.drop_p) ? v * (1.0f / (1.0f - p.drop_p)) : 0.f;
    if (R) v += to_f(R[row * p.ldr + col]);
    if (G) v = to_f(G[row * p.ldg + col]) > 0.f ? v : 0.f;
    if (p.beta != 0.f) v += p.beta * to_f(C[row * p.ldc + col]);
    C[row * p.ldc + col] = from_f<TO>(v);
}

// returns true when the skinny path took the problem (gemm_skinny.hip)
bool gemm_skinny_try(const GemmArgs& a, int in_dtype, int out_dtype, int64_t batch, hipStream_t s);
bool gemm_skinny_eligible(const GemmArgs& a, int in_dtype, int64_t batch);
// returns true when the k-major ("TN", weight-gradient) kernel took the problem (gemm_tn.hip)
bool gemm_tn_try(const GemmArgs& a, int in_dtype, int out_dtype, int64_t batch, hipStream_t s);
// returns true when the pipelined 128x128 LDS-DMA kernel took the problem (gemm_pipe.hip: large M, bf16)
bool gemm_pipe_try(const GemmArgs& a, int in_dtype, int out_dtype, int64_t batch, hipStream_t s);
bool gemm_pipe_eligible(const GemmArgs& a, int in_dtype, int out_dtype, int64_t batch);     // same test, no launch
// returns true when the all-rows x 48-columns kernel took the problem (gemm_wide.hip: vocabulary-wide N, few rows, K 256 / 512)
bool gemm_wide_try(const GemmArgs& a, int in_dtype, int out_dtype, int64_t batch, hipStream_t s);
// LayerNorm backward fused with the product behind it (a.lnb_* set): false when the shape is outside its reach
bool gemm_lnb_try(const GemmArgs& a, int out_dtype, hipStream_t s);
// mssr_gemm_desc -> GemmArgs (gemm.hip); the lnb_* fields are cleared
void gemm_fill_args(const mssr_gemm_desc* d, GemmArgs& a);
int gemm_skinny_ksplit(int64_t k);
int gemm_skinny_tile(int64_t m, int64_t n, int64_t k);     // 16 or 32: output tile edge the K-split kernel uses

}  // namespace mssr
