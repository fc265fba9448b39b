// Pieces shared by the two engines (decode.hip: step-wise soft decode; encoder.hip: encoder stack).
#pragma once
#include "common.hpp"

namespace mssr {

struct GemmCtx {
    hipStream_t s;
    int dt;
    size_t es;      // element size of the compute dtype
    void* ws;       // split-K workspace of mssr_gemm
    int64_t ws_bytes;
};

static inline char* at(void* base, int64_t elems, size_t es) { return (char*)base + elems * (int64_t)es; }

// C[m, n] = A[m,k] . W[n,k]^T (+ epilogue) -- every engine GEMM has this NT, k-contiguous form
static int gemm_nt(const GemmCtx& c, const void* a, int64_t lda, const void* w, int64_t ldw, void* out, int64_t ldc,
                   int64_t m, int64_t n, int64_t k, const float* bias, int relu, const void* residual, int64_t ldr,
                   const void* gate, int64_t ldg, float alpha, float beta, int out_dt, float drop_p, uint64_t seed,
                   const float* ln_colsum = nullptr, float* ln_mean = nullptr, float* ln_rstd = nullptr,
                   int64_t drop_row0 = 0) {
    mssr_gemm_desc d = {};
    d.a = a; d.b = w; d.c = out; d.bias = bias; d.residual = residual; d.gate = gate;
    d.m = m; d.n = n; d.k = k;
    d.a_rs = lda; d.a_cs = 1; d.b_rs = 1; d.b_cs = ldw;
    d.ldc = ldc; d.ldr = ldr; d.ldg = ldg;
    d.batch_outer = d.batch_inner = 1;
    d.a_bo = d.a_bi = d.b_bo = d.b_bi = d.c_bo = d.c_bi = 0;
    d.alpha = alpha; d.beta = beta; d.relu = relu;
    d.in_dtype = c.dt; d.out_dtype = out_dt;
    d.drop_p = drop_p; d.drop_seed = seed; d.drop_row0 = drop_row0;
    d.workspace = c.ws; d.workspace_bytes = c.ws_bytes; d.no_skinny = 0;
    d.ln_colsum = ln_colsum; d.ln_mean = ln_mean; d.ln_rstd = ln_rstd; d.ln_eps = 1e-6f;
    return mssr_gemm(&d, c.s);
}

// dx = dres + rstd * (a - mean(a) - xhat * mean(a * xhat)), a = gamma * dy   (no parameter gradients:
// those are contracted once over all steps by the caller).  dx2 = keep(seed, i) * dx * extra / (1 - p):
// the same gradient pushed back through the dropout of the block below (the mask is regenerated from
// the seed the forward used), i.e. the gradient of that block's last Linear output.
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const T* __restrict__ dres,
                                                        T* __restrict__ dx, T* __restrict__ dx2, int rows, int d,
                                                        float p, uint64_t seed, float extra, int row_off,
                                                        const uint64_t* __restrict__ seed_base = nullptr) {
    if (seed_base) seed += seed_word(seed_base);          // replayed sweeps keep the call's seed in memory (decode.hip)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float mu = mean[row], rs = rstd[row];
    const T* xr = x + (int64_t)row * d;
    const T* dr = dy + (int64_t)row * d;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        float xv[4], dv[4], g[4];
        V4<T>::load(xr + c, xv);
        V4<T>::load(dr + c, dv);
        V4<float>::load(gamma + c, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = dv[j] * g[j];
            s1 += a;
            s2 += a * (xv[j] - mu) * rs;
        }
    }
    const float c1 = wave_sum(s1) / (float)d, c2 = wave_sum(s2) / (float)d;
    const float sc = extra / (1.0f - p);
    const uint32_t thr = drop_thr16(p);
    for (int c = lane * 4; c < d; c += 256) {
        float xv[4], dv[4], g[4], rr[4] = {0.f, 0.f, 0.f, 0.f}, o[4], o2[4];
        V4<T>::load(xr + c, xv);
        V4<T>::load(dr + c, dv);
        V4<float>::load(gamma + c, g);
        if (dres) V4<T>::load(dres + (int64_t)row * d + c, rr);
        uint32_t kp[2] = {3u, 3u};          // row * d + c is a multiple of 4 (vector loads): two whole decision pairs
        if (p > 0.f) {
            const uint64_t i2 = ((uint64_t)(row + row_off) * d + c) >> 1;
            kp[0] = keep16_pair(seed, i2, thr);
            kp[1] = keep16_pair(seed, i2 + 1, thr);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = rr[j] + rs * (dv[j] * g[j] - c1 - (xv[j] - mu) * rs * c2);
            // the forward rounded nothing between dropout and the residual add, but dx IS stored in T:
            // replay from the stored value so that dx2 matches a separate pass over dx bit for bit
            const float ov = to_f(from_f<T>(o[j]));
            o2[j] = (p > 0.f) ? (((kp[j >> 1] >> (j & 1)) & 1u) ? ov * sc : 0.f) : ov * extra;
        }
        V4<T>::store(dx + (int64_t)row * d + c, o);
        if (dx2) V4<T>::store(dx2 + (int64_t)row * d + c, o2);
    }
}


template <typename T>
static void launch_ln_bwd_dx(hipStream_t s, const void* dy, const void* x, const float* g, const float* mean,
                             const float* rstd, const void* dres, void* dx, void* dx2, int rows, int d, float p,
                             uint64_t seed, float extra, int row_off = 0) {
    ln_bwd_dx_kernel<T><<<(rows + 3) / 4, 256, 0, s>>>((const T*)dy, (const T*)x, g, mean, rstd, (const T*)dres, (T*)dx,
                                                      (T*)dx2, rows, d, p, seed, extra, row_off);
}

}  // namespace mssr
