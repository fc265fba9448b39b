// ContextReps2's pooling (onmt/style_generator_builder.py:243-256) as one kernel family: the scores of the second
// Linear (m -> 1), the -inf mask past each sentence's length, the softmax over time and the weighted sum of the
// states -- forward, backward, and the backward's own backward (the WGAN-GP penalties differentiate the critics'
// input gradient, onmt/trainer_wgan_wh.py:1247-1250, 1315-1318).  The first Linear + ReLU (d -> m) stays a GEMM.
//
//   s_t = h_t . w2 + b2 ,  a = softmax_{t < len}(s) ,  pooled = sum_t a_t x_t
//   backward(dp, q):   e_t = x_t . dp + q_t ,  c = sum_t a_t e_t ,  ds_t = a_t (e_t - c)
//                      dx_t = a_t dp ,  dh_t = ds_t w2 ,  dw2 = sum_t ds_t h_t ,  db2 = sum_t ds_t
//   (q: a gradient arriving at `a` itself -- the second-order pass hands one back, see below)
//   backward of the backward, cotangents X_t (of dx_t), H_t (of dh_t), W (of dw2), beta (of db2):
//                      u_t = H_t . w2 + W . h_t + beta ,  ub = sum_t u_t a_t ,  ce_t = a_t (u_t - ub)
//                      g_a_t = X_t . dp + u_t (e_t - c) - ub e_t          (-> q of the forward's backward)
//                      g_dp = sum_t (a_t X_t + ce_t x_t) ,  g_x_t = ce_t dp ,  g_h_t = ds_t W ,  g_w2 = sum_t ds_t H_t
//
// One block (4 waves) per sentence; a wave takes every 4th time step with its lanes across the features, sums over
// time are combined through LDS.  Sentences are short (<= 18 words on the headline workload, <= 130 on `long`) and the
// whole family moves a few MB: the point is 3 launches where the composition of generic ops took ~45.
#include "common.hpp"

namespace mssr {

constexpr int CP_MAXL = 1024;      // time steps
constexpr int CP_MAXD = 1024;      // state width d (multiple of 4)
constexpr int CP_BLOCK = 256;
constexpr int CP_NC = CP_MAXD / 256;   // 4-wide column groups per lane

template <typename T>
struct CpRow {          // one row of d values spread over a wave: lane holds columns lane*4 + i*256 .. +3
    float v[CP_NC][4];
    __device__ __forceinline__ void load(const T* p, int d, int lane) {
#pragma unroll
        for (int i = 0; i < CP_NC; ++i) {
            const int c0 = lane * 4 + i * 256;
            if (c0 < d) V4<T>::load(p + c0, v[i]);
            else v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
        }
    }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < CP_NC; ++i) v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
    }
    __device__ __forceinline__ float dot(const CpRow& o) const {
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < CP_NC; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) r += v[i][k] * o.v[i][k];
        return r;
    }
    __device__ __forceinline__ void fma(float a, const CpRow& o) {
#pragma unroll
        for (int i = 0; i < CP_NC; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[i][k] += a * o.v[i][k];
    }
    __device__ __forceinline__ void store_scaled(T* p, float a, int d, int lane) const {
#pragma unroll
        for (int i = 0; i < CP_NC; ++i) {
            const int c0 = lane * 4 + i * 256;
            if (c0 < d) {
                const float o[4] = {a * v[i][0], a * v[i][1], a * v[i][2], a * v[i][3]};
                V4<T>::store(p + c0, o);
            }
        }
    }
    __device__ __forceinline__ void to_lds(float* r, int d, int lane) const {
#pragma unroll
        for (int i = 0; i < CP_NC; ++i) {
            const int c0 = lane * 4 + i * 256;
            if (c0 < d) *reinterpret_cast<float4*>(r + c0) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
        }
    }
};

__device__ __forceinline__ int cp_len(const int64_t* lengths, int b, int L) {
    const int64_t n = lengths[b];
    return n < 0 ? 0 : (n > L ? L : (int)n);
}

template <typename T>
__global__ __launch_bounds__(CP_BLOCK) void ctx_pool_fwd_kernel(const T* __restrict__ x, int64_t x_bs, int64_t x_ts,
                                                                const T* __restrict__ h, int64_t h_bs, int64_t h_ts,
                                                                const float* __restrict__ w2, const float* __restrict__ b2,
                                                                const int64_t* __restrict__ lengths, T* __restrict__ pooled,
                                                                float* __restrict__ a_out, int L, int d, int m) {
    __shared__ float s[CP_MAXL];
    __shared__ __align__(16) float red[4][CP_MAXD];
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int len = cp_len(lengths, b, L);
    const T* xb = x + (int64_t)b * x_bs;
    const T* hb = h + (int64_t)b * h_bs;
    const float bias = b2[0];
    for (int t = wave; t < len; t += 4) {
        const T* hr = hb + (int64_t)t * h_ts;
        float acc = 0.f;
        for (int j = lane; j < m; j += 64) acc += to_f(hr[j]) * w2[j];
        acc = wave_sum(acc);
        if (lane == 0) s[t] = acc + bias;
    }
    __syncthreads();
    float mx = -INFINITY;                        // every wave reduces the whole row itself: no second barrier round
    for (int t = lane; t < len; t += 64) mx = fmaxf(mx, s[t]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < len; t += 64) sum += expf(s[t] - mx);
    sum = wave_sum(sum);
    const float inv = len > 0 ? 1.0f / sum : 0.f;
    __syncthreads();
    for (int t = threadIdx.x; t < L; t += CP_BLOCK) {
        const float a = t < len ? expf(s[t] - mx) * inv : 0.f;
        s[t] = a;
        a_out[(int64_t)b * L + t] = a;
    }
    __syncthreads();
    CpRow<T> acc, xr;
    acc.zero();
    for (int t = wave; t < len; t += 4) {
        xr.load(xb + (int64_t)t * x_ts, d, lane);
        acc.fma(s[t], xr);
    }
    acc.to_lds(red[wave], d, lane);
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += CP_BLOCK)
        pooled[(int64_t)b * d + c] = from_f<T>((red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
}

template <typename T>
__global__ __launch_bounds__(CP_BLOCK) void ctx_pool_bwd_kernel(
    const T* __restrict__ dp, const float* __restrict__ q, const T* __restrict__ x, int64_t x_bs, int64_t x_ts,
    const T* __restrict__ h, int64_t h_bs, int64_t h_ts, const float* __restrict__ w2, const float* __restrict__ a,
    const int64_t* __restrict__ lengths, T* __restrict__ dx, int64_t dx_bs, int64_t dx_ts, T* __restrict__ dh,
    int64_t dh_bs, int64_t dh_ts, float* __restrict__ dw2b2, int L, int d, int m) {
    __shared__ float as[CP_MAXL], e[CP_MAXL], ds[CP_MAXL];
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int len = cp_len(lengths, b, L);
    const T* xb = x + (int64_t)b * x_bs;
    const T* hb = h + (int64_t)b * h_bs;
    for (int t = threadIdx.x; t < L; t += CP_BLOCK) as[t] = t < len ? a[(int64_t)b * L + t] : 0.f;
    CpRow<T> dpv, xr;
    if (dp) dpv.load(dp + (int64_t)b * d, d, lane);
    else dpv.zero();
    for (int t = wave; t < len; t += 4) {
        float acc = 0.f;
        if (dp) {
            xr.load(xb + (int64_t)t * x_ts, d, lane);
            acc = wave_sum(xr.dot(dpv));
        }
        if (lane == 0) e[t] = acc + (q ? q[(int64_t)b * L + t] : 0.f);
    }
    __syncthreads();
    float c = 0.f;
    for (int t = lane; t < len; t += 64) c += as[t] * e[t];
    c = wave_sum(c);
    for (int t = threadIdx.x; t < L; t += CP_BLOCK) ds[t] = t < len ? as[t] * (e[t] - c) : 0.f;
    __syncthreads();
    for (int t = wave; t < L; t += 4) {
        if (dx) dpv.store_scaled(dx + (int64_t)b * dx_bs + (int64_t)t * dx_ts, as[t], d, lane);
        const float dsv = ds[t];
        T* dhr = dh + (int64_t)b * dh_bs + (int64_t)t * dh_ts;
        for (int j = lane; j < m; j += 64) dhr[j] = from_f<T>(dsv * w2[j]);
    }
    for (int j = threadIdx.x; j < m; j += CP_BLOCK) {
        float acc = 0.f;
        for (int t = 0; t < len; ++t) acc += ds[t] * to_f(hb[(int64_t)t * h_ts + j]);
        atomicAdd(dw2b2 + j, acc);
    }
    if (wave == 3) {                              // wave 3: the threads least likely to hold one of the m columns
        float acc = 0.f;
        for (int t = lane; t < len; t += 64) acc += ds[t];
        acc = wave_sum(acc);
        if (lane == 0) atomicAdd(dw2b2 + m, acc);
    }
}

template <typename T>
__global__ __launch_bounds__(CP_BLOCK) void ctx_pool_bwd_bwd_kernel(
    const T* __restrict__ ddx, int64_t ddx_bs, int64_t ddx_ts, const T* __restrict__ ddh, int64_t ddh_bs, int64_t ddh_ts,
    const float* __restrict__ ddw2, const float* __restrict__ ddb2, const T* __restrict__ dp, const float* __restrict__ q,
    const T* __restrict__ x, int64_t x_bs, int64_t x_ts, const T* __restrict__ h, int64_t h_bs, int64_t h_ts,
    const float* __restrict__ w2, const float* __restrict__ a, const int64_t* __restrict__ lengths, T* __restrict__ g_dp,
    T* __restrict__ g_x, int64_t gx_bs, int64_t gx_ts, T* __restrict__ g_h, int64_t gh_bs, int64_t gh_ts,
    float* __restrict__ g_w2, float* __restrict__ g_a, int L, int d, int m) {
    __shared__ float as[CP_MAXL], e[CP_MAXL], p[CP_MAXL], u[CP_MAXL];
    __shared__ __align__(16) float red[4][CP_MAXD];
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int len = cp_len(lengths, b, L);
    const T* xb = x + (int64_t)b * x_bs;
    const T* hb = h + (int64_t)b * h_bs;
    for (int t = threadIdx.x; t < L; t += CP_BLOCK) as[t] = t < len ? a[(int64_t)b * L + t] : 0.f;
    CpRow<T> dpv, xr, acc;
    if (dp) dpv.load(dp + (int64_t)b * d, d, lane);
    else dpv.zero();
    const float beta = ddb2 ? ddb2[0] : 0.f;
    for (int t = wave; t < len; t += 4) {
        float ev = 0.f, pv = 0.f, uv = 0.f;
        if (dp) {
            xr.load(xb + (int64_t)t * x_ts, d, lane);
            ev = xr.dot(dpv);
            if (ddx) {
                xr.load(ddx + (int64_t)b * ddx_bs + (int64_t)t * ddx_ts, d, lane);
                pv = xr.dot(dpv);
            }
        }
        const T* hr = hb + (int64_t)t * h_ts;
        const T* Hr = ddh ? ddh + (int64_t)b * ddh_bs + (int64_t)t * ddh_ts : nullptr;
        for (int j = lane; j < m; j += 64) {
            if (Hr) uv += to_f(Hr[j]) * w2[j];
            if (ddw2) uv += ddw2[j] * to_f(hr[j]);
        }
        ev = wave_sum(ev);
        pv = wave_sum(pv);
        uv = wave_sum(uv);
        if (lane == 0) {
            e[t] = ev + (q ? q[(int64_t)b * L + t] : 0.f);
            p[t] = pv;
            u[t] = uv + beta;
        }
    }
    __syncthreads();
    float c = 0.f, ub = 0.f;
    for (int t = lane; t < len; t += 64) {
        c += as[t] * e[t];
        ub += as[t] * u[t];
    }
    c = wave_sum(c);
    ub = wave_sum(ub);
    __syncthreads();                             // every wave has read e / u before they are overwritten
    for (int t = threadIdx.x; t < L; t += CP_BLOCK) {
        float ga = 0.f, dsv = 0.f, ce = 0.f;
        if (t < len) {
            const float at = as[t], et = e[t], ut = u[t];
            ga = p[t] + ut * (et - c) - ub * et;
            dsv = at * (et - c);
            ce = at * (ut - ub);
        }
        g_a[(int64_t)b * L + t] = ga;
        e[t] = dsv;                              // e <- ds, u <- ce from here on
        u[t] = ce;
    }
    __syncthreads();
    acc.zero();
    for (int t = wave; t < L; t += 4) {
        const float ce = u[t];
        dpv.store_scaled(g_x + (int64_t)b * gx_bs + (int64_t)t * gx_ts, ce, d, lane);
        if (t < len) {
            xr.load(xb + (int64_t)t * x_ts, d, lane);
            acc.fma(ce, xr);
            if (ddx) {
                xr.load(ddx + (int64_t)b * ddx_bs + (int64_t)t * ddx_ts, d, lane);
                acc.fma(as[t], xr);
            }
        }
        if (g_h) {
            const float dsv = e[t];
            T* ghr = g_h + (int64_t)b * gh_bs + (int64_t)t * gh_ts;
            for (int j = lane; j < m; j += 64) ghr[j] = from_f<T>(ddw2 ? dsv * ddw2[j] : 0.f);
        }
    }
    acc.to_lds(red[wave], d, lane);
    __syncthreads();
    for (int cc = threadIdx.x; cc < d; cc += CP_BLOCK)
        g_dp[(int64_t)b * d + cc] = from_f<T>((red[0][cc] + red[1][cc]) + (red[2][cc] + red[3][cc]));
    if (g_w2 && ddh) {
        for (int j = threadIdx.x; j < m; j += CP_BLOCK) {
            float s = 0.f;
            for (int t = 0; t < len; ++t) s += e[t] * to_f(ddh[(int64_t)b * ddh_bs + (int64_t)t * ddh_ts + j]);
            atomicAdd(g_w2 + j, s);
        }
    }
}

static bool cp_rows_ok(const void* p, int64_t bs, int64_t ts, int dtype) {
    const int v = 4;                             // V4 accesses: 16 B (f32) / 8 B (bf16)
    const size_t al = dtype == MSSR_F32 ? 16 : 8;
    return !p || ((reinterpret_cast<uintptr_t>(p) % al) == 0 && bs % v == 0 && ts % v == 0);
}

static int cp_check(const char* who, int B, int L, int d, int m, int dtype) {
    MSSR_REQUIRE(dtype == MSSR_F32 || dtype == MSSR_BF16, "%s: bad dtype", who);
    MSSR_REQUIRE(B >= 0 && L >= 1 && L <= CP_MAXL && d >= 4 && d <= CP_MAXD && d % 4 == 0 && m >= 1 && m <= 65536,
                 "%s: unsupported shape B=%d L=%d d=%d m=%d (L <= %d, d <= %d and a multiple of 4)", who, B, L, d, m,
                 CP_MAXL, CP_MAXD);
    return MSSR_OK;
}

}  // namespace mssr

using namespace mssr;

extern "C" int mssr_ctx_pool_fwd(const void* x, int64_t x_bs, int64_t x_ts, const void* h, int64_t h_bs, int64_t h_ts,
                                 const float* w2, const float* b2, const int64_t* lengths, void* pooled, float* a, int B,
                                 int L, int d, int m, int dtype, void* stream) {
    if (int e = cp_check("ctx_pool_fwd", B, L, d, m, dtype)) return e;
    if (B == 0) return MSSR_OK;
    MSSR_REQUIRE(x && h && w2 && b2 && lengths && pooled && a, "ctx_pool_fwd: null pointer");
    MSSR_REQUIRE(cp_rows_ok(x, x_bs, x_ts, dtype) && cp_rows_ok(pooled, d, d, dtype),
                 "ctx_pool_fwd: state rows must start on 4-element boundaries");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSSR_F32)
        klaunch(ctx_pool_fwd_kernel<float>, dim3(B), dim3(CP_BLOCK), 0, s, (const float*)x, x_bs, x_ts, (const float*)h,
                h_bs, h_ts, w2, b2, lengths, (float*)pooled, a, L, d, m);
    else
        klaunch(ctx_pool_fwd_kernel<bf16>, dim3(B), dim3(CP_BLOCK), 0, s, (const bf16*)x, x_bs, x_ts, (const bf16*)h,
                h_bs, h_ts, w2, b2, lengths, (bf16*)pooled, a, L, d, m);
    MSSR_LAUNCH_OK("ctx_pool_fwd");
}

extern "C" int mssr_ctx_pool_bwd(const void* dp, const float* q, const void* x, int64_t x_bs, int64_t x_ts, const void* h,
                                 int64_t h_bs, int64_t h_ts, const float* w2, const float* a, const int64_t* lengths,
                                 void* dx, int64_t dx_bs, int64_t dx_ts, void* dh, int64_t dh_bs, int64_t dh_ts,
                                 float* dw2b2, int B, int L, int d, int m, int dtype, void* stream) {
    if (int e = cp_check("ctx_pool_bwd", B, L, d, m, dtype)) return e;
    if (B == 0) return MSSR_OK;
    MSSR_REQUIRE(x && h && w2 && a && lengths && dh && dw2b2 && (dp || q), "ctx_pool_bwd: null pointer");
    MSSR_REQUIRE(!dx || dp, "ctx_pool_bwd: dx needs dp");
    MSSR_REQUIRE(cp_rows_ok(x, x_bs, x_ts, dtype) && cp_rows_ok(dx, dx_bs, dx_ts, dtype) && cp_rows_ok(dp, d, d, dtype),
                 "ctx_pool_bwd: state rows must start on 4-element boundaries");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSSR_F32)
        klaunch(ctx_pool_bwd_kernel<float>, dim3(B), dim3(CP_BLOCK), 0, s, (const float*)dp, q, (const float*)x, x_bs, x_ts,
                (const float*)h, h_bs, h_ts, w2, a, lengths, (float*)dx, dx_bs, dx_ts, (float*)dh, dh_bs, dh_ts, dw2b2, L, d,
                m);
    else
        klaunch(ctx_pool_bwd_kernel<bf16>, dim3(B), dim3(CP_BLOCK), 0, s, (const bf16*)dp, q, (const bf16*)x, x_bs, x_ts,
                (const bf16*)h, h_bs, h_ts, w2, a, lengths, (bf16*)dx, dx_bs, dx_ts, (bf16*)dh, dh_bs, dh_ts, dw2b2, L, d, m);
    MSSR_LAUNCH_OK("ctx_pool_bwd");
}

extern "C" int mssr_ctx_pool_bwd_bwd(const void* ddx, int64_t ddx_bs, int64_t ddx_ts, const void* ddh, int64_t ddh_bs,
                                     int64_t ddh_ts, const float* ddw2, const float* ddb2, const void* dp, const float* q,
                                     const void* x, int64_t x_bs, int64_t x_ts, const void* h, int64_t h_bs, int64_t h_ts,
                                     const float* w2, const float* a, const int64_t* lengths, void* g_dp, void* g_x,
                                     int64_t gx_bs, int64_t gx_ts, void* g_h, int64_t gh_bs, int64_t gh_ts, float* g_w2,
                                     float* g_a, int B, int L, int d, int m, int dtype, void* stream) {
    if (int e = cp_check("ctx_pool_bwd_bwd", B, L, d, m, dtype)) return e;
    if (B == 0) return MSSR_OK;
    MSSR_REQUIRE(x && h && w2 && a && lengths && g_dp && g_x && g_a, "ctx_pool_bwd_bwd: null pointer");
    MSSR_REQUIRE(!ddh || g_w2, "ctx_pool_bwd_bwd: a cotangent of dh needs the g_w2 output");
    MSSR_REQUIRE(!ddw2 || g_h, "ctx_pool_bwd_bwd: a cotangent of dw2 needs the g_h output");
    MSSR_REQUIRE(cp_rows_ok(x, x_bs, x_ts, dtype) && cp_rows_ok(ddx, ddx_bs, ddx_ts, dtype) &&
                     cp_rows_ok(g_x, gx_bs, gx_ts, dtype) && cp_rows_ok(dp, d, d, dtype) && cp_rows_ok(g_dp, d, d, dtype),
                 "ctx_pool_bwd_bwd: state rows must start on 4-element boundaries");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MSSR_F32)
        klaunch(ctx_pool_bwd_bwd_kernel<float>, dim3(B), dim3(CP_BLOCK), 0, s, (const float*)ddx, ddx_bs, ddx_ts,
                (const float*)ddh, ddh_bs, ddh_ts, ddw2, ddb2, (const float*)dp, q, (const float*)x, x_bs, x_ts,
                (const float*)h, h_bs, h_ts, w2, a, lengths, (float*)g_dp, (float*)g_x, gx_bs, gx_ts, (float*)g_h, gh_bs,
                gh_ts, g_w2, g_a, L, d, m);
    else
        klaunch(ctx_pool_bwd_bwd_kernel<bf16>, dim3(B), dim3(CP_BLOCK), 0, s, (const bf16*)ddx, ddx_bs, ddx_ts,
                (const bf16*)ddh, ddh_bs, ddh_ts, ddw2, ddb2, (const bf16*)dp, q, (const bf16*)x, x_bs, x_ts,
                (const bf16*)h, h_bs, h_ts, w2, a, lengths, (bf16*)g_dp, (bf16*)g_x, gx_bs, gx_ts, (bf16*)g_h, gh_bs, gh_ts,
                g_w2, g_a, L, d, m);
    MSSR_LAUNCH_OK("ctx_pool_bwd_bwd");
}
