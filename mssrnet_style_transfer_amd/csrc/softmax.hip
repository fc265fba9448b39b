// Row softmax / log-softmax: forward, backward and the double-backward of the softmax.
// fp32 arithmetic whatever the storage type.  A row lives in the registers of one wave64
// (n <= 2048) or of one 256-thread block (n <= 16384, e.g. the 9606-word vocabulary), so each
// element is read once and written once: algorithmic traffic 2*rows*n elements forward,
// 3*rows*n backward.
#include "common.hpp"

namespace mssr {

constexpr float MASK_FILL = -1e18f;   // onmt/modules/multi_headed_attn.py:199

template <int G>
__device__ __forceinline__ float group_sum(float v, float* sm) {
    v = wave_sum(v);
    if (G > 64) {
        const int w = threadIdx.x >> 6;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm[w] = v;
        __syncthreads();
        v = 0.f;
#pragma unroll
        for (int i = 0; i < G / 64; ++i) v += sm[i];
    }
    return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v, float* sm) {
    v = wave_max(v);
    if (G > 64) {
        const int w = threadIdx.x >> 6;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm[w] = v;
        __syncthreads();
        v = sm[0];
#pragma unroll
        for (int i = 1; i < G / 64; ++i) v = fmaxf(v, sm[i]);
    }
    return v;
}

// G threads per row (64: one wave, rows = blockDim/64 per block; 256: one block per row)
template <typename TI, typename TO, int G, int NE>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const TI* __restrict__ x, TO* __restrict__ y,
                                                          const uint8_t* __restrict__ mask, int64_t rows, int n,
                                                          int64_t ldx, int64_t ldy, int64_t rows_per_batch,
                                                          int64_t q_len, int64_t mask_bs, int64_t mask_qs, float scale,
                                                          int log_mode) {
    __shared__ float sm[4];
    const int t = threadIdx.x % G;
    const int64_t row = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    const bool live = row < rows;      // G == 256: always true;  G == 64: whole wave uniform
    if (G == 64 && !live) return;
    const TI* xr = x + row * ldx;
    const uint8_t* mr = mask ? mask + (row / rows_per_batch) * mask_bs + (row % q_len) * mask_qs : nullptr;
    float v[NE];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = t + i * G;
        if (c < n) {
            float s = to_f(xr[c]) * scale;
            if (mr && mr[c]) s = MASK_FILL;
            v[i] = s;
            mx = fmaxf(mx, s);
        } else {
            v[i] = -INFINITY;
        }
    }
    mx = group_max<G>(mx, sm);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const float e = (t + i * G) < n ? expf(v[i] - mx) : 0.f;
        sum += e;
        if (!log_mode) v[i] = e;
    }
    sum = group_sum<G>(sum, sm);
    TO* yr = y + row * ldy;
    if (log_mode) {
        const float lse = mx + logf(sum);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = t + i * G;
            if (c < n) yr[c] = from_f<TO>(v[i] - lse);
        }
    } else {
        const float inv = 1.0f / sum;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = t + i * G;
            if (c < n) yr[c] = from_f<TO>(v[i] * inv);
        }
    }
}

template <typename TY, typename TD, typename TX, int G, int NE>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const TY* __restrict__ y, const TD* __restrict__ dy,
                                                          TX* __restrict__ dx, int64_t rows, int n, int64_t ldy,
                                                          int64_t lddy, int64_t lddx, float scale, int log_mode) {
    __shared__ float sm[4];
    const int t = threadIdx.x % G;
    const int64_t row = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    if (G == 64 && row >= rows) return;
    float yv[NE], dv[NE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = t + i * G;
        if (c < n) {
            yv[i] = to_f(y[row * ldy + c]);
            dv[i] = to_f(dy[row * lddy + c]);
            s += log_mode ? dv[i] : yv[i] * dv[i];
        } else {
            yv[i] = dv[i] = 0.f;
        }
    }
    s = group_sum<G>(s, sm);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = t + i * G;
        if (c < n) {
            const float o = log_mode ? dv[i] - expf(yv[i]) * s : yv[i] * (dv[i] - s);
            dx[row * lddx + c] = from_f<TX>(o * scale);
        }
    }
}

// Vocabulary-wide rows, few of them (the decode backward: 160 rows x 9606 per step): one 1024-thread block per row --
// sixteen waves keep a CU's memory pipe busy where the four-wave form left three quarters of the chip idle
// (160 x 9606 fp32 probabilities and gradients -> bf16: 22 -> 9 us).
template <typename TY, typename TD, typename TX, int NE>
__global__ __launch_bounds__(1024) void softmax_bwd_wide_kernel(const TY* __restrict__ y, const TD* __restrict__ dy,
                                                                TX* __restrict__ dx, int n, int64_t ldy, int64_t lddy,
                                                                int64_t lddx, float scale, int log_mode) {
    __shared__ float sm[16];
    const int t = threadIdx.x;
    const int64_t row = blockIdx.x;
    float yv[NE], dv[NE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = t + i * 1024;
        if (c < n) {
            yv[i] = to_f(y[row * ldy + c]);
            dv[i] = to_f(dy[row * lddy + c]);
            s += log_mode ? dv[i] : yv[i] * dv[i];
        } else {
            yv[i] = dv[i] = 0.f;
        }
    }
    s = wave_sum(s);
    if ((t & 63) == 0) sm[t >> 6] = s;
    __syncthreads();
    s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += sm[w];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = t + i * 1024;
        if (c < n) {
            const float o = log_mode ? dv[i] - expf(yv[i]) * s : yv[i] * (dv[i] - s);
            dx[row * lddx + c] = from_f<TX>(o * scale);
        }
    }
}

// g_y = scale * ( g*(dy - sum(y*dy)) - dy*sum(g*y) )
template <typename T, int G, int NE>
__global__ __launch_bounds__(256) void softmax_bwd_bwd_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                              const T* __restrict__ dy, T* __restrict__ gy,
                                                              int64_t rows, int n, int64_t ld, float scale) {
    __shared__ float sm[4];
    const int t = threadIdx.x % G;
    const int64_t row = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    if (G == 64 && row >= rows) return;
    float gv[NE], yv[NE], dv[NE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = t + i * G;
        if (c < n) {
            gv[i] = to_f(g[row * ld + c]);
            yv[i] = to_f(y[row * ld + c]);
            dv[i] = to_f(dy[row * ld + c]);
            s1 += yv[i] * dv[i];
            s2 += gv[i] * yv[i];
        } else {
            gv[i] = yv[i] = dv[i] = 0.f;
        }
    }
    s1 = group_sum<G>(s1, sm);
    s2 = group_sum<G>(s2, sm);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = t + i * G;
        if (c < n) gy[row * ld + c] = from_f<T>(scale * (gv[i] * (dv[i] - s1) - dv[i] * s2));
    }
}

// pick (G, NE) for a row length
struct RowCfg { int g, ne; };
static inline bool row_cfg(int n, RowCfg& c) {
    if (n <= 64) c = {64, 1};
    else if (n <= 128) c = {64, 2};
    else if (n <= 256) c = {64, 4};
    else if (n <= 512) c = {64, 8};
    else if (n <= 1024) c = {64, 16};
    else if (n <= 2048) c = {64, 32};
    else if (n <= 4096) c = {256, 16};
    else if (n <= 8192) c = {256, 32};
    else if (n <= 16384) c = {256, 64};
    else return false;
    return true;
}

template <int A, int B> struct GN { static constexpr int G = A, NE = B; };
template <typename F> static void row_dispatch(const RowCfg& c, F&& f) {
    if (c.g == 64) {
        switch (c.ne) {
            case 1: f(GN<64, 1>{}); break;
            case 2: f(GN<64, 2>{}); break;
            case 4: f(GN<64, 4>{}); break;
            case 8: f(GN<64, 8>{}); break;
            case 16: f(GN<64, 16>{}); break;
            default: f(GN<64, 32>{}); break;
        }
    } else {
        switch (c.ne) {
            case 16: f(GN<256, 16>{}); break;
            case 32: f(GN<256, 32>{}); break;
            default: f(GN<256, 64>{}); break;
        }
    }
}

}  // namespace mssr

using namespace mssr;

extern "C" int mssr_softmax_fwd(const void* x, void* y, const uint8_t* mask, int64_t rows, int n, int64_t ldx,
                                int64_t ldy, int64_t rows_per_batch, int64_t q_len, int64_t mask_bs, int64_t mask_qs,
                                float scale, int log_mode, int in_dtype, int out_dtype, void* stream) {
    MSSR_REQUIRE(valid_dtype(in_dtype) && valid_dtype(out_dtype), "softmax_fwd: bad dtype");
    MSSR_REQUIRE(rows >= 0 && n > 0 && ldx >= n && ldy >= n, "softmax_fwd: bad shape");
    RowCfg cfg;
    MSSR_REQUIRE(row_cfg(n, cfg), "softmax_fwd: n=%d exceeds 16384", n);
    if (rows == 0) return MSSR_OK;
    MSSR_REQUIRE(x && y, "softmax_fwd: null pointer");
    MSSR_REQUIRE(!mask || (rows_per_batch > 0 && q_len > 0), "softmax_fwd: mask needs rows_per_batch and q_len");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + (256 / cfg.g) - 1) / (256 / cfg.g)));
#define SM_FWD(TI, TO)                                                                                  \
    row_dispatch(cfg, [&](auto gn) {                                                                    \
        softmax_fwd_kernel<TI, TO, decltype(gn)::G, decltype(gn)::NE><<<grid, dim3(256), 0, s>>>(       \
            (const TI*)x, (TO*)y, mask, rows, n, ldx, ldy, rows_per_batch, q_len, mask_bs, mask_qs, scale, \
            log_mode);                                                                                  \
    })
    if (in_dtype == MSSR_F32 && out_dtype == MSSR_F32) { SM_FWD(float, float); }
    else if (in_dtype == MSSR_BF16 && out_dtype == MSSR_BF16) { SM_FWD(bf16, bf16); }
    else if (in_dtype == MSSR_BF16 && out_dtype == MSSR_F32) { SM_FWD(bf16, float); }
    else { SM_FWD(float, bf16); }
#undef SM_FWD
    MSSR_LAUNCH_OK("softmax_fwd");
}

extern "C" int mssr_softmax_bwd(const void* y, const void* dy, void* dx, int64_t rows, int n, int64_t ldy,
                                int64_t lddy, int64_t lddx, float scale, int log_mode, int y_dtype, int dy_dtype,
                                int dx_dtype, void* stream) {
    MSSR_REQUIRE(valid_dtype(y_dtype) && valid_dtype(dy_dtype) && valid_dtype(dx_dtype), "softmax_bwd: bad dtype");
    MSSR_REQUIRE(rows >= 0 && n > 0 && ldy >= n && lddy >= n && lddx >= n, "softmax_bwd: bad shape");
    RowCfg cfg;
    MSSR_REQUIRE(row_cfg(n, cfg), "softmax_bwd: n=%d exceeds 16384", n);
    if (rows == 0) return MSSR_OK;
    MSSR_REQUIRE(y && dy && dx, "softmax_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + (256 / cfg.g) - 1) / (256 / cfg.g)));
    const bool wide = n > 4096 && rows <= 2048;     // few vocabulary-wide rows: one 1024-thread block per row
#define SM_BWD_WIDE(TY, TD, TX, NEV)                                                                    \
    softmax_bwd_wide_kernel<TY, TD, TX, NEV><<<dim3((unsigned)rows), dim3(1024), 0, s>>>(              \
        (const TY*)y, (const TD*)dy, (TX*)dx, n, ldy, lddy, lddx, scale, log_mode)
#define SM_BWD(TY, TD, TX)                                                                              \
    if (wide) {                                                                                         \
        if (n <= 8192) SM_BWD_WIDE(TY, TD, TX, 8);                                                      \
        else if (n <= 10240) SM_BWD_WIDE(TY, TD, TX, 10);                                               \
        else SM_BWD_WIDE(TY, TD, TX, 16);                                                               \
    } else                                                                                              \
    row_dispatch(cfg, [&](auto gn) {                                                                    \
        softmax_bwd_kernel<TY, TD, TX, decltype(gn)::G, decltype(gn)::NE><<<grid, dim3(256), 0, s>>>(   \
            (const TY*)y, (const TD*)dy, (TX*)dx, rows, n, ldy, lddy, lddx, scale, log_mode);           \
    })
    const int key = y_dtype * 4 + dy_dtype * 2 + dx_dtype;
    switch (key) {
        case 0: SM_BWD(float, float, float); break;
        case 7: SM_BWD(bf16, bf16, bf16); break;
        case 2: SM_BWD(float, bf16, float); break;   // f32 probs, bf16 upstream grad
        case 1: SM_BWD(float, float, bf16); break;
        case 3: SM_BWD(float, bf16, bf16); break;
        case 4: SM_BWD(bf16, float, float); break;
        case 5: SM_BWD(bf16, float, bf16); break;
        default: SM_BWD(bf16, bf16, float); break;
    }
#undef SM_BWD
#undef SM_BWD_WIDE
    MSSR_LAUNCH_OK("softmax_bwd");
}

extern "C" int mssr_softmax_bwd_bwd(const void* g, const void* y, const void* dy, void* g_y, int64_t rows, int n,
                                    int64_t ld, float scale, int dtype, void* stream) {
    MSSR_REQUIRE(valid_dtype(dtype), "softmax_bwd_bwd: bad dtype");
    MSSR_REQUIRE(rows >= 0 && n > 0 && ld >= n, "softmax_bwd_bwd: bad shape");
    RowCfg cfg;
    MSSR_REQUIRE(row_cfg(n, cfg), "softmax_bwd_bwd: n=%d exceeds 16384", n);
    if (rows == 0) return MSSR_OK;
    MSSR_REQUIRE(g && y && dy && g_y, "softmax_bwd_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + (256 / cfg.g) - 1) / (256 / cfg.g)));
    row_dispatch(cfg, [&](auto gn) {
        constexpr int G = decltype(gn)::G, NE = decltype(gn)::NE;
        if (dtype == MSSR_F32)
            softmax_bwd_bwd_kernel<float, G, NE><<<grid, dim3(256), 0, s>>>(
                (const float*)g, (const float*)y, (const float*)dy, (float*)g_y, rows, n, ld, scale);
        else
            softmax_bwd_bwd_kernel<bf16, G, NE><<<grid, dim3(256), 0, s>>>(
                (const bf16*)g, (const bf16*)y, (const bf16*)dy, (bf16*)g_y, rows, n, ld, scale);
    });
    MSSR_LAUNCH_OK("softmax_bwd_bwd");
}
