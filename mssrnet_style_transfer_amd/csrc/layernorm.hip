// LayerNorm forward / backward / double-backward for gfx950.
// One wave64 per row, the row held in registers (4-element vectors per lane, coalesced 16-B /
// 8-B accesses), two-pass mean/variance by wave shuffles.  HBM-bound by design:
// fwd moves 2*rows*d elements, bwd 3*rows*d, bwd_bwd 5*rows*d (+ O(d) parameters).
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>

namespace mssr {

constexpr int LN_BLOCK = 256;
constexpr int LN_WPB = LN_BLOCK / WAVE;   // rows (waves) per block
constexpr int LN_MAX_PARTS = 256;

template <typename T, int NV>
__global__ __launch_bounds__(LN_BLOCK) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ y,
                                                          float* __restrict__ mean, float* __restrict__ rstd,
                                                          int64_t rows, int d, float eps) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * LN_WPB + wave;
    if (row >= rows) return;
    const T* xr = x + row * d;
    float v[NV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * WAVE + lane) * 4;
        if (c < d) {
            V4<T>::load(xr + c, v[i]);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        } else {
            v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
        }
    }
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * WAVE + lane) * 4;
        if (c < d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = v[i][j] - mu;
                q += t * t;
            }
        }
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
    T* yr = y + row * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * WAVE + lane) * 4;
        if (c < d) {
            float g[4], b[4], o[4];
            V4<float>::load(gamma + c, g);
            V4<float>::load(beta + c, b);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mu) * rs * g[j] + b[j];
            V4<T>::store(yr + c, o);
        }
    }
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
}

// dx = r*(a - mean(a) - xhat*mean(a*xhat)), a = gamma*dy ; dgamma = sum_rows dy*xhat ; dbeta = sum_rows dy
template <typename T, int NV>
__global__ __launch_bounds__(LN_BLOCK) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, T* __restrict__ dx,
                                                          float* __restrict__ partial, int64_t rows, int d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2*d]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * d; i += LN_BLOCK) smem[i] = 0.f;
    float ag[NV][4], ab[NV][4], g[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * WAVE + lane) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) ag[i][j] = ab[i][j] = g[i][j] = 0.f;
        if (c < d) V4<float>::load(gamma + c, g[i]);
    }
    const int64_t stride = (int64_t)gridDim.x * LN_WPB;
    for (int64_t row = (int64_t)blockIdx.x * LN_WPB + wave; row < rows; row += stride) {
        const float mu = mean[row], rs = rstd[row];
        float xh[NV][4], a[NV][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            if (c < d) {
                float xv[4], dv[4];
                V4<T>::load(x + row * d + c, xv);
                V4<T>::load(dy + row * d + c, dv);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[i][j] = (xv[j] - mu) * rs;
                    a[i][j] = dv[j] * g[i][j];
                    s1 += a[i][j];
                    s2 += a[i][j] * xh[i][j];
                    ag[i][j] += dv[j] * xh[i][j];
                    ab[i][j] += dv[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) xh[i][j] = a[i][j] = 0.f;
            }
        }
        const float c1 = wave_sum(s1) / (float)d, c2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            if (c < d) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rs * (a[i][j] - c1 - xh[i][j] * c2);
                V4<T>::store(dx + row * d + c, o);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * WAVE + lane) * 4;
        if (c < d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                atomicAdd(&smem[c + j], ag[i][j]);
                atomicAdd(&smem[d + c + j], ab[i][j]);
            }
        }
    }
    __syncthreads();
    float* out = partial + (int64_t)blockIdx.x * 2 * d;
    for (int i = threadIdx.x; i < 2 * d; i += LN_BLOCK) out[i] = smem[i];
}

// LayerNorm PARAMETER gradients alone, batched over layers: out[z][0][c] = sum_rows dy * xhat, out[z][1][c] = sum_rows dy
// for slab z = blockIdx.z of every operand (element strides *_bs).  The engines need no dx here (their backward sweeps
// produced it) and call this once per LayerNorm KIND over all layers instead of ln_bwd_kernel + ln_finalize_kernel
// once per LayerNorm (37 x 2 launches per decoder backward).  A lane owns one 16-byte column group, a wave one row per
// instruction; partial sums meet in LDS and leave as float atomics (out zeroed by the host call).
template <typename T>
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ out, int64_t dy_bs, int64_t x_bs,
                                                            int64_t stat_bs, int64_t out_bs, int64_t rows, int d) {
    constexpr int V = 16 / (int)sizeof(T);
    __shared__ float sm[4][2][64 * V];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    dy += (int64_t)blockIdx.z * dy_bs;
    x += (int64_t)blockIdx.z * x_bs;
    mean += (int64_t)blockIdx.z * stat_bs;
    rstd += (int64_t)blockIdx.z * stat_bs;
    out += (int64_t)blockIdx.z * out_bs;
    const int64_t stride = (int64_t)gridDim.y * 4;
    for (int c0 = blockIdx.x * 64 * V; c0 < d; c0 += gridDim.x * 64 * V) {
        const int c = c0 + lane * V;
        float ag[V], ab[V];
#pragma unroll
        for (int u = 0; u < V; ++u) ag[u] = ab[u] = 0.f;
        if (c < d) {
            int64_t r = (int64_t)blockIdx.y * 4 + wave;
            for (; r + stride < rows; r += 2 * stride) {
                const uint4 dv0 = *reinterpret_cast<const uint4*>(dy + r * d + c);
                const uint4 xv0 = *reinterpret_cast<const uint4*>(x + r * d + c);
                const uint4 dv1 = *reinterpret_cast<const uint4*>(dy + (r + stride) * d + c);
                const uint4 xv1 = *reinterpret_cast<const uint4*>(x + (r + stride) * d + c);
                const float mu0 = mean[r], rs0 = rstd[r], mu1 = mean[r + stride], rs1 = rstd[r + stride];
                const T *de0 = reinterpret_cast<const T*>(&dv0), *xe0 = reinterpret_cast<const T*>(&xv0);
                const T *de1 = reinterpret_cast<const T*>(&dv1), *xe1 = reinterpret_cast<const T*>(&xv1);
#pragma unroll
                for (int u = 0; u < V; ++u) {
                    const float g0 = to_f(de0[u]), g1 = to_f(de1[u]);
                    ag[u] += g0 * ((to_f(xe0[u]) - mu0) * rs0) + g1 * ((to_f(xe1[u]) - mu1) * rs1);
                    ab[u] += g0 + g1;
                }
            }
            for (; r < rows; r += stride) {
                const uint4 dv = *reinterpret_cast<const uint4*>(dy + r * d + c);
                const uint4 xv = *reinterpret_cast<const uint4*>(x + r * d + c);
                const float mu = mean[r], rs = rstd[r];
                const T *de = reinterpret_cast<const T*>(&dv), *xe = reinterpret_cast<const T*>(&xv);
#pragma unroll
                for (int u = 0; u < V; ++u) {
                    const float g = to_f(de[u]);
                    ag[u] += g * ((to_f(xe[u]) - mu) * rs);
                    ab[u] += g;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < V; ++u) {
            sm[wave][0][lane * V + u] = ag[u];
            sm[wave][1][lane * V + u] = ab[u];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * 64 * V; i += 256) {
            const int which = i / (64 * V), j = i - which * 64 * V;
            if (c0 + j < d)
                atomicAdd(out + (int64_t)which * d + c0 + j,
                          (sm[0][which][j] + sm[1][which][j]) + (sm[2][which][j] + sm[3][which][j]));
        }
    }
}

// sums `parts` rows of width `w` and scatters columns [0,d) to out0, [d,2d) to out1.
// 16 columns x 16 part-groups per block: every thread has its (<= 16) loads in flight at once, so the
// reduction is one memory round trip instead of a `parts`-long dependent chain.
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out0,
                                                          float* __restrict__ out1, int parts, int w, int d) {
    __shared__ float sm[16][17];
    const int cx = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < w) {
        int p = g;
        for (; p + 48 < parts; p += 64) {
            s0 += partial[(int64_t)p * w + c];
            s1 += partial[(int64_t)(p + 16) * w + c];
            s2 += partial[(int64_t)(p + 32) * w + c];
            s3 += partial[(int64_t)(p + 48) * w + c];
        }
        for (; p < parts; p += 16) s0 += partial[(int64_t)p * w + c];
    }
    sm[g][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && c < w) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += sm[i][cx];
        if (c < d) out0[c] = t;
        else out1[c - d] = t;
    }
}

// Backward of ln_bwd: see include/mssr_hip.h.  With xhat, a = gamma*dy, c1 = mean(a),
// c2 = mean(a*xhat), dxv = r(a - c1 - xhat c2), u = ggx, q = r(u - mean(u) - xhat mean(u xhat)):
//   g_dy    = gamma*q + ggg*xhat + ggb
//   g_gamma = sum_rows dy*q
//   g_x     = -r [ xhat*mean(u*dxv) + c2*q + mean(u*xhat)*dxv ] + r (w - mean(w) - xhat mean(w xhat)),
//             w = ggg*dy
template <typename T, int NV>
__global__ __launch_bounds__(LN_BLOCK) void ln_bwd_bwd_kernel(
    const T* __restrict__ ggx, const float* __restrict__ ggg, const float* __restrict__ ggb,
    const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ g_dy,
    T* __restrict__ g_x, float* __restrict__ partial, int64_t rows, int d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [d]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < d; i += LN_BLOCK) smem[i] = 0.f;
    float acc[NV][4], g[NV][4], vg[NV][4], vb[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * WAVE + lane) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = g[i][j] = vg[i][j] = vb[i][j] = 0.f;
        if (c < d) {
            V4<float>::load(gamma + c, g[i]);
            if (ggg) V4<float>::load(ggg + c, vg[i]);
            if (ggb) V4<float>::load(ggb + c, vb[i]);
        }
    }
    const float inv = 1.0f / (float)d;
    const int64_t stride = (int64_t)gridDim.x * LN_WPB;
    for (int64_t row = (int64_t)blockIdx.x * LN_WPB + wave; row < rows; row += stride) {
        const float mu = mean[row], rs = rstd[row];
        float xh[NV][4], a[NV][4], u[NV][4], dv[NV][4];
        float s_a = 0.f, s_ax = 0.f, s_u = 0.f, s_ux = 0.f, s_w = 0.f, s_wx = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            if (c < d) {
                float xv[4];
                V4<T>::load(x + row * d + c, xv);
                V4<T>::load(dy + row * d + c, dv[i]);
                if (ggx) V4<T>::load(ggx + row * d + c, u[i]);
                else u[i][0] = u[i][1] = u[i][2] = u[i][3] = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[i][j] = (xv[j] - mu) * rs;
                    a[i][j] = dv[i][j] * g[i][j];
                    const float w = vg[i][j] * dv[i][j];
                    s_a += a[i][j];
                    s_ax += a[i][j] * xh[i][j];
                    s_u += u[i][j];
                    s_ux += u[i][j] * xh[i][j];
                    s_w += w;
                    s_wx += w * xh[i][j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) xh[i][j] = a[i][j] = u[i][j] = dv[i][j] = 0.f;
            }
        }
        const float c1 = wave_sum(s_a) * inv, c2 = wave_sum(s_ax) * inv;
        const float m_u = wave_sum(s_u) * inv, m_ux = wave_sum(s_ux) * inv;
        const float m_w = wave_sum(s_w) * inv, m_wx = wave_sum(s_wx) * inv;
        float s_udx = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s_udx += u[i][j] * rs * (a[i][j] - c1 - xh[i][j] * c2);
        const float m_udx = wave_sum(s_udx) * inv;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * WAVE + lane) * 4;
            if (c < d) {
                float o1[4], o2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dxv = rs * (a[i][j] - c1 - xh[i][j] * c2);
                    const float q = rs * (u[i][j] - m_u - xh[i][j] * m_ux);
                    const float w = vg[i][j] * dv[i][j];
                    o1[j] = g[i][j] * q + vg[i][j] * xh[i][j] + vb[i][j];
                    o2[j] = -rs * (xh[i][j] * m_udx + c2 * q + m_ux * dxv) + rs * (w - m_w - xh[i][j] * m_wx);
                    acc[i][j] += dv[i][j] * q;
                }
                V4<T>::store(g_dy + row * d + c, o1);
                V4<T>::store(g_x + row * d + c, o2);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * WAVE + lane) * 4;
        if (c < d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(&smem[c + j], acc[i][j]);
        }
    }
    __syncthreads();
    float* out = partial + (int64_t)blockIdx.x * d;
    for (int i = threadIdx.x; i < d; i += LN_BLOCK) out[i] = smem[i];
}

static int ln_nv(int d) {
    if (d <= 256) return 1;
    if (d <= 512) return 2;
    if (d <= 1024) return 4;
    if (d <= 2048) return 8;
    return 16;
}

static int ln_check(const char* who, int64_t rows, int d, int dtype) {
    MSSR_REQUIRE(valid_dtype(dtype), "%s: bad dtype %d", who, dtype);
    MSSR_REQUIRE(rows >= 0 && rows < ((int64_t)1 << 40), "%s: bad rows %lld", who, (long long)rows);
    MSSR_REQUIRE(d > 0 && d % 4 == 0 && d <= 4096, "%s: d=%d must be a multiple of 4 in (0, 4096]", who, d);
    return MSSR_OK;
}

template <int N> using IC = std::integral_constant<int, N>;
template <typename F> static void dispatch_nv(int nv, F&& f) {
    switch (nv) {
        case 1: f(IC<1>{}); break;
        case 2: f(IC<2>{}); break;
        case 4: f(IC<4>{}); break;
        case 8: f(IC<8>{}); break;
        default: f(IC<16>{}); break;
    }
}

}  // namespace mssr

using namespace mssr;

extern "C" int mssr_layernorm_partial_rows(int64_t rows) {
    static const int64_t max_parts = [] {
        const char* e = getenv("MSSR_LN_PARTS");
        return e ? (int64_t)atoll(e) : (int64_t)LN_MAX_PARTS;
    }();
    int64_t b = (rows + LN_WPB - 1) / LN_WPB;
    if (b < 1) b = 1;
    // more slices shorten the per-wave row loop but lengthen the finalize pass: 4480 x 512 bf16 15 us with 256 slices,
    // 20 us with 512; 71680 x 512 91 us with 256, 61 us with 512 (tools/ln_bench.py)
    const int64_t cap = (rows > 32768 && max_parts == LN_MAX_PARTS) ? 2 * max_parts : max_parts;
    return (int)(b < cap ? b : cap);
}

extern "C" int mssr_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                  float* rstd, int64_t rows, int d, float eps, int dtype, void* stream) {
    if (int e = ln_check("layernorm_fwd", rows, d, dtype)) return e;
    if (rows == 0) return MSSR_OK;
    MSSR_REQUIRE(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
    MSSR_REQUIRE(aligned(x, 16) && aligned(y, 16) && aligned(gamma, 16) && aligned(beta, 16),
                 "layernorm_fwd: pointers must be 16-byte aligned");
    const dim3 grid((unsigned)((rows + LN_WPB - 1) / LN_WPB));
    hipStream_t s = (hipStream_t)stream;
    const int nv = ln_nv(d);
    dispatch_nv(nv, [&](auto nvc) {
        constexpr int NV = decltype(nvc)::value;
        if (dtype == MSSR_F32)
            ln_fwd_kernel<float, NV><<<grid, dim3(LN_BLOCK), 0, s>>>((const float*)x, gamma, beta, (float*)y, mean,
                                                                     rstd, rows, d, eps);
        else
            ln_fwd_kernel<bf16, NV><<<grid, dim3(LN_BLOCK), 0, s>>>((const bf16*)x, gamma, beta, (bf16*)y, mean, rstd,
                                                                    rows, d, eps);
    });
    MSSR_LAUNCH_OK("layernorm_fwd");
}

extern "C" int mssr_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                  const float* rstd, void* dx, float* dgamma, float* dbeta, float* partial,
                                  int64_t rows, int d, int dtype, void* stream) {
    if (int e = ln_check("layernorm_bwd", rows, d, dtype)) return e;
    MSSR_REQUIRE(dgamma && dbeta, "layernorm_bwd: null output");
    hipStream_t s = (hipStream_t)stream;
    if (rows == 0) {
        (void)hipMemsetAsync(dgamma, 0, sizeof(float) * d, s);
        (void)hipMemsetAsync(dbeta, 0, sizeof(float) * d, s);
        return MSSR_OK;
    }
    MSSR_REQUIRE(dy && x && gamma && mean && rstd && dx && partial, "layernorm_bwd: null pointer");
    MSSR_REQUIRE(aligned(x, 16) && aligned(dy, 16) && aligned(dx, 16) && aligned(gamma, 16),
                 "layernorm_bwd: pointers must be 16-byte aligned");
    const int parts = mssr_layernorm_partial_rows(rows);
    const size_t shm = sizeof(float) * 2 * d;
    const int nv = ln_nv(d);
    dispatch_nv(nv, [&](auto nvc) {
        constexpr int NV = decltype(nvc)::value;
        if (dtype == MSSR_F32)
            ln_bwd_kernel<float, NV><<<dim3(parts), dim3(LN_BLOCK), shm, s>>>(
                (const float*)dy, (const float*)x, gamma, mean, rstd, (float*)dx, partial, rows, d);
        else
            ln_bwd_kernel<bf16, NV><<<dim3(parts), dim3(LN_BLOCK), shm, s>>>(
                (const bf16*)dy, (const bf16*)x, gamma, mean, rstd, (bf16*)dx, partial, rows, d);
    });
    ln_finalize_kernel<<<dim3((2 * d + 15) / 16), dim3(256), 0, s>>>(partial, dgamma, dbeta, parts, 2 * d, d);
    MSSR_LAUNCH_OK("layernorm_bwd");
}

extern "C" int mssr_layernorm_bwd_bwd(const void* ggx, const float* ggg, const float* ggb, const void* dy,
                                      const void* x, const float* gamma, const float* mean, const float* rstd,
                                      void* g_dy, void* g_x, float* g_gamma, float* partial, int64_t rows, int d,
                                      int dtype, void* stream) {
    if (int e = ln_check("layernorm_bwd_bwd", rows, d, dtype)) return e;
    MSSR_REQUIRE(g_gamma, "layernorm_bwd_bwd: null output");
    hipStream_t s = (hipStream_t)stream;
    if (rows == 0) {
        (void)hipMemsetAsync(g_gamma, 0, sizeof(float) * d, s);
        return MSSR_OK;
    }
    MSSR_REQUIRE(dy && x && gamma && mean && rstd && g_dy && g_x && partial, "layernorm_bwd_bwd: null pointer");
    MSSR_REQUIRE(aligned(x, 16) && aligned(dy, 16) && aligned(g_dy, 16) && aligned(g_x, 16) && aligned(gamma, 16) &&
                     (!ggx || aligned(ggx, 16)) && (!ggg || aligned(ggg, 16)) && (!ggb || aligned(ggb, 16)),
                 "layernorm_bwd_bwd: pointers must be 16-byte aligned");
    const int parts = mssr_layernorm_partial_rows(rows);
    const size_t shm = sizeof(float) * d;
    const int nv = ln_nv(d);
    dispatch_nv(nv, [&](auto nvc) {
        constexpr int NV = decltype(nvc)::value;
        if (dtype == MSSR_F32)
            ln_bwd_bwd_kernel<float, NV><<<dim3(parts), dim3(LN_BLOCK), shm, s>>>(
                (const float*)ggx, ggg, ggb, (const float*)dy, (const float*)x, gamma, mean, rstd, (float*)g_dy,
                (float*)g_x, partial, rows, d);
        else
            ln_bwd_bwd_kernel<bf16, NV><<<dim3(parts), dim3(LN_BLOCK), shm, s>>>(
                (const bf16*)ggx, ggg, ggb, (const bf16*)dy, (const bf16*)x, gamma, mean, rstd, (bf16*)g_dy,
                (bf16*)g_x, partial, rows, d);
    });
    ln_finalize_kernel<<<dim3((d + 15) / 16), dim3(256), 0, s>>>(partial, g_gamma, g_gamma, parts, d, d);
    MSSR_LAUNCH_OK("layernorm_bwd_bwd");
}

extern "C" int mssr_layernorm_param_grads(const void* dy, const void* x, const float* mean, const float* rstd, float* out,
                                          int batch, int64_t dy_bs, int64_t x_bs, int64_t stat_bs, int64_t out_bs,
                                          int64_t rows, int d, int accumulate, int dtype, void* stream) {
    MSSR_REQUIRE(valid_dtype(dtype), "layernorm_param_grads: bad dtype");
    const int v = dtype == MSSR_BF16 ? 8 : 4;
    MSSR_REQUIRE(batch > 0 && batch <= 65535 && rows >= 0 && d > 0 && d % v == 0, "layernorm_param_grads: bad shape");
    MSSR_REQUIRE(out && out_bs >= 2 * (int64_t)d, "layernorm_param_grads: null / overlapping output");
    hipStream_t s = (hipStream_t)stream;
    if (accumulate) {
        // the caller zeroed `out` (one fill for all of an engine's small gradient outputs) or wants the sums added
    } else if (out_bs == 2 * (int64_t)d) {
        (void)hipMemsetAsync(out, 0, sizeof(float) * 2 * (size_t)d * (size_t)batch, s);
    } else {
        for (int z = 0; z < batch; ++z) (void)hipMemsetAsync(out + (int64_t)z * out_bs, 0, sizeof(float) * 2 * (size_t)d, s);
    }
    if (rows == 0) return MSSR_OK;
    MSSR_REQUIRE(dy && x && mean && rstd, "layernorm_param_grads: null pointer");
    MSSR_REQUIRE(aligned(dy, 16) && aligned(x, 16) && dy_bs % v == 0 && x_bs % v == 0,
                 "layernorm_param_grads: operands must be 16-byte aligned");
    const int chunks = (d + 64 * v - 1) / (64 * v);
    int64_t gy = (rows + 31) / 32;
    const int64_t want = (2048 + (int64_t)chunks * batch - 1) / ((int64_t)chunks * batch);
    if (gy > want) gy = want;
    if (gy < 1) gy = 1;
    const dim3 grid((unsigned)chunks, (unsigned)gy, (unsigned)batch);
    if (dtype == MSSR_F32)
        ln_param_grad_kernel<float><<<grid, 256, 0, s>>>((const float*)dy, (const float*)x, mean, rstd, out, dy_bs, x_bs,
                                                         stat_bs, out_bs, rows, d);
    else
        ln_param_grad_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)dy, (const bf16*)x, mean, rstd, out, dy_bs, x_bs,
                                                        stat_bs, out_bs, rows, d);
    MSSR_LAUNCH_OK("layernorm_param_grads");
}
