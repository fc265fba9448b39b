// Elementwise / reduction helpers: dropout(+residual), mask replay, ReLU gate, column sums,
// axpby, embedding lookup + positional encoding.  All HBM-bound streaming kernels:
// 4 elements per lane per access (16 B for f32, 8 B for bf16), grid capped at 2048 blocks.
#include "common.hpp"

namespace mssr {

constexpr int EW_BLOCK = 256;
static inline unsigned ew_grid(int64_t nvec) {
    int64_t b = (nvec + EW_BLOCK - 1) / EW_BLOCK;
    if (b < 1) b = 1;
    return (unsigned)(b < 2048 ? b : 2048);
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void dropout_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                               T* __restrict__ y, uint8_t* __restrict__ mask,
                                                               int64_t n, float p, uint64_t seed) {
    const float inv = 1.0f / (1.0f - p);
    const uint32_t thr = drop_thr16(p);
    const int64_t nv = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * EW_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < nv; i += stride) {
        float v[4], r[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
        V4<T>::load(x + i * 4, v);
        if (res) V4<T>::load(res + i * 4, r);
        uint32_t packed = 0;
        const uint32_t kp[2] = {keep16_pair(seed, (uint64_t)i * 2, thr), keep16_pair(seed, (uint64_t)i * 2 + 1, thr)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t keep = (kp[j >> 1] >> (j & 1)) & 1u;
            packed |= keep << (8 * j);
            o[j] = (keep ? v[j] * inv : 0.f) + r[j];
        }
        V4<T>::store(y + i * 4, o);
        *reinterpret_cast<uint32_t*>(mask + i * 4) = packed;
    }
    // tail (n % 4)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (nv << 2) + threadIdx.x;
        const bool keep = keep16(seed, (uint64_t)i, thr);
        mask[i] = keep ? 1 : 0;
        y[i] = from_f<T>((keep ? to_f(x[i]) * inv : 0.f) + (res ? to_f(res[i]) : 0.f));
    }
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void mask_scale_kernel(const T* __restrict__ x, const uint8_t* __restrict__ mask,
                                                              const T* __restrict__ res, T* __restrict__ y, int64_t n,
                                                              float scale) {
    const int64_t nv = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * EW_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < nv; i += stride) {
        float v[4], r[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
        V4<T>::load(x + i * 4, v);
        if (res) V4<T>::load(res + i * 4, r);
        const uint32_t m = *reinterpret_cast<const uint32_t*>(mask + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (((m >> (8 * j)) & 0xff) ? v[j] * scale : 0.f) + r[j];
        V4<T>::store(y + i * 4, o);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (nv << 2) + threadIdx.x;
        y[i] = from_f<T>((mask[i] ? to_f(x[i]) * scale : 0.f) + (res ? to_f(res[i]) : 0.f));
    }
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void gate_kernel(const T* __restrict__ x, const T* __restrict__ gate,
                                                        T* __restrict__ y, int64_t n) {
    const int64_t nv = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * EW_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < nv; i += stride) {
        float v[4], g[4], o[4];
        V4<T>::load(x + i * 4, v);
        V4<T>::load(gate + i * 4, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = g[j] > 0.f ? v[j] : 0.f;
        V4<T>::store(y + i * 4, o);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (nv << 2) + threadIdx.x;
        y[i] = from_f<T>(to_f(gate[i]) > 0.f ? to_f(x[i]) : 0.f);
    }
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void axpby_kernel(const T* __restrict__ x, const T* __restrict__ z,
                                                         T* __restrict__ y, int64_t n, float a, float b) {
    const int64_t nv = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * EW_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < nv; i += stride) {
        float v[4], w[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
        V4<T>::load(x + i * 4, v);
        if (z) V4<T>::load(z + i * 4, w);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = a * v[j] + b * w[j];
        V4<T>::store(y + i * 4, o);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (nv << 2) + threadIdx.x;
        y[i] = from_f<T>(a * to_f(x[i]) + (z ? b * to_f(z[i]) : 0.f));
    }
}

// Column sums, wide form (rows 16-byte aligned, n a multiple of the vector): a lane owns one 16-byte column group
// (8 bf16 / 4 fp32), a wave reads 1 KB of one row per instruction, four rows in flight per wave; the block's waves
// split the rows of its slice, partial sums meet in LDS and leave as one float atomic per column and block.
// 4480 x 1536 bf16 (bias gradients of the packed projections over all decode steps): 16 -> 6 us.
template <typename T>
__global__ __launch_bounds__(256) void colsum_wide_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                          int n, int64_t ldx, int64_t x_bs, int64_t out_bs) {
    constexpr int V = 16 / (int)sizeof(T);
    __shared__ float sm[4][64 * V];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * V;
    x += (int64_t)blockIdx.z * x_bs;
    out += (int64_t)blockIdx.z * out_bs;
    float acc[V];
#pragma unroll
    for (int u = 0; u < V; ++u) acc[u] = 0.f;
    if (c < n) {
        const int64_t stride = (int64_t)gridDim.y * 4;
        int64_t r = (int64_t)blockIdx.y * 4 + wave;
        for (; r + 3 * stride < rows; r += 4 * stride) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const uint4*>(x + (r + k * stride) * ldx + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const T* e = reinterpret_cast<const T*>(&v[k]);
#pragma unroll
                for (int u = 0; u < V; ++u) acc[u] += to_f(e[u]);
            }
        }
        for (; r < rows; r += stride) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + r * ldx + c);
            const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int u = 0; u < V; ++u) acc[u] += to_f(e[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < V; ++u) sm[wave][lane * V + u] = acc[u];
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * V; i += 256) {
        const int col = blockIdx.x * 64 * V + i;
        if (col < n) atomicAdd(out + col, (sm[0][i] + sm[1][i]) + (sm[2][i] + sm[3][i]));
    }
}

template <typename T>
static bool colsum_wide_ok(const void* x, int n, int64_t ldx, int64_t x_bs) {
    constexpr int V = 16 / (int)sizeof(T);
    return aligned(x, 16) && n % V == 0 && ldx % V == 0 && x_bs % V == 0;
}

// row slices for the wide kernel: enough blocks to cover the chip a few times, at least 8 rows per wave
static unsigned colsum_slices(int64_t rows, int64_t chunks, int64_t batch) {
    int64_t gy = (rows + 31) / 32;
    const int64_t want = (2048 + chunks * batch - 1) / (chunks * batch);
    if (gy > want) gy = want;
    if (gy < 1) gy = 1;
    if (gy > 65535) gy = 65535;
    return (unsigned)gy;
}

// Column sums: block (64 x 4): 64 consecutive columns x 4 row-lanes; rows split over gridDim.y,
// partial results combined with float atomics (out zeroed by the host call unless accumulate).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                     int n, int64_t ldx, int64_t x_bs, int64_t out_bs) {
    __shared__ float sm[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    x += (int64_t)blockIdx.z * x_bs;            // batch of matrices (gridDim.z), one output row each
    out += (int64_t)blockIdx.z * out_bs;
    float s = 0.f;
    if (c < n) {
        for (int64_t r = (int64_t)blockIdx.y * 4 + ty; r < rows; r += (int64_t)gridDim.y * 4)
            s += to_f(x[r * ldx + c]);
    }
    sm[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < n) atomicAdd(out + c, (sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx]));
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void embedding_fwd_kernel(const int64_t* __restrict__ ids,
                                                                 const float* __restrict__ weight,
                                                                 const float* __restrict__ pe, T* __restrict__ out,
                                                                 int64_t rows, int d, int64_t vocab, int64_t batch,
                                                                 int step, float scale) {
    const int vper = d >> 2;
    const int64_t total = rows * vper;
    const int64_t stride = (int64_t)gridDim.x * EW_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / vper;
        const int c = (int)(i - r * vper) * 4;
        int64_t id = ids[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const int64_t pos = step >= 0 ? step : r / batch;
        float w[4], p[4], o[4];
        V4<float>::load(weight + id * d + c, w);
        V4<float>::load(pe + pos * d + c, p);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = w[j] * scale + p[j];
        V4<T>::store(out + r * d + c, o);
    }
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void embedding_bwd_kernel(const T* __restrict__ dout,
                                                                 const int64_t* __restrict__ ids,
                                                                 float* __restrict__ dweight, int64_t rows, int d,
                                                                 int64_t vocab, int64_t pad_idx, float scale) {
    const int64_t total = rows * d;
    const int64_t stride = (int64_t)gridDim.x * EW_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        const int64_t id = ids[r];
        if (id == pad_idx || id < 0 || id >= vocab) continue;
        atomicAdd(dweight + id * d + c, scale * to_f(dout[i]));
    }
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void scale_add_pe_kernel(const T* __restrict__ x, const float* __restrict__ pe,
                                                                T* __restrict__ out, int64_t rows, int d, int64_t batch,
                                                                int step, float scale) {
    const int vper = d >> 2;
    const int64_t total = rows * vper;
    const int64_t stride = (int64_t)gridDim.x * EW_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / vper;
        const int c = (int)(i - r * vper) * 4;
        const int64_t pos = step >= 0 ? step : r / batch;
        float v[4], p[4], o[4];
        V4<T>::load(x + r * d + c, v);
        V4<float>::load(pe + pos * d + c, p);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = v[j] * scale + p[j];
        V4<T>::store(out + r * d + c, o);
    }
}

// ------------------------------------------------------------------ WGAN-GP penalty of per-token gradients
//   penalty = lamb * mean over the rows with mask != 0 of (||g_row||_2 - k)^2
// (onmt/trainer_wgan_wh.py:1253-1261, :1321-1329: gradients.norm(2, dim=1), ((norm - k) ** 2).mean() * lamb on the
// rows selected by the sequence mask).  Rows are addressed (i0, i1) -> g + i0*s0 + i1*s1, so the [T, B, V] slice the
// autograd graph hands over is read where it lies; the penalty is a sum over rows, their order does not matter.
// A block per row (grid-stride beyond 512 rows); the last block to finish (ticket) divides by the count and writes the
// scalar.
// acc: float[4] zeroed by the caller = {sum of (norm-k)^2, selected rows, ticket, result}.
template <typename T>
__global__ __launch_bounds__(256) void gp_penalty_fwd_kernel(const T* __restrict__ g, int64_t s0, int64_t s1, int n0,
                                                             int n1, int d, const uint8_t* __restrict__ mask,
                                                             int64_t m0, int64_t m1, float k, float lamb,
                                                             float* __restrict__ norms, float* __restrict__ acc) {
    __shared__ float part[4];
    float pen = 0.f, cnt = 0.f;                      // this block's rows (thread 0)
    for (int row = blockIdx.x; row < n0 * n1; row += gridDim.x) {
        const int i0 = row / n1, i1 = row - i0 * n1;
        const T* gr = g + (int64_t)i0 * s0 + (int64_t)i1 * s1;
        float ss = 0.f;
        for (int c0 = threadIdx.x; c0 < d; c0 += 256 * 8) {        // 8 independent loads in flight per thread
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u * 256;
                v[u] = to_f(gr[c < d ? c : d - 1]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c0 + u * 256 < d) ss += v[u] * v[u];
        }
        ss = wave_sum(ss);
        __syncthreads();                             // part[] of the previous row has been read
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float norm = sqrtf((part[0] + part[1]) + (part[2] + part[3]));
            norms[row] = norm;
            if (mask[(int64_t)i0 * m0 + (int64_t)i1 * m1]) {
                pen += (norm - k) * (norm - k);
                cnt += 1.0f;
            }
        }
    }
    if (threadIdx.x == 0) {
        // a few hundred blocks, three atomics each (one block per row: ~1 400 x 3 same-line atomics took 85 us)
        atomicAdd(acc, pen);
        atomicAdd(acc + 1, cnt);
        __threadfence();
        const unsigned done = atomicAdd(reinterpret_cast<unsigned*>(acc + 2), 1u);
        if (done == gridDim.x - 1) {
            __threadfence();
            const float sum = atomicAdd(acc, 0.f), n = atomicAdd(acc + 1, 0.f);       // device-scope reads
            acc[3] = sum / n * lamb;
        }
    }
}

// dg_row = dpen * lamb / count * 2 (norm - k) / norm * g_row on selected rows (0 where norm = 0, as vector_norm's
// backward), 0 elsewhere; dg is contiguous [rows, d]
template <typename T>
__global__ __launch_bounds__(256) void gp_penalty_bwd_kernel(const T* __restrict__ g, int64_t s0, int64_t s1, int n1,
                                                             int d, const uint8_t* __restrict__ mask, int64_t m0,
                                                             int64_t m1, float k, float lamb,
                                                             const float* __restrict__ norms,
                                                             const float* __restrict__ acc,
                                                             const float* __restrict__ dpen, T* __restrict__ dg) {
    const int row = blockIdx.x, i0 = row / n1, i1 = row - i0 * n1;
    const T* gr = g + (int64_t)i0 * s0 + (int64_t)i1 * s1;
    T* out = dg + (int64_t)row * d;
    const float norm = norms[row];
    float coef = 0.f;
    if (mask[(int64_t)i0 * m0 + (int64_t)i1 * m1] && norm > 0.f) coef = dpen[0] * lamb / acc[1] * 2.0f * (norm - k) / norm;
    for (int c0 = threadIdx.x; c0 < d; c0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u * 256;
            v[u] = to_f(gr[c < d ? c : d - 1]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c0 + u * 256 < d) out[c0 + u * 256] = from_f<T>(coef * v[u]);
    }
}

// ------------------------------------------------------------------ critic gain at the label column
//   picked_i = ratio * sigmoid(x[i, l_i])   (force_positive_gain_mode "sigmoid": onmt/trainer_wgan_wh.py:402-416)
//   picked_i = x[i, l_i]                    ("normal")
// followed by cal_disc_loss's per-row pick (:1192-1200), as one op with its first and second derivative: the
// gradient penalty differentiates the critic's input gradient, which passes through this op's backward.
// A label outside [0, C) yields NaN (the reference's indexing would raise).
__device__ __forceinline__ float gp_sig(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void gain_pick_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const int64_t* __restrict__ labels, float* __restrict__ out,
                                                            int n, int C, float ratio, int sig) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t l = labels[i];
    if (l < 0 || l >= C) { out[i] = NAN; return; }
    const float v = x[(int64_t)i * ldx + l];
    out[i] = sig ? ratio * gp_sig(v) : v;
}

// dx[i, c] = (c == l_i) dy_i f'(x[i, l_i])            (dx contiguous [n, C])
__global__ __launch_bounds__(256) void gain_pick_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            int64_t ldx, const int64_t* __restrict__ labels,
                                                            float* __restrict__ dx, int n, int C, float ratio, int sig) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n * C) return;
    const int i = e / C, c = e - i * C;
    const int64_t l = labels[i];
    float o = 0.f;
    if (l < 0 || l >= C) o = NAN;
    else if (c == l) {
        float d1 = 1.0f;
        if (sig) {
            const float sg = gp_sig(x[(int64_t)i * ldx + l]);
            d1 = ratio * sg * (1.0f - sg);
        }
        o = dy[i] * d1;
    }
    dx[e] = o;
}

// backward of the above w.r.t. (dy, x) given ddx [n, C] (row stride ldd):
//   ddy_i = ddx[i, l_i] f'(x_il) ,   dx2[i, c] = (c == l_i) ddx[i, l_i] dy_i f''(x_il)
__global__ __launch_bounds__(256) void gain_pick_bwd_bwd_kernel(const float* __restrict__ ddx, int64_t ldd,
                                                                const float* __restrict__ dy,
                                                                const float* __restrict__ x, int64_t ldx,
                                                                const int64_t* __restrict__ labels,
                                                                float* __restrict__ ddy, float* __restrict__ dx2, int n,
                                                                int C, float ratio, int sig) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n * C) return;
    const int i = e / C, c = e - i * C;
    const int64_t l = labels[i];
    if (l < 0 || l >= C) {
        dx2[e] = NAN;
        if (c == 0) ddy[i] = NAN;
        return;
    }
    float o = 0.f;
    if (c == l) {
        const float g = ddx[(int64_t)i * ldd + l];
        float d1 = 1.0f, d2 = 0.f;
        if (sig) {
            const float sg = gp_sig(x[(int64_t)i * ldx + l]);
            d1 = ratio * sg * (1.0f - sg);
            d2 = d1 * (1.0f - 2.0f * sg);
        }
        ddy[i] = g * d1;
        o = g * dy[i] * d2;
    }
    dx2[e] = o;
}

}  // namespace mssr

using namespace mssr;

#define EW_COMMON(who)                                                          \
    MSSR_REQUIRE(valid_dtype(dtype), who ": bad dtype %d", dtype);              \
    MSSR_REQUIRE(n >= 0, who ": bad n");                                        \
    if (n == 0) return MSSR_OK;                                                 \
    hipStream_t s = (hipStream_t)stream;

extern "C" int mssr_dropout_fwd(const void* x, const void* residual, void* y, uint8_t* mask_out, int64_t n, float p,
                                uint64_t seed, int dtype, void* stream) {
    EW_COMMON("dropout_fwd");
    MSSR_REQUIRE(x && y && mask_out, "dropout_fwd: null pointer");
    MSSR_REQUIRE(p >= 0.f && p < 1.f, "dropout_fwd: p=%f out of [0,1)", p);
    MSSR_REQUIRE(aligned(x, 16) && aligned(y, 16) && aligned(mask_out, 4) && (!residual || aligned(residual, 16)),
                 "dropout_fwd: misaligned pointer");
    const unsigned g = ew_grid(n >> 2);
    if (dtype == MSSR_F32)
        dropout_fwd_kernel<float><<<dim3(g), dim3(EW_BLOCK), 0, s>>>((const float*)x, (const float*)residual, (float*)y, mask_out, n, p, seed);
    else
        dropout_fwd_kernel<bf16><<<dim3(g), dim3(EW_BLOCK), 0, s>>>((const bf16*)x, (const bf16*)residual, (bf16*)y, mask_out, n, p, seed);
    MSSR_LAUNCH_OK("dropout_fwd");
}

extern "C" int mssr_mask_scale(const void* x, const uint8_t* mask, const void* residual, void* y, int64_t n,
                               float scale, int dtype, void* stream) {
    EW_COMMON("mask_scale");
    MSSR_REQUIRE(x && y && mask, "mask_scale: null pointer");
    MSSR_REQUIRE(aligned(x, 16) && aligned(y, 16) && aligned(mask, 4) && (!residual || aligned(residual, 16)),
                 "mask_scale: misaligned pointer");
    const unsigned g = ew_grid(n >> 2);
    if (dtype == MSSR_F32)
        mask_scale_kernel<float><<<dim3(g), dim3(EW_BLOCK), 0, s>>>((const float*)x, mask, (const float*)residual, (float*)y, n, scale);
    else
        mask_scale_kernel<bf16><<<dim3(g), dim3(EW_BLOCK), 0, s>>>((const bf16*)x, mask, (const bf16*)residual, (bf16*)y, n, scale);
    MSSR_LAUNCH_OK("mask_scale");
}

extern "C" int mssr_gate_gt0(const void* x, const void* gate, void* y, int64_t n, int dtype, void* stream) {
    EW_COMMON("gate_gt0");
    MSSR_REQUIRE(x && y && gate, "gate_gt0: null pointer");
    MSSR_REQUIRE(aligned(x, 16) && aligned(y, 16) && aligned(gate, 16), "gate_gt0: misaligned pointer");
    const unsigned g = ew_grid(n >> 2);
    if (dtype == MSSR_F32)
        gate_kernel<float><<<dim3(g), dim3(EW_BLOCK), 0, s>>>((const float*)x, (const float*)gate, (float*)y, n);
    else
        gate_kernel<bf16><<<dim3(g), dim3(EW_BLOCK), 0, s>>>((const bf16*)x, (const bf16*)gate, (bf16*)y, n);
    MSSR_LAUNCH_OK("gate_gt0");
}

extern "C" int mssr_axpby(const void* x, const void* z, void* y, int64_t n, float a, float b, int dtype,
                          void* stream) {
    EW_COMMON("axpby");
    MSSR_REQUIRE(x && y, "axpby: null pointer");
    MSSR_REQUIRE(aligned(x, 16) && aligned(y, 16) && (!z || aligned(z, 16)), "axpby: misaligned pointer");
    const unsigned g = ew_grid(n >> 2);
    if (dtype == MSSR_F32)
        axpby_kernel<float><<<dim3(g), dim3(EW_BLOCK), 0, s>>>((const float*)x, (const float*)z, (float*)y, n, a, b);
    else
        axpby_kernel<bf16><<<dim3(g), dim3(EW_BLOCK), 0, s>>>((const bf16*)x, (const bf16*)z, (bf16*)y, n, a, b);
    MSSR_LAUNCH_OK("axpby");
}

extern "C" int mssr_colsum(const void* x, float* out, int64_t rows, int n, int64_t ldx, int accumulate, int dtype,
                           void* stream) {
    MSSR_REQUIRE(valid_dtype(dtype), "colsum: bad dtype %d", dtype);
    MSSR_REQUIRE(rows >= 0 && n > 0 && ldx >= n, "colsum: bad shape rows=%lld n=%d ldx=%lld", (long long)rows, n,
                 (long long)ldx);
    MSSR_REQUIRE(out && (x || rows == 0), "colsum: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) (void)hipMemsetAsync(out, 0, sizeof(float) * n, s);
    if (rows == 0) return MSSR_OK;
    if (dtype == MSSR_BF16 && colsum_wide_ok<bf16>(x, n, ldx, 0)) {
        const int chunks = (n + 511) / 512;
        colsum_wide_kernel<bf16><<<dim3(chunks, colsum_slices(rows, chunks, 1)), dim3(256), 0, s>>>((const bf16*)x, out, rows, n, ldx, 0, 0);
        MSSR_LAUNCH_OK("colsum");
    }
    if (dtype == MSSR_F32 && colsum_wide_ok<float>(x, n, ldx, 0)) {
        const int chunks = (n + 255) / 256;
        colsum_wide_kernel<float><<<dim3(chunks, colsum_slices(rows, chunks, 1)), dim3(256), 0, s>>>((const float*)x, out, rows, n, ldx, 0, 0);
        MSSR_LAUNCH_OK("colsum");
    }
    int64_t gy = (rows + 63) / 64;
    if (gy > 64) gy = 64;
    const dim3 grid((n + 63) / 64, (unsigned)gy);
    if (dtype == MSSR_F32)
        colsum_kernel<float><<<grid, dim3(256), 0, s>>>((const float*)x, out, rows, n, ldx, 0, 0);
    else
        colsum_kernel<bf16><<<grid, dim3(256), 0, s>>>((const bf16*)x, out, rows, n, ldx, 0, 0);
    MSSR_LAUNCH_OK("colsum");
}

extern "C" int mssr_colsum_batched(const void* x, float* out, int batch, int64_t x_bs, int64_t rows, int n, int64_t ldx,
                                   int accumulate, int dtype, void* stream) {
    MSSR_REQUIRE(valid_dtype(dtype), "colsum_batched: bad dtype %d", dtype);
    MSSR_REQUIRE(batch > 0 && batch <= 65535 && rows >= 0 && n > 0 && ldx >= n && x_bs >= 0,
                 "colsum_batched: bad shape batch=%d rows=%lld n=%d ldx=%lld", batch, (long long)rows, n, (long long)ldx);
    MSSR_REQUIRE(out && (x || rows == 0), "colsum_batched: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) (void)hipMemsetAsync(out, 0, sizeof(float) * (size_t)n * batch, s);
    if (rows == 0) return MSSR_OK;
    if (dtype == MSSR_BF16 && colsum_wide_ok<bf16>(x, n, ldx, x_bs)) {
        const int chunks = (n + 511) / 512;
        colsum_wide_kernel<bf16><<<dim3(chunks, colsum_slices(rows, chunks, batch), batch), dim3(256), 0, s>>>(
            (const bf16*)x, out, rows, n, ldx, x_bs, n);
        MSSR_LAUNCH_OK("colsum_batched");
    }
    if (dtype == MSSR_F32 && colsum_wide_ok<float>(x, n, ldx, x_bs)) {
        const int chunks = (n + 255) / 256;
        colsum_wide_kernel<float><<<dim3(chunks, colsum_slices(rows, chunks, batch), batch), dim3(256), 0, s>>>(
            (const float*)x, out, rows, n, ldx, x_bs, n);
        MSSR_LAUNCH_OK("colsum_batched");
    }
    int64_t gy = (rows + 63) / 64;
    if (gy > 64) gy = 64;
    const dim3 grid((n + 63) / 64, (unsigned)gy, (unsigned)batch);
    if (dtype == MSSR_F32)
        colsum_kernel<float><<<grid, dim3(256), 0, s>>>((const float*)x, out, rows, n, ldx, x_bs, n);
    else
        colsum_kernel<bf16><<<grid, dim3(256), 0, s>>>((const bf16*)x, out, rows, n, ldx, x_bs, n);
    MSSR_LAUNCH_OK("colsum_batched");
}

extern "C" int mssr_embedding_fwd(const int64_t* ids, const float* weight, const float* pe, void* out, int64_t rows,
                                  int d, int64_t vocab, int64_t batch, int step, float scale, int dtype,
                                  void* stream) {
    MSSR_REQUIRE(valid_dtype(dtype), "embedding_fwd: bad dtype %d", dtype);
    MSSR_REQUIRE(rows >= 0 && d > 0 && d % 4 == 0 && vocab > 0 && batch > 0, "embedding_fwd: bad shape");
    if (rows == 0) return MSSR_OK;
    MSSR_REQUIRE(ids && weight && pe && out, "embedding_fwd: null pointer");
    MSSR_REQUIRE(aligned(weight, 16) && aligned(pe, 16) && aligned(out, 16), "embedding_fwd: misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = ew_grid(rows * (d >> 2));
    if (dtype == MSSR_F32)
        embedding_fwd_kernel<float><<<dim3(g), dim3(EW_BLOCK), 0, s>>>(ids, weight, pe, (float*)out, rows, d, vocab, batch, step, scale);
    else
        embedding_fwd_kernel<bf16><<<dim3(g), dim3(EW_BLOCK), 0, s>>>(ids, weight, pe, (bf16*)out, rows, d, vocab, batch, step, scale);
    MSSR_LAUNCH_OK("embedding_fwd");
}

extern "C" int mssr_embedding_bwd(const void* dout, const int64_t* ids, float* dweight, int64_t rows, int d,
                                  int64_t vocab, int64_t pad_idx, float scale, int dtype, void* stream) {
    MSSR_REQUIRE(valid_dtype(dtype), "embedding_bwd: bad dtype %d", dtype);
    MSSR_REQUIRE(rows >= 0 && d > 0 && vocab > 0, "embedding_bwd: bad shape");
    if (rows == 0) return MSSR_OK;
    MSSR_REQUIRE(ids && dweight && dout, "embedding_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = ew_grid(rows * d);
    if (dtype == MSSR_F32)
        embedding_bwd_kernel<float><<<dim3(g), dim3(EW_BLOCK), 0, s>>>((const float*)dout, ids, dweight, rows, d, vocab, pad_idx, scale);
    else
        embedding_bwd_kernel<bf16><<<dim3(g), dim3(EW_BLOCK), 0, s>>>((const bf16*)dout, ids, dweight, rows, d, vocab, pad_idx, scale);
    MSSR_LAUNCH_OK("embedding_bwd");
}

extern "C" int mssr_scale_add_pe(const void* x, const float* pe, void* out, int64_t rows, int d, int64_t batch,
                                 int step, float scale, int dtype, void* stream) {
    MSSR_REQUIRE(valid_dtype(dtype), "scale_add_pe: bad dtype %d", dtype);
    MSSR_REQUIRE(rows >= 0 && d > 0 && d % 4 == 0 && batch > 0, "scale_add_pe: bad shape");
    if (rows == 0) return MSSR_OK;
    MSSR_REQUIRE(x && pe && out, "scale_add_pe: null pointer");
    MSSR_REQUIRE(aligned(x, 16) && aligned(pe, 16) && aligned(out, 16), "scale_add_pe: misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    const unsigned g = ew_grid(rows * (d >> 2));
    if (dtype == MSSR_F32)
        scale_add_pe_kernel<float><<<dim3(g), dim3(EW_BLOCK), 0, s>>>((const float*)x, pe, (float*)out, rows, d, batch, step, scale);
    else
        scale_add_pe_kernel<bf16><<<dim3(g), dim3(EW_BLOCK), 0, s>>>((const bf16*)x, pe, (bf16*)out, rows, d, batch, step, scale);
    MSSR_LAUNCH_OK("scale_add_pe");
}

extern "C" int mssr_gp_penalty_fwd(const void* g, int64_t s0, int64_t s1, int n0, int n1, int d, const uint8_t* mask,
                                   int64_t m0, int64_t m1, float k, float lamb, float* norms, float* acc, int dtype,
                                   void* stream) {
    MSSR_REQUIRE(g && mask && norms && acc, "gp_penalty_fwd: null pointer");
    MSSR_REQUIRE(valid_dtype(dtype), "gp_penalty_fwd: bad dtype");
    MSSR_REQUIRE(n0 > 0 && n1 > 0 && d > 0 && (int64_t)n0 * n1 < ((int64_t)1 << 31), "gp_penalty_fwd: bad dimensions");
    const int rows = n0 * n1;
    const dim3 grid((unsigned)(rows < 512 ? rows : 512));
    if (dtype == MSSR_F32)
        gp_penalty_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)g, s0, s1, n0, n1, d, mask, m0,
                                                                            m1, k, lamb, norms, acc);
    else
        gp_penalty_fwd_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16*)g, s0, s1, n0, n1, d, mask, m0, m1,
                                                                           k, lamb, norms, acc);
    MSSR_LAUNCH_OK("gp_penalty_fwd");
}

extern "C" int mssr_gp_penalty_bwd(const void* g, int64_t s0, int64_t s1, int n0, int n1, int d, const uint8_t* mask,
                                   int64_t m0, int64_t m1, float k, float lamb, const float* norms, const float* acc,
                                   const float* dpen, void* dg, int dtype, void* stream) {
    MSSR_REQUIRE(g && mask && norms && acc && dpen && dg, "gp_penalty_bwd: null pointer");
    MSSR_REQUIRE(valid_dtype(dtype), "gp_penalty_bwd: bad dtype");
    MSSR_REQUIRE(n0 > 0 && n1 > 0 && d > 0 && (int64_t)n0 * n1 < ((int64_t)1 << 31), "gp_penalty_bwd: bad dimensions");
    const dim3 grid((unsigned)(n0 * n1));
    if (dtype == MSSR_F32)
        gp_penalty_bwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)g, s0, s1, n1, d, mask, m0, m1, k,
                                                                            lamb, norms, acc, dpen, (float*)dg);
    else
        gp_penalty_bwd_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16*)g, s0, s1, n1, d, mask, m0, m1, k,
                                                                           lamb, norms, acc, dpen, (bf16*)dg);
    MSSR_LAUNCH_OK("gp_penalty_bwd");
}

extern "C" int mssr_gain_pick_fwd(const float* x, int64_t ldx, const int64_t* labels, float* out, int n, int C,
                                  float ratio, int sigmoid, void* stream) {
    MSSR_REQUIRE(x && labels && out && n > 0 && C > 0 && ldx >= C, "gain_pick_fwd: bad arguments");
    gain_pick_fwd_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, (hipStream_t)stream>>>(x, ldx, labels, out, n, C,
                                                                                           ratio, sigmoid);
    MSSR_LAUNCH_OK("gain_pick_fwd");
}

extern "C" int mssr_gain_pick_bwd(const float* dy, const float* x, int64_t ldx, const int64_t* labels, float* dx, int n,
                                  int C, float ratio, int sigmoid, void* stream) {
    MSSR_REQUIRE(dy && x && labels && dx && n > 0 && C > 0 && ldx >= C && (int64_t)n * C < ((int64_t)1 << 31),
                 "gain_pick_bwd: bad arguments");
    gain_pick_bwd_kernel<<<dim3((unsigned)((n * C + 255) / 256)), 256, 0, (hipStream_t)stream>>>(dy, x, ldx, labels, dx, n,
                                                                                               C, ratio, sigmoid);
    MSSR_LAUNCH_OK("gain_pick_bwd");
}

extern "C" int mssr_gain_pick_bwd_bwd(const float* ddx, int64_t ldd, const float* dy, const float* x, int64_t ldx,
                                      const int64_t* labels, float* ddy, float* dx2, int n, int C, float ratio,
                                      int sigmoid, void* stream) {
    MSSR_REQUIRE(ddx && dy && x && labels && ddy && dx2 && n > 0 && C > 0 && ldx >= C && ldd >= C &&
                     (int64_t)n * C < ((int64_t)1 << 31),
                 "gain_pick_bwd_bwd: bad arguments");
    gain_pick_bwd_bwd_kernel<<<dim3((unsigned)((n * C + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
        ddx, ldd, dy, x, ldx, labels, ddy, dx2, n, C, ratio, sigmoid);
    MSSR_LAUNCH_OK("gain_pick_bwd_bwd");
}
