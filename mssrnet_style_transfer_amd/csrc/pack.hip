// Weight packs in ONE launch: every derived copy the engines read -- compute-dtype copies of the fp32 master
// parameters laid into packed [layer][rows][k] slabs, their transposes (the k-contiguous operand of dX = dY W),
// LayerNorm-folded weights gamma * W with the row sums of the ROUNDED values and W beta + b, fp32 vector copies --
// is described once per network as a device-resident job table and refreshed after each optimizer step by one
// kernel (mssr_pack_weights).  Replaces ~50 torch ops per network and step (cat / to / transpose().contiguous() /
// mul / sum / bmm in decode_engine._Pack, encoder_engine._Pack: 250 launches and ~2.5 ms of GPU time per WGAN step).
//
// One block = a strip of 32 rows of one job; columns are walked in chunks of 256.  Wave w owns rows 8w..8w+7 of the
// strip (16-byte loads, lane <-> 4 consecutive columns), so row sums are wave reductions in a fixed order; the
// transpose goes through an LDS tile and leaves as 64-byte column segments.
#include "common.hpp"

namespace mssr {

constexpr int PK_ROWS = 32, PK_COLS = 256, PK_LD = PK_COLS + 2;

__device__ __forceinline__ bool pk_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

template <typename T>
__device__ __forceinline__ void pk_store4(T* p, const float (&v)[4], bool vec, int n) {
    if (vec && n >= 4) {
        V4<T>::store(p, v);
    } else {
        for (int u = 0; u < 4 && u < n; ++u) p[u] = from_f<T>(v[u]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const mssr_pack_job* __restrict__ jobs, int n_jobs) {
    __shared__ __attribute__((aligned(16))) T tile[PK_ROWS * PK_LD];
    // job of this block: the last one whose first block is <= blockIdx.x
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const mssr_pack_job J = jobs[lo];
    const int r0 = ((int)blockIdx.x - J.block0) * PK_ROWS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rows = J.rows, cols = J.cols;
    T* dst = reinterpret_cast<T*>(J.dst);
    T* dst_t = reinterpret_cast<T*>(J.dst_t);
    T* dst_g = reinterpret_cast<T*>(J.dst_g);
    const bool src_vec = (cols & 3) == 0 && pk_aligned(J.src, 16);
    const bool dst_vec = dst && (J.dst_ld & 3) == 0 && pk_aligned(dst, 4 * sizeof(T));
    const bool dg_vec = dst_g && (J.dst_g_ld & 3) == 0 && pk_aligned(dst_g, 4 * sizeof(T));
    const bool df_vec = J.dst_f && (cols & 3) == 0 && pk_aligned(J.dst_f, 16);
    const bool t_vec = dst_t && (J.dst_t_ld & 7) == 0 && pk_aligned(dst_t, 16) && r0 + PK_ROWS <= rows;
    float cs[8], bf[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) cs[rr] = bf[rr] = 0.f;
    for (int c0 = 0; c0 < cols; c0 += PK_COLS) {
        const int c = c0 + 4 * lane, nc = cols - c;          // this lane's 4 columns; nc <= 0: beyond the row
        float g4[4] = {1.f, 1.f, 1.f, 1.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (nc > 0) {
            for (int u = 0; u < 4 && u < nc; ++u) {
                if (J.gamma) g4[u] = J.gamma[c + u];
                if (J.beta) b4[u] = J.beta[c + u];
            }
        }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int rl = 8 * wave + rr, r = r0 + rl;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (r < rows && nc > 0) {
                const float* s = J.src + (int64_t)r * cols + c;
                if (src_vec) V4<float>::load(s, v);
                else
                    for (int u = 0; u < 4 && u < nc; ++u) v[u] = s[u];
                if (dst) pk_store4<T>(dst + (int64_t)r * J.dst_ld + c, v, dst_vec, nc);
                if (J.dst_f) pk_store4<float>(J.dst_f + (int64_t)r * cols + c, v, df_vec, nc);
                if (dst_g || J.colsum) {
                    float w[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        w[u] = v[u] * g4[u];
                        cs[rr] += to_f(from_f<T>(w[u]));         // the sum of what the GEMM will read
                    }
                    if (dst_g) pk_store4<T>(dst_g + (int64_t)r * J.dst_g_ld + c, w, dg_vec, nc);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) bf[rr] += v[u] * b4[u];
            }
            if (dst_t && 4 * lane < PK_COLS) {
#pragma unroll
                for (int u = 0; u < 4; ++u) tile[rl * PK_LD + 4 * lane + u] = from_f<T>(v[u]);
            }
        }
        if (dst_t) {
            __syncthreads();
            const int ct = c0 + (int)threadIdx.x;            // one column of the chunk per thread
            if (ct < cols) {
                T* out = dst_t + (int64_t)ct * J.dst_t_ld + r0;
                if (t_vec) {
#pragma unroll
                    for (int q = 0; q < PK_ROWS * (int)sizeof(T) / 16; ++q) {
                        constexpr int PER = 16 / sizeof(T);
                        __attribute__((aligned(16))) T seg[PER];
#pragma unroll
                        for (int u = 0; u < PER; ++u) seg[u] = tile[(q * PER + u) * PK_LD + threadIdx.x];
                        *reinterpret_cast<uint4*>(out + q * PER) = *reinterpret_cast<const uint4*>(seg);
                    }
                } else {
                    for (int rl = 0; rl < PK_ROWS && r0 + rl < rows; ++rl) out[rl] = tile[rl * PK_LD + threadIdx.x];
                }
            }
            __syncthreads();
        }
    }
    if (J.colsum || J.bias_f) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int r = r0 + 8 * wave + rr;
            const float c_sum = wave_sum(cs[rr]), b_sum = wave_sum(bf[rr]);
            if (lane == 0 && r < rows) {
                if (J.colsum) J.colsum[r] = c_sum;
                if (J.bias_f) J.bias_f[r] = b_sum + (J.bias ? J.bias[r] : 0.f);
            }
        }
    }
}

}  // namespace mssr

using namespace mssr;

extern "C" int mssr_pack_blocks(int rows) { return rows > 0 ? (rows + PK_ROWS - 1) / PK_ROWS : 0; }

extern "C" int mssr_pack_weights(const mssr_pack_job* jobs_dev, int n_jobs, int n_blocks, int dtype, void* stream) {
    MSSR_REQUIRE(jobs_dev && n_jobs > 0 && n_blocks > 0, "pack_weights: empty job table");
    MSSR_REQUIRE(valid_dtype(dtype), "pack_weights: bad dtype");
    const dim3 grid((unsigned)n_blocks);
    if (dtype == MSSR_BF16) pack_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream>>>(jobs_dev, n_jobs);
    else pack_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(jobs_dev, n_jobs);
    MSSR_LAUNCH_OK("pack_weights");
}
