// Optimizer step of the WGAN phases: global gradient norm + clipping + Adam over ALL parameters of a network in a
// handful of launches.
//
// Replaces, per network and optimizer step, torch.nn.utils.clip_grad_norm_(params, max_grad_norm) followed by
// torch.optim.Adam(lr, betas).step() (onmt/trainer_wgan_wh.py:192-211 builds the optimizers, :623-628 and :973-977
// clip and step; the reference's torch 1.6 loops over ~400 tensors in Python for both).  Parameters and gradients stay
// the separate fp32 tensors torch owns (state-dict ABI, autograd); their device pointers travel in the kernel
// arguments, up to OPT_PER_LAUNCH tensors per launch, so a 400-tensor network takes 5 launches per pass and no
// pointer table has to be uploaded.  The Adam moments live in two flat fp32 buffers owned by the caller.
//
// HBM-bound: the norm reads 4 B per element, the update reads 16 B and writes 12 B per element.
#include "common.hpp"

namespace mssr {

constexpr int OPT_PER_LAUNCH = 96;         // tensors per launch: 96 x (8+8+8+4+4+4) B of kernel arguments
constexpr int OPT_CHUNK = 4096;            // elements per block pass (256 threads x 4 x 4)
constexpr int OPT_MAX_BLOCKS_PER_TENSOR = 128;

struct OptPack {
    float* p[OPT_PER_LAUNCH];
    const float* g[OPT_PER_LAUNCH];
    int64_t off[OPT_PER_LAUNCH];           // offset of the tensor's moments in the flat state buffers
    int n[OPT_PER_LAUNCH];
    float lr_over_bc1[OPT_PER_LAUNCH];     // lr / (1 - beta1^t_i): torch counts steps PER PARAMETER (a parameter that
    float inv_sqrt_bc2[OPT_PER_LAUNCH];    // got no gradient in some step lags behind), so the corrections are per tensor
    int count;
};

// partial[first_slot + blockIdx.y * gridDim.x + blockIdx.x] = sum of g^2 over this block's share of tensor blockIdx.y
// (every slot is written, zeros included: the reduction below reads them all, in a fixed order -> deterministic)
__global__ __launch_bounds__(256) void opt_sqnorm_kernel(const OptPack pk, float* __restrict__ partial, int first_slot) {
    __shared__ float sm[4];
    const int t = blockIdx.y;
    const float* g = pk.g[t];
    const int n = pk.n[t];
    float s = 0.f;
    if (g) {
        const int n4 = n & ~3;
        for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < n4; i += gridDim.x * 1024) {
            const float4 v = *reinterpret_cast<const float4*>(g + i);
            s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        if (blockIdx.x == 0 && threadIdx.x < n - n4) {
            const float v = g[n4 + threadIdx.x];
            s += v * v;
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[first_slot + t * gridDim.x + blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// out[0] = sqrt(sum of partials) = the global L2 norm; out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))
// (torch.nn.utils.clip_grad_norm_'s formula; max_norm <= 0: no clipping, coefficient 1)
__global__ __launch_bounds__(1024) void opt_norm_finish_kernel(const float* __restrict__ partial, int n,
                                                               float* __restrict__ out, float max_norm) {
    __shared__ float sm[16];
    // four independent loads in flight per thread and trip (a 51k-slot list read one dependent load at a time took 61 us)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = threadIdx.x;
    for (; i + 3072 < n; i += 4096) {
        s0 += partial[i];
        s1 += partial[i + 1024];
        s2 += partial[i + 2048];
        s3 += partial[i + 3072];
    }
    for (; i < n; i += 1024) s0 += partial[i];
    float s = wave_sum((s0 + s1) + (s2 + s3));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += sm[w];
        const float norm = sqrtf(t);
        out[0] = norm;
        float c = 1.0f;
        if (max_norm > 0.f) {
            c = max_norm / (norm + 1e-6f);
            c = c < 1.0f ? c : 1.0f;
        }
        out[1] = c;
    }
}

// Adam (torch.optim.Adam, no weight decay, no amsgrad) on tensor blockIdx.y:
//   g' = coef g ; m = b1 m + (1-b1) g' ; v = b2 v + (1-b2) g'^2 ; p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// and, when write_clipped, g <- g' (clip_grad_norm_ scales the gradients in place; callers may read them afterwards)
__global__ __launch_bounds__(256) void opt_adam_kernel(const OptPack pk, float* __restrict__ m_all,
                                                       float* __restrict__ v_all, const float* __restrict__ coef_ptr,
                                                       float b1, float b2, float eps, int write_clipped) {
    const int t = blockIdx.y;
    const float* g = pk.g[t];
    if (!g) return;                         // parameter without a gradient this step: untouched, as torch skips it
    float* p = pk.p[t];
    float* m = m_all + pk.off[t];
    float* v = v_all + pk.off[t];
    const int n = pk.n[t];
    const float lr_over_bc1 = pk.lr_over_bc1[t], inv_sqrt_bc2 = pk.inv_sqrt_bc2[t];
    const float coef = coef_ptr ? coef_ptr[1] : 1.0f;
    const int n4 = n & ~3;
    // the moments' offsets are multiples of 4 elements (16-byte aligned) by construction of the flat buffers
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < n4; i += gridDim.x * 1024) {
        float4 gv = *reinterpret_cast<const float4*>(g + i);
        float4 pv = *reinterpret_cast<const float4*>(p + i);
        float4 mv = *reinterpret_cast<const float4*>(m + i);
        float4 vv = *reinterpret_cast<const float4*>(v + i);
        float* ge = reinterpret_cast<float*>(&gv);
        float* pe = reinterpret_cast<float*>(&pv);
        float* me = reinterpret_cast<float*>(&mv);
        float* ve = reinterpret_cast<float*>(&vv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gg = ge[j] * coef;
            ge[j] = gg;
            me[j] = b1 * me[j] + (1.0f - b1) * gg;
            ve[j] = b2 * ve[j] + (1.0f - b2) * gg * gg;
            pe[j] -= lr_over_bc1 * me[j] / (sqrtf(ve[j]) * inv_sqrt_bc2 + eps);
        }
        *reinterpret_cast<float4*>(p + i) = pv;
        *reinterpret_cast<float4*>(m + i) = mv;
        *reinterpret_cast<float4*>(v + i) = vv;
        if (write_clipped) *reinterpret_cast<float4*>(const_cast<float*>(g) + i) = gv;
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4) {
        const int i = n4 + threadIdx.x;
        const float gg = g[i] * coef;
        const float mm = b1 * m[i] + (1.0f - b1) * gg;
        const float vv = b2 * v[i] + (1.0f - b2) * gg * gg;
        m[i] = mm;
        v[i] = vv;
        p[i] -= lr_over_bc1 * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
        if (write_clipped) const_cast<float*>(g)[i] = gg;
    }
}

// dst[t] = scale * src[t] for tensor blockIdx.y (src null: zeros) -- the gather of a gradient bucket into its flat
// all-reduce buffer (parallel.GradSync._fill; p = destination slices, g = the gradients)
__global__ __launch_bounds__(256) void opt_copy_kernel(const OptPack pk, float scale) {
    const int t = blockIdx.y;
    const float* g = pk.g[t];
    float* p = pk.p[t];
    const int n = pk.n[t];
    const int n4 = n & ~3;
    for (int i = (blockIdx.x * 256 + threadIdx.x) * 4; i < n4; i += gridDim.x * 1024) {
        float4 v = g ? *reinterpret_cast<const float4*>(g + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        *reinterpret_cast<float4*>(p + i) = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4) p[n4 + threadIdx.x] = g ? g[n4 + threadIdx.x] * scale : 0.f;
}

static int blocks_for(int64_t max_n) {
    int64_t b = (max_n + OPT_CHUNK - 1) / OPT_CHUNK;
    if (b < 1) b = 1;
    return (int)(b > OPT_MAX_BLOCKS_PER_TENSOR ? OPT_MAX_BLOCKS_PER_TENSOR : b);
}

static int check_list(const char* who, int n, void* const* params, void* const* grads, const int64_t* numels) {
    MSSR_REQUIRE(n >= 0 && (n == 0 || (grads && numels)), "%s: bad tensor list", who);
    for (int i = 0; i < n; ++i) {
        MSSR_REQUIRE(numels[i] >= 0 && numels[i] <= 0x7fffffff, "%s: tensor %d has %lld elements", who, i,
                     (long long)numels[i]);
        MSSR_REQUIRE(!grads[i] || aligned(grads[i], 16), "%s: gradient %d is not 16-byte aligned", who, i);
        MSSR_REQUIRE(!params || (params[i] && aligned(params[i], 16)), "%s: parameter %d is null or unaligned", who, i);
    }
    return MSSR_OK;
}

}  // namespace mssr

using namespace mssr;

extern "C" int64_t mssr_grad_norm_scratch_floats(int n_tensors) {
    return (int64_t)(n_tensors > 0 ? n_tensors : 0) * OPT_MAX_BLOCKS_PER_TENSOR + 256;
}

extern "C" int mssr_grad_norm(int n, void* const* grads, const int64_t* numels, float max_norm, float* partial,
                              float* out, void* stream) {
    if (int e = check_list("grad_norm", n, nullptr, grads, numels)) return e;
    MSSR_REQUIRE(partial && out, "grad_norm: null scratch / output");
    hipStream_t s = (hipStream_t)stream;
    int slots = 0;
    for (int base = 0; base < n; base += OPT_PER_LAUNCH) {
        OptPack pk;
        pk.count = n - base < OPT_PER_LAUNCH ? n - base : OPT_PER_LAUNCH;
        int64_t mx = 0;
        for (int i = 0; i < pk.count; ++i) {
            pk.p[i] = nullptr;
            pk.g[i] = (const float*)grads[base + i];
            pk.n[i] = (int)numels[base + i];
            pk.off[i] = 0;
            pk.lr_over_bc1[i] = pk.inv_sqrt_bc2[i] = 0.f;
            if (grads[base + i] && numels[base + i] > mx) mx = numels[base + i];
        }
        const int bx = blocks_for(mx);
        opt_sqnorm_kernel<<<dim3(bx, pk.count), 256, 0, s>>>(pk, partial, slots);
        slots += bx * pk.count;
    }
    opt_norm_finish_kernel<<<1, 1024, 0, s>>>(partial, slots, out, max_norm);
    MSSR_LAUNCH_OK("grad_norm");
}

extern "C" int mssr_adam_step(int n, void* const* params, void* const* grads, const int64_t* numels,
                              const int64_t* state_offsets, float* exp_avg, float* exp_avg_sq, const float* norm_coef,
                              float lr, float beta1, float beta2, float eps, const int32_t* steps, int write_clipped,
                              void* stream) {
    if (int e = check_list("adam_step", n, params, grads, numels)) return e;
    MSSR_REQUIRE(n == 0 || (state_offsets && exp_avg && exp_avg_sq), "adam_step: null state");
    MSSR_REQUIRE(n == 0 || steps, "adam_step: null step counts");
    MSSR_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f, "adam_step: bad hyper-parameters");
    for (int i = 0; i < n; ++i) {
        MSSR_REQUIRE(state_offsets[i] >= 0 && state_offsets[i] % 4 == 0, "adam_step: state offset %d must be a multiple of 4", i);
        MSSR_REQUIRE(!grads[i] || steps[i] >= 1, "adam_step: tensor %d: step counts from 1", i);
    }
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < n; base += OPT_PER_LAUNCH) {
        OptPack pk;
        pk.count = n - base < OPT_PER_LAUNCH ? n - base : OPT_PER_LAUNCH;
        int64_t mx = 0;
        for (int i = 0; i < pk.count; ++i) {
            pk.p[i] = (float*)params[base + i];
            pk.g[i] = (const float*)grads[base + i];
            pk.n[i] = (int)numels[base + i];
            pk.off[i] = state_offsets[base + i];
            const int t = steps[base + i] >= 1 ? steps[base + i] : 1;
            pk.lr_over_bc1[i] = (float)((double)lr / (1.0 - pow((double)beta1, (double)t)));
            pk.inv_sqrt_bc2[i] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)t)));
            if (grads[base + i] && numels[base + i] > mx) mx = numels[base + i];
        }
        if (mx == 0) continue;
        opt_adam_kernel<<<dim3(blocks_for(mx), pk.count), 256, 0, s>>>(pk, exp_avg, exp_avg_sq, norm_coef, beta1,
                                                                      beta2, eps, write_clipped);
    }
    MSSR_LAUNCH_OK("adam_step");
}

extern "C" int mssr_multi_copy(int n, void* const* dst, void* const* src, const int64_t* numels, float scale,
                               void* stream) {
    if (int e = check_list("multi_copy", n, dst, src, numels)) return e;
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < n; base += OPT_PER_LAUNCH) {
        OptPack pk;
        pk.count = n - base < OPT_PER_LAUNCH ? n - base : OPT_PER_LAUNCH;
        int64_t mx = 0;
        for (int i = 0; i < pk.count; ++i) {
            pk.p[i] = (float*)dst[base + i];
            pk.g[i] = (const float*)src[base + i];
            pk.n[i] = (int)numels[base + i];
            pk.off[i] = 0;
            pk.lr_over_bc1[i] = pk.inv_sqrt_bc2[i] = 0.f;
            if (numels[base + i] > mx) mx = numels[base + i];
        }
        if (mx == 0) continue;
        opt_copy_kernel<<<dim3(blocks_for(mx), pk.count), 256, 0, s>>>(pk, scale);
    }
    MSSR_LAUNCH_OK("multi_copy");
}
