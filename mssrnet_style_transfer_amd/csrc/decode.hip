// Step-wise soft-decode engine: one C call runs ONE decode step through all decoder layers
// (forward), or its exact reverse (backward), launching every kernel back to back on the stream.
//
// Replaces the Python loop of the reference's differentiable decoding
// (onmt/train_and_translate.py:202-231 -> TransformerDecoder.forward with step=s,
// onmt/decoders/transformer.py:289-337 -> 12 x TransformerDecoderLayer._forward :110-171 with
// the layer_cache protocol of onmt/modules/multi_headed_attn.py:133-160 -> generator Linear +
// softmax, :95-116).  All per-step tensors live in step-major slabs [T][B, width] that the
// caller allocates once (mssr_decode_plan); the K/V "cache" is the packed projection slab
// qkv[l] = [B][T, 3d] itself (sentence-major: a sentence's keys of all steps lie 3d apart, in one run of memory -- the
// step attention walks them every step, and a step-major slab put every key 2 MB and more from the next at the
// 128-word stress shape), so nothing is concatenated or copied as the sequence grows.
// Weight / bias / LayerNorm gradients are NOT formed per step: the backward stores the
// gradient of every GEMM output in a slab and the caller contracts each slab once at the end
// (dW = dY_all^T X_all with a long reduction dimension instead of 28 products of depth 80).
#include "engine_common.hpp"
#include "gemm_common.hpp"
#include <stdlib.h>
#include <thread>
#include <initializer_list>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>
#include <functional>
#include <string.h>

namespace mssr {

constexpr int AT_MAXK = 512;       // keys per (sentence, head) the step-attention kernels handle
constexpr int AT_PF = 16;          // one-row-per-lane kernels: value / key rows prefetched into registers per lane
constexpr int MAX_CHUNKS = 8;      // concurrent batch chunks (streams) of one decode

// ------------------------------------------------------------------ single-query attention
// One wave64 per (batch, head).  q [dh], keys/values addressed with strides, optional pad mask.
// ctx = dropout(softmax(scale * q.K^T (masked))) . V
template <typename T>
__global__ __launch_bounds__(256) void attn_step_fwd_kernel(
    const T* __restrict__ q, int64_t q_bs, const T* __restrict__ k, const T* __restrict__ v, int64_t kv_bs,
    int64_t kv_ts, const uint8_t* __restrict__ mask, int64_t mask_bs, T* __restrict__ ctx, int64_t ctx_bs,
    int B, int H, int dh, int Lk, float scale, float p_drop_all, uint64_t seed, int drop_b0,
    const uint64_t* __restrict__ seed_base) {
    if (seed_base) seed += seed_word(seed_base);
    __shared__ float sc[4][AT_MAXK];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh_raw = blockIdx.x * 4 + wave;
    const bool active = bh_raw < B * H;
    const int bh = active ? bh_raw : B * H - 1;     // idle waves shadow the last slice and store nothing
    const int b = bh / H, h = bh % H;
    const float p_drop = b >= drop_b0 ? p_drop_all : 0.f;     // eval-mode sentences of a mixed sweep: no dropout
    const T* qp = q + (int64_t)b * q_bs + h * dh;
    const T* kp = k + (int64_t)b * kv_bs + h * dh;
    const T* vp = v + (int64_t)b * kv_bs + h * dh;
    float* s = sc[wave];
    // the value rows this lane needs for the context product (lane <-> feature) are known now: issue the first
    // AT_PF of them before the scores, so the kernel exposes one memory round trip instead of two
    const int jv = lane < dh ? lane : dh - 1;
    float vpre[AT_PF];
#pragma unroll
    for (int t = 0; t < AT_PF; ++t) vpre[t] = to_f(vp[(int64_t)(t < Lk ? t : Lk - 1) * kv_ts + jv]);
    // scores: lane <-> key
    float mx = -INFINITY;
    for (int t = lane; t < Lk; t += 64) {
        const T* kr = kp + (int64_t)t * kv_ts;
        float a = 0.f;
        for (int j = 0; j < dh; j += 4) {
            float kq[4], qq[4];
            V4<T>::load(kr + j, kq);
            V4<T>::load(qp + j, qq);
            a += kq[0] * qq[0] + kq[1] * qq[1] + kq[2] * qq[2] + kq[3] * qq[3];
        }
        a *= scale;
        if (mask && mask[(int64_t)b * mask_bs + t]) a = -1e18f;
        s[t] = a;
        mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < Lk; t += 64) {
        const float e = expf(s[t] - mx);
        s[t] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    const float keep_scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint32_t thr = drop_thr16(p_drop);
    for (int t = lane; t < Lk; t += 64) {
        float p = s[t] * inv;
        if (p_drop > 0.f) p = keep16(seed, (uint64_t)bh * AT_MAXK + t, thr) ? p * keep_scale : 0.f;
        s[t] = p;
    }
    __syncthreads();
    if (!active) return;
    // context: lane <-> feature
    {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < AT_PF; ++t) a += (t < Lk ? s[t] : 0.f) * vpre[t];
        for (int t = AT_PF; t < Lk; ++t) a += s[t] * to_f(vp[(int64_t)t * kv_ts + jv]);
        if (lane < dh) ctx[(int64_t)b * ctx_bs + h * dh + lane] = from_f<T>(a);
    }
    for (int j = lane + 64; j < dh; j += 64) {          // heads wider than a wave
        float a = 0.f;
        for (int t = 0; t < Lk; ++t) a += s[t] * to_f(vp[(int64_t)t * kv_ts + j]);
        ctx[(int64_t)b * ctx_bs + h * dh + j] = from_f<T>(a);
    }
}

// Backward of the above for one step s.  Recomputes the probabilities (fp32) from q and K, then
//   dp = dctx.V^T (x dropout) ; ds = p (dp - sum p dp) scale ; dq = ds.K
// and leaves ds_t and pd_t = dropped p_t (what multiplied V) in the per-step slabs ds_out / pd_out [Lk]:
// key/value gradients are GATHERED from those slabs, never accumulated in memory --
//   dK[t] = sum_{s >= t} ds_s[t] q_s ,  dV[t] = sum_{s >= t} pd_s[t] dctx_s .
// Self attention (gather != 0): row t = s of K/V is final once step s runs (later steps ran before it
// in the reverse sweep), so this launch also forms dK[s], dV[s] from the entries of steps s .. T-1,
// the q rows of the packed cache (stride q_ss) and the saved dctx rows (stride dctx_ss), and writes
// them into the dqkv slot of step s.  For the gather the self-attention slab is KEY-major, pairs
// sl[bh][t][s] = {ds_s[t], pd_s[t]} (ds_out = its base, slab_bhs = T * T pairs, pd_out / slab_ss unused):
// a step writes one pair per key (8-byte stores, T pairs apart) and reads the coefficients of its own key
// row over the later steps as ONE contiguous run.  (Step-major rows, the layout until round 3, made every
// coefficient a lone 4-byte read of its own cache line: at 140 steps 2 x 140 lines per (sentence, head)
// and launch, up to 73 MB of line traffic beside the 143 MB of q / dctx rows.)  Context attention
// (gather == 0): rows ds_out / pd_out [Lk] of this step; the caller contracts the slabs over all steps
// at the end (one batched product per layer).
template <typename T>
__global__ __launch_bounds__(256) void attn_step_bwd_kernel(
    const T* __restrict__ q, int64_t q_bs, const T* __restrict__ k, const T* __restrict__ v, int64_t kv_bs,
    int64_t kv_ts, const uint8_t* __restrict__ mask, int64_t mask_bs, const T* __restrict__ dctx, int64_t dctx_bs,
    T* __restrict__ dq, int64_t dq_bs, float* __restrict__ ds_out, float* __restrict__ pd_out, int64_t slab_bhs,
    int gather, int step, int n_steps, int64_t slab_ss, int64_t q_ss, int64_t dctx_ss, T* __restrict__ dk_out,
    T* __restrict__ dv_out, int64_t out_bs, int B, int H, int dh, int Lk, float scale, float p_drop, uint64_t seed,
    int bh_off, const uint64_t* __restrict__ seed_base) {
    if (seed_base) seed += seed_word(seed_base);
    __shared__ float s_p[4][AT_MAXK];      // p, then ds
    __shared__ float s_dp[4][AT_MAXK];     // d(loss)/d(p); then the gathered ds coefficients of later steps
    __shared__ float s_pd[4][AT_MAXK];     // dropped p; then the gathered pd coefficients of later steps
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh_raw = blockIdx.x * 4 + wave;
    const bool active = bh_raw < B * H;
    const int bh = active ? bh_raw : B * H - 1;     // idle waves shadow the last slice and store nothing
    const int b = bh / H, h = bh % H;
    const T* qp = q + (int64_t)b * q_bs + h * dh;
    const T* kp = k + (int64_t)b * kv_bs + h * dh;
    const T* vp = v + (int64_t)b * kv_bs + h * dh;
    const T* dcp = dctx + (int64_t)b * dctx_bs + h * dh;
    // context attention (gather == 0): rows ds_out / pd_out [Lk] of this step.  Self attention: ONE slab of pairs
    // {ds_s[t], pd_s[t]} with the (sentence, head) index innermost, sl[(t * n_steps + s) * slab_bhs]: the waves of a
    // block write / read neighbouring pairs of the same (key, step) sector
    float* dso = gather ? nullptr : ds_out + (int64_t)bh * slab_bhs;
    float* pdo = gather ? nullptr : pd_out + (int64_t)bh * slab_bhs;
    float2* sl = gather ? reinterpret_cast<float2*>(ds_out) + bh : nullptr;
    float* s = s_p[wave];
    float* dpv = s_dp[wave];
    float* pdv = s_pd[wave];
    // key rows of the final lane <-> feature phase (dq): their addresses are known now, so issue them first
    // (prefetching the later steps' q / dctx rows of the gather as well measured slower: 17.7 vs 14.1 us)
    const int jv = lane < dh ? lane : dh - 1;
    const int n_later = gather ? n_steps - 1 - step : 0;
    float kpre[AT_PF];
#pragma unroll
    for (int t = 0; t < AT_PF; ++t) kpre[t] = to_f(kp[(int64_t)(t < Lk ? t : Lk - 1) * kv_ts + jv]);
    float mx = -INFINITY;
    for (int t = lane; t < Lk; t += 64) {
        const T* kr = kp + (int64_t)t * kv_ts;
        float a = 0.f;
        for (int j = 0; j < dh; j += 4) {
            float kq[4], qq[4];
            V4<T>::load(kr + j, kq);
            V4<T>::load(qp + j, qq);
            a += kq[0] * qq[0] + kq[1] * qq[1] + kq[2] * qq[2] + kq[3] * qq[3];
        }
        a *= scale;
        if (mask && mask[(int64_t)b * mask_bs + t]) a = -1e18f;
        s[t] = a;
        mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < Lk; t += 64) {
        const float e = expf(s[t] - mx);
        s[t] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    const float keep_scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint32_t thr = drop_thr16(p_drop);
    // dp_t = m_t * sum_j dctx_j V[t][j] ; r = sum_t p_t dp_t
    float r = 0.f;
    for (int t = lane; t < Lk; t += 64) {
        const float p = s[t] * inv;
        float m = 1.0f;
        if (p_drop > 0.f) m = keep16(seed, (uint64_t)(bh + bh_off) * AT_MAXK + t, thr) ? keep_scale : 0.f;
        const T* vr = vp + (int64_t)t * kv_ts;
        float a = 0.f;
        for (int j = 0; j < dh; j += 4) {
            float vv[4], dd[4];
            V4<T>::load(vr + j, vv);
            V4<T>::load(dcp + j, dd);
            a += vv[0] * dd[0] + vv[1] * dd[1] + vv[2] * dd[2] + vv[3] * dd[3];
        }
        const float dp = a * m;
        s[t] = p;
        dpv[t] = dp;
        pdv[t] = p * m;
        if (active && !gather) pdo[t] = p * m;
        r += p * dp;
    }
    r = wave_sum(r);
    for (int t = lane; t < Lk; t += 64) {
        float ds = s[t] * (dpv[t] - r) * scale;
        if (mask && mask[(int64_t)b * mask_bs + t]) ds = 0.f;     // masked_fill: no gradient to the score
        s[t] = ds;
        if (active) {
            if (gather) sl[((int64_t)t * n_steps + step) * slab_bhs] = make_float2(ds, pdv[t]);
            else dso[t] = ds;
        }
    }
    __syncthreads();
    const float pd_own = gather ? pdv[step] : 0.f;
    __syncthreads();
    // coefficients of key row `step` at the steps step+1 .. T-1
    for (int i = lane; i < n_later; i += 64) {
        const float2 cf = sl[((int64_t)step * n_steps + step + 1 + i) * slab_bhs];
        dpv[i] = cf.x;
        pdv[i] = cf.y;
    }
    __syncthreads();
    if (!active) return;
    // lane <-> feature: dq (and the gathered dK, dV of row `step`)
    if (lane < dh) {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < AT_PF; ++t) a += (t < Lk ? s[t] : 0.f) * kpre[t];
        for (int t = AT_PF; t < Lk; ++t) a += s[t] * to_f(kp[(int64_t)t * kv_ts + lane]);
        dq[(int64_t)b * dq_bs + h * dh + lane] = from_f<T>(a);
        if (gather) {
            float gk = s[step] * to_f(qp[lane]);
            float gv = pd_own * to_f(dcp[lane]);
            for (int i = 0; i < n_later; ++i) {
                gk += dpv[i] * to_f(qp[(int64_t)(i + 1) * q_ss + lane]);
                gv += pdv[i] * to_f(dcp[(int64_t)(i + 1) * dctx_ss + lane]);
            }
            dk_out[(int64_t)b * out_bs + h * dh + lane] = from_f<T>(gk);
            dv_out[(int64_t)b * out_bs + h * dh + lane] = from_f<T>(gv);
        }
    }
    for (int j = lane + 64; j < dh; j += 64) {            // heads wider than a wave
        float a = 0.f;
        for (int t = 0; t < Lk; ++t) a += s[t] * to_f(kp[(int64_t)t * kv_ts + j]);
        dq[(int64_t)b * dq_bs + h * dh + j] = from_f<T>(a);
        if (gather) {
            // own step: ds_step[step] q_step, pd_step[step] dctx_step; then the later steps
            float gk = s[step] * to_f(qp[j]);
            float gv = pd_own * to_f(dcp[j]);
            for (int i = 0; i < n_later; ++i) {
                gk += dpv[i] * to_f(qp[(int64_t)(i + 1) * q_ss + j]);
                gv += pdv[i] * to_f(dcp[(int64_t)(i + 1) * dctx_ss + j]);
            }
            dk_out[(int64_t)b * out_bs + h * dh + j] = from_f<T>(gk);
            dv_out[(int64_t)b * out_bs + h * dh + j] = from_f<T>(gv);
        }
    }
}

// ------------------------------------------------------------------ tiled single-query attention
// Same math as the two kernels above with a different lane mapping: every lane moves 16 bytes per load, LPR =
// dh / E lanes cover one key / value row (E = elements per 16 bytes) and a wave covers 64 / LPR rows per load
// instruction; AT_NL instructions (a "chunk" of rows) are in flight together.  Every address of the first chunk
// of K, of V (and, in the backward, of the later steps' q / dctx rows and slab coefficients) is known at kernel
// start, so for up to 32 bf16 rows the whole kernel exposes ONE memory round trip; the one-row-per-lane kernels
// above issue Lk dependent 2-byte loads in their lane <-> feature phases (measured at 140 keys: 55 / 140 us).
// AT_NL load instructions per chunk (template parameter): 4 -- Yelp lengths (<= 28 keys / steps) fit one chunk at
// 8 rows per instruction; longer operands loop over chunks in the forward (8-row-load chunks cost registers, 118
// against 71 VGPRs, and at 768 sentences x 128 words the occupancy is worth more: 532 -> 524 ms per step) and
// take the streaming kernel below in the backward (the 8-row-load backward instance remains for strides beyond
// that kernel's 32-bit offsets).

template <typename T> struct Row16;
template <> struct Row16<bf16> {
    static constexpr int E = 8;
    typedef bf16x8 V;
    static __device__ __forceinline__ V load(const bf16* p) { return *reinterpret_cast<const V*>(p); }
    static __device__ __forceinline__ void store(bf16* p, const float (&o)[8]) {
        V v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16)o[i];
        *reinterpret_cast<V*>(p) = v;
    }
};
template <> struct Row16<float> {
    static constexpr int E = 4;
    typedef f32x4 V;
    static __device__ __forceinline__ V load(const float* p) { return *reinterpret_cast<const V*>(p); }
    static __device__ __forceinline__ void store(float* p, const float (&o)[4]) {
        V v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = o[i];
        *reinterpret_cast<V*>(p) = v;
    }
};

template <typename T>
__device__ __forceinline__ float dot16(const typename Row16<T>::V& a, const typename Row16<T>::V& b) {
    float r = 0.f;
#pragma unroll
    for (int e = 0; e < Row16<T>::E; ++e) r += (float)a[e] * (float)b[e];
    return r;
}

// the same product on the packed-bf16 dot instruction (v_dot2c_f32_bf16: exact products, fp32 accumulation; no
// unpacked copies of the operands in registers)
template <typename T>
__device__ __forceinline__ float dot16_packed(const typename Row16<T>::V& a, const typename Row16<T>::V& b) {
    if constexpr (sizeof(T) == 2) {
        float r = 0.f;
        r = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), r, false);
        r = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), r, false);
        r = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), r, false);
        r = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), r, false);
        return r;
    } else {
        return dot16<T>(a, b);
    }
}

// acc += w * row, two elements per instruction (v_pk_fma_f32 on adjacent accumulator registers)
template <typename T>
__device__ __forceinline__ void axpy16(float (&acc)[Row16<T>::E], float w, const typename Row16<T>::V& x) {
    typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < Row16<T>::E / 2; ++q) {
        f2 a = {acc[2 * q], acc[2 * q + 1]};
        const f2 v = {(float)x[2 * q], (float)x[2 * q + 1]};
        a += w * v;
        acc[2 * q] = a[0];
        acc[2 * q + 1] = a[1];
    }
}

// sum over the 2^LS adjacent lanes that hold one row (every lane of the group gets the total): DPP moves, no LDS
// crossbar trip (quad_perm xor 1, xor 2, row_half_mirror, row_mirror)
template <int LS>
__device__ __forceinline__ float group_sum(float v) {
    if (LS >= 1) v += dpp_mov<0xB1>(v);
    if (LS >= 2) v += dpp_mov<0x4E>(v);
    if (LS >= 3) v += dpp_mov<0x141>(v);
    if (LS >= 4) v += dpp_mov<0x140>(v);
    return v;
}
// the value of lane ^ O: data-parallel moves where one exists for the pattern, the LDS crossbar otherwise
template <int O>
__device__ __forceinline__ float xor_get(float v) {
    if constexpr (O == 1) return dpp_mov<0xB1>(v);
    else if constexpr (O == 2) return dpp_mov<0x4E>(v);
    else if constexpr (O == 8) return dpp_mov<0x128>(v);      // row_ror:8 == xor 8 inside a 16-lane row
    else return __shfl_xor(v, O, 64);
}
// sum over the 64 >> LS row groups of a wave (lanes that differ in bits LS..5), N independent values
template <int LS, int N>
__device__ __forceinline__ void cross_group_sum(float (&a)[N]) {
#define MSSR_XSTEP(O)                                       \
    if constexpr ((1 << LS) <= O) {                         \
        _Pragma("unroll") for (int n = 0; n < N; ++n) a[n] += xor_get<O>(a[n]); \
    }
    MSSR_XSTEP(1)
    MSSR_XSTEP(2)
    MSSR_XSTEP(4)
    MSSR_XSTEP(8)
    MSSR_XSTEP(16)
    MSSR_XSTEP(32)
#undef MSSR_XSTEP
}

// The same sum for the 8-lanes-per-row layout (lane = 8 r + c, 8 values per lane) WITHOUT the LDS crossbar: a
// transposing reduction -- v_permlane32_swap / v_permlane16_swap exchange half-waves / 16-lane rows of two registers,
// so one swap + one add folds a lane bit for TWO values at once and halves the live values; the last bit goes
// through a DPP rotate.  7 swaps / rotates + 9 adds instead of 8 DPP moves + 16 ds_bpermute + 24 adds
// (cross_group_sum<3, 8>), and afterwards EVERY lane holds one finished value: element
// group8_elem(lane) = 4 b3 + 2 b4 + b5 (bk = bit k of the lane index) of its 8-column group c = lane & 7.
// (inline asm: hipcc 7.2 folds the two results of __builtin_amdgcn_permlane{16,32}_swap into one register when both
// feed the same add -- v_add v1, v2, v2 -- measured with a standalone probe; the s_nop pads are the instruction's
// VALU-write wait states, which the compiler cannot see inside the string.)
__device__ __forceinline__ float group8_transpose_sum(const float (&a)[8], int lane) {
    float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5], a6 = a[6], a7 = a[7];
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
                 "v_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\ts_nop 1"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    float t0 = a0 + a1, t1 = a2 + a3, t2 = a4 + a5, t3 = a6 + a7;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
    const float u0 = t0 + t1, u1 = t2 + t3;
    const float w0 = u0 + dpp_mov<0x128>(u0);
    const float w1 = u1 + dpp_mov<0x128>(u1);
    return (lane & 8) ? w1 : w0;
}
__device__ __forceinline__ int group8_elem(int lane) { return 4 * ((lane >> 3) & 1) + 2 * ((lane >> 4) & 1) + ((lane >> 5) & 1); }

// rows t0 + i*RPL + r (clamped to n_rows - 1) of a [n_rows, dh] operand with row stride `rs`; `base` already
// points at this lane's 16-byte column group
template <typename T, int AT_NL>
__device__ __forceinline__ void load_chunk(typename Row16<T>::V (&dst)[AT_NL], const T* base, int64_t rs, int t0, int r,
                                           int RPL, int n_rows) {
#pragma unroll
    for (int i = 0; i < AT_NL; ++i) {
        int t = t0 + i * RPL + r;
        t = t < n_rows ? t : n_rows - 1;
        dst[i] = Row16<T>::load(base + (int64_t)t * rs);
    }
}

// Every wave of these kernels owns its (sentence, head) slice and its own LDS rows: lanes exchange values through
// LDS inside ONE wave only, whose DS instructions execute in program order.  A compiler-level fence is all that is
// needed between a phase's writes and the next phase's reads; a block barrier would couple four unrelated waves.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <typename T, int LS, int AT_NL>
__global__ __launch_bounds__(256) void attn_step_fwd_tiled_kernel(
    const T* __restrict__ q, int64_t q_bs, const T* __restrict__ k, const T* __restrict__ v, int64_t kv_bs,
    int64_t kv_ts, const uint8_t* __restrict__ mask, int64_t mask_bs, T* __restrict__ ctx, int64_t ctx_bs,
    int B, int H, int dh, int Lk, float scale, float p_drop_all, uint64_t seed, int drop_b0,
    const uint64_t* __restrict__ seed_base) {
    uint64_t seed_w = 0;                       // the replayed sweep's seed word: requested now, added where it is used
    if (seed_base) seed_w = seed_word(seed_base);
    typedef Row16<T> R;
    typedef typename R::V V;
    constexpr int E = R::E;
    __shared__ float sc[4][AT_MAXK];
    // the wave index through the scalar unit: (sentence, head) and every base address derived from it live in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int bh_raw = blockIdx.x * 4 + wave;
    const bool active = bh_raw < B * H;
    const int bh = active ? bh_raw : B * H - 1;
    const int b = bh / H, h = bh % H;
    const float p_drop = b >= drop_b0 ? p_drop_all : 0.f;     // eval-mode sentences of a mixed sweep: no dropout
    constexpr int LPR = 1 << LS, RPL = 64 >> LS, RC = AT_NL * RPL;
    const int c = lane & (LPR - 1), r = lane >> LS;
    const T* qp = q + (int64_t)b * q_bs + h * dh + c * E;
    const T* kp = k + (int64_t)b * kv_bs + h * dh + c * E;
    const T* vp = v + (int64_t)b * kv_bs + h * dh + c * E;
    float* s = sc[wave];
    V k0[AT_NL], v0[AT_NL];
    load_chunk<T, AT_NL>(k0, kp, kv_ts, 0, r, RPL, Lk);
    load_chunk<T, AT_NL>(v0, vp, kv_ts, 0, r, RPL, Lk);
    const V qf = R::load(qp);
    // the key-padding byte of key `lane`: an unconditional load (no mask: a byte of the output row, unused), consumed after the
    // chunk loads -- a select on the loaded value here would wait for it, i.e. for every K / V load issued before it
    const uint8_t mk0 = *(mask ? mask + (int64_t)b * mask_bs + (lane < Lk ? lane : Lk - 1) : reinterpret_cast<const uint8_t*>(ctx + (int64_t)b * ctx_bs));
    // dropout decision of key `lane` (index + seed only): taken under the loads
    const uint32_t thr = drop_thr16(p_drop);
    seed += seed_w;
    const bool keep0 = p_drop > 0.f ? keep16(seed, (uint64_t)bh * AT_MAXK + lane, thr) : true;

    auto scores = [&](const V (&kt)[AT_NL], int t0) {
#pragma unroll
        for (int i = 0; i < AT_NL; ++i) {
            if (t0 + i * RPL >= Lk) break;                  // uniform: the rest of the chunk is past the last key
            const float a = group_sum<LS>(dot16_packed<T>(kt[i], qf));
            const int t = t0 + i * RPL + r;
            if (c == 0 && t < Lk) s[t] = a * scale;
        }
    };
    scores(k0, 0);
    for (int t0 = RC; t0 < Lk; t0 += RC) {
        V kt[AT_NL];
        load_chunk<T, AT_NL>(kt, kp, kv_ts, t0, r, RPL, Lk);
        scores(kt, t0);
    }
    wave_lds_sync();
    float mx = -INFINITY;
    for (int t = lane; t < Lk; t += 64) {
        float a = s[t];
        const bool masked = mask && (t < 64 ? mk0 : mask[(int64_t)b * mask_bs + t]);
        if (masked) a = -1e18f;
        s[t] = a;
        mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < Lk; t += 64) {
        const float e = expf(s[t] - mx);
        s[t] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    const float keep_scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int t = lane; t < Lk; t += 64) {
        float p = s[t] * inv;
        if (p_drop > 0.f) p = (t < 64 ? keep0 : keep16(seed, (uint64_t)bh * AT_MAXK + t, thr)) ? p * keep_scale : 0.f;
        s[t] = p;
    }
    wave_lds_sync();
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    auto ctxacc = [&](const V (&vt)[AT_NL], int t0) {
#pragma unroll
        for (int i = 0; i < AT_NL; ++i) {
            if (t0 + i * RPL >= Lk) break;
            const int t = t0 + i * RPL + r;
            const float pt = t < Lk ? s[t] : 0.f;
            axpy16<T>(acc, pt, vt[i]);
        }
    };
    ctxacc(v0, 0);
    for (int t0 = RC; t0 < Lk; t0 += RC) {
        V vt[AT_NL];
        load_chunk<T, AT_NL>(vt, vp, kv_ts, t0, r, RPL, Lk);
        ctxacc(vt, t0);
    }
    if constexpr (LS == 3 && E == 8) {        // bf16, 64-wide heads: every lane finishes one element (2-byte store)
        const float o = group8_transpose_sum(acc, lane);
        if (active) ctx[(int64_t)b * ctx_bs + h * dh + c * E + group8_elem(lane)] = from_f<T>(o);
    } else {
        cross_group_sum<LS, E>(acc);
        if (active && r == 0) R::store(ctx + (int64_t)b * ctx_bs + h * dh + c * E, acc);
    }
}

// Backward (see attn_step_bwd_kernel for the math and the gather form): chunk 0 of K, V, of the later steps'
// q / dctx rows and the slab coefficients are all requested before the first use.
template <typename T, int LS, int AT_NL>
__global__ __launch_bounds__(256) void attn_step_bwd_tiled_kernel(
    const T* __restrict__ q, int64_t q_bs, const T* __restrict__ k, const T* __restrict__ v, int64_t kv_bs,
    int64_t kv_ts, const uint8_t* __restrict__ mask, int64_t mask_bs, const T* __restrict__ dctx, int64_t dctx_bs,
    T* __restrict__ dq, int64_t dq_bs, float* __restrict__ ds_out, float* __restrict__ pd_out, int64_t slab_bhs,
    int gather, int step, int n_steps, int64_t slab_ss, int64_t q_ss, int64_t dctx_ss, T* __restrict__ dk_out,
    T* __restrict__ dv_out, int64_t out_bs, int B, int H, int dh, int Lk, float scale, float p_drop, uint64_t seed,
    int bh_off, const uint64_t* __restrict__ seed_base) {
    uint64_t seed_w = 0;
    if (seed_base) seed_w = seed_word(seed_base);
    typedef Row16<T> R;
    typedef typename R::V V;
    constexpr int E = R::E;
    __shared__ float s_p[4][AT_MAXK];      // scores, p, then ds
    __shared__ float s_dp[4][AT_MAXK];     // d(loss)/d(p); then the ds coefficients of steps step .. T-1
    __shared__ float s_pd[4][AT_MAXK];     // dropped p; then the pd coefficients of steps step .. T-1
    // the wave index through the scalar unit: (sentence, head) and every base address derived from it live in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int bh_raw = blockIdx.x * 4 + wave;
    const bool active = bh_raw < B * H;
    const int bh = active ? bh_raw : B * H - 1;
    const int b = bh / H, h = bh % H;
    constexpr int LPR = 1 << LS, RPL = 64 >> LS, RC = AT_NL * RPL;
    const int c = lane & (LPR - 1), r = lane >> LS;
    const T* qp = q + (int64_t)b * q_bs + h * dh + c * E;
    const T* kp = k + (int64_t)b * kv_bs + h * dh + c * E;
    const T* vp = v + (int64_t)b * kv_bs + h * dh + c * E;
    const T* dcp = dctx + (int64_t)b * dctx_bs + h * dh + c * E;
    // context attention (gather == 0): rows ds_out / pd_out [Lk] of this step.  Self attention: ONE slab of pairs
    // {ds_s[t], pd_s[t]} with the (sentence, head) index innermost, sl[(t * n_steps + s) * slab_bhs]: the waves of a
    // block write / read neighbouring pairs of the same (key, step) sector
    float* dso = gather ? nullptr : ds_out + (int64_t)bh * slab_bhs;
    float* pdo = gather ? nullptr : pd_out + (int64_t)bh * slab_bhs;
    float2* sl = gather ? reinterpret_cast<float2*>(ds_out) + bh : nullptr;
    float* s = s_p[wave];
    float* dpv = s_dp[wave];
    float* pdv = s_pd[wave];
    const int n_rows = gather ? n_steps - step : 0;      // this step and the later ones

    V k0[AT_NL], v0[AT_NL], q0[AT_NL], d0[AT_NL];
    load_chunk<T, AT_NL>(k0, kp, kv_ts, 0, r, RPL, Lk);
    load_chunk<T, AT_NL>(v0, vp, kv_ts, 0, r, RPL, Lk);
    const V qf = R::load(qp);
    const V df = R::load(dcp);
    // the key-padding byte of key `lane`: an unconditional load (no mask: a byte of the output row, unused), consumed after the
    // chunk loads -- a select on the loaded value here would wait for it, i.e. for every K / V load issued before it
    const uint8_t mk0 = *(mask ? mask + (int64_t)b * mask_bs + (lane < Lk ? lane : Lk - 1) : reinterpret_cast<const uint8_t*>(dq + (int64_t)b * dq_bs));
    const uint32_t thr = drop_thr16(p_drop);         // dropout decision of key `lane`: taken under the loads
    seed += seed_w;
    const bool keep0 = p_drop > 0.f ? keep16(seed, (uint64_t)(bh + bh_off) * AT_MAXK + lane, thr) : true;
    float ck0 = 0.f, cv0 = 0.f;
    if (gather) {
        load_chunk<T, AT_NL>(q0, qp, q_ss, 0, r, RPL, n_rows);
        load_chunk<T, AT_NL>(d0, dcp, dctx_ss, 0, r, RPL, n_rows);
        const int i = lane >= 1 && lane < n_rows ? lane : (n_rows > 1 ? 1 : 0);
        if (n_rows > 1) {                       // entries of the later steps exist (written by earlier launches)
            const float2 cf = sl[((int64_t)step * n_steps + step + i) * slab_bhs];
            ck0 = cf.x;
            cv0 = cf.y;
        }
    }

    auto phase_a = [&](const V (&kt)[AT_NL], const V (&vt)[AT_NL], int t0) {
#pragma unroll
        for (int i = 0; i < AT_NL; ++i) {
            if (t0 + i * RPL >= Lk) break;
            const float a = group_sum<LS>(dot16_packed<T>(kt[i], qf));
            const float g = group_sum<LS>(dot16_packed<T>(vt[i], df));
            const int t = t0 + i * RPL + r;
            if (c == 0 && t < Lk) {
                s[t] = a * scale;
                dpv[t] = g;
            }
        }
    };
    phase_a(k0, v0, 0);
    for (int t0 = RC; t0 < Lk; t0 += RC) {
        V kt[AT_NL], vt[AT_NL];
        load_chunk<T, AT_NL>(kt, kp, kv_ts, t0, r, RPL, Lk);
        load_chunk<T, AT_NL>(vt, vp, kv_ts, t0, r, RPL, Lk);
        phase_a(kt, vt, t0);
    }
    wave_lds_sync();
    float mx = -INFINITY;
    for (int t = lane; t < Lk; t += 64) {
        float a = s[t];
        const bool masked = mask && (t < 64 ? mk0 : mask[(int64_t)b * mask_bs + t]);
        if (masked) a = -1e18f;
        s[t] = a;
        mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < Lk; t += 64) {
        const float e = expf(s[t] - mx);
        s[t] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    const float keep_scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    float rr = 0.f;
    for (int t = lane; t < Lk; t += 64) {
        const float p = s[t] * inv;
        float m = 1.0f;
        if (p_drop > 0.f) m = (t < 64 ? keep0 : keep16(seed, (uint64_t)(bh + bh_off) * AT_MAXK + t, thr)) ? keep_scale : 0.f;
        const float dp = dpv[t] * m;
        s[t] = p;
        dpv[t] = dp;
        pdv[t] = p * m;
        if (active && !gather) pdo[t] = p * m;
        rr += p * dp;
    }
    rr = wave_sum(rr);
    for (int t = lane; t < Lk; t += 64) {
        float ds = s[t] * (dpv[t] - rr) * scale;
        const bool masked = mask && (t < 64 ? mk0 : mask[(int64_t)b * mask_bs + t]);
        if (masked) ds = 0.f;                    // masked_fill: no gradient to the score
        s[t] = ds;
        if (active) {
            if (gather) sl[((int64_t)t * n_steps + step) * slab_bhs] = make_float2(ds, pdv[t]);
            else dso[t] = ds;
        }
    }
    wave_lds_sync();
    const float ds_own = gather ? s[step] : 0.f;
    const float pd_own = gather ? pdv[step] : 0.f;
    wave_lds_sync();
    if (gather) {                                // coefficient of row `step` in the slab of step step + i
        if (lane == 0) {
            dpv[0] = ds_own;
            pdv[0] = pd_own;
        } else if (lane < n_rows) {
            dpv[lane] = ck0;
            pdv[lane] = cv0;
        }
        for (int i = lane + 64; i < n_rows; i += 64) {
            const float2 cf = sl[((int64_t)step * n_steps + step + i) * slab_bhs];
            dpv[i] = cf.x;
            pdv[i] = cf.y;
        }
    }
    wave_lds_sync();
    // dq = sum_t ds_t K[t]
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    auto dqacc = [&](const V (&kt)[AT_NL], int t0) {
#pragma unroll
        for (int i = 0; i < AT_NL; ++i) {
            if (t0 + i * RPL >= Lk) break;
            const int t = t0 + i * RPL + r;
            const float w = t < Lk ? s[t] : 0.f;
            axpy16<T>(acc, w, kt[i]);
        }
    };
    dqacc(k0, 0);
    for (int t0 = RC; t0 < Lk; t0 += RC) {
        V kt[AT_NL];
        load_chunk<T, AT_NL>(kt, kp, kv_ts, t0, r, RPL, Lk);
        dqacc(kt, t0);
    }
    if constexpr (LS == 3 && E == 8) {
        const float o = group8_transpose_sum(acc, lane);
        if (active) dq[(int64_t)b * dq_bs + h * dh + c * E + group8_elem(lane)] = from_f<T>(o);
    } else {
        cross_group_sum<LS, E>(acc);
        if (active && r == 0) R::store(dq + (int64_t)b * dq_bs + h * dh + c * E, acc);
    }
    if (!gather) return;
    // dK[step] = sum_i ds_{step+i}[step] q_{step+i} ; dV[step] = sum_i pd_{step+i}[step] dctx_{step+i}
    float gk[E], gv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) gk[e] = gv[e] = 0.f;
    auto gacc = [&](const V (&qt)[AT_NL], const V (&dt)[AT_NL], int t0) {
#pragma unroll
        for (int i = 0; i < AT_NL; ++i) {
            if (t0 + i * RPL >= n_rows) break;
            const int t = t0 + i * RPL + r;
            const float wk = t < n_rows ? dpv[t] : 0.f;
            const float wv = t < n_rows ? pdv[t] : 0.f;
            axpy16<T>(gk, wk, qt[i]);
            axpy16<T>(gv, wv, dt[i]);
        }
    };
    gacc(q0, d0, 0);
    for (int t0 = RC; t0 < n_rows; t0 += RC) {
        V qt[AT_NL], dt[AT_NL];
        load_chunk<T, AT_NL>(qt, qp, q_ss, t0, r, RPL, n_rows);
        load_chunk<T, AT_NL>(dt, dcp, dctx_ss, t0, r, RPL, n_rows);
        gacc(qt, dt, t0);
    }
    if constexpr (LS == 3 && E == 8) {
        const float ok = group8_transpose_sum(gk, lane), ov = group8_transpose_sum(gv, lane);
        if (active) {
            const int64_t o = (int64_t)b * out_bs + h * dh + c * E + group8_elem(lane);
            dk_out[o] = from_f<T>(ok);
            dv_out[o] = from_f<T>(ov);
        }
    } else {
        cross_group_sum<LS, E>(gk);
        cross_group_sum<LS, E>(gv);
        if (active && r == 0) {
            R::store(dk_out + (int64_t)b * out_bs + h * dh + c * E, gk);
            R::store(dv_out + (int64_t)b * out_bs + h * dh + c * E, gv);
        }
    }
}

// Backward for operands of more than one chunk (long sentences).  The kernel above reads K twice when the keys
// do not fit its first chunk (scores, then dq = sum_t ds_t K[t] once the softmax is known) and holds four operand
// chunks in registers (231 VGPRs with 8-row-load chunks: 2 waves per SIMD).  Here K and V stream through ONCE: with
// e_t = exp(a_t - m) against a running maximum m (rescaled when it moves, per row group -- no cross-lane traffic
// inside the loop),
//     dq = scale / L * ( sum_t e_t dp_t K[t]  -  rr * sum_t e_t K[t] ),   L = sum_t e_t,  rr = sum_t e_t dp_t / L,
// so both sums accumulate in the pass that produces the scores; the row groups' partial results are merged once at
// the end (split-softmax combine).  The score gradients / dropped probabilities of the slabs are finished from the
// raw scores kept in LDS.  No operand rows are held across phases, wave-uniform base addresses sit in SGPRs (lanes
// carry 32-bit byte offsets) and the bf16 dot products use the packed instruction: 97 VGPRs, 5 waves per SIMD, all
// waves of a 512-sentence launch resident at once.  Measured on --workload long (bs256, 128 words; ms per WGAN
// step): 560.1 with the kernel above, 545.8 with this one; variants that request the next chunk while the current
// one is worked on need the registers of a second chunk and lose more to occupancy than they gain (4-row-load
// chunks, double-buffered, 152 VGPRs: 550.3; 2-row-load chunks, 98 VGPRs: 548.8) -- thread-level parallelism
// hides the latency here, not software pipelining.
template <int LS>
__device__ __forceinline__ float cross_group_max(float v) {
#define MSSR_XMAX(O) \
    if constexpr ((1 << LS) <= O) v = fmaxf(v, xor_get<O>(v));
    MSSR_XMAX(1)
    MSSR_XMAX(2)
    MSSR_XMAX(4)
    MSSR_XMAX(8)
    MSSR_XMAX(16)
    MSSR_XMAX(32)
#undef MSSR_XMAX
    return v;
}

template <typename T, int LS, int AT_NL>
__global__ __launch_bounds__(256) void attn_step_bwd_stream_kernel(
    const T* __restrict__ q, int64_t q_bs, const T* __restrict__ k, const T* __restrict__ v, int64_t kv_bs,
    int64_t kv_ts, const uint8_t* __restrict__ mask, int64_t mask_bs, const T* __restrict__ dctx, int64_t dctx_bs,
    T* __restrict__ dq, int64_t dq_bs, float* __restrict__ ds_out, float* __restrict__ pd_out, int64_t slab_bhs,
    int gather, int step, int n_steps, int64_t slab_ss, int64_t q_ss, int64_t dctx_ss, T* __restrict__ dk_out,
    T* __restrict__ dv_out, int64_t out_bs, int B, int H, int dh, int Lk, float scale, float p_drop, uint64_t seed,
    int bh_off, const uint64_t* __restrict__ seed_base) {
    uint64_t seed_w = 0;
    if (seed_base) seed_w = seed_word(seed_base);
    typedef Row16<T> R;
    typedef typename R::V V;
    constexpr int E = R::E;
    __shared__ float s_p[4][AT_MAXK];      // masked scores, then ds
    __shared__ float s_dp[4][AT_MAXK];     // d(loss)/d(p) after dropout; then the ds coefficients of steps step .. T-1
    __shared__ float s_pd[4][AT_MAXK];     // dropout multiplier, then dropped p; then the pd coefficients
    // wave-uniform values through the scalar unit: base addresses live in SGPRs, lanes carry 32-bit byte offsets
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int bh_raw = blockIdx.x * 4 + wave;
    const bool active = bh_raw < B * H;
    const int bh = active ? bh_raw : B * H - 1;
    const int b = bh / H, h = bh % H;
    constexpr int LPR = 1 << LS, RPL = 64 >> LS, RC = AT_NL * RPL;
    const int c = lane & (LPR - 1), r = lane >> LS;
    const char* qp = reinterpret_cast<const char*>(q + (int64_t)b * q_bs + h * dh);
    const char* kp = reinterpret_cast<const char*>(k + (int64_t)b * kv_bs + h * dh);
    const char* vp = reinterpret_cast<const char*>(v + (int64_t)b * kv_bs + h * dh);
    const char* dcp = reinterpret_cast<const char*>(dctx + (int64_t)b * dctx_bs + h * dh);
    const uint32_t co = (uint32_t)(c * E * sizeof(T));                       // this lane's 16-byte column group
    const uint32_t kv_tsb = (uint32_t)(kv_ts * sizeof(T)), q_ssb = (uint32_t)(q_ss * sizeof(T)),
                   dctx_ssb = (uint32_t)(dctx_ss * sizeof(T));
    auto load_rows = [&](V (&dst)[AT_NL], const char* base, uint32_t rsb, int t0, int n) {
#pragma unroll
        for (int i = 0; i < AT_NL; ++i) {
            int t = t0 + i * RPL + r;
            t = t < n ? t : n - 1;
            dst[i] = *reinterpret_cast<const V*>(base + ((uint32_t)t * rsb + co));
        }
    };
    const uint8_t* mrow = mask ? mask + (int64_t)b * mask_bs : nullptr;
    // context attention (gather == 0): rows ds_out / pd_out [Lk] of this step.  Self attention: ONE slab of pairs
    // {ds_s[t], pd_s[t]} with the (sentence, head) index innermost, sl[(t * n_steps + s) * slab_bhs]: the waves of a
    // block write / read neighbouring pairs of the same (key, step) sector
    float* dso = gather ? nullptr : ds_out + (int64_t)bh * slab_bhs;
    float* pdo = gather ? nullptr : pd_out + (int64_t)bh * slab_bhs;
    float2* sl = gather ? reinterpret_cast<float2*>(ds_out) + bh : nullptr;
    float* s = s_p[wave];
    float* dpv = s_dp[wave];
    float* pdv = s_pd[wave];
    const int n_rows = gather ? n_steps - step : 0;      // this step and the later ones

    V kc[AT_NL], vc[AT_NL];
    uint8_t mc[AT_NL];
    auto load_kv = [&](V (&kk)[AT_NL], V (&vv)[AT_NL], uint8_t (&mm)[AT_NL], int t0) {
        load_rows(kk, kp, kv_tsb, t0, Lk);
        load_rows(vv, vp, kv_tsb, t0, Lk);
#pragma unroll
        for (int i = 0; i < AT_NL; ++i) mm[i] = 0;
        if (mrow) {
#pragma unroll
            for (int i = 0; i < AT_NL; ++i) {
                const int t = t0 + i * RPL + r;
                mm[i] = mrow[(uint32_t)(t < Lk ? t : Lk - 1)];
            }
        }
    };
    load_kv(kc, vc, mc, 0);
    const V qf = *reinterpret_cast<const V*>(qp + co);
    const V df = *reinterpret_cast<const V*>(dcp + co);
    float ck0 = 0.f, cv0 = 0.f;
    if (gather && n_rows > 1) {                 // slabs of the later steps exist (written by earlier launches)
        const int i = lane >= 1 && lane < n_rows ? lane : 1;
        const float2 cf = sl[((int64_t)step * n_steps + step + i) * slab_bhs];
        ck0 = cf.x;
        cv0 = cf.y;
    }
    const uint32_t thr = drop_thr16(p_drop);
    seed += seed_w;
    const float keep_scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint64_t hrow = (uint64_t)(bh + bh_off) * AT_MAXK;

    float m = -1e30f, l = 0.f, rn = 0.f, A[E], Bv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) A[e] = Bv[e] = 0.f;
    for (int t0 = 0;;) {
        float a[AT_NL], g[AT_NL];
        float cm = -INFINITY;
#pragma unroll
        for (int i = 0; i < AT_NL; ++i) {
            a[i] = -INFINITY;
            g[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < AT_NL; ++i) {
            if (t0 + i * RPL >= Lk) break;                  // uniform: the rest of the chunk is past the last key
            const int t = t0 + i * RPL + r;
            float av = group_sum<LS>(dot16_packed<T>(kc[i], qf)) * scale;
            float gv = group_sum<LS>(dot16_packed<T>(vc[i], df));
            float mul = 1.0f;
            if (p_drop > 0.f) mul = keep16(seed, hrow + t, thr) ? keep_scale : 0.f;
            if (mc[i]) av = -1e18f;
            if (t >= Lk) av = -INFINITY;
            gv *= mul;
            if (c == 0 && t < Lk) {
                s[t] = av;
                dpv[t] = gv;
                pdv[t] = mul;
            }
            a[i] = av;
            g[i] = gv;
            cm = fmaxf(cm, av);
        }
        const float m_new = fmaxf(m, cm);
        const float alpha = expf(m - m_new);
        m = m_new;
        l *= alpha;
        rn *= alpha;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            A[e] *= alpha;
            Bv[e] *= alpha;
        }
#pragma unroll
        for (int i = 0; i < AT_NL; ++i) {
            if (t0 + i * RPL >= Lk) break;
            const float ex = expf(a[i] - m);
            l += ex;
            rn += ex * g[i];
            const float ek = mc[i] ? 0.f : ex;          // masked_fill: no gradient to a masked score
            const float wk = ek * g[i];
            axpy16<T>(A, wk, kc[i]);
            axpy16<T>(Bv, ek, kc[i]);
        }
        t0 += RC;
        if (t0 >= Lk) break;
        load_kv(kc, vc, mc, t0);
    }
    // the later steps' q / dctx rows: first chunk requested before the merge and the slab arithmetic
    V qa[AT_NL], da[AT_NL];
    if (gather) {
        load_rows(qa, qp, q_ssb, 0, n_rows);
        load_rows(da, dcp, dctx_ssb, 0, n_rows);
    }
    // merge the row groups
    const float M = cross_group_max<LS>(m);
    const float f = expf(m - M);
    float lr[2] = {l * f, rn * f};
    cross_group_sum<LS, 2>(lr);
    const float inv = 1.0f / lr[0];
    const float rr = lr[1] * inv;
    {
        float o[E];
        const float w = f * inv * scale;
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = (A[e] - rr * Bv[e]) * w;
        if constexpr (LS == 3 && E == 8) {
            const float x = group8_transpose_sum(o, lane);
            if (active) dq[(int64_t)b * dq_bs + h * dh + c * E + group8_elem(lane)] = from_f<T>(x);
        } else {
            cross_group_sum<LS, E>(o);
            if (active && r == 0) R::store(dq + (int64_t)b * dq_bs + h * dh + c * E, o);
        }
    }
    wave_lds_sync();
    for (int t = lane; t < Lk; t += 64) {
        const float av = s[t];
        const float p = expf(av - M) * inv;
        const float pm = p * pdv[t];
        float ds = p * (dpv[t] - rr) * scale;
        if (av == -1e18f) ds = 0.f;              // masked_fill: no gradient to the score
        s[t] = ds;
        pdv[t] = pm;
        if (active) {
            if (gather) {
                sl[((int64_t)t * n_steps + step) * slab_bhs] = make_float2(ds, pm);
            } else {
                dso[t] = ds;
                pdo[t] = pm;
            }
        }
    }
    if (!gather) return;
    wave_lds_sync();
    const float ds_own = s[step];
    const float pd_own = pdv[step];
    wave_lds_sync();
    if (lane == 0) {                             // coefficient of row `step` in the slab of step step + i
        dpv[0] = ds_own;
        pdv[0] = pd_own;
    } else if (lane < n_rows) {
        dpv[lane] = ck0;
        pdv[lane] = cv0;
    }
    for (int i = lane + 64; i < n_rows; i += 64) {
        const float2 cf = sl[((int64_t)step * n_steps + step + i) * slab_bhs];
        dpv[i] = cf.x;
        pdv[i] = cf.y;
    }
    wave_lds_sync();
    // dK[step] = sum_i ds_{step+i}[step] q_{step+i} ; dV[step] = sum_i pd_{step+i}[step] dctx_{step+i}
    float gk[E], gv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) gk[e] = gv[e] = 0.f;
    for (int t0 = 0;;) {
#pragma unroll
        for (int i = 0; i < AT_NL; ++i) {
            if (t0 + i * RPL >= n_rows) break;
            const int t = t0 + i * RPL + r;
            const float wk = t < n_rows ? dpv[t] : 0.f;
            const float wv = t < n_rows ? pdv[t] : 0.f;
            axpy16<T>(gk, wk, qa[i]);
            axpy16<T>(gv, wv, da[i]);
        }
        t0 += RC;
        if (t0 >= n_rows) break;
        load_rows(qa, qp, q_ssb, t0, n_rows);
        load_rows(da, dcp, dctx_ssb, t0, n_rows);
    }
    if constexpr (LS == 3 && E == 8) {
        const float ok = group8_transpose_sum(gk, lane), ov = group8_transpose_sum(gv, lane);
        if (active) {
            const int64_t o = (int64_t)b * out_bs + h * dh + c * E + group8_elem(lane);
            dk_out[o] = from_f<T>(ok);
            dv_out[o] = from_f<T>(ov);
        }
    } else {
        cross_group_sum<LS, E>(gk);
        cross_group_sum<LS, E>(gv);
        if (active && r == 0) {
            R::store(dk_out + (int64_t)b * out_bs + h * dh + c * E, gk);
            R::store(dv_out + (int64_t)b * out_bs + h * dh + c * E, gv);
        }
    }
}

// Gradient of the memory keys / values of a decode sweep: the backward leaves, per step s, the score gradients ds_s and
// the dropped probabilities pd_s of every (sentence, head) in the slabs ds_ctx [n_layers][2][T][B*H][Ls] (float); the
// keys' and values' gradients are their contraction over the steps with the saved queries / context gradients,
//     dK[b, j, l, h, :] = sum_s ds_s[b,h,j] qc_s[b,h,:] ,   dV[b, j, l, h, :] = sum_s pd_s[b,h,j] dctx2_s[b,h,:] .
// One wave per (layer, K|V, sentence, head): lane <-> feature, 32 keys of accumulators per pass, the step's 32
// coefficients fetched by the lanes and broadcast through the scalar lane-read path.  One launch for all layers
// (before: 2 x n_layers batched products of [Ls, T] x [T, dh] tiles plus a float -> bf16 copy of the slabs).
template <typename T>
__global__ __launch_bounds__(256) void ctx_kv_grad_kernel(const float* __restrict__ ds, const T* __restrict__ qc,
                                                          int64_t qc_ls, const T* __restrict__ dctx, int64_t dctx_ls,
                                                          T* __restrict__ out, int NL, int Tn, int B, int b0, int H,
                                                          int dh, int Ls) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t id = (int64_t)blockIdx.x * 4 + wave;
    const int64_t total = (int64_t)NL * 2 * (B - b0) * H;
    if (id >= total) return;
    const int h = (int)(id % H);
    const int b = b0 + (int)((id / H) % (B - b0));
    const int which = (int)((id / ((int64_t)H * (B - b0))) % 2);
    const int l = (int)(id / ((int64_t)H * (B - b0) * 2));
    const int d = H * dh;
    const T* rhs = (which ? dctx + (int64_t)l * dctx_ls : qc + (int64_t)l * qc_ls) + (int64_t)b * d + h * dh +
                   (lane < dh ? lane : dh - 1);
    const float* cf = ds + ((((int64_t)l * 2 + which) * Tn) * B * H + (int64_t)b * H + h) * Ls;
    T* o = out + ((((int64_t)b * Ls) * NL + l) * 2 + which) * d + h * dh + lane;
    for (int j0 = 0; j0 < Ls; j0 += 32) {
        float acc[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = 0.f;
        const int jl = j0 + (lane & 31) < Ls ? j0 + (lane & 31) : Ls - 1;
        for (int t = 0; t < Tn; ++t) {
            const float c = cf[(int64_t)t * B * H * Ls + jl];
            const float r = to_f(rhs[(int64_t)t * B * d]);
#pragma unroll
            for (int j = 0; j < 32; ++j) acc[j] += lane_read(c, j) * r;
        }
        if (lane < dh) {
#pragma unroll
            for (int j = 0; j < 32; ++j)
                if (j0 + j < Ls) o[(int64_t)(j0 + j) * NL * 2 * d] = from_f<T>(acc[j]);
        }
    }
}

// x0 = dropout(e * s + pe[step])   (Embeddings soft path epilogue, onmt/modules/embeddings.py:48-53)
template <typename T>
__global__ __launch_bounds__(256) void embed_epilogue_kernel(const T* __restrict__ e, const float* __restrict__ pe,
                                                             T* __restrict__ out, int rows, int d, int step, float s,
                                                             float p, uint64_t seed, int drop_row0,
                                                             const uint64_t* __restrict__ seed_base) {
    if (seed_base) seed += seed_word(seed_base);
    const float sc = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const uint32_t thr = drop_thr16(p);
    const int64_t n = (int64_t)rows * d;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
        const uint32_t row = (uint32_t)i / (uint32_t)d;      // rows * d < 2^31 (a step's [B, d] slab): 32-bit division
        const int c = (int)((uint32_t)i - row * (uint32_t)d);
        float v[4], pp[4], o[4];
        V4<T>::load(e + i, v);
        V4<float>::load(pe + (int64_t)step * d + c, pp);
        uint32_t kp[2] = {3u, 3u};          // i is a multiple of 4: two whole decision pairs
        const bool drop = p > 0.f && (int)row >= drop_row0;
        if (drop) {
            kp[0] = keep16_pair(seed, (uint64_t)i >> 1, thr);
            kp[1] = keep16_pair(seed, ((uint64_t)i >> 1) + 1, thr);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = v[j] * s + pp[j];
            o[j] = drop ? (((kp[j >> 1] >> (j & 1)) & 1u) ? t * sc : 0.f) : t;
        }
        V4<T>::store(out + i, o);
    }
}

// Greedy search, steps > 0: the input is a token id, so its embedding is a row of the table, not a one-hot product
// (onmt/modules/embeddings.py:282 lookup path): x0[r,:] = E[ids[r],:] * s + pe[step]  (eval mode: no dropout).
// Bit-identical to the one-hot GEMM it replaces: that product is the table row in the compute dtype.
template <typename T>
__global__ __launch_bounds__(256) void embed_lookup_kernel(const T* __restrict__ emb, const int* __restrict__ ids,
                                                           const float* __restrict__ pe, T* __restrict__ out, int rows,
                                                           int d, int V, int step, float s) {
    const int64_t n = (int64_t)rows * d;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
        const int r = (int)(i / d), c = (int)(i % d);
        int id = ids[r];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        float v[4], pp[4], o[4];
        V4<T>::load(emb + (int64_t)id * d + c, v);
        V4<float>::load(pe + (int64_t)step * d + c, pp);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = v[j] * s + pp[j];
        V4<T>::store(out + i, o);
    }
}

// Greedy search: ids[row] = argmax of the row's logits (lowest index on ties, as torch.topk(1) / argmax;
// onmt/translate/greedy_search.py:135-165 with beam 1).  softmax is monotone, so no probabilities are formed.
template <typename T>
__global__ __launch_bounds__(256) void vocab_argmax_kernel(const T* __restrict__ logits, int64_t ldl, int n,
                                                           int* __restrict__ ids) {
    __shared__ float sm[4];
    __shared__ int si[4];
    const int t = threadIdx.x, row = blockIdx.x;
    float mx = -INFINITY;
    int mi = 0x7fffffff;
    for (int c = t; c < n; c += 256) {
        const float v = to_f(logits[(int64_t)row * ldl + c]);
        if (v > mx) { mx = v; mi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
    }
    if ((t & 63) == 0) { sm[t >> 6] = mx; si[t >> 6] = mi; }
    __syncthreads();
    if (t == 0) {
        mx = sm[0]; mi = si[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sm[w] > mx || (sm[w] == mx && si[w] < mi)) { mx = sm[w]; mi = si[w]; }
        ids[row] = mi == 0x7fffffff ? 0 : mi;
    }
}

// probs = softmax(logits * inv_t): fp32 result (the user-visible [B,V] slab) + a copy in the
// compute dtype into the next step's padded GEMM operand row.  One 256-thread block per row.
// greedy: the next input is the one-hot of the row's argmax instead (lowest index on ties, as
// torch.topk(1) / argmax) and the index goes to ids[row]: the greedy search of
// onmt/translate/greedy_search.py:135-165 with beam 1.
// NT threads per row: 256, or 1024 for vocabulary-sized rows (sixteen waves per CU instead of four: 17.7 -> ~8 us at
// 240 x 9606)
template <typename T, int NE, int NT = 256>
__global__ __launch_bounds__(NT) void vocab_softmax_kernel(const T* __restrict__ logits, int64_t ldl,
                                                           float* __restrict__ probs, int64_t ldp,
                                                           T* __restrict__ copy, int64_t ldc, int n, float inv_t,
                                                           int greedy, int* __restrict__ ids, int copy_row0) {
    constexpr int NW = NT / 64;
    __shared__ float sm[NW];
    __shared__ int si[NW];
    const int t = threadIdx.x, row = blockIdx.x;
    float v[NE];
    float mx = -INFINITY;
    int mi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = t + i * NT;
        v[i] = c < n ? to_f(logits[(int64_t)row * ldl + c]) * inv_t : -INFINITY;
        if (v[i] > mx) { mx = v[i]; mi = c; }
    }
    // (max, lowest index attaining it) over the wave, then over the 4 waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
    }
    if ((t & 63) == 0) { sm[t >> 6] = mx; si[t >> 6] = mi; }
    __syncthreads();
    mx = sm[0]; mi = si[0];
#pragma unroll
    for (int w = 1; w < NW; ++w)
        if (sm[w] > mx || (sm[w] == mx && si[w] < mi)) { mx = sm[w]; mi = si[w]; }
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        v[i] = (t + i * NT) < n ? expf(v[i] - mx) : 0.f;
        sum += v[i];
    }
    sum = wave_sum(sum);
    if ((t & 63) == 0) sm[t >> 6] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += sm[w];
    const float inv = 1.0f / tot;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = t + i * NT;
        if (c < n) {
            const float p = v[i] * inv;
            probs[(int64_t)row * ldp + c] = p;
            if (copy && row >= copy_row0) copy[(int64_t)row * ldc + c] = from_f<T>(greedy ? (c == mi ? 1.0f : 0.f) : p);
        }
    }
    if (greedy && ids && t == 0) ids[row] = mi;
}

// One chunk of the batch (rows b0 .. b0+Bc of every [B, .] slab) on its own stream: sentences are
// independent, every kernel of a decode step is latency-bound and fills a fraction of the 256 CUs,
// so the engine runs several chunks' dependency chains concurrently.
struct Ctx : GemmCtx {
    const mssr_decode_plan* p;
    int b0, Bc, chunk;
    int eval_rows;      // forward: rows of this chunk below it skip dropout (mixed eval / train sweep)
    int hash_row_off;   // backward over a row slice of an unchunked forward: offset of the slice in the forward's rows
    // dropout seed of the call = seed_val + *seed_base (seed_base null: seed_val alone).  Eager sweeps carry the plan's
    // seed by value; a sweep recorded for hipGraph replay carries 0 and reads the seed from memory, so one recording
    // serves every later call of the same plan shape
    uint64_t seed_val;
    const uint64_t* seed_base;
};

static inline uint64_t site_seed(const Ctx& c, int step, int layer, int site) {
    return c.seed_val + (uint64_t)step * 0x100000001B3ull + (uint64_t)(layer + 1) * 0x9E3779B1ull +
           (uint64_t)site * 7919ull + (uint64_t)c.chunk * 0xC2B2AE3D27D4EB4Full;
}

// log2(lanes per row) when the tiled kernels apply: dh = E * 2^k with k = 1..4, every pointer 16-byte aligned and every
// stride a multiple of E elements (E = elements per 16 bytes); -1 otherwise (one-row-per-lane kernels)
template <typename T>
static int tiled_shift(int dh, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> strides) {
    static const bool enabled = [] {
        const char* e = getenv("MSSR_ATTN_STEP_TILED");
        return !(e && e[0] == '0');
    }();
    if (!enabled) return -1;
    constexpr int E = 16 / (int)sizeof(T);
    if (dh % E != 0) return -1;
    const int lpr = dh / E;
    if (lpr < 1 || lpr > 64 || (lpr & (lpr - 1)) != 0) return -1;
    for (const void* q : ptrs)
        if (q && !aligned(q, 16)) return -1;
    for (int64_t st : strides)
        if (st % E != 0) return -1;
    int sh = 0;
    while ((1 << sh) < lpr) ++sh;
    return sh;
}

template <typename T>
static void launch_attn_fwd(const Ctx& c, const void* q, int64_t q_bs, const void* k, const void* v, int64_t kv_bs,
                            int64_t kv_ts, const uint8_t* mask, int64_t mask_bs, void* ctx, int64_t ctx_bs, int Lk,
                            float p_drop, uint64_t seed) {
    const mssr_decode_plan* p = c.p;
    const int dh = p->d / p->heads;
    const int bh = c.Bc * p->heads;
    const int sh = tiled_shift<T>(dh, {q, k, v, ctx}, {q_bs, kv_bs, kv_ts, ctx_bs});
#define MSSR_FWD_TILED_N(LS, NL)                                                                                    \
    attn_step_fwd_tiled_kernel<T, LS, NL><<<dim3((bh + 3) / 4), 256, 0, c.s>>>(                                    \
        (const T*)q, q_bs, (const T*)k, (const T*)v, kv_bs, kv_ts, mask, mask_bs, (T*)ctx, ctx_bs, c.Bc, p->heads, dh, \
        Lk, 1.0f / sqrtf((float)dh), p_drop, seed, c.eval_rows, c.seed_base)
#define MSSR_FWD_TILED(LS)                            \
    do {                                              \
        MSSR_FWD_TILED_N(LS, 4);                      \
    } while (0)
    switch (sh) {
        case 1: MSSR_FWD_TILED(1); return;
        case 2: MSSR_FWD_TILED(2); return;
        case 3: MSSR_FWD_TILED(3); return;
        case 4: MSSR_FWD_TILED(4); return;
        default: break;
    }
#undef MSSR_FWD_TILED
#undef MSSR_FWD_TILED_N
    attn_step_fwd_kernel<T><<<dim3((bh + 3) / 4), 256, 0, c.s>>>(
        (const T*)q, q_bs, (const T*)k, (const T*)v, kv_bs, kv_ts, mask, mask_bs, (T*)ctx, ctx_bs, c.Bc, p->heads, dh,
        Lk, 1.0f / sqrtf((float)dh), p_drop, seed, c.eval_rows, c.seed_base);
}

template <typename T>
static void launch_attn_bwd(const Ctx& c, const void* q, int64_t q_bs, const void* k, const void* v, int64_t kv_bs,
                            int64_t kv_ts, const uint8_t* mask, int64_t mask_bs, const void* dctx, int64_t dctx_bs,
                            void* dq, int64_t dq_bs, float* ds_out, float* pd_out, int64_t slab_bhs, int gather,
                            int step, int64_t slab_ss, int64_t q_ss, int64_t dctx_ss, void* dk_out, void* dv_out,
                            int64_t out_bs, int Lk, float p_drop, uint64_t seed) {
    const mssr_decode_plan* p = c.p;
    const int dh = p->d / p->heads;
    const int bh = c.Bc * p->heads;
    const int sh = tiled_shift<T>(dh, {q, k, v, dctx, dq, dk_out, dv_out},
                                  {q_bs, kv_bs, kv_ts, dctx_bs, dq_bs, q_ss, dctx_ss, out_bs});
    // the streaming kernel's lanes carry 32-bit byte offsets from per-(sentence, head) bases
    const int64_t es = (int64_t)sizeof(T), lim = (int64_t)1 << 32;
    const bool offsets32 = (int64_t)Lk * kv_ts * es < lim && (!gather || ((int64_t)p->T * q_ss * es < lim &&
                                                                          (int64_t)p->T * dctx_ss * es < lim));
#define MSSR_BWD_TILED_N(LS, NL)                                                                                    \
    attn_step_bwd_tiled_kernel<T, LS, NL><<<dim3((bh + 3) / 4), 256, 0, c.s>>>(                                    \
        (const T*)q, q_bs, (const T*)k, (const T*)v, kv_bs, kv_ts, mask, mask_bs, (const T*)dctx, dctx_bs, (T*)dq,    \
        dq_bs, ds_out, pd_out, slab_bhs, gather, step, p->T, slab_ss, q_ss, dctx_ss, (T*)dk_out, (T*)dv_out, out_bs,  \
        c.Bc, p->heads, dh, Lk, 1.0f / sqrtf((float)dh), p_drop, seed, c.hash_row_off * p->heads, c.seed_base)
#define MSSR_BWD_STREAM(LS)                                                                                         \
    attn_step_bwd_stream_kernel<T, LS, 4><<<dim3((bh + 3) / 4), 256, 0, c.s>>>(                                     \
        (const T*)q, q_bs, (const T*)k, (const T*)v, kv_bs, kv_ts, mask, mask_bs, (const T*)dctx, dctx_bs, (T*)dq,    \
        dq_bs, ds_out, pd_out, slab_bhs, gather, step, p->T, slab_ss, q_ss, dctx_ss, (T*)dk_out, (T*)dv_out, out_bs,  \
        c.Bc, p->heads, dh, Lk, 1.0f / sqrtf((float)dh), p_drop, seed, c.hash_row_off * p->heads, c.seed_base)
#define MSSR_BWD_TILED(LS)                                                              \
    do {                                                                                \
        const int need = gather && p->T - step > Lk ? p->T - step : Lk;                 \
        if (need <= 4 * (64 >> LS)) MSSR_BWD_TILED_N(LS, 4);                            \
        else if (offsets32) MSSR_BWD_STREAM(LS);                                        \
        else MSSR_BWD_TILED_N(LS, 8);                                                   \
    } while (0)
    switch (sh) {
        case 1: MSSR_BWD_TILED(1); return;
        case 2: MSSR_BWD_TILED(2); return;
        case 3: MSSR_BWD_TILED(3); return;
        case 4: MSSR_BWD_TILED(4); return;
        default: break;
    }
#undef MSSR_BWD_TILED
#undef MSSR_BWD_TILED_N
#undef MSSR_BWD_STREAM
    attn_step_bwd_kernel<T><<<dim3((bh + 3) / 4), 256, 0, c.s>>>(
        (const T*)q, q_bs, (const T*)k, (const T*)v, kv_bs, kv_ts, mask, mask_bs, (const T*)dctx, dctx_bs, (T*)dq,
        dq_bs, ds_out, pd_out, slab_bhs, gather, step, p->T, slab_ss, q_ss, dctx_ss, (T*)dk_out, (T*)dv_out, out_bs,
        c.Bc, p->heads, dh, Lk, 1.0f / sqrtf((float)dh), p_drop, seed, c.hash_row_off * p->heads, c.seed_base);
}

#define DT_CALL(c, fn, ...)                                   \
    do {                                                      \
        if ((c).dt == MSSR_F32) fn<float>(__VA_ARGS__);       \
        else fn<bf16>(__VA_ARGS__);                           \
    } while (0)

static int check_plan(const mssr_decode_plan* p, int step, const char* who) {
    MSSR_REQUIRE(p, "%s: null plan", who);
    MSSR_REQUIRE(valid_dtype(p->dtype), "%s: bad dtype", who);
    MSSR_REQUIRE(p->B > 0 && p->d > 0 && p->heads > 0 && p->d % p->heads == 0 && (p->d / p->heads) % 4 == 0 &&
                     p->d % 8 == 0 && p->ff % 8 == 0 && p->n_layers > 0 && p->V > 0 && p->Vp >= p->V &&
                     p->Vp % 32 == 0 && p->Ls > 0 && p->T > 0,
                 "%s: bad dimensions", who);
    MSSR_REQUIRE(step >= 0 && step < p->T, "%s: step %d outside [0, %d)", who, step, p->T);
    MSSR_REQUIRE(p->T <= AT_MAXK && p->Ls <= AT_MAXK, "%s: more than %d keys", who, AT_MAXK);
    MSSR_REQUIRE(p->V <= 16384, "%s: vocabulary above 16384", who);
    MSSR_REQUIRE(p->d <= 4096, "%s: model dim above 4096", who);
    MSSR_REQUIRE(p->layers && p->emb_t && p->pe && p->w_gen && p->b_gen && p->lnF_g && p->lnF_b && p->mem_kv &&
                     p->src_mask && p->in_all && p->probs && p->X && p->qkv && p->actD && p->actF && p->stats &&
                     p->nf && p->logits && p->tmp,
                 "%s: null buffer in plan", who);
    MSSR_REQUIRE(p->temperature > 0.f, "%s: temperature must be positive", who);
    MSSR_REQUIRE(!p->greedy || (!p->save && p->ids), "%s: greedy decoding is forward-only and needs the ids buffer", who);
    if (p->fuse_ln) {
        MSSR_REQUIRE(p->dtype == MSSR_BF16 && p->d % 32 == 0 && p->d <= 2048,
                     "%s: fuse_ln needs bf16 and a model dim that is a multiple of 32, <= 2048", who);
        MSSR_REQUIRE(p->wg_gen && p->cs_gen && p->bf_gen, "%s: fuse_ln needs the folded generator weights", who);
        for (int l = 0; l < p->n_layers; ++l) {
            const mssr_dec_layer& L = p->layers[l];
            MSSR_REQUIRE(L.wg_qkv && L.wg_cq && L.wg_1 && L.cs_qkv && L.cs_cq && L.cs_1 && L.bf_qkv && L.bf_cq && L.bf_1,
                         "%s: fuse_ln needs the folded weights of layer %d", who, l);
        }
    }
    MSSR_REQUIRE(p->n_chunks >= 1 && p->n_chunks <= MAX_CHUNKS, "%s: n_chunks must be in [1, %d]", who, MAX_CHUNKS);
    MSSR_REQUIRE(p->p_drop >= 0.f && p->p_drop < 1.f && p->p_attn >= 0.f && p->p_attn < 1.f, "%s: bad dropout", who);
    MSSR_REQUIRE(p->eval_rows >= 0 && p->eval_rows <= p->B && p->bwd_row0 >= 0 && p->bwd_row0 < p->B,
                 "%s: eval_rows / bwd_row0 outside the batch", who);
    MSSR_REQUIRE((p->eval_rows == 0 && p->bwd_row0 == 0) || (p->n_chunks == 1 && p->eval_rows <= p->bwd_row0),
                 "%s: a mixed eval / train sweep needs n_chunks = 1 and eval_rows <= bwd_row0", who);
    MSSR_REQUIRE(p->late_row0 >= 0 && p->late_row0 < p->B && p->late_step >= 0 && p->late_step <= p->T &&
                     (p->late_row0 == 0 || p->n_chunks == 1),
                 "%s: late_row0 / late_step outside the batch / the steps (or with n_chunks > 1)", who);
    MSSR_REQUIRE(p->forced_rows >= 0 && p->forced_rows <= p->B && p->forced_steps >= 0 && p->forced_steps <= p->T,
                 "%s: forced_rows / forced_steps outside the batch / the steps", who);
    MSSR_REQUIRE(p->forced_rows == 0 || (!p->greedy && p->eval_rows == 0 && p->bwd_row0 == 0),
                 "%s: teacher-forced sentences cannot be combined with greedy decoding or eval_rows", who);
    return MSSR_OK;
}

// Slab geometry (elements), shared by forward and backward.  Ts = save ? T : 1, slot = save ? step : 0.
//   X     [n_layers+1][Ts][B,d]      residual stream entering layer l (l = n_layers: decoder output)
//   qkv   [n_layers][B][T,3d]        packed self-attention projections = the K/V cache (sentence-major)
//   actD  [n_layers][8][Ts][B,d]     n1 | ctx1 | h1 | n2 | qc | ctx2 | h2 | n3
//   actF  [n_layers][Ts][B,ff]       FFN inner activation (after ReLU and dropout)
//   stats [3*n_layers+1][2][Ts][B]   LayerNorm mean, rstd (float)
//   nf    [Ts][B,d]                  final LayerNorm output
// Every quantity is contiguous over (step, batch), so the caller contracts weight gradients over
// all steps with ONE product per parameter.
enum { K_N1 = 0, K_CTX1, K_H1, K_N2, K_QC, K_CTX2, K_H2, K_N3, K_NKINDS };
enum { G_DN1 = 0, G_DSO, G_DN2, G_DCQ, G_DCO, G_DN3, G_DF2, G_NKINDS };

// A step block of a slab holds B rows; this chunk owns rows r0 .. r0+R of it.
struct Geo {
    int B, R, r0, d, ff, NL, T, Ts, slot, V, Vp, Ls;
    size_t es;
    const mssr_decode_plan* p;
    Geo(const Ctx& c, int step)
        : B(c.p->B), R(c.Bc), r0(c.b0), d(c.p->d), ff(c.p->ff), NL(c.p->n_layers), T(c.p->T),
          Ts(c.p->save ? c.p->T : 1), slot(c.p->save ? step : 0), V(c.p->V), Vp(c.p->Vp), Ls(c.p->Ls), es(c.es),
          p(c.p) {}
    // rows of block `blk` of a slab whose rows are `w` elements wide
    char* rows(void* base, int64_t blk, int64_t w) const { return at(base, (blk * B + r0) * w, es); }
    char* X(int l) const { return rows(p->X, (int64_t)l * Ts + slot, d); }
    char* act(int l, int kind) const { return rows(p->actD, ((int64_t)l * K_NKINDS + kind) * Ts + slot, d); }
    char* inter(int l) const { return rows(p->actF, (int64_t)l * Ts + slot, ff); }
    float* stat(int idx, int which) const { return p->stats + (((int64_t)idx * 2 + which) * Ts + slot) * B + r0; }
    // row of sentence r0 at `step`; sentences are qkv_bs() elements apart, steps 3d
    char* qkv(int l, int step) const { return at(p->qkv, (((int64_t)l * B + r0) * T + step) * 3 * d, es); }
    int64_t qkv_bs() const { return (int64_t)T * 3 * d; }
    char* dact(int l, int kind, int step) const { return rows(p->dactD, ((int64_t)l * G_NKINDS + kind) * T + step, d); }
    char* dinter(int l, int step) const { return rows(p->dactF, (int64_t)l * T + step, ff); }
    char* dqkv(int l, int step) const { return rows(p->dqkv, (int64_t)l * T + step, 3 * d); }
    char* dctx(int l, int which, int step) const { return rows(p->dctx, ((int64_t)l * 2 + which) * T + step, d); }
    // per-step score-gradient slabs [n_layers][2 = ds|pd][T][B*H][W], W = T (self) or Ls (context)
    float* ds_self(int l) const {               // pairs [key][step][B * heads] of layer l, first (sentence, head) of this chunk
        return p->ds_self + (((int64_t)l * T * T * B + r0) * p->heads) * 2;
    }
    float* ds_ctx(int l, int which, int step) const {
        return p->ds_ctx + ((((int64_t)l * 2 + which) * T + step) * B + r0) * p->heads * Ls;
    }
    const char* mem_kv(int l) const { return at((void*)p->mem_kv, ((int64_t)r0 * Ls * NL + l) * 2 * d, es); }
    const uint8_t* src_mask() const { return p->src_mask + (int64_t)r0 * Ls; }
    const uint8_t* self_mask() const { return p->self_mask ? p->self_mask + (int64_t)r0 * T : nullptr; }
    char* tmp() const { return at(p->tmp, (int64_t)r0 * d, es); }
    char* logits() const { return at(p->logits, (int64_t)r0 * Vp, es); }
    char* dX(int which) const { return at(p->dX, ((int64_t)which * B + r0) * d, es); }     // two buffers, used in turn
};

static int step_fwd(const Ctx& c, int step) {
    const mssr_decode_plan* p = c.p;
    const Geo G(c, step);
    const int B = G.B, R = G.R, d = G.d, ff = G.ff, NL = G.NL;
    const size_t es = c.es;
    const int64_t Rd = (int64_t)R * d;
    const float pd = p->training ? p->p_drop : 0.f, pa = p->training ? p->p_attn : 0.f;
    const bool fuse = p->fuse_ln != 0;
    int rc;
    // embedding of the (soft) input: e = in[step] . E ; x0 = dropout(e*sqrt(d) + pe[step])
    const void* in = G.rows(p->in_all, step, p->Vp);
    if (p->greedy && step > 0) {        // the input is last step's token: a row of the table
        const unsigned g = (unsigned)((Rd / 4 + 255) / 256);
        const int* prev = p->ids + (int64_t)(step - 1) * B + G.r0;
        if (c.dt == MSSR_F32)
            embed_lookup_kernel<float><<<g, 256, 0, c.s>>>((const float*)p->emb, prev, p->pe, (float*)G.X(0), R, d, p->V,
                                                           step, sqrtf((float)d));
        else
            embed_lookup_kernel<bf16><<<g, 256, 0, c.s>>>((const bf16*)p->emb, prev, p->pe, (bf16*)G.X(0), R, d, p->V, step,
                                                          sqrtf((float)d));
    } else {
    if ((rc = gemm_nt(c, in, p->Vp, p->emb_t, p->Vp, G.tmp(), d, R, d, p->Vp, nullptr, 0, nullptr, 0, nullptr, 0, 1.f,
                      0.f, c.dt, 0.f, 0)))
        return rc;
    {
        const unsigned g = (unsigned)((Rd / 4 + 255) / 256);
        if (c.dt == MSSR_F32)
            embed_epilogue_kernel<float><<<g, 256, 0, c.s>>>((const float*)G.tmp(), p->pe, (float*)G.X(0), R, d, step,
                                                             sqrtf((float)d), pd, site_seed(c, step, -1, 0), c.eval_rows, c.seed_base);
        else
            embed_epilogue_kernel<bf16><<<g, 256, 0, c.s>>>((const bf16*)G.tmp(), p->pe, (bf16*)G.X(0), R, d, step,
                                                            sqrtf((float)d), pd, site_seed(c, step, -1, 0), c.eval_rows, c.seed_base);
    }
    }
    for (int l = 0; l < NL; ++l) {
        const mssr_dec_layer& L = p->layers[l];
        char* x = G.X(l);
        char *n1 = G.act(l, K_N1), *ctx1 = G.act(l, K_CTX1), *h1 = G.act(l, K_H1), *n2 = G.act(l, K_N2),
             *qc = G.act(l, K_QC), *ctx2 = G.act(l, K_CTX2), *h2 = G.act(l, K_H2), *n3 = G.act(l, K_N3);
        char* inter = G.inter(l);
        char* qkv_0 = G.qkv(l, 0);               // this chunk's rows of step 0: the cache base
        char* qkv_s = G.qkv(l, step);
        // --- self attention block
        if (fuse) {     // LayerNorm folded into the projection (stats written by the GEMM, n1 not materialised)
            if ((rc = gemm_nt(c, x, d, L.wg_qkv, d, qkv_s, G.qkv_bs(), R, 3 * d, d, L.bf_qkv, 0, nullptr, 0, nullptr, 0, 1.f,
                              0.f, c.dt, 0.f, 0, L.cs_qkv, G.stat(l * 3 + 0, 0), G.stat(l * 3 + 0, 1))))
                return rc;
        } else {
            if ((rc = mssr_layernorm_fwd(x, L.ln1_g, L.ln1_b, n1, G.stat(l * 3 + 0, 0), G.stat(l * 3 + 0, 1), R, d,
                                         1e-6f, c.dt, c.s)))
                return rc;
            if ((rc = gemm_nt(c, n1, d, L.w_qkv, d, qkv_s, G.qkv_bs(), R, 3 * d, d, L.b_qkv, 0, nullptr, 0, nullptr, 0, 1.f,
                              0.f, c.dt, 0.f, 0)))
                return rc;
        }
        DT_CALL(c, launch_attn_fwd, c, qkv_s, G.qkv_bs(), qkv_0 + (int64_t)d * es, qkv_0 + 2LL * d * es, G.qkv_bs(),
                (int64_t)3 * d, G.self_mask(), p->T, ctx1, d, step + 1, pa, site_seed(c, step, l, 1));
        if ((rc = gemm_nt(c, ctx1, d, L.w_so, d, h1, d, R, d, d, L.b_so, 0, x, d, nullptr, 0, 1.f, 0.f, c.dt, pd,
                          site_seed(c, step, l, 2), nullptr, nullptr, nullptr, c.eval_rows)))
            return rc;
        // --- context attention block (keys / values of the memory bank were projected once)
        if (fuse) {
            if ((rc = gemm_nt(c, h1, d, L.wg_cq, d, qc, d, R, d, d, L.bf_cq, 0, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt,
                              0.f, 0, L.cs_cq, G.stat(l * 3 + 1, 0), G.stat(l * 3 + 1, 1))))
                return rc;
        } else {
            if ((rc = mssr_layernorm_fwd(h1, L.ln2_g, L.ln2_b, n2, G.stat(l * 3 + 1, 0), G.stat(l * 3 + 1, 1), R, d,
                                         1e-6f, c.dt, c.s)))
                return rc;
            if ((rc = gemm_nt(c, n2, d, L.w_cq, d, qc, d, R, d, d, L.b_cq, 0, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt,
                              0.f, 0)))
                return rc;
        }
        const char* mkv = G.mem_kv(l);                                      // [B, Ls, NL, 2d], layer l
        DT_CALL(c, launch_attn_fwd, c, qc, d, mkv, mkv + (int64_t)d * es, (int64_t)p->Ls * NL * 2 * d,
                (int64_t)NL * 2 * d, G.src_mask(), p->Ls, ctx2, d, p->Ls, pa, site_seed(c, step, l, 3));
        if ((rc = gemm_nt(c, ctx2, d, L.w_co, d, h2, d, R, d, d, L.b_co, 0, h1, d, nullptr, 0, 1.f, 0.f, c.dt, pd,
                          site_seed(c, step, l, 4), nullptr, nullptr, nullptr, c.eval_rows)))
            return rc;
        // --- feed forward block
        if (fuse) {
            if ((rc = gemm_nt(c, h2, d, L.wg_1, d, inter, ff, R, ff, d, L.bf_1, 1, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt,
                              pd, site_seed(c, step, l, 5), L.cs_1, G.stat(l * 3 + 2, 0), G.stat(l * 3 + 2, 1),
                              c.eval_rows)))
                return rc;
        } else {
            if ((rc = mssr_layernorm_fwd(h2, L.lnf_g, L.lnf_b, n3, G.stat(l * 3 + 2, 0), G.stat(l * 3 + 2, 1), R, d,
                                         1e-6f, c.dt, c.s)))
                return rc;
            if ((rc = gemm_nt(c, n3, d, L.w_1, d, inter, ff, R, ff, d, L.b_1, 1, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt,
                              pd, site_seed(c, step, l, 5), nullptr, nullptr, nullptr, c.eval_rows)))
                return rc;
        }
        if ((rc = gemm_nt(c, inter, ff, L.w_2, ff, G.X(l + 1), d, R, d, ff, L.b_2, 0, h2, d, nullptr, 0, 1.f, 0.f, c.dt,
                          pd, site_seed(c, step, l, 6), nullptr, nullptr, nullptr, c.eval_rows)))
            return rc;
    }
    // final LayerNorm, generator Linear, softmax(/T)
    char* nf = G.rows(p->nf, G.slot, d);
    if (fuse) {
        if ((rc = gemm_nt(c, G.X(NL), d, p->wg_gen, d, G.logits(), p->Vp, R, p->V, d, p->bf_gen, 0, nullptr, 0, nullptr,
                          0, 1.f, 0.f, c.dt, 0.f, 0, p->cs_gen, G.stat(NL * 3, 0), G.stat(NL * 3, 1))))
            return rc;
    } else {
        if ((rc = mssr_layernorm_fwd(G.X(NL), p->lnF_g, p->lnF_b, nf, G.stat(NL * 3, 0), G.stat(NL * 3, 1), R, d, 1e-6f,
                                     c.dt, c.s)))
            return rc;
        if ((rc = gemm_nt(c, nf, d, p->w_gen, d, G.logits(), p->Vp, R, p->V, d, p->b_gen, 0, nullptr, 0, nullptr, 0, 1.f,
                          0.f, c.dt, 0.f, 0)))
            return rc;
    }
    if (p->greedy) {                    // the token is all the next step needs (embed_lookup_kernel)
        int* out_ids = p->ids + (int64_t)step * B + G.r0;
        if (c.dt == MSSR_F32) vocab_argmax_kernel<float><<<R, 256, 0, c.s>>>((const float*)G.logits(), p->Vp, p->V, out_ids);
        else vocab_argmax_kernel<bf16><<<R, 256, 0, c.s>>>((const bf16*)G.logits(), p->Vp, p->V, out_ids);
        MSSR_LAUNCH_OK("decode_step_fwd");
    }
    float* probs = p->probs + ((int64_t)step * B + G.r0) * p->V;
    void* copy = G.rows(p->in_all, step + 1, p->Vp);                        // in_all has T+1 slots
    // teacher-forced sentences: their next input was preset by the caller, the step's output must not replace it
    const int keep_rows = step + 1 < p->forced_steps && p->forced_rows > G.r0 ? p->forced_rows - G.r0 : 0;
    const int ne = (p->V + 255) / 256;
#define VS(T, NEc)                                                                                             \
    vocab_softmax_kernel<T, NEc><<<R, 256, 0, c.s>>>((const T*)G.logits(), p->Vp, probs, p->V, (T*)copy, p->Vp, \
                                                     p->V, 1.0f / p->temperature, p->greedy,                    \
                                                     p->ids ? p->ids + (int64_t)step * B + G.r0 : nullptr, keep_rows)
#define VSW(T, NEc)                                                                                                   \
    vocab_softmax_kernel<T, NEc, 1024><<<R, 1024, 0, c.s>>>((const T*)G.logits(), p->Vp, probs, p->V, (T*)copy, p->Vp, \
                                                            p->V, 1.0f / p->temperature, p->greedy,                    \
                                                            p->ids ? p->ids + (int64_t)step * B + G.r0 : nullptr, keep_rows)
    const bool wide = p->V > 4096 && p->V <= 16384 && R <= 2048;      // few vocabulary-wide rows: 1024 threads per row
    if (c.dt == MSSR_F32) {
        if (wide) { if (p->V <= 10240) VSW(float, 10); else VSW(float, 16); }
        else if (ne <= 8) VS(float, 8); else if (ne <= 16) VS(float, 16); else if (ne <= 40) VS(float, 40); else VS(float, 64);
    } else {
        if (wide) { if (p->V <= 10240) VSW(bf16, 10); else VSW(bf16, 16); }
        else if (ne <= 8) VS(bf16, 8); else if (ne <= 16) VS(bf16, 16); else if (ne <= 40) VS(bf16, 40); else VS(bf16, 64);
    }
#undef VSW
#undef VS
    MSSR_LAUNCH_OK("decode_step_fwd");
}

static int step_bwd(const Ctx& c, int step) {
    const mssr_decode_plan* p = c.p;
    const Geo G(c, step);
    const int B = G.B, R = G.R, d = G.d, ff = G.ff, NL = G.NL;
    const size_t es = c.es;
    const float pd = p->training ? p->p_drop : 0.f, pa = p->training ? p->p_attn : 0.f;
    const float keep = pd > 0.f ? 1.0f / (1.0f - pd) : 1.0f;
    int rc;
    // gradient slabs: dactD [n_layers][7][T][B,d] = dn1 | d_so | dn2 | d_cq | d_co | dn3 | d_f2 (gradients of the
    // outputs of LN1, self out-proj, LN2, ctx query proj, ctx out-proj, LN3, W2); dactF [n_layers][T][B,ff] = d_f1
    // LayerNorm backward into the running residual gradient dx, plus dx2 = that gradient pushed through the
    // dropout (site seed) of the block below: the gradient of the block's last Linear output
    auto ln_bwd = [&](const void* dy, const void* x, const float* g, int idx, const void* dres, void* dx, void* dx2,
                      uint64_t seed, float extra) {
        if (c.dt == MSSR_F32)
            ln_bwd_dx_kernel<float><<<(R + 3) / 4, 256, 0, c.s>>>((const float*)dy, (const float*)x, g, G.stat(idx, 0),
                                                                 G.stat(idx, 1), (const float*)dres, (float*)dx,
                                                                 (float*)dx2, R, d, pd, seed, extra, c.hash_row_off, c.seed_base);
        else
            ln_bwd_dx_kernel<bf16><<<(R + 3) / 4, 256, 0, c.s>>>((const bf16*)dy, (const bf16*)x, g, G.stat(idx, 0),
                                                                G.stat(idx, 1), (const bf16*)dres, (bf16*)dx,
                                                                (bf16*)dx2, R, d, pd, seed, extra, c.hash_row_off, c.seed_base);
    };

    // softmax backward: dlogits[step] (compute dtype, padded columns stay zero)
    char* dlg = G.rows(p->dlogits, step, p->Vp);
    const float* probs = p->probs + ((int64_t)step * B + G.r0) * p->V;
    const float* dpr = p->dprobs + ((int64_t)step * B + G.r0) * p->V;
    if ((rc = mssr_softmax_bwd(probs, dpr, dlg, R, p->V, p->V, p->V, p->Vp, 1.0f / p->temperature, 0, MSSR_F32,
                               MSSR_F32, c.dt, c.s)))
        return rc;
    // generator Linear: dnf = dlogits . W_gen   (W as [N=d, K=Vp] = wt_gen)
    char* dnf = G.rows(p->dnf, step, d);
    if ((rc = gemm_nt(c, dlg, p->Vp, p->wt_gen, p->Vp, dnf, d, R, d, p->Vp, nullptr, 0, nullptr, 0, nullptr, 0, 1.f,
                      0.f, c.dt, 0.f, 0)))
        return rc;
    // running gradient of the residual stream [R,d]: two buffers, every LayerNorm backward reads one and writes the
    // other (in the fused form other thread blocks still read the old one while the new one is written)
    char* dxb[2] = {G.dX(0), G.dX(1)};
    int cur = 0;                               // dxb[cur] holds the current gradient (after the first ln step)
    // LayerNorm backward + the Linear backward behind it.  Fused (bf16, d <= 512): ONE launch, the product's A
    // operand dx2 is formed in its prologue (gemm_lnb.hip); otherwise two launches.
    static const bool fuse_env = [] {
        const char* e = getenv("MSSR_DECODE_FUSE_LNB");
        return !(e && e[0] == '0');
    }();
    const bool fuse_lnb = fuse_env && c.dt == MSSR_BF16;
    auto ln_gemm = [&](const void* dy, const void* x, const float* g, int idx, bool first, void* dx2, uint64_t seed,
                       const void* w, int64_t ldw, void* out, int64_t n, const void* gate, int64_t ldg,
                       float alpha) -> int {
        const void* dres = first ? nullptr : dxb[cur];
        void* dxo = first ? dxb[cur] : dxb[cur ^ 1];
        if (!first) cur ^= 1;
        if (fuse_lnb) {
            mssr_gemm_desc ds = {};
            ds.a = dy; ds.b = w; ds.c = out; ds.bias = nullptr; ds.residual = nullptr; ds.gate = gate;
            ds.m = R; ds.n = n; ds.k = d;
            ds.a_rs = d; ds.a_cs = 1; ds.b_rs = 1; ds.b_cs = ldw;
            ds.ldc = n; ds.ldr = 0; ds.ldg = ldg;
            ds.batch_outer = ds.batch_inner = 1;
            ds.a_bo = ds.a_bi = ds.b_bo = ds.b_bi = ds.c_bo = ds.c_bi = 0;
            ds.alpha = alpha; ds.beta = 0.f; ds.relu = 0;
            ds.in_dtype = c.dt; ds.out_dtype = c.dt;
            ds.drop_p = 0.f; ds.drop_seed = 0;
            ds.workspace = nullptr; ds.workspace_bytes = 0; ds.no_skinny = 0;
            ds.ln_colsum = nullptr; ds.ln_mean = nullptr; ds.ln_rstd = nullptr; ds.ln_eps = 0.f; ds.drop_row0 = 0;
            GemmArgs a;
            gemm_fill_args(&ds, a);
            a.lnb_x = x; a.lnb_gamma = g; a.lnb_mean = G.stat(idx, 0); a.lnb_rstd = G.stat(idx, 1);
            a.lnb_dres = dres; a.lnb_dx = dxo; a.lnb_dx2 = dx2;
            a.lnb_p = pd; a.lnb_extra = 1.f; a.lnb_seed = seed; a.lnb_row_off = c.hash_row_off;
            const int pi = prof_begin(c.s);
            const bool done = gemm_lnb_try(a, c.dt, c.s);
            prof_end(pi, c.s, 3, c.dt, c.dt, 32, 1, R, n, d, 1, 2);        // nothing launched: the record keeps ms = -1
            if (done) return MSSR_OK;
        }
        ln_bwd(dy, x, g, idx, dres, dxo, dx2, seed, 1.f);
        return gemm_nt(c, dx2, d, w, ldw, out, n, R, n, d, nullptr, 0, nullptr, 0, gate, ldg, alpha, 0.f, c.dt, 0.f, 0);
    };
    // final LayerNorm -> d_f2 of the last layer -> d_f1 = (d_f2 . W2) * (inter > 0) / (1-p)   [W as [N=ff, K=d] = wt_2]
    if ((rc = ln_gemm(dnf, G.X(NL), p->lnF_g, NL * 3, true, G.dact(NL - 1, G_DF2, step), site_seed(c, step, NL - 1, 6),
                      p->layers[NL - 1].wt_2, d, G.dinter(NL - 1, step), ff, G.inter(NL - 1), ff, keep)))
        return rc;
    for (int l = NL - 1; l >= 0; --l) {
        const mssr_dec_layer& L = p->layers[l];
        char *h1 = G.act(l, K_H1), *qc = G.act(l, K_QC), *h2 = G.act(l, K_H2);
        char *dn1 = G.dact(l, G_DN1, step), *d_so = G.dact(l, G_DSO, step), *dn2 = G.dact(l, G_DN2, step),
             *d_cq = G.dact(l, G_DCQ, step), *d_co = G.dact(l, G_DCO, step), *dn3 = G.dact(l, G_DN3, step);
        char* d_f1 = G.dinter(l, step);
        char* qkv_0 = G.qkv(l, 0);
        char* qkv_s = G.qkv(l, step);
        char* dqkv_s = G.dqkv(l, step);
        // --- feed forward block: y = h2 + drop(W2 inter + b2), inter = drop(relu(W1 n3 + b1)); d_f1 (gradient of
        //     inter) was formed together with the LayerNorm backward above this layer
        if ((rc = gemm_nt(c, d_f1, ff, L.wt_1, ff, dn3, d, R, d, ff, nullptr, 0, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt,
                          0.f, 0)))
            return rc;
        // --- context attention block: h2 = h1 + drop(Wco ctx2 + bco); LN3 backward (dx <- grad wrt h2) + out-proj
        char* dctx2 = G.dctx(l, 1, step);
        if ((rc = ln_gemm(dn3, h2, L.lnf_g, l * 3 + 2, false, d_co, site_seed(c, step, l, 4), L.wt_co, d, dctx2, d, nullptr,
                          0, 1.f)))
            return rc;
        {
            const char* mkv = G.mem_kv(l);
            DT_CALL(c, launch_attn_bwd, c, qc, d, mkv, mkv + (int64_t)d * es, (int64_t)p->Ls * NL * 2 * d,
                    (int64_t)NL * 2 * d, G.src_mask(), p->Ls, dctx2, d, d_cq, d, G.ds_ctx(l, 0, step),
                    G.ds_ctx(l, 1, step), p->Ls, 0, step, 0, 0, 0, nullptr, nullptr, 0, p->Ls, pa,
                    site_seed(c, step, l, 3));
        }
        if ((rc = gemm_nt(c, d_cq, d, L.wt_cq, d, dn2, d, R, d, d, nullptr, 0, nullptr, 0, nullptr, 0, 1.f, 0.f, c.dt,
                          0.f, 0)))
            return rc;
        // --- self attention block: h1 = x + drop(Wso ctx1 + bso); LN2 backward (dx <- grad wrt h1) + out-proj
        char* dctx1 = G.dctx(l, 0, step);
        if ((rc = ln_gemm(dn2, h1, L.ln2_g, l * 3 + 1, false, d_so, site_seed(c, step, l, 2), L.wt_so, d, dctx1, d, nullptr,
                          0, 1.f)))
            return rc;
        DT_CALL(c, launch_attn_bwd, c, qkv_s, G.qkv_bs(), qkv_0 + (int64_t)d * es, qkv_0 + 2LL * d * es, G.qkv_bs(),
                (int64_t)3 * d, G.self_mask(), p->T, dctx1, d, dqkv_s, 3 * d, G.ds_self(l), nullptr,
                (int64_t)B * p->heads, 1, step, 0, (int64_t)3 * d, (int64_t)B * d,
                dqkv_s + (int64_t)d * es, dqkv_s + 2LL * d * es, 3 * d, step + 1, pa, site_seed(c, step, l, 1));
        if ((rc = gemm_nt(c, dqkv_s, 3 * d, L.wt_qkv, 3 * d, dn1, d, R, d, 3 * d, nullptr, 0, nullptr, 0, nullptr, 0, 1.f,
                          0.f, c.dt, 0.f, 0)))
            return rc;
        // LN1 backward: dx <- grad wrt the layer input; below it sits layer l-1's W2 output (its gradient d_f2 and the
        // product d_f1 = (d_f2 . W2) * gate in the same launch), or (l = 0) the embedding x0 = drop(e*sqrt(d) + pe),
        // whose gradient d_e also takes the sqrt(d)
        if (l > 0) {
            if ((rc = ln_gemm(dn1, G.X(l), L.ln1_g, l * 3 + 0, false, G.dact(l - 1, G_DF2, step),
                              site_seed(c, step, l - 1, 6), p->layers[l - 1].wt_2, d, G.dinter(l - 1, step), ff,
                              G.inter(l - 1), ff, keep)))
                return rc;
        } else {
            ln_bwd(dn1, G.X(l), L.ln1_g, l * 3 + 0, dxb[cur], dxb[cur ^ 1], G.rows(p->d_e, step, d),
                   site_seed(c, step, -1, 0), sqrtf((float)d));
            cur ^= 1;
        }
    }
    char* de = G.rows(p->d_e, step, d);
    if (step > 0) {
        // dprobs[step-1] += d_e . E^T      (W as [N=V, K=d] = E itself); teacher-forced sentences took a preset input at
        // this step, not their previous output
        const int skip = step < p->forced_steps ? p->forced_rows - G.r0 : 0;
        const int f0 = skip > 0 ? (skip < R ? skip : R) : 0;
        float* dprev = p->dprobs + ((int64_t)(step - 1) * B + G.r0 + f0) * p->V;
        if (R - f0 > 0 &&
            (rc = gemm_nt(c, de + (int64_t)f0 * d * es, d, p->emb, d, dprev, p->V, R - f0, p->V, d, nullptr, 0, nullptr, 0,
                          nullptr, 0, 1.f, 1.f, MSSR_F32, 0.f, 0)))
            return rc;
    }
    MSSR_LAUNCH_OK("decode_step_bwd");
}

// ---- chunk streams: created once per device, reused by every decode
struct StreamPool {
    bool ready = false;
    hipStream_t s[MAX_CHUNKS];
    hipEvent_t fork, join[MAX_CHUNKS];
};
static StreamPool g_pools[16];

static StreamPool* pool_for_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    StreamPool& P = g_pools[dev];
    if (!P.ready) {
        for (int i = 0; i < MAX_CHUNKS; ++i) {
            if (hipStreamCreateWithFlags(&P.s[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
            if (hipEventCreateWithFlags(&P.join[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
        if (hipEventCreateWithFlags(&P.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        P.ready = true;
    }
    return &P;
}

// Runs steps [s_begin, s_end) (forward) or the same range downwards (backward) for every chunk.
static int run_steps(const mssr_decode_plan* p, int s_begin, int s_end, bool backward, hipStream_t stream,
                     const char* who, const uint64_t* seed_base = nullptr) {
    if (int e = check_plan(p, s_begin, who)) return e;
    MSSR_REQUIRE(s_end > s_begin && s_end <= p->T, "%s: bad step range", who);
    if (backward) {
        MSSR_REQUIRE(p->save, "%s: the forward did not save activations (save = 0)", who);
        MSSR_REQUIRE(p->emb && p->wt_gen && p->dprobs && p->dlogits && p->dX && p->dqkv && p->dactD && p->dactF &&
                         p->dctx && p->ds_self && p->ds_ctx && p->d_e && p->dnf,
                     "%s: null gradient buffer in plan", who);
    }
    // backward of a mixed sweep: ONE chunk = sentences [bwd_row0, B), addressed as a slice of the unchunked forward
    const int row0 = backward ? p->bwd_row0 : 0;
    const int per = row0 ? p->B - row0 : (p->B + p->n_chunks - 1) / p->n_chunks;
    const int nch = row0 ? 1 : (p->B + per - 1) / per;
    Ctx cs[MAX_CHUNKS];
    StreamPool* P = nullptr;
    if (nch > 1) {
        P = pool_for_current_device();
        MSSR_REQUIRE(P, "%s: could not create the chunk streams", who);
        if (hipEventRecord(P->fork, stream) != hipSuccess) { set_error("%s: fork failed", who); return MSSR_ERR_LAUNCH; }
    }
    const int64_t ws_each = p->gemm_ws ? (p->gemm_ws_bytes / nch) & ~(int64_t)255 : 0;
    for (int i = 0; i < nch; ++i) {
        Ctx& c = cs[i];
        c.p = p; c.dt = p->dtype; c.es = dsize(p->dtype);
        c.b0 = row0 + i * per; c.Bc = (c.b0 + per <= p->B) ? per : p->B - c.b0; c.chunk = i;
        c.eval_rows = backward ? 0 : p->eval_rows;
        c.hash_row_off = row0;
        c.seed_val = seed_base ? 0 : p->seed;
        c.seed_base = seed_base;
        c.ws = p->gemm_ws ? (char*)p->gemm_ws + i * ws_each : nullptr;
        c.ws_bytes = ws_each;
        c.s = stream;
        if (nch > 1) {
            c.s = P->s[i];
            (void)hipStreamWaitEvent(c.s, P->fork, 0);
        }
    }
    // every GEMM descriptor filled on this thread while the sweep is issued adds the same word to its seeds
    struct SeedBaseGuard {
        const uint64_t* prev;
        explicit SeedBaseGuard(const uint64_t* b) : prev(tl_seed_base) { tl_seed_base = b; }
        ~SeedBaseGuard() { tl_seed_base = prev; }
    } seed_guard(seed_base);
    int rc = MSSR_OK;
    // steps from late_step on cover sentences [late_row0, B) only: backward (the others' gradients are zero there)
    // and, with late_fwd, forward too (their decodes have ended).  Dropout decisions are indexed by the row inside the
    // launch: a forward that ran on the suffix is replayed with offset 0.
    auto ctx_at = [&](int i, int step) -> Ctx {
        Ctx c = cs[i];
        if (p->late_row0 > c.b0 && step >= p->late_step && (backward || p->late_fwd)) {
            c.Bc = c.b0 + c.Bc - p->late_row0;
            c.b0 = p->late_row0;
            c.hash_row_off = p->late_fwd ? 0 : c.b0;
            c.eval_rows = c.eval_rows > c.b0 ? c.eval_rows - c.b0 : 0;
        }
        return c;
    };
    auto run_chunk = [&](int i) -> int {
        int r = MSSR_OK;
        for (int k = 0; k < s_end - s_begin && r == MSSR_OK; ++k) {
            const int step = backward ? s_end - 1 - k : s_begin + k;
            r = backward ? step_bwd(ctx_at(i, step), step) : step_fwd(ctx_at(i, step), step);
        }
        return r;
    };
    // under stream capture (hipGraph) the chunks are recorded from this thread: interleaved step by step
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    static const bool use_threads = [] { const char* e = getenv("MSSR_DECODE_THREADS"); return !e || atoi(e) != 0; }();
    if (nch == 1) {
        rc = run_chunk(0);
    } else if (cap != hipStreamCaptureStatusNone || !use_threads) {
        for (int k = 0; k < s_end - s_begin && rc == MSSR_OK; ++k) {
            const int step = backward ? s_end - 1 - k : s_begin + k;
            for (int i = 0; i < nch && rc == MSSR_OK; ++i)
                rc = backward ? step_bwd(ctx_at(i, step), step) : step_fwd(ctx_at(i, step), step);
        }
    } else {
        // one host thread per chunk: a launch costs the host about as much as these kernels run, so a
        // single thread could not keep several streams busy
        int dev = 0;
        (void)hipGetDevice(&dev);
        int rcs[MAX_CHUNKS];
        char msgs[MAX_CHUNKS][256];
        std::thread th[MAX_CHUNKS];
        for (int i = 1; i < nch; ++i)
            th[i] = std::thread([&, i]() {
                (void)hipSetDevice(dev);
                tl_seed_base = seed_base;
                rcs[i] = run_chunk(i);
                snprintf(msgs[i], sizeof(msgs[i]), "%s", mssr_last_error());
            });
        rcs[0] = run_chunk(0);
        for (int i = 1; i < nch; ++i) th[i].join();
        for (int i = nch - 1; i >= 0; --i)
            if (rcs[i] != MSSR_OK) {
                rc = rcs[i];
                if (i > 0) set_error("%s", msgs[i]);
            }
    }
    if (nch > 1) {            // join even after an error, so the caller's stream never runs ahead of a chunk
        for (int i = 0; i < nch; ++i) {
            (void)hipEventRecord(P->join[i], cs[i].s);
            (void)hipStreamWaitEvent(stream, P->join[i], 0);
        }
    }
    return rc;
}

// ------------------------------------------------------------------ whole sweeps as hipGraphs
// A sweep is ~2 800 launches that the host issues in ~9 ms (3.3 us each) and the GPU runs in ~17 ms: with two sweeps
// per WGAN step the host spends 40 % of a step inside these two calls and the step is host-bound.  A sweep's launches
// depend only on the plan (pointers, sizes, flags): everything that changes from call to call is the dropout seed,
// which recorded sweeps read from a device word (Ctx::seed_base).  So the SECOND time a plan with the same contents
// arrives on a stream, its launches are recorded once (stream capture on a stream of the library's own: the caller's
// may be the legacy default stream, which cannot be captured) and from then on the call is: one tiny kernel that
// stores the seed + hipGraphLaunch -- 0.9 ms of host time (tools/graph_host_bench.py), identical launches, identical
// bits (tests/test_gpu_decode_engine.py).  Plans whose buffers move between calls simply never match and stay eager;
// MSSR_DECODE_GRAPH=0 switches the mechanism off.  Nothing is allocated: the seed words live in a static device array.
constexpr int GRAPH_SLOTS = 16;                    // recorded sweeps kept per process (least recently used goes)
static __device__ __align__(16) uint64_t graph_seed_words[GRAPH_SLOTS * 2];     // one 16-byte line part per slot

__global__ void graph_seed_kernel(uint64_t* word, uint64_t seed) { *word = seed; }

struct SweepGraph {
    bool used = false;
    std::string key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t stamp = 0;
};
struct GraphCache {
    std::mutex mu;
    SweepGraph slot[GRAPH_SLOTS];
    uint64_t* words[16] = {};                      // device address of graph_seed_words, per device
    hipStream_t cap_stream[16] = {};
    std::vector<size_t> seen;                      // hashes of plans met once (recorded at the second meeting)
    long recorded = 0;                             // recordings made since the process started
    uint64_t clock = 0;
};
static GraphCache g_graphs;

static std::string sweep_key(const mssr_decode_plan* p, bool backward, hipStream_t stream, int dev) {
    mssr_decode_plan q = *p;
    q.seed = 0;
    std::string k(reinterpret_cast<const char*>(&q), sizeof(q));
    k.append(reinterpret_cast<const char*>(p->layers), sizeof(mssr_dec_layer) * (size_t)p->n_layers);
    const uintptr_t tail[3] = {(uintptr_t)stream, (uintptr_t)dev, (uintptr_t)(backward ? 1 : 0)};
    k.append(reinterpret_cast<const char*>(tail), sizeof(tail));
    return k;
}

// true: the sweep was launched from a recorded graph (rc holds the result); false: the caller runs it eagerly
static std::atomic<int> g_graph_switch{1};         // mssr_decode_graphs_enable (the MSSR_DECODE_GRAPH=0 veto stays)

static bool run_sweep_graphed(const mssr_decode_plan* p, bool backward, hipStream_t stream, const char* who, int& rc) {
    static const bool enabled = [] { const char* e = getenv("MSSR_DECODE_GRAPH"); return !e || atoi(e) != 0; }();
    if (!enabled || !g_graph_switch.load(std::memory_order_relaxed) || prof_active() || !p || !p->layers || p->n_layers <= 0 || p->n_chunks != 1) return false;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
    if (check_plan(p, 0, who) != MSSR_OK) return false;            // the eager path reports the error
    GraphCache& C = g_graphs;
    std::lock_guard<std::mutex> lock(C.mu);
    if (!C.words[dev]) {
        void* w = nullptr;
        if (hipGetSymbolAddress(&w, HIP_SYMBOL(graph_seed_words)) != hipSuccess || !w) return false;
        C.words[dev] = (uint64_t*)w;
        if (hipStreamCreateWithFlags(&C.cap_stream[dev], hipStreamNonBlocking) != hipSuccess) { C.words[dev] = nullptr; return false; }
    }
    const std::string key = sweep_key(p, backward, stream, dev);
    static const bool debug = [] { const char* e = getenv("MSSR_DECODE_GRAPH_DEBUG"); return e && atoi(e) != 0; }();
    if (debug) {            // which bytes of the plan moved since the last sweep of this direction
        static std::string last[2];
        std::string& prev = last[backward ? 1 : 0];
        if (prev.size() == key.size()) {
            int nd = 0;
            char buf[256];
            int o = snprintf(buf, sizeof(buf), "[mssr] %s plan vs previous:", who);
            for (size_t i = 0; i + 8 <= key.size(); i += 8)
                if (memcmp(&key[i], &prev[i], 8) != 0 && ++nd <= 12) o += snprintf(buf + o, sizeof(buf) - o, " +%zu", i);
            fprintf(stderr, "%s (%d words differ of %zu; plan struct %zu bytes)\n", buf, nd, key.size() / 8, sizeof(mssr_decode_plan));
        }
        prev = key;
    }
    int hit = -1, victim = 0;
    for (int i = 0; i < GRAPH_SLOTS; ++i) {
        if (C.slot[i].used && C.slot[i].key == key) { hit = i; break; }
        if (!C.slot[i].used) victim = i;
        else if (C.slot[victim].used && C.slot[i].stamp < C.slot[victim].stamp) victim = i;
    }
    if (hit < 0) {
        const size_t h = std::hash<std::string>()(key);
        bool met = false;
        for (size_t v : C.seen) met = met || v == h;
        if (!met) {                                 // first meeting: eager, remember it
            if (C.seen.size() >= 64) C.seen.erase(C.seen.begin());
            C.seen.push_back(h);
            return false;                           // (building and hashing the 1.5 KB key costs microseconds)
        }
        SweepGraph& S = C.slot[victim];
        if (S.used) {                               // a replay of the evicted recording may still be running
            (void)hipDeviceSynchronize();
            (void)hipGraphExecDestroy(S.exec);
            (void)hipGraphDestroy(S.graph);
            S = SweepGraph();
        }
        hipStream_t cs = C.cap_stream[dev];
        if (hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); return false; }
        const int r = run_steps(p, 0, p->T, backward, cs, who, C.words[dev] + 2 * victim);
        hipGraph_t g = nullptr;
        const hipError_t ee = hipStreamEndCapture(cs, &g);
        if (r != MSSR_OK || ee != hipSuccess || !g) {
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            if (r != MSSR_OK) { rc = r; return true; }        // an argument error: report it, nothing ran
            return false;
        }
        hipGraphExec_t ex = nullptr;
        if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess || !ex) {
            (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            return false;
        }
        S.used = true; S.key = key; S.graph = g; S.exec = ex;
        ++C.recorded;
        hit = victim;
    }
    SweepGraph& S = C.slot[hit];
    S.stamp = ++C.clock;
    graph_seed_kernel<<<1, 1, 0, stream>>>(C.words[dev] + 2 * hit, p->seed);
    if (hipGraphLaunch(S.exec, stream) != hipSuccess) {
        set_error("%s: hipGraphLaunch failed: %s", who, hipGetErrorString(hipGetLastError()));
        rc = MSSR_ERR_LAUNCH;
        return true;
    }
    rc = MSSR_OK;
    return true;
}

// number of recorded sweeps alive (tests / tools)
static int sweep_graph_count() {
    std::lock_guard<std::mutex> lock(g_graphs.mu);
    int n = 0;
    for (const SweepGraph& s : g_graphs.slot) n += s.used ? 1 : 0;
    return n;
}

}  // namespace mssr

using namespace mssr;

extern "C" int mssr_decode_graphs(void) { return sweep_graph_count(); }
extern "C" int mssr_decode_graphs_enable(int on) {
    return g_graph_switch.exchange(on ? 1 : 0);
}
extern "C" long mssr_decode_graphs_recorded(void) {
    std::lock_guard<std::mutex> lock(g_graphs.mu);
    return g_graphs.recorded;
}

extern "C" int mssr_decode_step_fwd(const mssr_decode_plan* p, int step, void* stream) {
    MSSR_REQUIRE(p, "decode_step_fwd: null plan");
    return run_steps(p, step, step + 1, false, (hipStream_t)stream, "decode_step_fwd");
}

extern "C" int mssr_decode_step_bwd(const mssr_decode_plan* p, int step, void* stream) {
    MSSR_REQUIRE(p, "decode_step_bwd: null plan");
    return run_steps(p, step, step + 1, true, (hipStream_t)stream, "decode_step_bwd");
}

// All T steps forward (0 .. T-1) / backward (T-1 .. 0) in one call: the whole decode is one host
// round trip, every kernel enqueued back to back, the batch chunks on concurrent streams that fork
// from and join back into `stream`.
extern "C" int mssr_decode_fwd(const mssr_decode_plan* p, void* stream) {
    MSSR_REQUIRE(p, "decode_fwd: null plan");
    int rc = MSSR_OK;
    if (run_sweep_graphed(p, false, (hipStream_t)stream, "decode_fwd", rc)) return rc;
    return run_steps(p, 0, p->T, false, (hipStream_t)stream, "decode_fwd");
}

extern "C" int mssr_decode_bwd(const mssr_decode_plan* p, void* stream) {
    MSSR_REQUIRE(p, "decode_bwd: null plan");
    int rc = MSSR_OK;
    if (run_sweep_graphed(p, true, (hipStream_t)stream, "decode_bwd", rc)) return rc;
    return run_steps(p, 0, p->T, true, (hipStream_t)stream, "decode_bwd");
}

extern "C" int mssr_ctx_kv_grad(const float* ds_ctx, const void* qc, int64_t qc_layer_stride, const void* dctx,
                                int64_t dctx_layer_stride, void* out, int n_layers, int T, int B, int b0, int heads,
                                int dh, int Ls, int dtype, void* stream) {
    MSSR_REQUIRE(ds_ctx && qc && dctx && out, "ctx_kv_grad: null pointer");
    MSSR_REQUIRE(valid_dtype(dtype), "ctx_kv_grad: bad dtype");
    MSSR_REQUIRE(n_layers > 0 && T > 0 && B > 0 && b0 >= 0 && b0 < B && heads > 0 && dh > 0 && dh <= 64 && Ls > 0,
                 "ctx_kv_grad: bad dimensions (head width must be <= 64)");
    const int64_t waves = (int64_t)n_layers * 2 * (B - b0) * heads;
    MSSR_REQUIRE((waves + 3) / 4 < ((int64_t)1 << 31), "ctx_kv_grad: too many slices");
    const dim3 grid((unsigned)((waves + 3) / 4));
    if (dtype == MSSR_F32)
        ctx_kv_grad_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(ds_ctx, (const float*)qc, qc_layer_stride,
                                                                         (const float*)dctx, dctx_layer_stride,
                                                                         (float*)out, n_layers, T, B, b0, heads, dh, Ls);
    else
        ctx_kv_grad_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream>>>(ds_ctx, (const bf16*)qc, qc_layer_stride,
                                                                        (const bf16*)dctx, dctx_layer_stride, (bf16*)out,
                                                                        n_layers, T, B, b0, heads, dh, Ls);
    MSSR_LAUNCH_OK("ctx_kv_grad");
}
