// Shared device/host helpers for the gfx950 kernels of libmssr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mssr_hip.h"

namespace mssr {

void set_error(const char* fmt, ...);
// bench instrumentation (core.hip): HIP events around a launch when mssr_prof_start() is active
int prof_begin(hipStream_t s);
void prof_end(int i, hipStream_t s, int kernel, int in_dt, int out_dt, int tile, int64_t batch, int64_t m, int64_t n,
              int64_t k, int ksplit = 1, int ln = 0);
// Profiling launches: prof_begin arms the calling thread with the record's event pair and the NEXT klaunch hands
// it to hipExtLaunchKernelGGL, which stamps the events with the dispatch's own begin / end times (the same
// clock rocprofv3's kernel trace reads), so a record holds the kernel alone and not the markers around it.
bool prof_take(hipEvent_t* start, hipEvent_t* stop);
bool prof_active();     // mssr_prof_start() .. mssr_prof_stop(): launches carry events, nothing may be replayed from a graph
template <typename... KArgs, typename... Args>
inline void klaunch(void (*kern)(KArgs...), dim3 grid, dim3 block, unsigned shmem, hipStream_t s, Args... args) {
    hipEvent_t e0, e1;
    if (prof_take(&e0, &e1)) hipExtLaunchKernelGGL(kern, grid, block, shmem, s, e0, e1, 0, args...);
    else kern<<<grid, block, shmem, s>>>(args...);
}

// A launch with explicit profiling events (either may be null): products that take two launches -- split-K main kernel
// + reduction, memset + atomic accumulation -- hand the record's start event to the first and its stop event to the
// last, so the record covers the whole product.
template <typename... KArgs, typename... Args>
inline void klaunch_events(void (*kern)(KArgs...), dim3 grid, dim3 block, unsigned shmem, hipStream_t s, hipEvent_t e0,
                           hipEvent_t e1, Args... args) {
    if (e0 || e1) hipExtLaunchKernelGGL(kern, grid, block, shmem, s, e0, e1, 0, args...);
    else kern<<<grid, block, shmem, s>>>(args...);
}

#define MSSR_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            ::mssr::set_error(__VA_ARGS__);     \
            return MSSR_ERR_ARG;                \
        }                                       \
    } while (0)

#define MSSR_LAUNCH_OK(name)                                                        \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            ::mssr::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return MSSR_ERR_LAUNCH;                                                 \
        }                                                                           \
        return MSSR_OK;                                                             \
    } while (0)

constexpr int WAVE = 64;

typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }

// 4 consecutive elements <-> float[4]  (16-B access for f32, 8-B for bf16)
template <typename T> struct V4;
template <> struct V4<float> {
    static __device__ __forceinline__ void load(const float* p, float (&o)[4]) {
        float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&o)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    }
};
template <> struct V4<bf16> {
    static __device__ __forceinline__ void load(const bf16* p, float (&o)[4]) {
        bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
        o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
    }
    static __device__ __forceinline__ void store(bf16* p, const float (&o)[4]) {
        bf16x4 v;
        v[0] = (bf16)o[0]; v[1] = (bf16)o[1]; v[2] = (bf16)o[2]; v[3] = (bf16)o[3];
        *reinterpret_cast<bf16x4*>(p) = v;
    }
};

// Wave64 reductions, result in every lane.  The 16 lanes of a DPP row are folded with data-parallel moves
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: plain VALU latency), the four row totals are read
// through the scalar lane-read path.  The ds_bpermute form (__shfl_xor) costs an LDS crossbar trip per step:
// six dependent trips per reduction were the largest part of the single-query attention kernels' time.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_read(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return (lane_read(v, 0) + lane_read(v, 16)) + (lane_read(v, 32) + lane_read(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return fmaxf(fmaxf(lane_read(v, 0), lane_read(v, 16)), fmaxf(lane_read(v, 32), lane_read(v, 48)));
}

// The per-call seed word of a replayed sweep (decode.hip), read through the CONSTANT address space: a uniform pointer
// there becomes a scalar load (s_load_dwordx2, lgkmcnt).  As an ordinary global load it was a VECTOR load whose
// `s_waitcnt vmcnt(0)` also waited for every LDS-DMA / fragment load the kernel had issued before it - the dropout
// hashes, placed "under the loads", ran after them instead: +0.25 .. 0.6 us on every replayed kernel (0.9 ms per WGAN
// step; rocprofv3, replayed vs eager sweeps).  The word is written by an earlier kernel of the stream and constant for
// this one; kernel boundaries invalidate the scalar cache like they do for kernel arguments.
__device__ __forceinline__ uint64_t seed_word(const uint64_t* p) {
    typedef const __attribute__((address_space(4))) uint64_t* cptr;
    return *(cptr)(uintptr_t)p;
}

// counter-based RNG for dropout: one 32-bit hash per element index.  32-bit arithmetic only (two multiplies:
// the "lowbias32" finaliser over index ^ seed mix): the 64-bit splitmix it replaces cost ~400 cycles per wave
// and element on CDNA (64-bit multiplies are sequences of quarter-rate 32-bit ones), which made the mask the
// most expensive part of every kernel that replays dropout.  The seed mix is uniform (scalar unit).
__device__ __forceinline__ uint32_t mix_u32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t idx) {
    // seeds of neighbouring sites / steps differ in a few bits: mix them first (uniform, scalar unit), so that two
    // sites' index spaces are xor-shifted apart by a full-width random constant
    const uint32_t s = mix_u32((uint32_t)seed) ^ mix_u32((uint32_t)(seed >> 32) + 0x9E3779B9u);
    const uint32_t hi = (uint32_t)(idx >> 32);
    return mix_u32((uint32_t)idx ^ s ^ ((hi << 16) | (hi >> 16)));
}

// Dropout decisions come in PAIRS: elements 2i and 2i+1 share one hash, 16 bits each against a 16-bit threshold (p is
// quantised to 1/65536: 0.1 -> 0.09999).  The hash's two quarter-rate 32-bit multiplies dominate the integer work of
// the kernels that replay dropout (gemm_lnb_kernel: 76 v_mul_lo_u32 per lane before, 44 after); kernels that own
// consecutive elements take both decisions of a pair from one hash (keep16_pair), the others use keep16 - same decisions.
__device__ __forceinline__ uint32_t drop_thr16(float p) { return (uint32_t)((double)p * 65536.0); }
__device__ __forceinline__ bool keep16(uint64_t seed, uint64_t idx, uint32_t thr16) {
    const uint32_t h = hash_u32(seed, idx >> 1);
    return ((idx & 1) ? (h >> 16) : (h & 0xFFFFu)) >= thr16;
}
// both decisions of the pair (2 idx2, 2 idx2 + 1): bit 0 and bit 1 of the result
__device__ __forceinline__ uint32_t keep16_pair(uint64_t seed, uint64_t idx2, uint32_t thr16) {
    const uint32_t h = hash_u32(seed, idx2);
    return ((h & 0xFFFFu) >= thr16 ? 1u : 0u) | ((h >> 16) >= thr16 ? 2u : 0u);
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
inline bool valid_dtype(int d) { return d == MSSR_F32 || d == MSSR_BF16; }
inline size_t dsize(int d) { return d == MSSR_BF16 ? 2 : 4; }

}  // namespace mssr
