// Shared by attention.hip (scalar fused kernels, L <= 64, any dtype) and attention_mfma.hip (matrix-core kernels, bf16).
#pragma once
#include "common.hpp"

namespace mssr {

constexpr int FA_KSTRIDE = 256;   // dropout decision index = ((b*H + h)*Lq + i)*FA_KSTRIDE + key: the SAME in both kernel families

struct AttnArgs {
    const void *q, *k, *v;
    const uint8_t* mask;
    int B, H, Lq, Lk, dh;
    int64_t q_bs, q_ls, k_bs, k_ls, v_bs, v_ls, mask_bs, mask_qs;
    float scale, p_drop;
    uint64_t seed;
    int drop_b0;        // sentences below it are not dropped
};

__device__ __forceinline__ float fa_keep(const AttnArgs& a, int bh, int i, int t) {
    if (a.p_drop <= 0.f || bh / a.H < a.drop_b0) return 1.0f;
    const uint32_t thr = drop_thr16(a.p_drop);
    return keep16(a.seed, ((uint64_t)bh * a.Lq + i) * FA_KSTRIDE + t, thr) ? 1.0f / (1.0f - a.p_drop) : 0.f;
}

// attention_mfma.hip: 1 when the matrix-core kernels can take this problem (bf16, dh 32 / 64, L <= 160, 16-byte operands)
bool attn_mfma_fits(int Lq, int Lk, int dh, int dtype, int backward);
bool attn_mfma_fwd_try(const AttnArgs& a, int dtype, void* ctx, int64_t ctx_bs, int64_t ctx_ls, void* probs, hipStream_t s);
bool attn_mfma_bwd_try(const AttnArgs& a, int dtype, const void* dctx, int64_t dc_bs, int64_t dc_ls, void* dq, int64_t dq_bs,
                       int64_t dq_ls, void* dk, int64_t dk_bs, int64_t dk_ls, void* dv, int64_t dv_bs, int64_t dv_ls,
                       hipStream_t s);

}  // namespace mssr
