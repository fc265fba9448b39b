"""ctypes binding of libmssr_hip.so (include/mssr_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is NO
fallback: if the shared object is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmssr_hip.so")

F32, BF16 = 0, 1
_DT = {torch.float32: F32, torch.bfloat16: BF16}


class GemmDesc(Structure):
    _fields_ = [("a", c_void_p), ("b", c_void_p), ("c", c_void_p), ("bias", c_void_p),
                ("residual", c_void_p), ("gate", c_void_p),
                ("m", c_int64), ("n", c_int64), ("k", c_int64),
                ("a_rs", c_int64), ("a_cs", c_int64), ("b_rs", c_int64), ("b_cs", c_int64),
                ("ldc", c_int64), ("ldr", c_int64), ("ldg", c_int64),
                ("batch_outer", c_int64), ("batch_inner", c_int64),
                ("a_bo", c_int64), ("a_bi", c_int64), ("b_bo", c_int64), ("b_bi", c_int64),
                ("c_bo", c_int64), ("c_bi", c_int64),
                ("alpha", c_float), ("beta", c_float), ("relu", c_int),
                ("in_dtype", c_int), ("out_dtype", c_int),
                ("drop_p", c_float), ("drop_seed", c_uint64),
                ("workspace", c_void_p), ("workspace_bytes", c_int64), ("no_skinny", c_int),
                ("ln_colsum", c_void_p), ("ln_mean", c_void_p), ("ln_rstd", c_void_p), ("ln_eps", c_float),
                ("drop_row0", c_int64), ("k_seg_len", c_int64), ("k_seg_stride", c_int64)]


class DecLayer(Structure):
    """mssr_dec_layer (include/mssr_hip.h)."""
    _fields_ = [(n, c_void_p) for n in (
        "w_qkv", "wt_qkv", "w_so", "wt_so", "w_cq", "wt_cq", "w_co", "wt_co", "w_1", "wt_1", "w_2", "wt_2",
        "b_qkv", "b_so", "b_cq", "b_co", "b_1", "b_2",
        "ln1_g", "ln1_b", "ln2_g", "ln2_b", "lnf_g", "lnf_b",
        "wg_qkv", "wg_cq", "wg_1", "cs_qkv", "cs_cq", "cs_1", "bf_qkv", "bf_cq", "bf_1")]


class EncLayer(Structure):
    """mssr_enc_layer (include/mssr_hip.h)."""
    _fields_ = [(n, c_void_p) for n in (
        "w_qkv", "wt_qkv", "w_o", "wt_o", "w_1", "wt_1", "w_2", "wt_2", "b_qkv", "b_o", "b_1", "b_2",
        "ln1_g", "ln1_b", "lnf_g", "lnf_b", "wg_qkv", "wg_1", "cs_qkv", "cs_1", "bf_qkv", "bf_1")]


class EncoderPlan(Structure):
    """mssr_encoder_plan (include/mssr_hip.h)."""
    _fields_ = ([(n, c_int) for n in ("dtype", "B", "L", "d", "heads", "ff", "n_layers", "save", "training", "fuse_ln")]
                + [("p_drop", c_float), ("p_attn", c_float), ("seed", c_uint64), ("layers", POINTER(EncLayer))]
                + [(n, c_void_p) for n in ("lnF_g", "lnF_b", "mask", "X", "out", "qkv", "actD", "actF", "stats",
                                           "gemm_ws")]
                + [("gemm_ws_bytes", c_int64)]
                + [(n, c_void_p) for n in ("dout", "dX", "dqkv", "dactD", "dactF", "dctx")]
                + [("eval_rows", c_int)])


class DecodePlan(Structure):
    """mssr_decode_plan (include/mssr_hip.h)."""
    _fields_ = ([(n, c_int) for n in ("dtype", "B", "d", "heads", "ff", "n_layers", "V", "Vp", "Ls", "T", "save",
                                      "training", "n_chunks", "fuse_ln")]
                + [("p_drop", c_float), ("p_attn", c_float), ("temperature", c_float), ("seed", c_uint64),
                   ("layers", POINTER(DecLayer))]
                + [(n, c_void_p) for n in ("emb_t", "emb", "pe", "w_gen", "wt_gen", "b_gen", "lnF_g", "lnF_b",
                                           "wg_gen", "cs_gen", "bf_gen", "mem_kv", "src_mask", "in_all", "probs", "X", "qkv", "actD", "actF",
                                           "stats", "nf", "logits", "tmp", "gemm_ws")]
                + [("gemm_ws_bytes", c_int64)]
                + [(n, c_void_p) for n in ("dprobs", "dlogits", "dX", "dqkv", "dactD", "dactF", "dctx",
                                           "ds_self", "ds_ctx", "d_e", "dnf", "ids")]
                + [("greedy", c_int), ("eval_rows", c_int), ("bwd_row0", c_int), ("forced_rows", c_int),
                   ("forced_steps", c_int), ("self_mask", c_void_p), ("late_step", c_int), ("late_row0", c_int), ("late_fwd", c_int)])


class DecStackPlan(Structure):
    """mssr_decstack_plan (include/mssr_hip.h)."""
    _fields_ = ([(n, c_int) for n in ("dtype", "B", "T", "Ls", "d", "heads", "ff", "n_layers", "save", "training")]
                + [("p_drop", c_float), ("p_attn", c_float), ("seed", c_uint64), ("layers", POINTER(DecLayer))]
                + [(n, c_void_p) for n in ("lnF_g", "lnF_b", "self_mask", "src_mask", "X", "out", "mem_kv", "probs", "qkv",
                                           "actD", "actF", "stats", "gemm_ws")]
                + [("gemm_ws_bytes", c_int64)]
                + [(n, c_void_p) for n in ("dout", "dX", "dqkv", "dactD", "dactF", "dctx", "dmem_kv")])


class AttnDesc(Structure):
    """mssr_attn_desc (include/mssr_hip.h)."""
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("mask", c_void_p),
                ("B", c_int), ("H", c_int), ("Lq", c_int), ("Lk", c_int), ("dh", c_int),
                ("q_bs", c_int64), ("q_ls", c_int64), ("k_bs", c_int64), ("k_ls", c_int64),
                ("v_bs", c_int64), ("v_ls", c_int64), ("mask_bs", c_int64), ("mask_qs", c_int64),
                ("p_drop", c_float), ("seed", c_uint64), ("dtype", c_int), ("drop_b0", c_int)]


class PackJob(Structure):
    """mssr_pack_job (include/mssr_hip.h)."""
    _fields_ = [("src", c_void_p), ("rows", c_int), ("cols", c_int), ("dst", c_void_p), ("dst_ld", c_int64),
                ("dst_t", c_void_p), ("dst_t_ld", c_int64), ("dst_g", c_void_p), ("dst_g_ld", c_int64),
                ("gamma", c_void_p), ("beta", c_void_p), ("bias", c_void_p), ("colsum", c_void_p), ("bias_f", c_void_p),
                ("dst_f", c_void_p), ("block0", c_int), ("pad_", c_int)]


class ProfRecord(Structure):
    """mssr_prof_record (include/mssr_hip.h)."""
    _fields_ = [("kernel", c_int), ("in_dtype", c_int), ("out_dtype", c_int), ("tile", c_int), ("ksplit", c_int), ("ln", c_int),
                ("batch", c_int64), ("m", c_int64), ("n", c_int64), ("k", c_int64), ("ms", c_float)]


_P, _I, _L, _F, _U = c_void_p, c_int, c_int64, c_float, c_uint64
SIGNATURES = {
    "mssr_abi_version": ([], c_int),
    "mssr_last_error": ([], c_char_p),
    "mssr_layernorm_fwd": ([_P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P], c_int),
    "mssr_layernorm_partial_rows": ([_L], c_int),
    "mssr_layernorm_bwd": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P], c_int),
    "mssr_layernorm_bwd_bwd": ([_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P], c_int),
    "mssr_layernorm_param_grads": ([_P, _P, _P, _P, _P, _I, _L, _L, _L, _L, _L, _I, _I, _I, _P], c_int),
    "mssr_gemm": ([POINTER(GemmDesc), _P], c_int),
    "mssr_gemm_noalloc": ([POINTER(GemmDesc), _P], c_int),
    "mssr_gemm_noalloc_scratch_bytes": ([], c_int64),
    "mssr_gemm_tile": ([_L, _L, _L], c_int),
    "mssr_gemm_is_skinny": ([POINTER(GemmDesc)], c_int),
    "mssr_softmax_fwd": ([_P, _P, _P, _L, _I, _L, _L, _L, _L, _L, _L, _F, _I, _I, _I, _P], c_int),
    "mssr_softmax_bwd": ([_P, _P, _P, _L, _I, _L, _L, _L, _F, _I, _I, _I, _I, _P], c_int),
    "mssr_softmax_bwd_bwd": ([_P, _P, _P, _P, _L, _I, _L, _F, _I, _P], c_int),
    "mssr_dropout_fwd": ([_P, _P, _P, _P, _L, _F, _U, _I, _P], c_int),
    "mssr_mask_scale": ([_P, _P, _P, _P, _L, _F, _I, _P], c_int),
    "mssr_gate_gt0": ([_P, _P, _P, _L, _I, _P], c_int),
    "mssr_gain_pick_fwd": ([_P, _L, _P, _P, _I, _I, _F, _I, _P], c_int),
    "mssr_gain_pick_bwd": ([_P, _P, _L, _P, _P, _I, _I, _F, _I, _P], c_int),
    "mssr_gain_pick_bwd_bwd": ([_P, _L, _P, _P, _L, _P, _P, _P, _I, _I, _F, _I, _P], c_int),
    "mssr_ctx_pool_fwd": ([_P, _L, _L, _P, _L, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P], c_int),
    "mssr_ctx_pool_bwd": ([_P, _P, _P, _L, _L, _P, _L, _L, _P, _P, _P, _P, _L, _L, _P, _L, _L, _P, _I, _I, _I, _I, _I, _P],
                          c_int),
    "mssr_ctx_pool_bwd_bwd": ([_P, _L, _L, _P, _L, _L, _P, _P, _P, _P, _P, _L, _L, _P, _L, _L, _P, _P, _P, _P, _P, _L, _L,
                               _P, _L, _L, _P, _P, _I, _I, _I, _I, _I, _P], c_int),
    "mssr_gp_penalty_fwd": ([_P, _L, _L, _I, _I, _I, _P, _L, _L, _F, _F, _P, _P, _I, _P], c_int),
    "mssr_gp_penalty_bwd": ([_P, _L, _L, _I, _I, _I, _P, _L, _L, _F, _F, _P, _P, _P, _P, _I, _P], c_int),
    "mssr_colsum": ([_P, _P, _L, _I, _L, _I, _I, _P], c_int),
    "mssr_colsum_batched": ([_P, _P, _I, _L, _L, _I, _L, _I, _I, _P], c_int),
    "mssr_axpby": ([_P, _P, _P, _L, _F, _F, _I, _P], c_int),
    "mssr_grad_norm_scratch_floats": ([_I], c_int64),
    "mssr_grad_norm": ([_I, _P, _P, _F, _P, _P, _P], c_int),
    "mssr_adam_step": ([_I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _P, _I, _P], c_int),
    "mssr_pack_blocks": ([_I], c_int),
    "mssr_pack_weights": ([_P, _I, _I, _I, _P], c_int),
    "mssr_multi_copy": ([_I, _P, _P, _P, _F, _P], c_int),
    "mssr_embedding_fwd": ([_P, _P, _P, _P, _L, _I, _L, _L, _I, _F, _I, _P], c_int),
    "mssr_embedding_bwd": ([_P, _P, _P, _L, _I, _L, _L, _F, _I, _P], c_int),
    "mssr_scale_add_pe": ([_P, _P, _P, _L, _I, _L, _I, _F, _I, _P], c_int),
    "mssr_attention_fits": ([_I, _I, _I, _I, _I], c_int),
    "mssr_attention_fwd": ([POINTER(AttnDesc), _P, _L, _L, _P, _P], c_int),
    "mssr_attention_bwd": ([POINTER(AttnDesc), _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P], c_int),
    "mssr_prof_start": ([_I], c_int),
    "mssr_prof_stop": ([], c_int),
    "mssr_prof_get": ([_I, POINTER(ProfRecord)], c_int),
    "mssr_encoder_fwd": ([POINTER(EncoderPlan), _P], c_int),
    "mssr_encoder_bwd": ([POINTER(EncoderPlan), _P], c_int),
    "mssr_decstack_fwd": ([POINTER(DecStackPlan), _P], c_int),
    "mssr_decstack_bwd": ([POINTER(DecStackPlan), _P], c_int),
    "mssr_decode_step_fwd": ([POINTER(DecodePlan), _I, _P], c_int),
    "mssr_decode_step_bwd": ([POINTER(DecodePlan), _I, _P], c_int),
    "mssr_decode_fwd": ([POINTER(DecodePlan), _P], c_int),
    "mssr_decode_bwd": ([POINTER(DecodePlan), _P], c_int),
    "mssr_decode_graphs": ([], c_int),
    "mssr_decode_graphs_recorded": ([], ctypes.c_long),
    "mssr_decode_graphs_enable": ([_I], c_int),
    "mssr_debug_stamps": ([_P], None),
    "mssr_ctx_kv_grad": ([_P, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P], c_int),
}

_lib = None


def load():
    """dlopen the library once and attach the C signatures.  Fails loudly if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "mssrnet_style_transfer_amd: %s is missing -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "There is no CPU or eager fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch: also loud
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().mssr_last_error().decode("utf-8", "replace")
        raise RuntimeError("libmssr_hip %s failed (code %d): %s" % (what, rc, msg))


def dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError("unsupported dtype %s (need float32 or bfloat16)" % t.dtype)


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mssrnet_style_transfer_amd kernels need tensors on the GPU (got %s); "
                               "there is no CPU path" % t.device)
