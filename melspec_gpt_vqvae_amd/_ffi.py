"""ctypes binding of libmelgpt_hip.so (the C ABI declared in include/melgpt.h).

PyTorch is only plumbing here: it owns device memory and the HIP stream; every launch goes
through the C ABI with raw device pointers.  There is NO fallback: if the shared library is
missing or a call is made without a GPU tensor, this module raises."""
from __future__ import annotations

import ctypes as C
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# The 16-bit lane's storage format is a property of the LIBRARY (csrc/common.h): bfloat16 (libmelgpt_hip.so, default)
# or IEEE half (libmelgpt_hip_fp16.so, MELGPT_HALF=fp16 - BASELINE configs[4] names fp16).  One format per process.
HALF = os.environ.get("MELGPT_HALF", "bf16").lower()
if HALF not in ("bf16", "fp16"):
    raise ImportError(f"MELGPT_HALF={HALF!r}: expected bf16 or fp16")
HALF_DTYPE = torch.float16 if HALF == "fp16" else torch.bfloat16
LIB_PATH = os.path.join(_PKG, "lib", "libmelgpt_hip_fp16.so" if HALF == "fp16" else "libmelgpt_hip.so")
if os.environ.get("MELGPT_LAB_LIB"):          # lab A/B builds of the same ABI (tools/lab): never set in production
    LIB_PATH = os.environ["MELGPT_LAB_LIB"]

F32, BF16 = 0, 1
_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
ERR_UNSUPPORTED = -2  # MELGPT_ERR_UNSUPPORTED (include/melgpt.h)
_u64 = C.c_uint64


class MelgptError(RuntimeError):
    pass


# name -> argtypes (restype is int unless listed in _RESTYPE)
_PROTOS = {
    "melgpt_abi_version": [],
    "melgpt_strerror": [_i],
    "melgpt_set_reserved_cus": [_i],
    "melgpt_get_reserved_cus": [],
    "melgpt_set_gemm_pingpong": [_i],
    "melgpt_get_gemm_pingpong": [],
    "melgpt_gemm_loop_launches": [_p, _p],
    "melgpt_vq_argmin_fwd": [_p, _i, _l, _i, _l, _l, _l, _l, _p, _i, _p, _p, _p, _p, _p, _p],
    "melgpt_vq_argmin_fwd_ex": [_p, _i, _l, _i, _l, _l, _l, _l, _p, _i, _p, _p, _p, _p, _p, _p, _p],
    "melgpt_vq_max_grid": [],
    "melgpt_vq_image_bytes": [_i],
    "melgpt_vq_prepare_image": [_p, _i, _i, _p, _p, _i, _p, _p],
    "melgpt_vq_lookup_image": [_p, _l, _i, _l, _l, _l, _l, _p, _i, _i, _p, _p, _p],
    "melgpt_vq_finalize": [_p, _i, _p, _i, _l, _i, _f, _p, _p],
    "melgpt_vq_gather": [_p, _l, _p, _i, _i, _p, _i, _l, _l, _l, _l, _p],
    "melgpt_vq_onehot": [_p, _l, _i, _p, _p],
    "melgpt_vq_bwd": [_p, _p, _i, _l, _i, _l, _l, _l, _l, _p, _i, _p, _p, _f, _p, _p, _p],
    "melgpt_gemm": [_p, _i, _l, _l, _p, _i, _l, _l, _p, _l, _l, _i, _i, _i, _i, _i, _i, _i, _f, _p, _i, _p, _l, _l,
                    _p, _f, _u64, C.c_uint, _p],
    "melgpt_wgrad_rowsum_rows": [_i, _i],
    "melgpt_wgrad_rowsum": [_p, _l, _l, _p, _l, _l, _p, _l, _l, _i, _i, _i, _i, _i, _p, _l, _p],
    "melgpt_conv2d_nhwc": [_p, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p],
    "melgpt_attn_fwd": [_p, _p, _p, _l, _p, _l, _p, _p, _i, _i, _i, _i, _i, _f, _u64, C.c_uint, _i, _p],
    "melgpt_attn_decode": [_p, _l, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p],
    "melgpt_embed_decode": [_p, _p, _p, _p, _i, _i, _i, _p, _i, _p],
    "melgpt_incr_i32": [_p, _p],
    "melgpt_pad1d_act": [_p, _p, _i, _i, _i, _i, _i, _f, _i, _p],
    "melgpt_conv1d_nlc": [_p, _i, _i, _i, _p, _i, _i, _i, _i, _i, _f, _p, _p, _l, _i, _f, _p, _l, _i, _p],
    "melgpt_conv1d_out1": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "melgpt_conv1d_out1_fused": [_p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _i, _p],
    "melgpt_resblock_narrow": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p],
    "melgpt_gemv_rows": [_p, _l, _p, _l, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _i, _p, _p, _f, _p],
    "melgpt_linear_skinny": [_p, _l, _p, _l, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _i, _p, _p, _f, _p],
    "melgpt_linear_lds_workspace": [_i, _i, _i],
    "melgpt_linear_lds": [_p, _l, _p, _l, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _i, _p, _p, _f, _p, _p],
    "melgpt_ln_fold_prepare": [_p, _l, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p],
    "melgpt_attn_bwd": [_p, _p, _p, _l, _p, _p, _l, _p, _p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _f, _u64, C.c_uint,
                        _i, _p],
    "melgpt_set_attn_bwd_two_pass": [_i],
    "melgpt_set_attn_fwd32": [_i],
    "melgpt_layernorm_fwd": [_p, _p, _p, _p, _p, _p, _l, _i, _f, _i, _p],
    "melgpt_layernorm_bwd_nwaves": [_l],
    "melgpt_layernorm_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _l, _i, _i, _p],
    "melgpt_layernorm_bwd_masked": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _l, _i, _p, _f, _u64, C.c_uint, _i, _p],
    "melgpt_colsum_rows": [],
    "melgpt_colsum": [_p, _l, _i, _l, _p, _i, _p, _i, _p],
    "melgpt_embed_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _l, _i, _i, _p, _i, _f, _u64, C.c_uint, _p],
    "melgpt_embed_bwd": [_p, _p, _l, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _i, _f, _u64, C.c_uint, _p],
    "melgpt_cross_entropy_fwd": [_p, _l, _p, _l, _i, _p, _p, _p],
    "melgpt_cross_entropy_bwd": [_p, _l, _p, _p, _p, _i, _p, _f, _l, _i, _p, _l, _i, _p],
    "melgpt_group_sum_f32": [_p, _l, _i, _f, _p, _p],
    "melgpt_sample_logits": [_p, _l, _i, _i, _f, _i, _i, _u64, C.c_uint, _p, _p, _p],
    "melgpt_sample_logits_dev": [_p, _l, _i, _i, _f, _i, _i, _u64, _p, _i, _p, _p, _l, _p],
    "melgpt_vae_reparam_fwd": [_p, _p, _i, _u64, _i, _i, _i, _p, _p, _p],
    "melgpt_vae_reparam_bwd": [_p, _p, _p, _p, _i, _i, _i, _p, _p],
    "melgpt_gauss_log_density": [_p, _p, _p, _l, _i, _i, _i, _i, _p, _p],
    "melgpt_vae_calc_mi": [_p, _p, _l, _p, _i, _u64, _i, _i, _p, _p, _p],
    "melgpt_sum_f32": [_p, _l, _f, _p, _i, _p],
    "melgpt_dropout_apply": [_p, _p, _l, _f, _u64, C.c_uint, _i, _p],
    "melgpt_dropout_apply_colsum": [_p, _p, _l, _i, _f, _u64, C.c_uint, _p, _i, _p, _i, _p],
    "melgpt_cast": [_p, _i, _p, _i, _l, _p],
    "melgpt_zero_bytes": [_p, _l, _p],
    "melgpt_adamw": [_p, _p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _i, _f, _p],
    "melgpt_groupnorm_nchunks": [_i],
    "melgpt_groupnorm_stats": [_p, _i, _i, _i, _f, _p, _p, _p, _i, _p],
    "melgpt_groupnorm_apply": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "melgpt_groupnorm_fused": [_p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _p, _i, _p],
    "melgpt_conv_in_c1": [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "melgpt_conv_in_c1_stats_workspace": [_i, _i, _i],
    "melgpt_conv_in_c1_stats": [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p],
    "melgpt_conv_out_c1": [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "melgpt_softmax_rows": [_p, _l, _i, _l, _f, _p, _l, _i, _p],
    "melgpt_repack_conv_weight": [_p, _p, _i, _i, _i, _i, _i, _p],
    "melgpt_permute_nchw_nhwc": [_p, _i, _p, _i, _i, _i, _i, _i, _p],
    "melgpt_codes_permute": [_p, _p, _i, _i, _i, _i, _p],
    "melgpt_onehot_rows": [_p, _l, _i, _i, _i, _i, _p, _i, _p],
    "melgpt_reduce_rows": [_p, _i, _l, _l, _p, _i, _f, _p],
    "melgpt_reduce_rows_pair": [_p, _i, _l, _l, _p, _i, _p, _i, _l, _l, _p, _i, _p],
    "melgpt_mel_frontend_fwd": [_p, _i, _l, _i, _i, _p, _p, _p, _i, _f, _f, _f, _f, _f, _f, _f, _p, _i, _p, _i, _i, _i,
                                _p],
    "melgpt_mel_transforms_fwd": [_p, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _p, _i, _p, _i, _i, _i, _p],
    "melgpt_conv3x3_gn_nhwc": [_p, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _i, _p, _p, _p, _i, _p],
    "melgpt_conv3x3_gn_stats_workspace": [_i, _i, _i],
    "melgpt_conv3x3_gn_nhwc_stats": [_p, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _i, _p, _p, _p, _i, _f, _p, _p, _p, _p],
    "melgpt_groupnorm_finalize": [_p, _i, _i, C.c_double, _f, _p, _p, _p],
    "melgpt_conv3x3_bwd_workspace": [_i, _i, _i, _i, _i, _i],
    "melgpt_conv3x3_bwd_data": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p],
    "melgpt_conv3x3_bwd_weight": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p],
    "melgpt_conv3x3_s2_bwd_workspace": [_i, _i, _i, _i, _i, _i],
    "melgpt_conv3x3_s2_bwd_data": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p],
    "melgpt_conv3x3_s2_bwd_weight": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p],
    "melgpt_upsample2_nhwc": [_p, _p, _i, _i, _i, _i, _i, _p],
    "melgpt_sumpool2_nhwc": [_p, _p, _i, _i, _i, _i, _i, _p],
    "melgpt_softmax_bwd_rows": [_p, _l, _p, _l, _i, _l, _f, _p, _l, _i, _p],
    "melgpt_im2col_c1": [_p, _p, _i, _i, _i, _i, _p],
    "melgpt_groupnorm_swish_bwd_workspace": [_i, _i, _i],
    "melgpt_groupnorm_swish_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p],
}
_RESTYPE = {"melgpt_strerror": C.c_char_p, "melgpt_vq_image_bytes": C.c_int64, "melgpt_linear_lds_workspace": C.c_int64,
            "melgpt_conv3x3_bwd_workspace": C.c_int64, "melgpt_groupnorm_swish_bwd_workspace": C.c_int64,
            "melgpt_conv3x3_s2_bwd_workspace": C.c_int64}

_lib = None


def lib():
    """The loaded shared library; raises MelgptError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MelgptError(
                f"{LIB_PATH} is missing - build it with `python -m melspec_gpt_vqvae_amd.build` "
                "(or __graft_entry__.build()). There is no CPU / eager fallback.")
        L = C.CDLL(LIB_PATH)
        for name, args in _PROTOS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, C.c_int)
        _lib = L
    return _lib


def check(code, what=""):
    if code != 0:
        msg = lib().melgpt_strerror(code).decode()
        raise MelgptError(f"{what or 'melgpt call'} failed: {msg} ({code})")


def call(name, *args):
    check(getattr(lib(), name)(*args), name)


def ptr(t):
    """device pointer of a tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MelgptError("melgpt kernels need GPU tensors (got a CPU tensor); there is no CPU fallback")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == HALF_DTYPE:
        return BF16          # "the library's 16-bit format" (include/melgpt.h)
    if dt in (torch.bfloat16, torch.float16):
        raise MelgptError(f"{dt} tensors need the other library flavour: this process loaded the {HALF} one "
                          f"(MELGPT_HALF={HALF}); set MELGPT_HALF before importing the package")
    raise MelgptError(f"unsupported dtype {dt}: the kernels are built for float32 and a 16-bit format")
