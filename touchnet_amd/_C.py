"""ctypes binding of libtouchnet_amd.so — the C ABI declared in include/touchnet_amd.h.

This is the stub a TouchNet maintainer would add to bind the MI355X kernels;
the reference itself is pure Python and calls everything through torch / transformers.
There is NO fallback: if the library is missing or a kernel returns an error we raise.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# TN_AMD_LIB: kernel-development hook (scripts/build_variant.sh builds -D variants of the same library)
LIB_PATH = os.environ.get("TN_AMD_LIB") or os.path.join(_HERE, "_lib", "libtouchnet_amd.so")

_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong

# name -> argtypes (all return int unless listed in _RESTYPE)
PROTOTYPES = {
    "tn_version": [],
    "tn_norm_bwd_workspace_floats": [_i, _i],
    "tn_rmsnorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp],
    "tn_rmsnorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "tn_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp],
    "tn_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "tn_swiglu_fwd": [_vp, _vp, _vp, _ll, _i, _vp],
    "tn_swiglu_bwd": [_vp, _vp, _vp, _vp, _vp, _ll, _i, _vp],
    "tn_gelu_fwd": [_vp, _vp, _ll, _i, _vp],
    "tn_gelu_bwd": [_vp, _vp, _vp, _ll, _i, _vp],
    "tn_rope_table": [_vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp],
    "tn_rope_apply": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "tn_ce_forward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _ll, _i, _vp],
    "tn_ce_reduce": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _vp],
    "tn_ce_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _ll, _i, _vp],
    "tn_attn_meta_ints": [_i, _i],
    "tn_attn_build_meta": [_vp, _vp, _i, _i, _vp],
    "tn_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp],
    "tn_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp],
    "tn_attn_bwd_rope": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp],
    "tn_attn_fwd_bidir": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp],
    "tn_attn_bwd_bidir": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp],
    "tn_attn_fwd_seg": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _i, _vp],
    "tn_attn_fwd_seg_chunks": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _i, _i, C.c_ulonglong, _vp],
    "tn_attn_merge": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "tn_attn_bwd_seg": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp,
                        _i, _vp],
    "tn_fbank_frames": [_i],
    "tn_kaldi_fbank": [_vp, _vp, _i, _i, _vp],
    "tn_log_mel": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "tn_audiofeat_stack": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "tn_resample_polyphase": [_vp, _vp, _vp, _ll, _ll, _i, _i, _i, _vp],
    "tn_feat_augment": [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp],
    "tn_sumsq_scratch_floats": [],
    "tn_sumsq": [_vp, _vp, _vp, _ll, _i, _vp],
    "tn_adamw_step": [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _f, _f, _f, _f, _i, _vp],
    "tn_transpose_bf16": [_vp, _vp, _i, _i, _ll, _ll, _vp],
    "tn_swiglu_fwd_t": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "tn_swiglu_bwd_t": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "tn_colsum_workspace_floats": [_i, _i],
    "tn_colsum_bf16": [_vp, _vp, _vp, _i, _i, _ll, _vp],
    "tn_pcm16_to_f32": [_vp, _vp, _ll, _vp],
    "tn_bestrq_tokenize": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "tn_sumsq_multi_chunk": [],
    "tn_sumsq_multi": [_vp, _vp, _vp, _i, _ll, _vp, _vp, _i, _vp],
    "tn_adamw_multi_chunk": [],
    "tn_adamw_prepare": [_vp, _vp, _f, _f, _f, _vp],
    "tn_adamw_multi": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _vp, _f, _f, _f, _f, _f, _i, _vp],
    "tn_adamw_multi_bounded": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _ll, _vp, _f, _f, _f, _f, _f, _i, _i, _vp],
    "tn_pack_plan": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "tn_pack_fill": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _ll, _ll, _ll, _vp, _vp, _vp,
                     _vp, _vp, _vp, _vp, _vp],
    "tn_gemm_bf16_tn": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _ll, _ll, _ll, _ll, _i, _vp],
    "tn_gemm_bf16_addend": [_vp, _vp, _ll, _ll, _i, _i, _i, _vp, _vp, _vp, _ll, _i, _i, _ll, _vp],
    "tn_gemm_bf16_splitk": [_vp, _vp, _ll, _ll, _i, _i, _i, _vp, _vp, _i, _i, _ll, _i, _i, _i, _vp, _ll, _vp],
    "tn_gemm_bf16_wgrad_f32": [_vp, _vp, _ll, _ll, _i, _vp, _i, _i, _ll, _i, _i, _vp, _ll, _vp],
    "tn_gemm_bf16": [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_ll), C.POINTER(_ll), C.POINTER(_i), _i, _i, _i, _vp, _vp,
                     _vp, _i, _i, _ll, _ll, _i, _vp],
    "tn_gemm_grouped_workspace_bytes": [C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _i],
    "tn_gemm_bf16_grouped": [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_ll), C.POINTER(_ll), C.POINTER(_i), C.POINTER(_vp),
                             C.POINTER(_ll), C.POINTER(_i), C.POINTER(_i), _i, _i, _i, _i, _i, _vp, _ll, _vp],
    "tn_gemm_bf16_wgrad_bias": [_vp, _vp, _ll, _ll, _i, _vp, _vp, _i, _i, _ll, _i, _i, _i, _vp, _ll, _vp],
    "tn_gemm_bf16_rope": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _ll, _ll, _ll, _i, _vp],
    "tn_gemm_bf16_gelu_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _ll, _ll, _ll, _vp],
    "tn_gemm_bf16_gelu_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _ll, _ll, _ll, _vp],
    "tn_gemm_bf16_swiglu_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _ll, _ll, _ll, _vp],
    "tn_gemm_bf16_swiglu_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _ll, _ll, _ll, _vp],
    "tn_gemm_set_persistent": [_i],
    "tn_gemm_get_persistent": [],
}
_RESTYPE = {"tn_gemm_set_persistent": None, "tn_version": C.c_char_p, "tn_sumsq_multi_chunk": C.c_longlong, "tn_adamw_multi_chunk": C.c_longlong,
            "tn_colsum_workspace_floats": C.c_longlong, "tn_gemm_grouped_workspace_bytes": C.c_longlong}

# kernel-development entry points: exported by the library, deliberately NOT part of the C ABI (include/touchnet_amd.h)
DEV_PROTOTYPES = {
    "tn_attn_fwd_ablate": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],   # scripts/attn_ablate.py
    "tn_attn_fwd_stream_trace": [_vp],         # scripts/r06_attn_trace.py
    "tn_attn_set_fwd_schedule": [_i],
    "tn_attn_set_bwd_dq": [_i],                # A/B of the dQ kernels inside one process (scripts/r06_attn_bwd_ab.py)      # A/B of the forward schedules inside one process (scripts/r06_attn_ab.py)
}

_lib = None


class KernelError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) the HIP library.  Raises if it is absent — there is no CPU/eager fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -m touchnet_amd.build` "
                "(touchnet_amd has no fallback path; the HIP extension is mandatory)")
        handle = C.CDLL(LIB_PATH)
        for name, argtypes in {**PROTOTYPES, **DEV_PROTOTYPES}.items():
            fn = getattr(handle, name)      # AttributeError if the ABI and this stub ever drift
            fn.argtypes = argtypes
            fn.restype = _RESTYPE.get(name, C.c_int)
        _lib = handle
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        raise KernelError(f"{what} failed with code {code} "
                          f"({'EINVAL: unsupported shape/dtype' if code == -22 else 'hipError_t'})")


DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1}


def dcode(t: torch.Tensor) -> int:
    try:
        return DTYPE_CODE[t.dtype]
    except KeyError:
        raise KernelError(f"unsupported dtype {t.dtype} (float32 / bfloat16 only)") from None


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise KernelError("touchnet_amd kernels need device (HIP) tensors; got a CPU tensor")
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def version() -> str:
    return lib().tn_version().decode()
