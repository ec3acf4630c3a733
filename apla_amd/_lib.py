"""ctypes binding of libapla_hip.so (the C-ABI declared in include/apla_hip.h).

There is deliberately NO fallback: if the library has not been built (``python -m apla_amd.build``) or a kernel
returns an error, an exception is raised.  Nothing here imports the CPU oracle.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libapla_hip.so")

APLA_BF16, APLA_F32 = 0, 2
EPI_STORE, EPI_GELU, EPI_RESIDUAL, EPI_MUL, EPI_SWIGLU, EPI_SWIGLU_BWD = 0, 1, 2, 3, 4, 5

# name -> (restype, argtypes); must list every symbol of include/apla_hip.h (tests/test_cabi.py checks this)
SIGNATURES = {
    "apla_last_error": (c_char_p, []),
    "apla_version": (c_int, []),
    "apla_gemm_nt": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                             c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "apla_gemm_set_variant": (c_int, [c_int]),
    "apla_attn_set_variant": (c_int, [c_int]),
    "apla_layernorm_fwd": (c_int, [c_void_p, c_int, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                   c_void_p, c_int, c_int, c_float, c_void_p, c_long, c_void_p, c_long, c_void_p]),
    "apla_layernorm_bwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_long, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_int, c_long, c_void_p, c_long, c_void_p, c_int, c_void_p,
                                   c_int, c_int, c_void_p]),
    "apla_gather_cols": (c_int, [c_void_p, c_int, c_long, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "apla_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "apla_attn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                              c_void_p]),
    "apla_attn_varlen_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "apla_attn_varlen_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                     c_int, c_float, c_void_p]),
    "apla_attn_bwd_cls": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "apla_attn_probs": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "apla_dw_workspace_bytes": (c_long, [c_int, c_int, c_int]),
    "apla_proj_dw": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                             c_int, c_void_p]),
    "apla_pack_proj_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                    c_void_p]),
    "apla_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float,
                                c_float, c_float, c_int, c_float, c_float, c_void_p, c_void_p]),
    "apla_patchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "apla_assemble_tokens": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     c_void_p]),
    "apla_sgemm_small": (c_int, [c_void_p, c_long, c_long, c_void_p, c_long, c_long, c_void_p, c_void_p, c_long, c_int,
                                 c_int, c_int, c_int, c_void_p]),
    "apla_cross_entropy": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "apla_colsum": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, c_void_p]),
}

_lib = None


class AplaHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AplaHipError(
                f"{LIB_PATH} not found: the APLA HIP kernels are not built. Run `python -m apla_amd.build` "
                "(needs hipcc, targets gfx950). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().apla_last_error()
        raise AplaHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
