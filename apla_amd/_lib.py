"""ctypes binding of libapla_hip.so (the C-ABI declared in include/apla_hip.h).

There is deliberately NO fallback: if the library has not been built (``python -m apla_amd.build``) or a kernel
returns an error, an exception is raised.  Nothing here imports the CPU oracle.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_uint, c_ulonglong, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# bf16 operands (default).  APLA_LIB=<path> substitutes another build of the library (tools/build_ablations.sh: A/B timing)
LIB_PATH = os.environ.get("APLA_LIB") or os.path.join(HERE, "libapla_hip.so")
LIB_PATH_F16 = os.path.join(HERE, "libapla_hip_f16.so")  # fp16 operands (same sources, -DAPLA_FP16)

APLA_BF16, APLA_F16, APLA_F32 = 0, 1, 2
EPI_STORE, EPI_GELU, EPI_RESIDUAL, EPI_MUL, EPI_SWIGLU, EPI_SWIGLU_BWD, EPI_GELU_FWD = 0, 1, 2, 3, 4, 5, 6

# name -> (restype, argtypes); must list every symbol of include/apla_hip.h (tests/test_cabi.py checks this)
SIGNATURES = {
    "apla_last_error": (c_char_p, []),
    "apla_version": (c_int, []),
    "apla_operand_dtype": (c_int, []),
    "apla_gemm_nt": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                             c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "apla_gemm_nt_ex": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "apla_gemm_small_workspace_bytes": (c_long, [c_int, c_int, c_int]),
    "apla_gemm_nt_small": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_long, c_void_p]),
    "apla_gemm_nt_splitk_workspace_bytes": (c_long, [c_int, c_int, c_int]),
    "apla_gemm_nt_splitk": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_long,
                                    c_void_p]),
    "apla_attn_fwd_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "apla_attn_bwd_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                 c_int, c_void_p]),
    "apla_attn_varlen_fwd_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "apla_attn_varlen_bwd_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                        c_int, c_float, c_int, c_void_p]),
    "apla_layernorm_fwd": (c_int, [c_void_p, c_int, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                   c_void_p, c_int, c_int, c_float, c_void_p, c_long, c_void_p, c_long, c_void_p]),
    "apla_layernorm_bwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_long, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_int, c_long, c_void_p, c_long, c_void_p, c_int, c_void_p,
                                   c_int, c_int, c_void_p]),
    "apla_layernorm_bwd_ex": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_long, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_int, c_void_p, c_int, c_long, c_void_p, c_long, c_void_p, c_int, c_void_p,
                                      c_int, c_int, c_void_p]),
    "apla_layernorm_fwd_dp": (c_int, [c_void_p, c_int, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                      c_void_p, c_int, c_int, c_float, c_void_p, c_long, c_void_p, c_long, c_void_p, c_int, c_void_p]),
    "apla_layernorm_bwd_dp": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_long, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_int, c_void_p, c_int, c_long, c_void_p, c_long, c_void_p, c_int, c_void_p,
                                      c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "apla_layernorm_fwd_drop": (c_int, [c_void_p, c_int, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                        c_void_p, c_int, c_int, c_float, c_void_p, c_long, c_void_p, c_long, c_void_p, c_int,
                                        c_void_p, c_ulonglong, c_uint, c_float, c_long, c_void_p]),
    "apla_layernorm_bwd_drop": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_long, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int, c_void_p, c_int, c_long, c_void_p, c_long, c_void_p, c_int, c_void_p,
                                        c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_long, c_void_p, c_void_p, c_ulonglong,
                                        c_uint, c_float, c_void_p]),
    "apla_dropout_fwd_dev": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_long, c_float, c_void_p, c_ulonglong, c_uint, c_void_p]),
    "apla_gemm_nt_gelu_drop": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int,
                                       c_void_p, c_ulonglong, c_uint, c_float, c_void_p]),
    "apla_gemm_nt_kernel_name": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_char_p, c_int]),
    "apla_attn_kernel_name": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_char_p, c_int]),
    "apla_attn_fwd_dropout": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_ulonglong, c_uint, c_void_p]),
    "apla_attn_bwd_dropout": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float,
                                      c_ulonglong, c_uint, c_void_p]),
    "apla_probe_occupy": (c_int, [c_int, c_int, c_int, c_int, c_void_p]),
    "apla_gather_cols": (c_int, [c_void_p, c_int, c_long, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "apla_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "apla_attn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                              c_void_p]),
    "apla_attn_varlen_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "apla_attn_varlen_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                     c_int, c_float, c_void_p]),
    "apla_attn_fwd_cls": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "apla_attn_bwd_cls": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "apla_attn_probs": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "apla_attn_probs_dropout": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_ulonglong, c_uint, c_void_p]),
    "apla_dw_workspace_bytes": (c_long, [c_int, c_int, c_int]),
    "apla_proj_dw": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                             c_int, c_void_p]),
    "apla_gemm_nt_panel_ok": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "apla_gemm_nt_out_image_ok": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "apla_pack_k_panels": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, c_void_p]),
    "apla_pack_proj_rows_batched_ex": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_int, c_int, c_int, c_void_p]),
    "apla_dw_workspace_bytes_batched": (c_long, [c_int, c_int, c_int, c_int]),
    "apla_proj_dw_batched": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                     c_int, c_void_p]),
    "apla_proj_workspace_bytes": (c_long, [c_int, c_int, c_int]),
    "apla_proj_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "apla_proj_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int,
                              c_int, c_int, c_void_p]),
    "apla_pack_proj_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                    c_void_p]),
    "apla_pack_proj_rows_batched": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                            c_int, c_void_p]),
    "apla_adamw_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float,
                                c_float, c_float, c_int, c_float, c_float, c_void_p, c_void_p]),
    "apla_ema_update": (c_int, [c_void_p, c_void_p, c_long, c_double, c_void_p]),
    "apla_grad_sumsq": (c_int, [c_void_p, c_long, c_float, c_void_p, c_void_p]),
    "apla_adamw_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float,
                                 c_float, c_float, c_int, c_float, c_float, c_void_p, c_void_p]),
    "apla_adamw_step_dynamic": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float,
                                        c_float, c_float, c_float, c_float, c_void_p, c_int, c_float, c_float, c_int,
                                        c_void_p, c_void_p]),
    "apla_patchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "apla_softmax_center": (c_int, [c_void_p, c_int, c_long, c_void_p, c_float, c_void_p, c_long, c_int, c_int, c_void_p]),
    "apla_distill_ce": (c_int, [c_void_p, c_int, c_long, c_void_p, c_long, c_float, c_void_p, c_float, c_void_p, c_long, c_int,
                                c_void_p, c_int, c_int, c_void_p]),
    "apla_distill_ce_bcast": (c_int, [c_void_p, c_int, c_long, c_void_p, c_long, c_int, c_float, c_void_p, c_float, c_void_p, c_int, c_long,
                                      c_int, c_void_p, c_int, c_int, c_void_p]),
    "apla_distill_ce_ex": (c_int, [c_void_p, c_int, c_long, c_void_p, c_long, c_float, c_void_p, c_float, c_void_p, c_int, c_long, c_int,
                                   c_void_p, c_int, c_int, c_void_p]),
    "apla_distill_ce_centered": (c_int, [c_void_p, c_int, c_long, c_void_p, c_int, c_long, c_void_p, c_float, c_float, c_void_p, c_float,
                                         c_void_p, c_int, c_long, c_void_p, c_int, c_int, c_void_p]),
    "apla_augment_images": (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "apla_assemble_tokens": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     c_void_p]),
    "apla_weight_norm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "apla_weight_norm_fwd_t": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "apla_weight_norm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "apla_koleo_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "apla_koleo_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "apla_dropout_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_long, c_float, c_ulonglong, c_ulonglong, c_void_p]),
    "apla_dropout_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_long, c_float, c_void_p]),
    "apla_scale_samples": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_long, c_long, c_void_p]),
    "apla_assemble_tokens_masked": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                            c_int, c_void_p]),
    "apla_sgemm_small": (c_int, [c_void_p, c_long, c_long, c_void_p, c_long, c_long, c_void_p, c_void_p, c_long, c_int,
                                 c_int, c_int, c_int, c_void_p]),
    "apla_cross_entropy": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "apla_cross_entropy_soft": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "apla_colsum": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, c_void_p]),
    "apla_colsum_h16": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, c_void_p]),
}

_libs = {}      # operand code (APLA_BF16 / APLA_F16) -> ctypes handle
_current = APLA_BF16  # which library lib() returns: switched by ops.use_half


class AplaHipError(RuntimeError):
    pass


def lib(operand=None):
    """Load (once) and return the ctypes handle of the library built for `operand` (default: the current one, bf16 unless
    inside ops.use_half(torch.float16)); raises if that library is missing."""
    code = _current if operand is None else operand
    if code not in _libs:
        path = LIB_PATH if code == APLA_BF16 else LIB_PATH_F16
        if not os.path.exists(path):
            raise AplaHipError(
                f"{path} not found: the APLA HIP kernels are not built. Run `python -m apla_amd.build` "
                "(needs hipcc, targets gfx950). There is no CPU fallback.")
        # torch first: it ships its own libamdhip64; loaded before ours, the dynamic linker binds libapla_hip.so to that same
        # runtime (same SONAME).  The other order leaves two HIP runtimes in the process and ours without a device context
        # ("no ROCm-capable device is detected" at the first launch).
        import torch  # noqa: F401
        handle = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.apla_operand_dtype() != code:
            raise AplaHipError(f"{path} was built for operand dtype {handle.apla_operand_dtype()}, expected {code}")
        _libs[code] = handle
    return _libs[code]


def set_current(code):
    global _current
    old, _current = _current, code
    return old


def check(rc, what):
    if rc != 0:
        msg = lib().apla_last_error()
        raise AplaHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
