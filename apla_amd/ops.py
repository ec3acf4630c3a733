"""Thin torch-tensor front ends of the C-ABI kernels (include/apla_hip.h).

PyTorch is used here for device memory and streams only: every function checks its operands on the host (device,
dtype, contiguity, shapes — a mis-shaped launch can fault the GPU), passes raw pointers and the current HIP stream to
libapla_hip.so and raises on a non-zero return code.  No function has a CPU path.
"""
import os
from typing import Optional

import torch

from . import _lib
from ._lib import (APLA_BF16, APLA_F16, APLA_F32, EPI_GELU, EPI_GELU_FWD, EPI_MUL, EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU,  # noqa: F401
                   EPI_SWIGLU_BWD, check, lib)

_DT = {torch.bfloat16: APLA_BF16, torch.float16: APLA_F16, torch.float32: APLA_F32}
_HALF = torch.bfloat16  # the 16-bit operand type the wrappers currently accept (and the library lib() returns)


def half():
    """The current 16-bit operand dtype: torch.bfloat16 unless inside ``use_half(torch.float16)``."""
    return _HALF


class use_half:
    """Context manager: run the wrappers against the fp16 build of the library (libapla_hip_f16.so; every "bf16" operand is
    then torch.float16).  The kernels, layouts and entry points are identical; only the operand rounding differs (8x
    smaller), and fp16 gradients need a loss scale (AplaTrainEngine(loss_scale=...))."""

    def __init__(self, dtype):
        if dtype not in (torch.bfloat16, torch.float16):
            raise TypeError("operand dtype must be torch.bfloat16 or torch.float16")
        self.dtype = dtype

    def __enter__(self):
        global _HALF
        self._old = (_HALF, _lib.set_current(_DT[self.dtype]))
        _HALF = self.dtype
        lib()  # fail here, loudly, if that build is missing
        return self

    def __exit__(self, *exc):
        global _HALF
        _HALF = self._old[0]
        _lib.set_current(self._old[1])
        return False


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str, ndim: Optional[int] = None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.AplaHipError(f"{name}: expected a CUDA/HIP tensor (the APLA kernels have no CPU path)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if ndim is not None and t.ndim != ndim:
        raise ValueError(f"{name}: expected {ndim}-D, got shape {tuple(t.shape)}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")
    return t


def _rows2d(t: torch.Tensor, name: str):
    """2-D view metadata (rows, cols, leading dimension) of a row-major matrix."""
    if t.ndim != 2:
        raise ValueError(f"{name}: expected 2-D, got {tuple(t.shape)}")
    return t.shape[0], t.shape[1], t.stride(0)


# Kernel-schedule choice of the GEMM / attention wrappers: test and benchmark hooks kept HERE, in Python (the C-ABI itself is
# stateless: the choice travels as a per-call argument of the *_ex entry points).  0 = auto.
_GEMM_VARIANT = 0
_DW_PAD = {}         # proj_dw: padded operands of widths that are multiples of 64 only
_ATTN_VARIANT = int(os.environ.get("APLA_ATTN_VARIANT", "0"))
_GEMM_EXP = 0       # experiment selector of apla_gemm_nt_ex (flags bits 28-30): tools/gemm_bench.py only
_RESERVED_CUS = 0   # CUs the persistent GEMM launches leave free (apla_gemm_nt_ex flags bits 20-27): see reserved_cus()
# profiling tags of apla_gemm_nt_ex (kernel names in a rocprofv3 trace): call sites of the training step
TAG_QKV, TAG_PROJ, TAG_FC2, TAG_DFC1, TAG_DPROJ, TAG_DQKV, TAG_PATCH = 2, 3, 4, 5, 6, 7, 8


def k_panels(m: torch.Tensor) -> torch.Tensor:
    """The K-panel image [K/32, rows, 32] of a 16-bit [rows, K] matrix (include/apla_hip.h:apla_pack_k_panels)."""
    _req(m, half(), "m", 2)
    rows, K, ld = _rows2d(m, "m")
    if K % 32:
        raise ValueError("k_panels: K % 32 != 0")
    out = torch.empty(K // 32, rows, 32, device=m.device, dtype=m.dtype)
    check(lib().apla_pack_k_panels(m.data_ptr(), ld, out.data_ptr(), rows, K, _stream()), "apla_pack_k_panels")
    return out


def gemm_out_image_ok(M: int, N: int, K: int, epilogue: int, out_dtype=None) -> bool:
    """May this problem write its output as a K-panel image? (apla_gemm_nt_out_image_ok)"""
    return bool(lib().apla_gemm_nt_out_image_ok(M, N, K, epilogue, _DT[out_dtype or half()]))


def gemm_panel_ok(M: int, N: int, K: int, epilogue: int = 0, out_dtype=None) -> bool:
    """May this problem take K-panel operand images? (apla_gemm_nt_panel_ok)"""
    return bool(lib().apla_gemm_nt_panel_ok(M, N, K, epilogue, _DT[out_dtype or half()]))


class reserved_cus:
    """Context manager: the GEMM launches issued inside leave `n` CUs free for a concurrent collective (data-parallel step)."""

    def __init__(self, n: int):
        if not 0 <= int(n) < 192:
            raise ValueError("reserved_cus: 0..191")
        self.n = int(n)

    def __enter__(self):
        global _RESERVED_CUS
        self._old, _RESERVED_CUS = _RESERVED_CUS, self.n
        return self

    def __exit__(self, *exc):
        global _RESERVED_CUS
        _RESERVED_CUS = self._old
        return False


def set_gemm_variant(v: int) -> int:
    """Pin the GEMM kernel schedule for subsequent ops.gemm_nt calls (include/apla_hip.h:apla_gemm_nt_ex); returns the old value."""
    global _GEMM_VARIANT
    old, _GEMM_VARIANT = _GEMM_VARIANT, int(v)
    return old


def set_attn_variant(v: int) -> int:
    """Pin the attention kernel choice for subsequent ops.attn_* calls (apla_attn_*_ex); returns the old value."""
    global _ATTN_VARIANT
    old, _ATTN_VARIANT = _ATTN_VARIANT, int(v)
    return old


def gemm_nt(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, epilogue: int = EPI_STORE,
            out: Optional[torch.Tensor] = None, out_dtype=None, aux_in: Optional[torch.Tensor] = None,
            aux_out: Optional[torch.Tensor] = None, tag: int = 0, drop=None) -> torch.Tensor:
    """out[M,Nout] = epilogue(a[M,K] @ w[N,K]^T + bias).  See include/apla_hip.h:apla_gemm_nt / apla_gemm_nt_ex (`tag`: profiling
    tag of the call site, TAG_*).  ``drop`` = (rng, stride, site, p) with EPI_GELU: Mlp.drop after the activation inside the epilogue
    (apla_gemm_nt_gelu_drop; rng = device int64 [2] {seed, step}).  A 3-D contiguous operand [K/32, rows, 32] is taken as its K-panel image (k_panels()); a 3-D `out`
    [N/32, M, 32] is written as one (GELU / GELU_FWD / MUL epilogues)."""
    _req(a, half(), "a", None), _req(w, half(), "w", None)
    panel = 0
    if w.ndim == 3:
        if w.shape[2] != 32 or not w.is_contiguous():
            raise ValueError("gemm_nt: a K-panel image is a contiguous [K/32, N, 32] tensor")
        N, Kw, ldw, panel = w.shape[1], w.shape[0] * 32, 32, 1
    else:
        _req(w, half(), "w", 2)
        N, Kw, ldw = _rows2d(w, "w")
    if a.ndim == 3:
        if a.shape[2] != 32 or not a.is_contiguous():
            raise ValueError("gemm_nt: a K-panel image is a contiguous [K/32, M, 32] tensor")
        M, K, lda, panel = a.shape[1], a.shape[0] * 32, 32, panel | 2
    else:
        _req(a, half(), "a", 2)
        M, K, lda = _rows2d(a, "a")
    if K != Kw:
        raise ValueError(f"gemm_nt: K mismatch {K} vs {Kw}")
    if bias is not None:
        _req(bias, torch.float32, "bias", 1)
        if bias.numel() != N:
            raise ValueError("gemm_nt: bias length != N")
    n_out = {EPI_SWIGLU: N // 2, EPI_SWIGLU_BWD: 2 * N}.get(epilogue, N)
    if out is None:
        out = torch.empty(M, n_out, device=a.device, dtype=out_dtype or half())
    ldc = None
    if out.ndim == 3:   # the output as a K-panel image [n_out/32, M, 32]: the A operand of the next GEMM
        _req(out, half(), "out", 3)
        if tuple(out.shape) != (n_out // 32, M, 32) or n_out % 32 or not out.is_contiguous():
            raise ValueError(f"gemm_nt: an output image is a contiguous [{n_out // 32}, {M}, 32] tensor")
        panel, ldc = panel | 4, n_out
    else:
        _req(out, None, "out", 2)
        if out.shape != (M, n_out):
            raise ValueError(f"gemm_nt: out shape {tuple(out.shape)} != {(M, n_out)}")
        ldc = out.stride(0)
    ld_in = ld_out = 0
    aux = aux_in if aux_in is not None else aux_out
    if aux is not None and aux.ndim == 3:   # gelu' kept as a K-panel image [N/32, M, 32] between the GELU and MUL epilogues
        _req(aux, half(), "aux", 3)
        if epilogue not in (EPI_GELU, EPI_MUL) or tuple(aux.shape) != (N // 32, M, 32) or N % 32 or not aux.is_contiguous():
            raise ValueError(f"gemm_nt: a second-operand image is a contiguous [{N // 32}, {M}, 32] tensor (GELU / MUL epilogues)")
        panel |= 8
        ld_in, ld_out = (N, 0) if aux_in is not None else (0, N)
        aux_view = aux.view(M, N)       # shape bookkeeping below only
        aux_in, aux_out = (aux_view, None) if aux_in is not None else (None, aux_view)
    if aux_in is not None:
        _req(aux_in, None, "aux_in", 2)
        need = {EPI_RESIDUAL: (M, N), EPI_MUL: (M, N), EPI_SWIGLU_BWD: (M, 2 * N)}.get(epilogue)
        if need is None or tuple(aux_in.shape) != need:
            raise ValueError(f"gemm_nt: aux_in shape {tuple(aux_in.shape)} invalid for epilogue {epilogue}")
        if epilogue == EPI_RESIDUAL and aux_in.dtype != out.dtype:
            raise TypeError("gemm_nt: residual and output dtypes differ")
        if epilogue in (EPI_MUL, EPI_SWIGLU_BWD) and aux_in.dtype != half():
            raise TypeError("gemm_nt: aux_in must be bf16")
        ld_in = aux_in.stride(0)
    if aux_out is not None:
        _req(aux_out, half(), "aux_out", 2)
        if tuple(aux_out.shape) != (M, N):
            raise ValueError("gemm_nt: aux_out shape")
        ld_out = aux_out.stride(0)
    if epilogue in (EPI_GELU, EPI_SWIGLU) and aux_out is None:
        raise ValueError("gemm_nt: epilogue needs aux_out")
    if epilogue in (EPI_RESIDUAL, EPI_MUL, EPI_SWIGLU_BWD) and aux_in is None:
        raise ValueError("gemm_nt: epilogue needs aux_in")
    if epilogue not in (EPI_STORE, EPI_RESIDUAL) and out.dtype != half():
        raise TypeError("gemm_nt: this epilogue writes bf16")
    if epilogue == EPI_GELU_FWD and aux_out is not None:
        raise ValueError("gemm_nt: EPI_GELU_FWD saves nothing (use EPI_GELU to get gelu')")
    if drop is not None:
        rng, dstride, dsite, dp_ = drop
        if epilogue != EPI_GELU or panel & 3 or rng.dtype != torch.int64 or rng.numel() != 2 or not rng.is_cuda:
            raise ValueError("gemm_nt: drop needs the EPI_GELU epilogue, row-major operands and a device int64 [2] rng tensor")
        check(lib().apla_gemm_nt_gelu_drop(a.data_ptr(), lda, w.data_ptr(), ldw, _ptr(bias), out.data_ptr(), ldc, M, N, K, _ptr(aux_out), ld_out,
                                           (int(tag) & 0xff) | (panel << 16) | (_RESERVED_CUS << 20), rng.data_ptr(), int(dstride), int(dsite),
                                           float(dp_), _stream()), "apla_gemm_nt_gelu_drop")
        return out
    rc = lib().apla_gemm_nt_ex(a.data_ptr(), lda, w.data_ptr(), ldw, _ptr(bias), out.data_ptr(), ldc, M, N, K,
                               epilogue, _DT[out.dtype], _ptr(aux_in), ld_in, _ptr(aux_out), ld_out,
                               (int(tag) & 0xff) | (_GEMM_VARIANT << 8) | (panel << 16) | (_RESERVED_CUS << 20) | (_GEMM_EXP << 28), _stream())
    check(rc, "apla_gemm_nt")
    return out


_SPLITK_WS = {}


def gemm_nt_splitk(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, out: Optional[torch.Tensor] = None,
                   out_dtype=None) -> torch.Tensor:
    """out[M,N] = a[M,K] @ w[N,K]^T (+ bias) for few output tiles and a long K (include/apla_hip.h:apla_gemm_nt_splitk): the K axis is
    cut into parts that run side by side, fp32 partials summed in a fixed order.  The workspace is cached per (device, size)."""
    _req(a, half(), "a", 2), _req(w, half(), "w", 2)
    (M, K, lda), (N, Kw, ldw) = _rows2d(a, "a"), _rows2d(w, "w")
    if K != Kw:
        raise ValueError(f"gemm_nt_splitk: K mismatch {K} vs {Kw}")
    if bias is not None:
        _req(bias, torch.float32, "bias", 1)
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=out_dtype or half())
    nbytes = lib().apla_gemm_nt_splitk_workspace_bytes(M, N, K)
    if nbytes < 0:
        raise ValueError(f"gemm_nt_splitk: shape not covered (N % 256 == 0, K % 32 == 0): {(M, N, K)}")
    key = (a.device, nbytes)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        _SPLITK_WS.clear()     # one live workspace: the shapes of a training loop repeat
        ws = _SPLITK_WS[key] = torch.empty(nbytes // 4, device=a.device, dtype=torch.float32)
    check(lib().apla_gemm_nt_splitk(a.data_ptr(), lda, w.data_ptr(), ldw, _ptr(bias), out.data_ptr(), out.stride(0), M, N, K,
                                    _DT[out.dtype], ws.data_ptr(), nbytes, _stream()), "apla_gemm_nt_splitk")
    return out


def gemm_splitk_wanted(M: int, N: int, K: int) -> bool:
    """Few tiles of the tiled kernels (fewer than half the CUs) and a K long enough to cut."""
    return N % 256 == 0 and K % 32 == 0 and K >= 4096 and ((M + 159) // 160) * (N // 256) < 128


def gemm_small_workspace(M: int, N: int, K: int, device) -> Optional[torch.Tensor]:
    """Workspace of gemm_nt_small for this shape, or None when the few-row kernel does not take it."""
    nbytes = lib().apla_gemm_small_workspace_bytes(M, N, K)
    return None if nbytes < 0 else torch.empty(nbytes // 4, device=device, dtype=torch.float32)


def gemm_nt_small(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, workspace: torch.Tensor,
                  epilogue: int = EPI_STORE, out: Optional[torch.Tensor] = None, out_dtype=None,
                  aux_in: Optional[torch.Tensor] = None, aux_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """gemm_nt for few rows (split-K, two launches); see include/apla_hip.h:apla_gemm_nt_small."""
    _req(a, half(), "a", 2), _req(w, half(), "w", 2), _req(workspace, torch.float32, "workspace", 1)
    M, K, lda = _rows2d(a, "a")
    N, Kw, ldw = _rows2d(w, "w")
    if K != Kw:
        raise ValueError(f"gemm_nt_small: K mismatch {K} vs {Kw}")
    if bias is not None:
        _req(bias, torch.float32, "bias", 1)
        if bias.numel() != N:
            raise ValueError("gemm_nt_small: bias length != N")
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=out_dtype or half())
    _req(out, None, "out", 2)
    if tuple(out.shape) != (M, N):
        raise ValueError("gemm_nt_small: out shape")
    for t_, nm in ((aux_in, "aux_in"), (aux_out, "aux_out")):
        if t_ is not None:
            _req(t_, None, nm, 2)
            if tuple(t_.shape) != (M, N):
                raise ValueError(f"gemm_nt_small: {nm} shape")
    check(lib().apla_gemm_nt_small(a.data_ptr(), lda, w.data_ptr(), ldw, _ptr(bias), out.data_ptr(), out.stride(0), M, N, K, epilogue,
                                   _DT[out.dtype], _ptr(aux_in), aux_in.stride(0) if aux_in is not None else 0, _ptr(aux_out),
                                   aux_out.stride(0) if aux_out is not None else 0, workspace.data_ptr(), workspace.numel() * 4,
                                   _stream()), "apla_gemm_nt_small")
    return out


def gemm_kernel_name(M: int, N: int, K: int, epilogue: int = EPI_STORE, out_dtype=None, *, out_image: bool = False,
                     aux_image: bool = False) -> str:
    """Name of the kernel apla_gemm_nt_ex dispatches this problem to under the current schedule choice (apla_gemm_nt_kernel_name)."""
    import ctypes
    buf = ctypes.create_string_buffer(96)
    panel = (4 if out_image else 0) | (8 if aux_image else 0)
    check(lib().apla_gemm_nt_kernel_name(M, N, K, epilogue, _DT[out_dtype or half()], (_GEMM_VARIANT << 8) | (panel << 16) | (_RESERVED_CUS << 20) | (_GEMM_EXP << 28), buf, 96),
          "apla_gemm_nt_kernel_name")
    return buf.value.decode()


def layernorm_fwd(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float = 1e-6, *,
                  out_dtype=None, rows: Optional[int] = None, row_stride: Optional[int] = None,
                  out: Optional[torch.Tensor] = None, mean: Optional[torch.Tensor] = None,
                  rstd: Optional[torch.Tensor] = None, add: Optional[torch.Tensor] = None,
                  x_out: Optional[torch.Tensor] = None, add_row_stride: Optional[int] = None, D: Optional[int] = None,
                  add_scale: Optional[torch.Tensor] = None, scale_period: int = 1, drop=None):
    """x: [M,D] residual stream (fp32|bf16).  With rows/row_stride given, x (and add / x_out) are flat buffers and row m
    starts at m*row_stride (CLS-row selection).  With ``add`` (bf16 branch output) the residual update x_out = x + add is
    fused (x_out may alias x).  gamma = beta = None: y is the normalised row itself (pass ``D``).
    ``add_scale`` (fp32, one per sample; ``scale_period`` rows per sample): stochastic depth, x_out = x + add_scale[m // scale_period] * add.
    ``drop`` = (rng, stride, site, p, index_row_stride): element-wise dropout of ``add`` inside the kernel (apla_layernorm_fwd_drop);
    ``rng`` is a device int64 [2] tensor {seed, step}.
    Returns (y [M,D], mean [M], rstd [M])."""
    _req(x, None, "x")
    if (gamma is None) != (beta is None) or (gamma is None and D is None and rows is not None):
        raise ValueError("layernorm_fwd: gamma and beta go together; without them give D for strided rows")
    D = gamma.numel() if gamma is not None else (D if D is not None else x.shape[-1])
    if rows is None:
        M, Dx, xs = _rows2d(x, "x")
        if Dx != D:
            raise ValueError("layernorm_fwd: feature size mismatch")
    else:
        M, xs = rows, row_stride
        if (M - 1) * xs + D > x.numel():
            raise ValueError("layernorm_fwd: strided rows exceed the buffer")
    if gamma is not None:
        _req(gamma, torch.float32, "gamma", 1), _req(beta, torch.float32, "beta", 1)
        if beta.numel() != D:
            raise ValueError("layernorm_fwd: beta length")
    if out is None:
        out = torch.empty(M, D, device=x.device, dtype=out_dtype or half())
    if mean is None:
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
    if rstd is None:
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    _req(out, None, "out", 2), _req(mean, torch.float32, "mean", 1), _req(rstd, torch.float32, "rstd", 1)
    if tuple(out.shape) != (M, D) or mean.numel() != M or rstd.numel() != M:
        raise ValueError("layernorm_fwd: bad output buffers")
    adds = xouts = 0
    if add is not None:
        _req(add, half(), "add"), _req(x_out, x.dtype, "x_out")
        if rows is None:
            if tuple(add.shape) != (M, D) or tuple(x_out.shape) != (M, D):
                raise ValueError("layernorm_fwd: add / x_out shape")
            adds, xouts = add.stride(0), x_out.stride(0)
        else:
            xouts = row_stride
            adds = row_stride if add_row_stride is None else add_row_stride  # e.g. a compact [M, D] branch added into strided rows
            if (M - 1) * adds + D > add.numel() or (M - 1) * row_stride + D > x_out.numel():
                raise ValueError("layernorm_fwd: strided add / x_out exceed their buffers")
    if add_scale is not None:
        _req(add_scale, torch.float32, "add_scale", 1)
        if add is None or scale_period < 1 or add_scale.numel() * scale_period < M:
            raise ValueError("layernorm_fwd: add_scale needs `add` and one entry per sample (M <= len * scale_period)")
    rng, dstride, dsite, dp_, drow = (None, 0, 0, 0.0, D) if drop is None else drop
    if rng is not None and (rng.dtype != torch.int64 or rng.numel() != 2 or not rng.is_cuda or add is None):
        raise ValueError("layernorm_fwd: drop needs a device int64 [2] rng tensor {seed, step} and `add`")
    rc = lib().apla_layernorm_fwd_drop(x.data_ptr(), _DT[x.dtype], xs, _ptr(gamma), _ptr(beta), out.data_ptr(),
                                       _DT[out.dtype], out.stride(0), mean.data_ptr(), rstd.data_ptr(), M, D, float(eps),
                                       _ptr(add), adds, _ptr(x_out), xouts, _ptr(add_scale), int(scale_period),
                                       _ptr(rng), int(dstride), int(dsite), float(dp_), int(drow), _stream())
    check(rc, "apla_layernorm_fwd")
    return out, mean, rstd


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: Optional[torch.Tensor], mean: Optional[torch.Tensor],
                  rstd: torch.Tensor, *,
                  dres: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                  out_bf16: Optional[torch.Tensor] = None, inds: Optional[torch.Tensor] = None, r: int = 0,
                  gathered: Optional[torch.Tensor] = None, rows: Optional[int] = None,
                  row_stride: Optional[int] = None, x_row_stride: Optional[int] = None, dres_period: int = 0,
                  dy_scale: Optional[torch.Tensor] = None, gather_scale: Optional[torch.Tensor] = None, scale_period: int = 1,
                  masked: Optional[torch.Tensor] = None, mask_scale: Optional[torch.Tensor] = None, drop=None):
    """dx = dres + LN_bwd_dx(dy).  Returns (dx, gathered | None).  ``out`` (the gradient stream, fp32|bf16) may alias
    ``dres``; ``out_bf16`` optionally receives a bf16 copy.  With rows/row_stride: x, out (and out_bf16) are flat
    buffers whose row m starts at m*row_stride (only those rows are read/written); ``x_row_stride`` gives x its own pitch.
    mean = None: ``x`` is the normalised row saved by layernorm_fwd(gamma=None); gamma = None: no affine part;
    dres_period p > 1: dres is read in rows m % p == 0 only and taken as zero elsewhere (apla_layernorm_bwd_ex).
    Stochastic depth (apla_layernorm_bwd_dp): dx = dres + dy_scale[m // scale_period] * LN_bwd_dx(dy), gathered columns times
    gather_scale[m // scale_period]; fp32 vectors with one entry per sample.
    ``masked`` [M,D] 16-bit + ``drop`` = (rng, stride, site, p) (+ ``mask_scale`` per sample): also writes dx through the dropout mask of
    the branch that consumes it next, and gathers from that copy (apla_layernorm_bwd_drop; normalised-row 16-bit form only)."""
    _req(dy, None, "dy", 2), _req(x, None, "x")
    M, D, lddy = _rows2d(dy, "dy")
    if rows is None:
        Mx, Dx, xs = _rows2d(x, "x")
        if (Mx, Dx) != (M, D):
            raise ValueError("layernorm_bwd: x/dy shape mismatch")
        if out is None:
            out = torch.empty(M, D, device=x.device, dtype=dres.dtype if dres is not None else x.dtype)
        if tuple(out.shape) != (M, D):
            raise ValueError("layernorm_bwd: out shape")
        dxs = out.stride(0)
        cbs = out_bf16.stride(0) if out_bf16 is not None else 0
        if out_bf16 is not None and tuple(out_bf16.shape) != (M, D):
            raise ValueError("layernorm_bwd: out_bf16 shape")
    else:
        if rows != M or out is None:
            raise ValueError("layernorm_bwd: strided mode needs rows == dy rows and an out buffer")
        dxs = cbs = row_stride
        xs = row_stride if x_row_stride is None else x_row_stride
        if (M - 1) * xs + D > x.numel() or (M - 1) * dxs + D > out.numel() or \
                (out_bf16 is not None and (M - 1) * cbs + D > out_bf16.numel()):
            raise ValueError("layernorm_bwd: strided rows exceed the buffer")
    _req(out, None, "out")
    if dres is not None and (dres.dtype != out.dtype or dres.shape != out.shape or dres.stride() != out.stride()):
        raise TypeError("layernorm_bwd: dres must match out (dtype, shape, strides)")
    if out_bf16 is not None:
        _req(out_bf16, half(), "out_bf16")
    if inds is not None:
        _req(inds, torch.int32, "inds", 1)
        if not (0 < r <= D) or inds.numel() < r:
            raise ValueError("layernorm_bwd: bad r")
        if gathered is None:
            gathered = torch.empty(M, r, device=x.device, dtype=half())
        _req(gathered, half(), "gathered", 2)
        if tuple(gathered.shape) != (M, r) or not gathered.is_contiguous():
            raise ValueError("layernorm_bwd: gathered buffer")
    else:
        gathered = None
    if gamma is not None:
        _req(gamma, torch.float32, "gamma", 1)
    if mean is not None:
        _req(mean, torch.float32, "mean", 1)
    _req(rstd, torch.float32, "rstd", 1)
    if rstd.numel() < M or (mean is not None and mean.numel() < M) or (gamma is not None and gamma.numel() != D) or dres_period < 0:
        raise ValueError("layernorm_bwd: statistics / gamma sizes")
    for sc, nm in ((dy_scale, "dy_scale"), (gather_scale, "gather_scale")):
        if sc is not None:
            _req(sc, torch.float32, nm, 1)
            if scale_period < 1 or sc.numel() * scale_period < M:
                raise ValueError(f"layernorm_bwd: {nm} needs one entry per sample (M <= len * scale_period)")
    rng, dstride, dsite, dp_ = (None, 0, 0, 0.0) if drop is None else drop
    if masked is not None:
        _req(masked, half(), "masked", 2)
        if rng is None or rng.dtype != torch.int64 or rng.numel() != 2 or tuple(masked.shape) != (M, D) or rows is not None:
            raise ValueError("layernorm_bwd: masked needs drop=(rng int64 [2], stride, site, p), an [M, D] buffer and dense rows")
        if mask_scale is not None:
            _req(mask_scale, torch.float32, "mask_scale", 1)
    rc = lib().apla_layernorm_bwd_drop(dy.data_ptr(), _DT[dy.dtype], lddy, x.data_ptr(), _DT[x.dtype], xs, _ptr(gamma),
                                       _ptr(mean), rstd.data_ptr(), _ptr(dres), int(dres_period), out.data_ptr(), _DT[out.dtype], dxs,
                                       _ptr(out_bf16), cbs, _ptr(inds), r, _ptr(gathered), M, D, _ptr(dy_scale), _ptr(gather_scale),
                                       int(scale_period), _ptr(masked), masked.stride(0) if masked is not None else 0, _ptr(mask_scale),
                                       _ptr(rng if masked is not None else None), int(dstride), int(dsite), float(dp_), _stream())
    check(rc, "apla_layernorm_bwd")
    return out, gathered


def gather_cols(src: torch.Tensor, inds: torch.Tensor, r: int, *, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(src, None, "src", 2), _req(inds, torch.int32, "inds", 1)
    M, D, ss = _rows2d(src, "src")
    if out is None:
        out = torch.empty(M, r, device=src.device, dtype=half())
    if tuple(out.shape) != (M, r) or out.dtype != half() or not out.is_contiguous():
        raise ValueError("gather_cols: bad output buffer")
    check(lib().apla_gather_cols(src.data_ptr(), _DT[src.dtype], ss, inds.data_ptr(), r, out.data_ptr(), M, D,
                                 _stream()), "apla_gather_cols")
    return out


def attn_kernel_name(which: str, B: int, N: int, H: int, *, packed: bool = False) -> str:
    """Name of the kernel apla_attn_fwd / apla_attn_bwd (which = "fwd" | "bwd") dispatch this problem to under the current variant."""
    import ctypes
    buf = ctypes.create_string_buffer(96)
    check(lib().apla_attn_kernel_name(1 if which == "bwd" else 0, 1 if packed else 0, B, N, H, _ATTN_VARIANT & 0xff, buf, 96), "apla_attn_kernel_name")
    return buf.value.decode()


def attn_fwd(qkv: torch.Tensor, B: int, N: int, H: int, scale: float, *, o: Optional[torch.Tensor] = None,
             lse: Optional[torch.Tensor] = None):
    """qkv: [B*N, 3*H*64] bf16 contiguous.  Returns (o [B*N, H*64] bf16, lse [B,H,N] fp32)."""
    _req(qkv, half(), "qkv", 2)
    if tuple(qkv.shape) != (B * N, 3 * H * 64) or not qkv.is_contiguous():
        raise ValueError(f"attn_fwd: qkv must be contiguous [{B * N}, {3 * H * 64}], got {tuple(qkv.shape)}")
    if o is None:
        o = torch.empty(B * N, H * 64, device=qkv.device, dtype=half())
    if lse is None:
        lse = torch.empty(B, H, N, device=qkv.device, dtype=torch.float32)
    _req(o, half(), "o", 2), _req(lse, torch.float32, "lse", 3)
    if tuple(o.shape) != (B * N, H * 64) or tuple(lse.shape) != (B, H, N) or not (o.is_contiguous() and lse.is_contiguous()):
        raise ValueError("attn_fwd: bad output buffers")
    check(lib().apla_attn_fwd_ex(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, N, H, float(scale), _ATTN_VARIANT, _stream()),
          "apla_attn_fwd")
    return o, lse


def attn_fwd_dropout(qkv: torch.Tensor, B: int, N: int, H: int, scale: float, p: float, seed: int, offset: int = 0, *,
                     o: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None):
    """attn_fwd with dropout on the attention probabilities (appla_attn.py:56-58): (o, lse); the mask is regenerated from
    (seed, offset) by attn_bwd_dropout, nothing is stored (include/apla_hip.h: apla_attn_fwd_dropout)."""
    _req(qkv, half(), "qkv", 2)
    if tuple(qkv.shape) != (B * N, 3 * H * 64) or not qkv.is_contiguous():
        raise ValueError(f"attn_fwd_dropout: qkv must be contiguous [{B * N}, {3 * H * 64}], got {tuple(qkv.shape)}")
    if o is None:
        o = torch.empty(B * N, H * 64, device=qkv.device, dtype=half())
    if lse is None:
        lse = torch.empty(B, H, N, device=qkv.device, dtype=torch.float32)
    _req(o, half(), "o", 2), _req(lse, torch.float32, "lse", 3)
    if tuple(o.shape) != (B * N, H * 64) or tuple(lse.shape) != (B, H, N) or not (o.is_contiguous() and lse.is_contiguous()):
        raise ValueError("attn_fwd_dropout: bad output buffers")
    check(lib().apla_attn_fwd_dropout(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, N, H, float(scale), float(p),
                                      int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset) & 0xFFFFFFFF, _stream()), "apla_attn_fwd_dropout")
    return o, lse


def attn_bwd_dropout(qkv: torch.Tensor, o: torch.Tensor, do: torch.Tensor, lse: torch.Tensor, B: int, N: int, H: int, scale: float,
                     p: float, seed: int, offset: int = 0, *, dqkv: Optional[torch.Tensor] = None, delta: Optional[torch.Tensor] = None):
    _req(qkv, half(), "qkv", 2), _req(o, half(), "o", 2), _req(do, half(), "do", 2), _req(lse, torch.float32, "lse", 3)
    if tuple(qkv.shape) != (B * N, 3 * H * 64) or tuple(o.shape) != (B * N, H * 64) or tuple(do.shape) != tuple(o.shape) or \
            tuple(lse.shape) != (B, H, N) or not (qkv.is_contiguous() and o.is_contiguous() and do.is_contiguous() and lse.is_contiguous()):
        raise ValueError("attn_bwd_dropout: bad operand shapes")
    if delta is None:
        delta = torch.empty_like(lse)
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    _req(dqkv, half(), "dqkv", 2), _req(delta, torch.float32, "delta", 3)
    if dqkv.shape != qkv.shape or delta.shape != lse.shape or not (dqkv.is_contiguous() and delta.is_contiguous()):
        raise ValueError("attn_bwd_dropout: bad output buffers")
    check(lib().apla_attn_bwd_dropout(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), B, N, H,
                                      float(scale), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset) & 0xFFFFFFFF, _stream()),
          "apla_attn_bwd_dropout")
    return dqkv


def attn_bwd(qkv: torch.Tensor, o: torch.Tensor, do: torch.Tensor, lse: torch.Tensor, B: int, N: int, H: int,
             scale: float, *, dqkv: Optional[torch.Tensor] = None, delta: Optional[torch.Tensor] = None):
    _req(qkv, half(), "qkv", 2), _req(o, half(), "o", 2), _req(do, half(), "do", 2)
    _req(lse, torch.float32, "lse", 3)
    if tuple(qkv.shape) != (B * N, 3 * H * 64) or tuple(o.shape) != (B * N, H * 64) or o.shape != do.shape \
            or tuple(lse.shape) != (B, H, N):
        raise ValueError("attn_bwd: shape mismatch")
    if not (qkv.is_contiguous() and o.is_contiguous() and do.is_contiguous() and lse.is_contiguous()):
        raise ValueError("attn_bwd: operands must be contiguous")
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    if delta is None:
        delta = torch.empty(B, H, N, device=qkv.device, dtype=torch.float32)
    if dqkv.shape != qkv.shape or not dqkv.is_contiguous() or delta.numel() < B * H * N:
        raise ValueError("attn_bwd: bad dqkv/delta buffers")
    check(lib().apla_attn_bwd_ex(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                 dqkv.data_ptr(), B, N, H, float(scale), _ATTN_VARIANT, _stream()), "apla_attn_bwd")
    return dqkv


def _check_cu(cu: torch.Tensor, total: int, who: str):
    _req(cu, torch.int32, "cu_seqlens", 1)
    if cu.numel() < 2 or not cu.is_contiguous():
        raise ValueError(f"{who}: cu_seqlens must be a contiguous int32 vector of S+1 offsets")
    return cu.numel() - 1


def attn_varlen_fwd(qkv: torch.Tensor, cu_seqlens: torch.Tensor, max_n: int, H: int, scale: float, *,
                    o: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None):
    """Block-diagonal attention over packed sequences.  qkv: [total, 3*H*64] bf16; cu_seqlens: int32[S+1] on the device
    (cu[0] = 0, cu[S] = total; the caller guarantees it, the kernels trust it); max_n = longest sequence.
    Returns (o [total, H*64] bf16, lse [H, total] fp32)."""
    _req(qkv, half(), "qkv", 2)
    total = qkv.shape[0]
    S = _check_cu(cu_seqlens, total, "attn_varlen_fwd")
    if qkv.shape[1] != 3 * H * 64 or not qkv.is_contiguous() or max_n <= 0 or max_n > total:
        raise ValueError(f"attn_varlen_fwd: qkv must be contiguous [total, {3 * H * 64}] and 0 < max_n <= total")
    if o is None:
        o = torch.empty(total, H * 64, device=qkv.device, dtype=half())
    if lse is None:
        lse = torch.empty(H, total, device=qkv.device, dtype=torch.float32)
    _req(o, half(), "o", 2), _req(lse, torch.float32, "lse", 2)
    if tuple(o.shape) != (total, H * 64) or tuple(lse.shape) != (H, total) or not (o.is_contiguous() and lse.is_contiguous()):
        raise ValueError("attn_varlen_fwd: bad output buffers")
    check(lib().apla_attn_varlen_fwd_ex(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), cu_seqlens.data_ptr(), S, total,
                                        int(max_n), H, float(scale), _ATTN_VARIANT, _stream()), "apla_attn_varlen_fwd")
    return o, lse


def attn_varlen_bwd(qkv: torch.Tensor, o: torch.Tensor, do: torch.Tensor, lse: torch.Tensor, cu_seqlens: torch.Tensor,
                    max_n: int, H: int, scale: float, *, dqkv: Optional[torch.Tensor] = None,
                    delta: Optional[torch.Tensor] = None):
    _req(qkv, half(), "qkv", 2), _req(o, half(), "o", 2), _req(do, half(), "do", 2)
    _req(lse, torch.float32, "lse", 2)
    total = qkv.shape[0]
    S = _check_cu(cu_seqlens, total, "attn_varlen_bwd")
    if qkv.shape[1] != 3 * H * 64 or tuple(o.shape) != (total, H * 64) or o.shape != do.shape \
            or tuple(lse.shape) != (H, total) or max_n <= 0 or max_n > total:
        raise ValueError("attn_varlen_bwd: shape mismatch")
    if not (qkv.is_contiguous() and o.is_contiguous() and do.is_contiguous() and lse.is_contiguous()):
        raise ValueError("attn_varlen_bwd: operands must be contiguous")
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    if delta is None:
        delta = torch.empty(H, total, device=qkv.device, dtype=torch.float32)
    if dqkv.shape != qkv.shape or not dqkv.is_contiguous() or delta.numel() < H * total:
        raise ValueError("attn_varlen_bwd: bad dqkv/delta buffers")
    check(lib().apla_attn_varlen_bwd_ex(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                        dqkv.data_ptr(), cu_seqlens.data_ptr(), S, total, int(max_n), H, float(scale),
                                        _ATTN_VARIANT, _stream()), "apla_attn_varlen_bwd")
    return dqkv


def attn_fwd_cls(qkv: torch.Tensor, B: int, N: int, H: int, scale: float, *, o: torch.Tensor, lse: torch.Tensor):
    """Attention output of the CLS query of every sequence only: fills row b*N of o [B*N, H*64] and lse[:, :, 0]."""
    _req(qkv, half(), "qkv", 2), _req(o, half(), "o", 2), _req(lse, torch.float32, "lse", 3)
    if tuple(qkv.shape) != (B * N, 3 * H * 64) or tuple(o.shape) != (B * N, H * 64) or tuple(lse.shape) != (B, H, N) \
            or not (qkv.is_contiguous() and o.is_contiguous() and lse.is_contiguous()):
        raise ValueError("attn_fwd_cls: shape mismatch / non-contiguous operand")
    check(lib().apla_attn_fwd_cls(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, N, H, float(scale), _stream()),
          "apla_attn_fwd_cls")
    return o, lse


def attn_bwd_cls(qkv: torch.Tensor, o: torch.Tensor, do_cls: torch.Tensor, lse: torch.Tensor, B: int, N: int, H: int,
                 scale: float, *, dqkv: Optional[torch.Tensor] = None):
    """Attention backward for dO that is non-zero only at token 0 of every sequence; do_cls: [B, H*64] bf16."""
    _req(qkv, half(), "qkv", 2), _req(o, half(), "o", 2), _req(do_cls, half(), "do_cls", 2)
    _req(lse, torch.float32, "lse", 3)
    if tuple(qkv.shape) != (B * N, 3 * H * 64) or tuple(o.shape) != (B * N, H * 64) or tuple(do_cls.shape) != (B, H * 64) \
            or tuple(lse.shape) != (B, H, N) or not (qkv.is_contiguous() and o.is_contiguous() and do_cls.is_contiguous() and lse.is_contiguous()):
        raise ValueError("attn_bwd_cls: shape mismatch / non-contiguous operand")
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    if dqkv.shape != qkv.shape or not dqkv.is_contiguous() or dqkv.dtype != half():
        raise ValueError("attn_bwd_cls: bad dqkv buffer")
    check(lib().apla_attn_bwd_cls(qkv.data_ptr(), o.data_ptr(), do_cls.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, N, H,
                                  float(scale), _stream()), "apla_attn_bwd_cls")
    return dqkv


def attn_probs(qkv: torch.Tensor, lse: torch.Tensor, B: int, N: int, H: int, scale: float, p: float = 0.0, seed: int = 0,
               offset: int = 0) -> torch.Tensor:
    """softmax(q k^T scale) [B, H, N, N] fp32 from qkv and the forward's lse; ``p`` > 0: the matrix after the dropout that
    attn_fwd_dropout applied with the same (p, seed, offset) — kept entries / (1 - p), the others 0 (appla_attn.py:56-58, 83)."""
    _req(qkv, half(), "qkv", 2), _req(lse, torch.float32, "lse", 3)
    if tuple(qkv.shape) != (B * N, 3 * H * 64) or not qkv.is_contiguous() or tuple(lse.shape) != (B, H, N) or not lse.is_contiguous():
        raise ValueError("attn_probs: shape mismatch")
    attn = torch.empty(B, H, N, N, device=qkv.device, dtype=torch.float32)
    if p and p > 0.0:
        check(lib().apla_attn_probs_dropout(qkv.data_ptr(), lse.data_ptr(), attn.data_ptr(), B, N, H, float(scale), float(p),
                                            int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset) & 0xFFFFFFFF, _stream()), "apla_attn_probs_dropout")
    else:
        check(lib().apla_attn_probs(qkv.data_ptr(), lse.data_ptr(), attn.data_ptr(), B, N, H, float(scale), _stream()),
              "apla_attn_probs")
    return attn


def dw_workspace(M: int, r: int, D: int, device) -> torch.Tensor:
    if D % 64 == 0:
        D = (D + 127) // 128 * 128        # widths that are multiples of 64 only run padded (proj_dw)
    nbytes = lib().apla_dw_workspace_bytes(M, r, D)
    if nbytes < 0:
        raise ValueError(f"apla_proj_dw needs r%64==0 and D%128==0 (r={r}, D={D})")
    return torch.empty(nbytes // 4, device=device, dtype=torch.float32)


DW_MAX_BATCH = 8


def dw_workspace_batched(M: int, r: int, D: int, nb: int, device) -> torch.Tensor:
    nbytes = lib().apla_dw_workspace_bytes_batched(M, r, D, nb)
    if nbytes < 0:
        raise ValueError(f"apla_proj_dw_batched needs r%64==0, D%128==0 and 1..{DW_MAX_BATCH} layers (r={r}, D={D}, nb={nb})")
    return torch.empty(nbytes // 4, device=device, dtype=torch.float32)


def proj_dw_batched(dyg, x, dW1, db1, *, row_scale=None, workspace: Optional[torch.Tensor] = None, accumulate: bool = False):
    """proj_dw for several projections of equal shape in one launch pair: lists of tensors (row_scale: list with None entries,
    or None).  All layers share M, r, D and the row stride of x."""
    import ctypes
    nb = len(dyg)
    if not (1 <= nb <= DW_MAX_BATCH) or len(x) != nb or len(dW1) != nb or len(db1) != nb or (row_scale is not None and len(row_scale) != nb):
        raise ValueError(f"proj_dw_batched: 1..{DW_MAX_BATCH} layers, lists of equal length")
    M, r = dyg[0].shape
    _, D, ldx = _rows2d(x[0], "x")
    for l in range(nb):
        _req(dyg[l], half(), "dyg", 2), _req(x[l], half(), "x", 2)
        _req(dW1[l], torch.float32, "dW1", 2), _req(db1[l], torch.float32, "db1", 1)
        Mx, Dx, ldl = _rows2d(x[l], "x")
        if tuple(dyg[l].shape) != (M, r) or (Mx, Dx, ldl) != (M, D, ldx) or tuple(dW1[l].shape) != (r, D) or db1[l].numel() != r \
                or not dyg[l].is_contiguous() or not dW1[l].is_contiguous():
            raise ValueError("proj_dw_batched: shape mismatch")
        if row_scale is not None and row_scale[l] is not None:
            _req(row_scale[l], torch.float32, "row_scale", 1)
            if row_scale[l].numel() != r:
                raise ValueError("proj_dw_batched: row_scale length != r")
    if workspace is None:
        workspace = dw_workspace_batched(M, r, D, nb, x[0].device)
    if workspace.numel() * 4 < lib().apla_dw_workspace_bytes_batched(M, r, D, nb):
        raise ValueError("proj_dw_batched: workspace too small")
    arr = lambda ts: (ctypes.c_void_p * nb)(*[None if t is None else t.data_ptr() for t in ts])
    rs = arr(row_scale) if row_scale is not None else None
    check(lib().apla_proj_dw_batched(nb, arr(dyg), arr(x), ldx, rs, arr(dW1), arr(db1), workspace.data_ptr(), M, r, D,
                                     int(accumulate), _stream()), "apla_proj_dw_batched")


def proj_dw(dyg: torch.Tensor, x: torch.Tensor, dW1: torch.Tensor, db1: torch.Tensor, *,
            row_scale: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None,
            accumulate: bool = False):
    """dW1[r,D] (+)= row_scale * dyg[M,r]^T @ x[M,D]; db1[r] (+)= row_scale * colsum(dyg)."""
    _req(dyg, half(), "dyg", 2), _req(x, half(), "x", 2)
    _req(dW1, torch.float32, "dW1", 2), _req(db1, torch.float32, "db1", 1)
    M, r = dyg.shape
    Mx, D, ldx = _rows2d(x, "x")
    if Mx != M or tuple(dW1.shape) != (r, D) or db1.numel() != r or not dyg.is_contiguous() or not dW1.is_contiguous():
        raise ValueError("proj_dw: shape mismatch")
    if D % 128 != 0 and D % 64 == 0:
        # The kernel tiles the feature axis of x in 128s.  A width that is a multiple of 64 only (the reference's vit_tiny, D = 192:
        # utils/transformers/vit.py:511-525) runs on a zero-padded copy of x and a padded fp32 result whose first D columns are the answer
        # (buffers cached per shape: stable addresses under hipGraph capture).  A toy-model path: two extra copies per call.
        Dp = (D + 127) // 128 * 128
        key = (M, r, D, x.device, half())
        if key not in _DW_PAD:
            _DW_PAD[key] = (torch.zeros(M, Dp, device=x.device, dtype=half()), torch.empty(r, Dp, device=x.device, dtype=torch.float32),
                            torch.empty(r, device=x.device, dtype=torch.float32))
        xp, dWp, dbp = _DW_PAD[key]
        xp[:, :D].copy_(x)
        proj_dw(dyg, xp, dWp, dbp, row_scale=row_scale, workspace=workspace)
        if accumulate:
            dW1.add_(dWp[:, :D]), db1.add_(dbp)
        else:
            dW1.copy_(dWp[:, :D]), db1.copy_(dbp)
        return
    if row_scale is not None:
        _req(row_scale, torch.float32, "row_scale", 1)
        if row_scale.numel() != r:
            raise ValueError("proj_dw: row_scale length != r")
    if workspace is None:
        workspace = dw_workspace(M, r, D, x.device)
    if workspace.numel() * 4 < lib().apla_dw_workspace_bytes(M, r, D):
        raise ValueError("proj_dw: workspace too small")
    check(lib().apla_proj_dw(dyg.data_ptr(), x.data_ptr(), ldx, _ptr(row_scale), dW1.data_ptr(), db1.data_ptr(),
                             workspace.data_ptr(), M, r, D, int(accumulate), _stream()), "apla_proj_dw")


def pack_proj_rows(W1: torch.Tensor, b1: Optional[torch.Tensor], inds: torch.Tensor, gamma: Optional[torch.Tensor],
                   Wnat: torch.Tensor, WnatT: torch.Tensor, bnat: Optional[torch.Tensor]):
    _req(W1, torch.float32, "W1", 2), _req(inds, torch.int32, "inds", 1)
    _req(Wnat, half(), "Wnat", 2), _req(WnatT, half(), "WnatT", 2)
    r, D = W1.shape
    if tuple(Wnat.shape) != (D, D) or tuple(WnatT.shape) != (D, D) or inds.numel() < r \
            or not (W1.is_contiguous() and Wnat.is_contiguous() and WnatT.is_contiguous()):
        raise ValueError("pack_proj_rows: shape mismatch")
    if b1 is not None and (bnat is None or bnat.numel() != D or b1.numel() != r):
        raise ValueError("pack_proj_rows: bias buffers")
    if gamma is not None and gamma.numel() != D:
        raise ValueError("pack_proj_rows: gamma length")
    check(lib().apla_pack_proj_rows(W1.data_ptr(), _ptr(b1), inds.data_ptr(), _ptr(gamma), Wnat.data_ptr(),
                                    WnatT.data_ptr(), _ptr(bnat), r, D, _stream()), "apla_pack_proj_rows")


def pack_proj_rows_batched(flat: torch.Tensor, block_stride: int, inds_all: torch.Tensor, gamma_all: Optional[torch.Tensor],
                           Wnat_all: torch.Tensor, WnatT_all: torch.Tensor, bnat_all: torch.Tensor, r: int,
                           Wnat_panels: Optional[torch.Tensor] = None, WnatT_panels: Optional[torch.Tensor] = None):
    """pack_proj_rows for all L blocks in one launch; see include/apla_hip.h:apla_pack_proj_rows_batched(_ex).  The optional
    K-panel images are [L, D/32, D, 32]."""
    _req(flat, torch.float32, "flat", 1), _req(inds_all, torch.int32, "inds_all", 2)
    _req(Wnat_all, half(), "Wnat_all", 3), _req(WnatT_all, half(), "WnatT_all", 3), _req(bnat_all, torch.float32, "bnat_all", 2)
    L, D = inds_all.shape
    if tuple(Wnat_all.shape) != (L, D, D) or tuple(WnatT_all.shape) != (L, D, D) or tuple(bnat_all.shape) != (L, D) \
            or flat.numel() < (L - 1) * block_stride + r * D + r or block_stride < r * D + r \
            or not (inds_all.is_contiguous() and Wnat_all.is_contiguous() and WnatT_all.is_contiguous() and bnat_all.is_contiguous()):
        raise ValueError("pack_proj_rows_batched: shape mismatch")
    if gamma_all is not None:
        _req(gamma_all, torch.float32, "gamma_all", 2)
        if tuple(gamma_all.shape) != (L, D) or not gamma_all.is_contiguous():
            raise ValueError("pack_proj_rows_batched: gamma_all shape")
    for t_ in (Wnat_panels, WnatT_panels):
        if t_ is not None:
            _req(t_, half(), "panels", 4)
            if tuple(t_.shape) != (L, D // 32, D, 32) or not t_.is_contiguous():
                raise ValueError("pack_proj_rows_batched: K-panel images are [L, D/32, D, 32]")
    check(lib().apla_pack_proj_rows_batched_ex(flat.data_ptr(), int(block_stride), inds_all.data_ptr(), _ptr(gamma_all),
                                               Wnat_all.data_ptr(), WnatT_all.data_ptr(), bnat_all.data_ptr(),
                                               _ptr(Wnat_panels), _ptr(WnatT_panels), L, r, D, _stream()),
          "apla_pack_proj_rows_batched")


def adamw_step(params, grads, exp_avg, exp_avg_sq, decay_mask, *, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8,
               step: int, max_norm: float = 0.0, grad_scale: float = 1.0, norm_ws: torch.Tensor):
    for t_, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(t_, torch.float32, nm, 1)
    _req(decay_mask, torch.uint8, "decay_mask", 1), _req(norm_ws, torch.float32, "norm_ws", 1)
    n = params.numel()
    if not (grads.numel() == exp_avg.numel() == exp_avg_sq.numel() == decay_mask.numel() == n) or norm_ws.numel() < 512:
        raise ValueError("adamw_step: buffer sizes")
    check(lib().apla_adamw_step(params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                decay_mask.data_ptr(), n, float(lr), float(weight_decay), float(betas[0]),
                                float(betas[1]), float(eps), int(step), float(max_norm), float(grad_scale),
                                norm_ws.data_ptr(), _stream()), "apla_adamw_step")


def adamw_step_dynamic(params, grads, exp_avg, exp_avg_sq, decay_mask, scaler, parity: int, *, lr, weight_decay,
                       betas=(0.9, 0.999), eps=1e-8, max_norm: float = 0.0, grad_scale: float = 1.0,
                       growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000,
                       norm_ws: torch.Tensor):
    """Fused unscale + clip + AdamW + GradScaler.update on the device; `scaler` is the float32[8] state described at
    include/apla_hip.h:apla_adamw_step_dynamic (use ``new_scaler_state``)."""
    for t_, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(t_, torch.float32, nm, 1)
    _req(decay_mask, torch.uint8, "decay_mask", 1), _req(norm_ws, torch.float32, "norm_ws", 1)
    _req(scaler, torch.float32, "scaler", 1)
    n = params.numel()
    if not (grads.numel() == exp_avg.numel() == exp_avg_sq.numel() == decay_mask.numel() == n) or norm_ws.numel() < 512 \
            or scaler.numel() < 8 or not scaler.is_contiguous() or parity not in (0, 1):
        raise ValueError("adamw_step_dynamic: buffer sizes / parity")
    check(lib().apla_adamw_step_dynamic(params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                        decay_mask.data_ptr(), n, float(lr), float(weight_decay), float(betas[0]),
                                        float(betas[1]), float(eps), float(max_norm), float(grad_scale), scaler.data_ptr(),
                                        int(parity), float(growth_factor), float(backoff_factor), int(growth_interval),
                                        norm_ws.data_ptr(), _stream()), "apla_adamw_step_dynamic")


def new_scaler_state(device, init_scale: float = 65536.0, steps: int = 0) -> torch.Tensor:
    """Device state of the dynamic loss scaler: both slots start at (init_scale, 0, steps); [6] = init_scale, [7] = 0."""
    return torch.tensor([init_scale, 0.0, float(steps), init_scale, 0.0, float(steps), init_scale, 0.0],
                        dtype=torch.float32, device=device)


def patchify(images: torch.Tensor, patch: int, Kp: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(images, torch.float32, "images", 4)
    B, C, S, S2 = images.shape
    if C != 3 or S != S2 or not images.is_contiguous():
        raise ValueError("patchify: expected contiguous [B,3,S,S]")
    Np = (S // patch) ** 2
    cols = out if out is not None else torch.empty(B * Np, Kp, device=images.device, dtype=half())
    if tuple(cols.shape) != (B * Np, Kp) or cols.dtype != half() or not cols.is_contiguous():
        raise ValueError("patchify: bad out buffer")
    check(lib().apla_patchify(images.data_ptr(), cols.data_ptr(), B, S, patch, Kp, _stream()), "apla_patchify")
    return cols


def assemble_tokens(patches: torch.Tensor, cls_token: torch.Tensor, pos: torch.Tensor, B: int, Np: int,
                    res_dtype=torch.float32, out: Optional[torch.Tensor] = None, masked: Optional[torch.Tensor] = None,
                    mask_token: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tokens[b] = [cls + pos[0] | patches[b] + pos[1:]] in the residual dtype; with ``masked`` (bool / uint8 [B, Np]) the masked
    patches are replaced by ``mask_token`` (fp32 [D]) first (iBOT).  ``out`` [B*(Np+1), D] may be a row slice of a larger buffer."""
    _req(patches, half(), "patches", 2), _req(cls_token, torch.float32, "cls"), _req(pos, torch.float32, "pos", 2)
    D = patches.shape[1]
    if patches.shape[0] != B * Np or cls_token.numel() != D or tuple(pos.shape) != (Np + 1, D) or not pos.is_contiguous() \
            or not cls_token.is_contiguous():
        raise ValueError("assemble_tokens: shape mismatch")
    if out is None:
        out = torch.empty(B * (Np + 1), D, device=patches.device, dtype=res_dtype)
    _req(out, None, "out", 2)
    if tuple(out.shape) != (B * (Np + 1), D) or not out.is_contiguous() or out.dtype not in (torch.float32, half()):
        raise ValueError("assemble_tokens: bad out buffer")
    if (masked is None) != (mask_token is None):
        raise ValueError("assemble_tokens: masked and mask_token come together")
    if masked is not None:
        if masked.dtype == torch.bool:
            masked = masked.view(torch.uint8)
        _req(masked, torch.uint8, "masked"), _req(mask_token, torch.float32, "mask_token")
        if masked.numel() != B * Np or not masked.is_contiguous() or mask_token.numel() != D or not mask_token.is_contiguous():
            raise ValueError("assemble_tokens: masked must be [B, Np], mask_token [D]")
    check(lib().apla_assemble_tokens_masked(patches.data_ptr(), patches.stride(0), cls_token.data_ptr(), pos.data_ptr(), _ptr(masked),
                                            _ptr(mask_token), out.data_ptr(), _DT[out.dtype], B, Np, D, _stream()),
          "apla_assemble_tokens")
    return out


def weight_norm_fwd(v: torch.Tensor, g: torch.Tensor, transposed: bool = False):
    """(W 16-bit [K, D], row norms fp32 [K]) of torch.nn.utils.weight_norm(dim=0): W = v * g / ||v|| (include/apla_hip.h); with
    ``transposed`` (W, norms, W^T 16-bit [D, K]) from the same launch where the kernel covers it (K % 64 == 0, D <= 1024; else W^T is None)."""
    _req(v, torch.float32, "v", 2), _req(g, torch.float32, "g")
    K, D = v.shape
    if not v.is_contiguous() or g.numel() != K or not g.is_contiguous() or D % 4:
        raise ValueError("weight_norm_fwd: contiguous v [K, D] with D % 4 == 0 and g with K entries expected")
    w = torch.empty(K, D, device=v.device, dtype=half())
    norm = torch.empty(K, device=v.device, dtype=torch.float32)
    if transposed:
        wt = torch.empty(D, K, device=v.device, dtype=half()) if K % 64 == 0 and D <= 1024 else None
        check(lib().apla_weight_norm_fwd_t(v.data_ptr(), g.data_ptr(), w.data_ptr(), _ptr(wt), norm.data_ptr(), K, D, _stream()),
              "apla_weight_norm_fwd_t")
        return w, norm, wt
    check(lib().apla_weight_norm_fwd(v.data_ptr(), g.data_ptr(), w.data_ptr(), norm.data_ptr(), K, D, _stream()), "apla_weight_norm_fwd")
    return w, norm


def weight_norm_bwd(dw: torch.Tensor, v: torch.Tensor, g: torch.Tensor, norm: torch.Tensor, want_dg: bool = True):
    """(dv [K, D], dg [K] | None) from dW (fp32) — torch's _weight_norm_interface_backward."""
    _req(dw, torch.float32, "dw", 2), _req(v, torch.float32, "v", 2), _req(g, torch.float32, "g"), _req(norm, torch.float32, "norm", 1)
    K, D = v.shape
    if tuple(dw.shape) != (K, D) or not (dw.is_contiguous() and v.is_contiguous() and g.is_contiguous()) or norm.numel() != K or g.numel() != K:
        raise ValueError("weight_norm_bwd: shape mismatch")
    dv = torch.empty_like(v)
    dg = torch.empty(K, device=v.device, dtype=torch.float32) if want_dg else None
    check(lib().apla_weight_norm_bwd(dw.data_ptr(), v.data_ptr(), g.data_ptr(), norm.data_ptr(), dv.data_ptr(), _ptr(dg), K, D, _stream()),
          "apla_weight_norm_bwd")
    return dv, dg


def ema_update(teacher: torch.Tensor, student: torch.Tensor, m: float):
    """teacher = m * teacher + (1 - m) * student in place over two flat fp32 buffers (include/apla_hip.h:apla_ema_update)."""
    _req(teacher, torch.float32, "teacher", 1), _req(student, torch.float32, "student", 1)
    if teacher.numel() != student.numel() or not (teacher.is_contiguous() and student.is_contiguous()):
        raise ValueError("ema_update: two contiguous flat buffers of one size expected")
    check(lib().apla_ema_update(teacher.data_ptr(), student.data_ptr(), teacher.numel(), float(m), _stream()), "apla_ema_update")
    return teacher


_KOLEO_TICKET = {}   # (device index, stream) -> the int32 ticket of apla_koleo_fwd (0 between launches: launches of one stream are ordered)


def koleo_fwd(x: torch.Tensor, groups: int, eps: float = 1e-8):
    """KoLeoLoss of ``groups`` equal groups of rows of x [G*B, D] (fp32 or the build's 16-bit type): (out fp32 [G + 1] — the group losses
    and their sum —, saved = (nn_idx, dist, nrm)) — include/apla_hip.h:apla_koleo_fwd."""
    _req(x, None, "x", 2)
    R, D = x.shape
    if x.dtype not in (torch.float32, half()) or not x.is_contiguous() or groups <= 0 or R % groups or D % 4 or R == 0:
        raise ValueError("koleo_fwd: contiguous x [G*B, D] (fp32 or the build's 16-bit type) with D % 4 == 0 expected")
    dev = x.device
    tkey = (dev.index, _stream())
    ticket = _KOLEO_TICKET.get(tkey)
    if ticket is None:
        ticket = _KOLEO_TICKET[tkey] = torch.zeros(1, device=dev, dtype=torch.int32)
    nn_idx = torch.empty(R, device=dev, dtype=torch.int32)
    f = torch.empty(3, R, device=dev, dtype=torch.float32)    # dist, nrm, terms
    out = torch.empty(groups + 1, device=dev, dtype=torch.float32)
    check(lib().apla_koleo_fwd(x.data_ptr(), _DT[x.dtype], groups, R // groups, D, eps, nn_idx.data_ptr(), f[0].data_ptr(), f[1].data_ptr(),
                               f[2].data_ptr(), out.data_ptr(), ticket.data_ptr(), _stream()), "apla_koleo_fwd")
    return out, (nn_idx, f[0], f[1])


def koleo_bwd(x: torch.Tensor, groups: int, saved, gout: torch.Tensor, eps: float = 1e-8):
    """dx (x's type) = gout * d(sum of the group losses) / dx; ``gout``: one fp32 element on the device."""
    nn_idx, dist, nrm = saved
    _req(gout, torch.float32, "gout")
    R, D = x.shape
    if gout.numel() != 1 or nn_idx.numel() != R:
        raise ValueError("koleo_bwd: one gradient element and the saved tensors of koleo_fwd expected")
    dx = torch.empty_like(x)
    check(lib().apla_koleo_bwd(x.data_ptr(), _DT[x.dtype], groups, R // groups, D, eps, nn_idx.data_ptr(), dist.data_ptr(), nrm.data_ptr(),
                               gout.data_ptr(), dx.data_ptr(), _stream()), "apla_koleo_bwd")
    return dx


def dropout_fwd(x: torch.Tensor, p: float, seed: int, offset: int = 0, *, out: Optional[torch.Tensor] = None, keep: Optional[torch.Tensor] = None):
    """(y, keep) of include/apla_hip.h:apla_dropout_fwd for a contiguous fp32 / 16-bit tensor whose size is a multiple of 8
    (``out`` may be ``x`` itself: element-wise, in place)."""
    _req(x, None, "x")
    if x.dtype not in (torch.float32, half()) or not x.is_contiguous() or x.numel() % 8 or not 0.0 <= p < 1.0:
        raise ValueError("dropout_fwd: contiguous fp32 / 16-bit tensor with numel % 8 == 0 and 0 <= p < 1 expected")
    y = torch.empty_like(x) if out is None else out
    if keep is None:
        keep = torch.empty(x.numel(), device=x.device, dtype=torch.uint8)
    if y.dtype != x.dtype or y.numel() != x.numel() or not y.is_contiguous() or keep.dtype != torch.uint8 or keep.numel() < x.numel() or not keep.is_contiguous():
        raise ValueError("dropout_fwd: bad output buffers")
    check(lib().apla_dropout_fwd(x.data_ptr(), _DT[x.dtype], y.data_ptr(), keep.data_ptr(), x.numel(), float(p), int(seed) & (2 ** 64 - 1),
                                 int(offset) & (2 ** 64 - 1), _stream()), "apla_dropout_fwd")
    return y, keep


def dropout_dev(x: torch.Tensor, p: float, rng: torch.Tensor, stride: int, site: int, *, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = keep ? x / (1 - p) : 0 with the mask of (rng = device int64 [2] {seed, step}, offset = step * stride + site): apla_dropout_fwd_dev
    (no keep bytes; ``out`` may be ``x``).  Forward of a site and — applied to the gradient — its backward."""
    _req(x, None, "x")
    if x.dtype not in (torch.float32, half()) or not x.is_contiguous() or x.numel() % 8 or not 0.0 <= p < 1.0 or \
            rng.dtype != torch.int64 or rng.numel() != 2 or not rng.is_cuda:
        raise ValueError("dropout_dev: contiguous fp32 / 16-bit tensor with numel % 8 == 0, 0 <= p < 1 and a device int64 [2] rng expected")
    y = torch.empty_like(x) if out is None else out
    if y.dtype != x.dtype or y.numel() != x.numel() or not y.is_contiguous():
        raise ValueError("dropout_dev: bad output buffer")
    check(lib().apla_dropout_fwd_dev(x.data_ptr(), _DT[x.dtype], y.data_ptr(), None, x.numel(), float(p), rng.data_ptr(), int(stride), int(site),
                                     _stream()), "apla_dropout_fwd_dev")
    return y


def dropout_bwd(dy: torch.Tensor, keep: torch.Tensor, p: float, *, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(dy, None, "dy"), _req(keep, torch.uint8, "keep", 1)
    if dy.dtype not in (torch.float32, half()) or not dy.is_contiguous() or keep.numel() < dy.numel() or dy.numel() % 8:
        raise ValueError("dropout_bwd: dy must match the forward's tensor")
    dx = torch.empty_like(dy) if out is None else out
    if dx.dtype != dy.dtype or dx.numel() != dy.numel() or not dx.is_contiguous():
        raise ValueError("dropout_bwd: bad output buffer")
    check(lib().apla_dropout_bwd(dy.data_ptr(), _DT[dy.dtype], keep.data_ptr(), dx.data_ptr(), dy.numel(), float(p), _stream()), "apla_dropout_bwd")
    return dx


def scale_samples(x: torch.Tensor, scale: torch.Tensor, *, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y[s] = x[s] * scale[s] over the leading dimension (include/apla_hip.h:apla_scale_samples); ``out`` may be ``x``."""
    _req(x, None, "x"), _req(scale, torch.float32, "scale", 1)
    S = x.shape[0]
    per = x.numel() // max(S, 1)
    if x.dtype not in (torch.float32, half()) or not x.is_contiguous() or scale.numel() != S or per % 8 or not scale.is_contiguous():
        raise ValueError("scale_samples: contiguous [S, ...] tensor with a multiple of 8 elements per sample and S scales expected")
    y = torch.empty_like(x) if out is None else out
    check(lib().apla_scale_samples(x.data_ptr(), _DT[x.dtype], y.data_ptr(), scale.data_ptr(), S, per, _stream()), "apla_scale_samples")
    return y


def sgemm_small(A: torch.Tensor, Bm: torch.Tensor, *, trans_a=False, trans_b=False, bias=None, out=None,
                accumulate=False) -> torch.Tensor:
    """fp32: out[M,N] (+)= op(A) @ op(Bm) (+bias); op = transpose when trans_* is set."""
    _req(A, torch.float32, "A", 2), _req(Bm, torch.float32, "B", 2)
    M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
    Kb, N = (Bm.shape[1], Bm.shape[0]) if trans_b else Bm.shape
    if K != Kb:
        raise ValueError("sgemm_small: inner dimension mismatch")
    sai, sak = (1, A.stride(0)) if trans_a else (A.stride(0), 1)
    sbk, sbj = (1, Bm.stride(0)) if trans_b else (Bm.stride(0), 1)
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    if tuple(out.shape) != (M, N) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise ValueError("sgemm_small: bad out")
    check(lib().apla_sgemm_small(A.data_ptr(), sai, sak, Bm.data_ptr(), sbk, sbj, _ptr(bias), out.data_ptr(),
                                 out.stride(0), M, N, K, int(accumulate), _stream()), "apla_sgemm_small")
    return out


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor, *, dlogits=None, row_loss=None, loss=None):
    """Returns (loss [1], dlogits [B,C], row_loss [B]); mean reduction.  `labels`: int32 [B] class ids, or float32 [B, C]
    probability targets (Mixup / label smoothing), as nn.CrossEntropyLoss accepts both."""
    _req(logits, torch.float32, "logits", 2)
    B, C = logits.shape
    soft = labels.ndim == 2
    if soft:
        _req(labels, torch.float32, "targets", 2)
        if tuple(labels.shape) != (B, C):
            raise ValueError("cross_entropy: probability targets must be [B, C]")
    else:
        _req(labels, torch.int32, "labels", 1)
    if (not soft and labels.numel() != B) or not logits.is_contiguous():
        raise ValueError("cross_entropy: labels length / logits must be contiguous")
    if dlogits is None:
        dlogits = torch.empty(B, C, device=logits.device, dtype=torch.float32)
    if row_loss is None:
        row_loss = torch.empty(B, device=logits.device, dtype=torch.float32)
    if loss is None:
        loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    if tuple(dlogits.shape) != (B, C) or not dlogits.is_contiguous() or row_loss.numel() != B or loss.numel() < 1:
        raise ValueError("cross_entropy: bad output buffers")
    if soft:
        check(lib().apla_cross_entropy_soft(logits.data_ptr(), logits.stride(0), labels.data_ptr(), labels.stride(0),
                                            dlogits.data_ptr(), row_loss.data_ptr(), loss.data_ptr(), B, C, _stream()),
              "apla_cross_entropy_soft")
        return loss, dlogits, row_loss
    check(lib().apla_cross_entropy(logits.data_ptr(), logits.stride(0), labels.data_ptr(), dlogits.data_ptr(),
                                   row_loss.data_ptr(), loss.data_ptr(), B, C, _stream()), "apla_cross_entropy")
    return loss, dlogits, row_loss


def colsum(X: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[j] = sum_i X[i, j] in fp32; X fp32, or the build's 16-bit type with N % 4 == 0 (apla_colsum_h16)."""
    _req(X, None, "X", 2)
    M, N = X.shape
    if out is None:
        out = torch.empty(N, device=X.device, dtype=torch.float32)
    if X.dtype == torch.float32:
        check(lib().apla_colsum(X.data_ptr(), X.stride(0), out.data_ptr(), M, N, _stream()), "apla_colsum")
    elif X.dtype == half():
        check(lib().apla_colsum_h16(X.data_ptr(), X.stride(0), out.data_ptr(), M, N, _stream()), "apla_colsum_h16")
    else:
        raise TypeError(f"colsum: fp32 or {half()} expected, got {X.dtype}")
    return out
