"""Autograd Functions that put the HIP kernels behind ordinary nn.Module.forward calls (the drop-in path).

This is what lets ``APLA_Attention`` swap into a timm / dinov2-style ``Attention`` inside somebody else's model and
train under plain ``loss.backward()``: each Function's forward/backward is a short sequence of C-ABI kernel calls on
the current HIP stream.  Activations run in bf16 with fp32 accumulation; inputs/outputs keep the caller's dtype.
Frozen weights are converted to the kernel layout (bf16, plus a transposed bf16 copy for dX) once and cached against
the parameter's version counter.  No function has a CPU path: CPU tensors raise AplaHipError.

The fused whole-step path (apla_amd/engine.py) does not use autograd at all.
"""
import os
import weakref

from typing import Optional

import torch

from . import ops
from ._lib import AplaHipError

def _h():
    """The 16-bit operand dtype of the loaded library build (bf16, or fp16 inside ``ops.use_half(torch.float16)``)."""
    return ops.half()


def require_no_dropout(drop_module, training: bool):
    """The PACKED (block-diagonal) attention path has no attention-probability dropout (the dense path has: attention_core(attn_drop=);
    0 in every shipped APLA config, and the reference's own dinov2 blocks never pass one: dinov2/layers/attention.py)."""
    p = getattr(drop_module, "p", 0.0)
    if training and p and p > 0.0:
        raise NotImplementedError(f"attention-probability dropout p={p} is not supported for packed (block-diagonal) batches on the HIP "
                                  "path; dense batches, proj_drop / drop_rate / drop_path_rate are")


def active_p(drop_module, training: bool) -> float:
    """The probability an nn.Dropout would apply right now (0 in evaluation mode)."""
    p = float(getattr(drop_module, "p", 0.0) or 0.0)
    return p if (training and p > 0.0) else 0.0


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        _require_cuda(x, "dropout")
        x2 = x.contiguous()
        if x2.dtype not in (torch.float32, _h()):
            x2 = x2.float()
        y, keep = ops.dropout_fwd(x2, p, seed)
        ctx.save_for_backward(keep)
        ctx.p, ctx.dtype = p, x.dtype
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        (keep,) = ctx.saved_tensors
        d = dy.contiguous()
        if d.dtype not in (torch.float32, _h()):
            d = d.float()
        return ops.dropout_bwd(d, keep, ctx.p).to(ctx.dtype), None, None


def dropout(x, drop, training: bool):
    """nn.Dropout on the HIP path (Mlp.drop vit.py:152-168, proj_drop appla_attn.py:82, pos_drop vit.py:395): ``drop`` is the module or
    its p.  The mask comes from a counter-based generator (apla_dropout_fwd) keyed by one 63-bit draw from torch's default CPU
    generator per call, so ``torch.manual_seed`` makes a run repeatable; it is not torch's own CUDA mask stream."""
    p = float(getattr(drop, "p", drop) or 0.0)
    if not training or p == 0.0:
        return x
    if p >= 1.0:
        return torch.zeros_like(x)
    seed = int(torch.empty((), dtype=torch.int64).random_())
    return _DropoutFn.apply(x, p, seed)


class _ScaleSamplesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        _require_cuda(x, "drop_path")
        x2 = x.contiguous()
        if x2.dtype not in (torch.float32, _h()):
            x2 = x2.float()
        ctx.save_for_backward(scale)
        ctx.dtype = x.dtype
        return ops.scale_samples(x2, scale).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        d = dy.contiguous()
        if d.dtype not in (torch.float32, _h()):
            d = d.float()
        return ops.scale_samples(d, scale).to(ctx.dtype), None


def drop_path(x, drop_prob: float, training: bool):
    """Stochastic depth per sample (utils/transformers/vit.py:74-82): x / keep_prob * floor(keep_prob + u_b), one u per sample of the
    leading dimension (drawn with torch on the device: B numbers), applied by apla_scale_samples forward and backward."""
    if not drop_prob or not training:
        return x
    keep = 1.0 - float(drop_prob)
    scale = torch.floor(keep + torch.rand(x.shape[0], device=x.device, dtype=torch.float32)) / keep
    return _ScaleSamplesFn.apply(x, scale)


def _require_cuda(x: torch.Tensor, what: str):
    if not x.is_cuda:
        raise AplaHipError(f"{what}: got a CPU tensor; the APLA path runs on MI355X only (no CPU fallback)")


# ------------------------------------------------------------------------------------------------ weight cache
class _WeightCache:
    """16-bit (and transposed 16-bit) copies of parameters keyed by identity + version + the operand dtype in force."""

    def __init__(self):
        self._store = {}

    def get(self, p: torch.Tensor, kind: str, make):
        key = (id(p), kind, ops.half())
        ent = self._store.get(key)
        ver = p._version
        if ent is not None and ent[0] == ver and ent[1]() is p and ent[2].device == p.device:
            return ent[2]
        val = make()
        self._store[key] = (ver, weakref.ref(p), val)
        return val

    def clear(self):
        self._store.clear()


CACHE = _WeightCache()


def w_bf16(p):
    return CACHE.get(p, "bf16", lambda: p.detach().to(_h()).contiguous())


def w_bf16_t(p):
    return CACHE.get(p, "bf16_t", lambda: p.detach().t().to(_h()).contiguous())


def b_f32(p):
    return None if p is None else CACHE.get(p, "f32", lambda: p.detach().float().contiguous())


_IMAGES = os.environ.get("APLA_W_PANELS", "1") != "0"


def _img(p, kind: str, make, M: int):
    """The K-panel image (ops.k_panels) of a cached 16-bit weight copy for a plain-store GEMM with M rows, where the ping-pong
    kernel covers the problem; else the row-major copy.  Cached like the copy itself (a frozen weight is converted once)."""
    w = CACHE.get(p, kind, make)
    if not (_IMAGES and ops.gemm_panel_ok(M, w.shape[0], w.shape[1])):
        return w
    return CACHE.get(p, kind + "_img", lambda: ops.k_panels(w))


def _as2d_bf16(x):
    return x.reshape(-1, x.shape[-1]).to(_h()).contiguous()


# ------------------------------------------------------------------------------------------------ LayerNorm
class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        _require_cuda(x, "layer_norm")
        if weight.requires_grad or bias.requires_grad:
            raise NotImplementedError("trainable LayerNorm affine is outside the APLA path (all norms are frozen)")
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.dtype in (torch.float32, _h()) else x2.float()
        x2 = x2.contiguous()
        y, mean, rstd = ops.layernorm_fwd(x2, b_f32(weight), b_f32(bias), eps)
        ctx.save_for_backward(x2, mean, rstd, weight)
        ctx.shape, ctx.dtype = x.shape, x.dtype
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd, weight = ctx.saved_tensors
        dx, _ = ops.layernorm_bwd(_as2d_bf16(dy), x2, b_f32(weight), mean, rstd)
        return dx.reshape(ctx.shape).to(ctx.dtype), None, None, None


def layer_norm(x, norm_module):
    """bf16 output (feeds the next GEMM)."""
    return _LayerNormFn.apply(x, norm_module.weight, norm_module.bias, norm_module.eps)


class _AddLayerNormFn(torch.autograd.Function):
    """(x_new, y) = (x + branch, LayerNorm(x + branch)) in ONE pass over the residual stream (the residual update of
    vit.py:284-285 fused into the following norm, as the fused engine does), and in backward ONE pass that adds the residual
    gradient to the LayerNorm input gradient: d(x) = d(branch) = d(x_new) + LN_bwd(dy)."""

    @staticmethod
    def forward(ctx, x, branch, weight, bias, eps):
        _require_cuda(x, "add_layer_norm")
        if weight.requires_grad or bias.requires_grad:
            raise NotImplementedError("trainable LayerNorm affine is outside the APLA path (all norms are frozen)")
        x2 = x.reshape(-1, x.shape[-1])
        x2 = (x2 if x2.dtype in (torch.float32, _h()) else x2.float()).contiguous()
        b2 = _as2d_bf16(branch)
        x_new = torch.empty_like(x2)
        y, mean, rstd = ops.layernorm_fwd(x2, b_f32(weight), b_f32(bias), eps, add=b2, x_out=x_new)
        ctx.save_for_backward(x_new, mean, rstd, weight)
        ctx.shape, ctx.bdtype = x.shape, branch.dtype
        return x_new.reshape(x.shape), y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dx_new, dy):
        x_new, mean, rstd, weight = ctx.saved_tensors
        if dy is None:
            d = dx_new.reshape(x_new.shape)
        else:
            dres = None if dx_new is None else dx_new.reshape(x_new.shape).to(x_new.dtype).contiguous()
            if ctx.bdtype == _h() and x_new.dtype == torch.float32:
                # the branch gradient is the 16-bit copy of the same rows: written by the kernel in the same pass instead of a
                # cast over the whole [tokens, dim] stream afterwards (1.4 ms per self-supervised iteration at config 4)
                db = torch.empty(x_new.shape, device=x_new.device, dtype=_h())
                d, _ = ops.layernorm_bwd(_as2d_bf16(dy), x_new, b_f32(weight), mean, rstd, dres=dres, out_bf16=db)
                return d.reshape(ctx.shape), db.reshape(ctx.shape), None, None, None
            d, _ = ops.layernorm_bwd(_as2d_bf16(dy), x_new, b_f32(weight), mean, rstd, dres=dres)
        d = d.reshape(ctx.shape)
        return d, d.to(ctx.bdtype), None, None, None


def add_layer_norm(x, branch, norm_module):
    """Returns (x + branch [dtype of x], LayerNorm(x + branch) [bf16])."""
    return _AddLayerNormFn.apply(x, branch, norm_module.weight, norm_module.bias, norm_module.eps)


# ------------------------------------------------------------------------------------------------ residual stream with a 16-bit gradient
class ResidualStream:
    """The fp32 residual stream of VisionTransformer.run_blocks together with a 16-bit autograd PROXY for it.

    autograd hands a tensor a gradient of the tensor's own dtype, so an fp32 stream means an fp32 gradient stream: 16 bytes per
    element and LayerNorm backward (read dy, x, the incoming stream; write the stream and its 16-bit copy for the branch).  The fused
    engine keeps the gradient stream in 16 bits.  Here the VALUES stay fp32 and outside the graph; what the graph
    carries from block to block is an uninitialised 16-bit tensor of the stream's shape whose only role is to own the gradient:
    every ``stream_add_layer_norm`` takes the proxy in and gives a new one out, its backward reads and writes the 16-bit gradient
    (10 bytes per element, one output that serves the stream and the branch alike).  Values are never read from a proxy."""

    def __init__(self, values: torch.Tensor):
        self.values = values.detach().float()
        self.proxy: Optional[torch.Tensor] = None

    def select_rows(self, rows: torch.Tensor):
        """Keep the token rows `rows` of a packed stream [1, T, D] (the last block of a backbone whose consumers read a subset)."""
        self.values = self.values.index_select(1, rows)
        if self.proxy is not None:
            self.proxy = self.proxy.index_select(1, rows)      # backward: the 16-bit gradient scattered into zero rows

    def exit(self) -> torch.Tensor:
        """The stream as an ordinary fp32 tensor that still carries gradient into the blocks (x_prenorm of the reference's API)."""
        if self.proxy is None or not self.proxy.requires_grad:
            return self.values
        return _StreamExitFn.apply(self.proxy, self.values)


class _StreamExitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, proxy, values):
        return values.view_as(values)

    @staticmethod
    def backward(ctx, g):
        return g.to(_h()), None


class _StreamAddLayerNormFn(torch.autograd.Function):
    """stream += branch; y = LayerNorm(stream) (vit.py:284-285 fused into the following norm), the gradient of the stream carried by
    its 16-bit proxy: d(stream_in) = d(branch) = d(stream_out) + LN_bwd(dy), ONE 16-bit tensor."""

    @staticmethod
    def forward(ctx, proxy_in, branch, weight, bias, stream, eps):
        _require_cuda(branch, "add_layer_norm")
        if weight.requires_grad or bias.requires_grad:
            raise NotImplementedError("trainable LayerNorm affine is outside the APLA path (all norms are frozen)")
        shape = stream.values.shape
        x2 = stream.values.reshape(-1, shape[-1]).contiguous()
        b2 = _as2d_bf16(branch)
        x_new = torch.empty_like(x2)
        y, mean, rstd = ops.layernorm_fwd(x2, b_f32(weight), b_f32(bias), eps, add=b2, x_out=x_new)
        stream.values = x_new.reshape(shape)
        ctx.save_for_backward(x_new, mean, rstd, weight)
        ctx.shape, ctx.bdtype = shape, branch.dtype
        return torch.empty(shape, device=branch.device, dtype=_h()), y.reshape(shape)

    @staticmethod
    def backward(ctx, d_proxy, dy):
        x_new, mean, rstd, weight = ctx.saved_tensors
        dres = None if d_proxy is None else d_proxy.reshape(x_new.shape).to(_h()).contiguous()
        if dy is None:
            d = dres
        else:
            out = torch.empty(x_new.shape, device=x_new.device, dtype=_h())
            d, _ = ops.layernorm_bwd(_as2d_bf16(dy), x_new, b_f32(weight), mean, rstd, dres=dres, out=out)
        if d is None:
            return None, None, None, None, None, None
        d = d.reshape(ctx.shape)
        return (d if ctx.needs_input_grad[0] else None), (d if ctx.bdtype == d.dtype else d.to(ctx.bdtype)), None, None, None, None


def stream_add_layer_norm(stream: ResidualStream, branch, norm_module):
    """stream += branch, returns LayerNorm(stream) in 16 bits; ``stream`` is updated in place (values and proxy)."""
    proxy, y = _StreamAddLayerNormFn.apply(stream.proxy, branch, norm_module.weight, norm_module.bias, stream, norm_module.eps)
    stream.proxy = proxy
    return y


# ------------------------------------------------------------------------------------------------ Linear
class _LinearFn(torch.autograd.Function):
    """y = x W^T + b through apla_gemm_nt.  dX through the transposed copy; dW/db (full-rank, only for the
    ``partial_size: full`` mode of apla_vit.py:66-75) through the dW kernel with r = out_features."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_cuda(x, "linear")
        x2 = _as2d_bf16(x)
        y = ops.gemm_nt(x2, _img(weight, "bf16", lambda: weight.detach().to(_h()).contiguous(), x2.shape[0]), b_f32(bias))
        ctx.save_for_backward(x2, weight, bias if bias is not None else torch.empty(0))
        ctx.has_bias = bias is not None
        ctx.shape = x.shape
        return y.reshape(x.shape[:-1] + (weight.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias = ctx.saved_tensors
        dy2 = _as2d_bf16(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(dy2, _img(weight, "bf16_t", lambda: weight.detach().t().to(_h()).contiguous(), dy2.shape[0])).reshape(ctx.shape)
        dW = db = None
        if weight.requires_grad:
            dW = torch.empty(weight.shape, device=dy.device, dtype=torch.float32)
            db_ = torch.empty(weight.shape[0], device=dy.device, dtype=torch.float32)
            ops.proj_dw(dy2, x2, dW, db_)
            db = db_.to(bias.dtype) if ctx.has_bias and bias.requires_grad else None
            dW = dW.to(weight.dtype)
        return dx, dW, db


def linear(x, weight, bias=None):
    return _LinearFn.apply(x, weight, bias)


class _LinearGeluFn(torch.autograd.Function):
    """h = GELU(x W^T + b) with the activation in the GEMM epilogue (exact erf form, nn.GELU()): the trainable Linear + GELU pairs of
    the DINO head (dinov2/layers/dino_head.py:24-31).  Training saves gelu'(a) from the same epilogue; backward is da = dh * gelu',
    dX on the transposed weight, dW / db on the TN kernel.  Under no_grad the forward-only epilogue runs (nothing saved)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_cuda(x, "linear_gelu")
        x2 = _as2d_bf16(x)
        w = w_bf16(weight)      # cached against the parameter's version counter (FlatAdamW increments it)
        train = any(ctx.needs_input_grad)   # (grad mode itself is off inside Function.forward)
        if train:
            g = torch.empty(x2.shape[0], weight.shape[0], device=x2.device, dtype=_h())
            h = ops.gemm_nt(x2, w, b_f32(bias), epilogue=ops.EPI_GELU, aux_out=g)
            ctx.save_for_backward(x2, g, weight, bias if bias is not None else torch.empty(0))
        else:
            h = ops.gemm_nt(x2, w, b_f32(bias), epilogue=ops.EPI_GELU_FWD)
        ctx.has_bias, ctx.shape = bias is not None, x.shape
        return h.reshape(x.shape[:-1] + (weight.shape[0],))

    @staticmethod
    def backward(ctx, dh):
        x2, g, weight, bias = ctx.saved_tensors
        da = _as2d_bf16(dh) * g
        dx = ops.gemm_nt(da, w_bf16_t(weight)).reshape(ctx.shape) if ctx.needs_input_grad[0] else None
        dW = db = None
        if weight.requires_grad:
            n_out, n_in = weight.shape
            if n_out % 64 == 0 and n_in % 128 == 0:
                dW = torch.empty(n_out, n_in, device=da.device, dtype=torch.float32)
                db_ = torch.empty(n_out, device=da.device, dtype=torch.float32)
                ops.proj_dw(da, x2, dW, db_)
            else:
                dW, db_ = torch.mm(da.float().t(), x2.float()), da.float().sum(0)
            db = db_.to(bias.dtype) if ctx.has_bias and bias.requires_grad else None
            dW = dW.to(weight.dtype)
        return dx, dW, db


def linear_gelu(x, weight, bias=None):
    """GELU(linear(x)) in one launch (needs out_features % 128 == 0 and in_features % 64 == 0, else falls back to two steps)."""
    if weight.shape[0] % 128 or weight.shape[1] % 64:
        return torch.nn.functional.gelu(linear(x, weight, bias).float()).to(_h())
    return _LinearGeluFn.apply(x, weight, bias)


# ------------------------------------------------------------------------------------------------ attention core
class _AttnCoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, B, N, H, scale):
        qkv2 = qkv.reshape(B * N, -1).contiguous()
        o, lse = ops.attn_fwd(qkv2, B, N, H, scale)
        ctx.save_for_backward(qkv2, o, lse)
        ctx.dims = (B, N, H, scale)
        ctx.mark_non_differentiable(lse)
        return o.reshape(B, N, H * 64), lse

    @staticmethod
    def backward(ctx, do, _dlse):
        qkv2, o, lse = ctx.saved_tensors
        B, N, H, scale = ctx.dims
        dqkv = ops.attn_bwd(qkv2, o, _as2d_bf16(do), lse, B, N, H, scale)
        return dqkv.reshape(B, N, -1), None, None, None, None


class _AttnVarlenFn(torch.autograd.Function):
    """Block-diagonal attention over a packed batch (xformers memory_efficient_attention with a BlockDiagonalMask in the
    reference, appla_attn_mem_eff.py:40-42)."""

    @staticmethod
    def forward(ctx, qkv, cu_seqlens, max_n, H, scale):
        shape = qkv.shape
        qkv2 = qkv.reshape(-1, shape[-1]).contiguous()
        o, lse = ops.attn_varlen_fwd(qkv2, cu_seqlens, max_n, H, scale)
        ctx.save_for_backward(qkv2, o, lse, cu_seqlens)
        ctx.dims = (shape, max_n, H, scale)
        return o.reshape(*shape[:-1], H * 64)

    @staticmethod
    def backward(ctx, do):
        qkv2, o, lse, cu = ctx.saved_tensors
        shape, max_n, H, scale = ctx.dims
        dqkv = ops.attn_varlen_bwd(qkv2, o, _as2d_bf16(do), lse, cu, max_n, H, scale)
        return dqkv.reshape(shape), None, None, None, None


class _AttnRunsFn(torch.autograd.Function):
    """The same block-diagonal attention for a packed batch made of a FEW runs of equal-length sequences (the multi-crop batch:
    global crops then local crops): one uniform launch per run on its slice of the packed rows.  Each run then gets the kernel and
    the workgroup size of its own length (the packed launch sizes every workgroup and its LDS for the longest sequence: at config 4
    512 of the 640 sequences are 50 tokens long next to 257) and the persistent backward where it applies."""

    @staticmethod
    def forward(ctx, qkv, runs, H, scale):
        shape = qkv.shape
        qkv2 = qkv.reshape(-1, shape[-1]).contiguous()
        o = torch.empty(qkv2.shape[0], H * 64, device=qkv.device, dtype=_h())
        lses, a = [], 0
        for cnt, n in runs:
            b = a + cnt * n
            lses.append(ops.attn_fwd(qkv2[a:b], cnt, n, H, scale, o=o[a:b])[1])
            a = b
        ctx.save_for_backward(qkv2, o, *lses)
        ctx.dims = (shape, runs, H, scale)
        return o.reshape(*shape[:-1], H * 64)

    @staticmethod
    def backward(ctx, do):
        qkv2, o, *lses = ctx.saved_tensors
        shape, runs, H, scale = ctx.dims
        do2 = _as2d_bf16(do)
        dqkv = torch.empty_like(qkv2)
        a = 0
        for (cnt, n), lse in zip(runs, lses):
            b = a + cnt * n
            ops.attn_bwd(qkv2[a:b], o[a:b], do2[a:b], lse, cnt, n, H, scale, dqkv=dqkv[a:b])
            a = b
        return dqkv.reshape(shape), None, None, None


MAX_ATTN_RUNS = 4   # more runs than this: one packed launch over the offsets instead


def attention_core_varlen(qkv, cu_seqlens, max_n, H, scale, runs=None):
    """qkv [..., total, 3*H*64] with all leading dims of size 1 (a packed batch); cu_seqlens int32[S+1] on qkv's device.
    ``runs``: the host-side description of the same offsets as (count, length) pairs of consecutive equal-length sequences
    (BlockDiagonalMask.runs()); with a few runs each goes through the uniform-batch kernels on its slice."""
    if qkv.shape[-1] != 3 * H * 64:
        raise NotImplementedError(f"the HIP attention kernel needs head_dim 64 (got {qkv.shape[-1] // (3 * H)})")
    if qkv.numel() != qkv.shape[-2] * qkv.shape[-1]:
        raise ValueError("a packed (block-diagonal) batch must have batch size 1: [1, total, 3*dim]")
    if runs is not None and len(runs) <= MAX_ATTN_RUNS:
        if sum(c * n for c, n in runs) != qkv.shape[-2]:
            raise ValueError("runs do not cover the packed batch")
        return _AttnRunsFn.apply(qkv, tuple(runs), H, scale)
    return _AttnVarlenFn.apply(qkv, cu_seqlens, max_n, H, scale)


class _AttnCoreDropFn(torch.autograd.Function):
    """Attention with dropout on the probabilities (appla_attn.py:56-58).  The mask is a function of (seed, row, key): the backward
    regenerates it, nothing but qkv, o and lse is saved."""

    @staticmethod
    def forward(ctx, qkv, B, N, H, scale, p, seed):
        qkv2 = qkv.reshape(B * N, -1).contiguous()
        o, lse = ops.attn_fwd_dropout(qkv2, B, N, H, scale, p, seed)
        ctx.save_for_backward(qkv2, o, lse)
        ctx.dims = (B, N, H, scale, p, seed)
        ctx.mark_non_differentiable(lse)
        return o.reshape(B, N, H * 64), lse

    @staticmethod
    def backward(ctx, do, _dlse):
        qkv2, o, lse = ctx.saved_tensors
        B, N, H, scale, p, seed = ctx.dims
        dqkv = ops.attn_bwd_dropout(qkv2, o, _as2d_bf16(do), lse, B, N, H, scale, p, seed)
        return dqkv.reshape(B, N, -1), None, None, None, None, None, None


def draw_seed() -> int:
    """A fresh 64-bit seed from torch's CPU generator (``torch.manual_seed`` makes a run repeatable), as ``dropout`` draws its own."""
    return int(torch.empty((), dtype=torch.int64).random_())


def attention_core(qkv, B, N, H, scale, attn_drop: float = 0.0, seed: Optional[int] = None):
    """(o, lse).  ``attn_drop`` > 0: dropout on the attention probabilities under ``seed`` (default: a fresh ``draw_seed()``)."""
    if qkv.shape[-1] != 3 * H * 64:
        raise NotImplementedError(f"the HIP attention kernel needs head_dim 64 (got {qkv.shape[-1] // (3 * H)})")
    if attn_drop and attn_drop > 0.0:
        if attn_drop >= 1.0:
            raise ValueError("attn_drop must be < 1")
        return _AttnCoreDropFn.apply(qkv, B, N, H, scale, float(attn_drop), draw_seed() if seed is None else int(seed))
    return _AttnCoreFn.apply(qkv, B, N, H, scale)


def attention_module_forward(x, qkv_w, qkv_b, proj_w, proj_b, num_heads, scale, want_attn, attn_drop: float = 0.0):
    """Plain (non-APLA) Attention.forward, vit.py:184-196: returns (x, attn|None).  The attention matrix returned on demand is
    rebuilt from qkv and the kernels' log-sum-exp — after its dropout when ``attn_drop`` is active, as the reference returns it."""
    B, N, _ = x.shape
    qkv = linear(x, qkv_w, qkv_b)
    seed = draw_seed() if attn_drop and attn_drop > 0.0 else 0
    o, lse = attention_core(qkv, B, N, num_heads, scale, attn_drop, seed)
    y = linear(o, proj_w, proj_b).to(x.dtype)
    attn = ops.attn_probs(qkv.detach().reshape(B * N, -1), lse, B, N, num_heads, scale, attn_drop or 0.0, seed) if want_attn else None
    return y, attn


# ------------------------------------------------------------------------------------------------ APLA projection
class AplaProjState:
    """Kernel-layout copies of one APLA projection: natural-order bf16 weight, its transpose, fp32 bias, int32 indices.
    Frozen rows are written once; the r trainable rows are re-scattered from the fp32 masters before every forward."""

    def __init__(self):
        self.key = None
        self.Wnat = self.WnatT = self.bnat = self.inds32 = None

    def refresh(self, W1, b1, W2, b2, inds, gamma=None):
        """``gamma``: a frozen LayerScale vector folded into the merged weight and bias (y = gamma * (o W^T + b))."""
        D = W1.shape[1]
        r = W1.shape[0]
        key = (W2.data_ptr(), W2._version, b2._version, inds._version, str(W1.device), r, D,
               None if gamma is None else (gamma.data_ptr(), gamma._version), _h())   # (_h(): the 16-bit images are of ONE operand type)
        if key != self.key:
            idx = inds.to(W1.device).long()
            Wn = torch.zeros(D, D, device=W1.device, dtype=torch.float32)
            Wn[idx[r:]] = W2.detach().float()
            bn = torch.zeros(D, device=W1.device, dtype=torch.float32)
            bn[idx[r:]] = b2.detach().float()
            self.gamma = None
            if gamma is not None:
                self.gamma = gamma.detach().float().contiguous()
                Wn *= self.gamma[:, None]
                bn *= self.gamma
                self.gamma_r = self.gamma[idx[:(r + 63) // 64 * 64]].contiguous()   # row scales of the dW step (padded rank)
            self.Wnat = Wn.to(_h())
            self.WnatT = Wn.t().contiguous().to(_h())
            self.bnat = bn
            self.inds32 = idx.int().contiguous()
            self.key = key
        ops.pack_proj_rows(W1.detach().float().contiguous(), b1.detach().float().contiguous(), self.inds32, self.gamma,
                           self.Wnat, self.WnatT, self.bnat)


class _AplaProjFn(torch.autograd.Function):
    """APLA output projection (appla_attn.py:62-79).  Forward: one GEMM over the natural-order merged weight (the two
    scatter_ calls are folded into the weight layout).  Backward: dX over the transposed merged weight; dW1/db1 from
    the r gathered columns of dY only."""

    @staticmethod
    def forward(ctx, o, W1, b1, state: AplaProjState):
        o2 = _as2d_bf16(o)
        y = ops.gemm_nt(o2, state.Wnat, state.bnat)
        ctx.save_for_backward(o2, W1, b1)
        ctx.state, ctx.shape = state, o.shape
        return y.reshape(o.shape)

    @staticmethod
    def backward(ctx, dy):
        o2, W1, b1 = ctx.saved_tensors
        st = ctx.state
        dy2 = _as2d_bf16(dy)
        do = ops.gemm_nt(dy2, st.WnatT).reshape(ctx.shape) if ctx.needs_input_grad[0] else None
        r, D = W1.shape
        r_pad = (r + 63) // 64 * 64  # the dW kernel works on multiples of 64 rows: pad with the next (frozen) indices, drop their rows
        dyg = ops.gather_cols(dy2, st.inds32, r_pad)
        dW1 = torch.empty(r_pad, D, device=dy.device, dtype=torch.float32)
        db1 = torch.empty(r_pad, device=dy.device, dtype=torch.float32)
        ops.proj_dw(dyg, o2, dW1, db1, row_scale=st.gamma_r if st.gamma is not None else None)
        return do, dW1[:r].to(W1.dtype), db1[:r].to(b1.dtype), None


def apla_projection(o, W1, b1, W2, b2, inds, state: AplaProjState, gamma=None):
    """``gamma`` (optional, frozen): the block's LayerScale vector, folded into the projection (ls1(proj(o)) in one GEMM)."""
    if W1.shape[1] % 64 != 0:
        raise NotImplementedError(f"APLA HIP projection needs dim % 64 == 0 (got dim={W1.shape[1]})")
    if gamma is not None and gamma.requires_grad:
        raise NotImplementedError("a trainable LayerScale cannot be folded into the projection")
    with torch.no_grad():
        state.refresh(W1, b1, W2, b2, inds, gamma)
    return _AplaProjFn.apply(o, W1, b1, state)


# ------------------------------------------------------------------------------------------------ MLPs
def _scaled_w(w, gamma):      # gamma[:, None] * w  as bf16 (frozen LayerScale folded into the producing Linear)
    return CACHE.get(w, f"bf16_g{id(gamma)}_{gamma._version}", lambda: (w.detach().float() * gamma.detach().float()[:, None]).to(_h()).contiguous())


def _scaled_w_t(w, gamma):
    return CACHE.get(w, f"bf16_gt{id(gamma)}_{gamma._version}",
                     lambda: (w.detach().float() * gamma.detach().float()[:, None]).t().to(_h()).contiguous())


def _scaled_b(b, gamma):
    return None if b is None else CACHE.get(b, f"f32_g{id(gamma)}_{gamma._version}", lambda: (b.detach().float() * gamma.detach().float()).contiguous())


def _hidden_buffer(M, F, n_next, epilogue, device):
    """Output buffer of the MLP's first GEMM of a pass ([M, F], epilogue GELU / GELU_FWD / MUL) whose only reader is a plain-store
    GEMM with n_next outputs: a K-panel image [F/32, M, 32] where the producing epilogue can write one and the consumer reads one."""
    if _IMAGES and ops.gemm_out_image_ok(M, F, n_next, epilogue) and ops.gemm_panel_ok(M, n_next, F):
        return torch.empty(F // 32, M, 32, device=device, dtype=_h())
    return torch.empty(M, F, device=device, dtype=_h())


def _fc2_weight(w2, gamma, M):
    if gamma is None:
        return _img(w2, "bf16", lambda: w2.detach().to(_h()).contiguous(), M)
    return _img(w2, f"bf16_g{id(gamma)}_{gamma._version}",
                lambda: (w2.detach().float() * gamma.detach().float()[:, None]).to(_h()).contiguous(), M)


class _MlpGeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma):
        _require_cuda(x, "mlp")
        x2 = _as2d_bf16(x)
        M, F = x2.shape[0], w1.shape[0]
        h = _hidden_buffer(M, F, w2.shape[0], ops.EPI_GELU, x.device)   # a K-panel image where fc1's epilogue can write one
        # GELU' is private to this epilogue and the backward's MUL epilogue: an image whenever h is one (as the fused engine keeps it) —
        # whole-line stores / loads, and what lets the dispatch take the tile-alternating kernel for the packed student batch (round 6)
        gp = torch.empty(F // 32, M, 32, device=x.device, dtype=_h()) if h.ndim == 3 else torch.empty(M, F, device=x.device, dtype=_h())
        w1h = w_bf16(w1)
        if _IMAGES and ops.gemm_panel_ok(M, F, w1.shape[1], ops.EPI_GELU):   # the two-output GELU above 40 000 rows: ping-pong kernel
            w1h = CACHE.get(w1, "bf16_img", lambda: ops.k_panels(w1h))
        ops.gemm_nt(x2, w1h, b_f32(b1), epilogue=ops.EPI_GELU, aux_out=gp, out=h)
        y = ops.gemm_nt(h, _fc2_weight(w2, gamma, M), b_f32(b2) if gamma is None else _scaled_b(b2, gamma))
        ctx.save_for_backward(gp, w1, w2, gamma if gamma is not None else torch.empty(0))
        ctx.shape, ctx.has_gamma = x.shape, gamma is not None
        return y.reshape(x.shape[:-1] + (w2.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        gp, w1, w2, gamma = ctx.saved_tensors
        if w1.requires_grad or w2.requires_grad or (ctx.has_gamma and gamma.requires_grad):
            raise NotImplementedError("trainable MLP weights / LayerScale are outside the APLA path")
        w2t = _scaled_w_t(w2, gamma) if ctx.has_gamma else w_bf16_t(w2)
        M, F = (gp.shape[1], gp.shape[0] * 32) if gp.ndim == 3 else gp.shape
        da = _hidden_buffer(M, F, w1.shape[1], ops.EPI_MUL, dy.device)
        ops.gemm_nt(_as2d_bf16(dy), w2t, epilogue=ops.EPI_MUL, aux_in=gp, out=da)
        dx = ops.gemm_nt(da, _img(w1, "bf16_t", lambda: w1.detach().t().to(_h()).contiguous(), M))
        return dx.reshape(ctx.shape), None, None, None, None, None


def mlp_gelu(x, w1, b1, w2, b2, gamma=None):
    """fc2(GELU(fc1(x))); with ``gamma`` (a frozen LayerScale vector) the scale is folded into fc2: ls2(mlp(x)).  Without
    autograd (evaluation, the EMA teacher) fc1 runs the forward-only GELU epilogue: GELU' is neither computed nor stored."""
    if not torch.is_grad_enabled():
        _require_cuda(x, "mlp")
        x2 = _as2d_bf16(x)
        M, F = x2.shape[0], w1.shape[0]
        h = _hidden_buffer(M, F, w2.shape[0], ops.EPI_GELU_FWD, x.device)
        ops.gemm_nt(x2, w_bf16(w1), b_f32(b1), epilogue=ops.EPI_GELU_FWD, out=h)
        y = ops.gemm_nt(h, _fc2_weight(w2, gamma, M), b_f32(b2) if gamma is None else _scaled_b(b2, gamma))
        return y.reshape(x.shape[:-1] + (w2.shape[0],)).to(x.dtype)
    return _MlpGeluFn.apply(x, w1, b1, w2, b2, gamma).to(x.dtype)


def _interleave_rows(w):
    h = w.shape[0] // 2
    return torch.stack([w[:h], w[h:]], 1).reshape(w.shape).contiguous()


class _MlpSwigluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w12, b12, w3, b3):
        _require_cuda(x, "mlp")
        x2 = _as2d_bf16(x)
        w12i = CACHE.get(w12, "il", lambda: _interleave_rows(w12.detach()).to(_h()))
        b12i = None if b12 is None else CACHE.get(b12, "il", lambda: _interleave_rows(b12.detach().float()))
        saved = torch.empty(x2.shape[0], w12.shape[0], device=x.device, dtype=_h())
        h = ops.gemm_nt(x2, w12i, b12i, epilogue=ops.EPI_SWIGLU, aux_out=saved)
        y = ops.gemm_nt(h, w_bf16(w3), b_f32(b3))
        ctx.save_for_backward(saved, w12, w3)
        ctx.shape = x.shape
        return y.reshape(x.shape[:-1] + (w3.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        saved, w12, w3 = ctx.saved_tensors
        if w12.requires_grad or w3.requires_grad:
            raise NotImplementedError("trainable MLP weights are outside the APLA path")
        dx12 = ops.gemm_nt(_as2d_bf16(dy), w_bf16_t(w3), epilogue=ops.EPI_SWIGLU_BWD, aux_in=saved)
        w12it = CACHE.get(w12, "il_t", lambda: _interleave_rows(w12.detach()).t().contiguous().to(_h()))
        dx = ops.gemm_nt(dx12, w12it)
        return dx.reshape(ctx.shape), None, None, None, None


def mlp_swiglu(x, w12, b12, w3, b3):
    return _MlpSwigluFn.apply(x, w12, b12, w3, b3).to(x.dtype)


# ------------------------------------------------------------------------------------------------ patch embedding
def patch_embed(images, conv_w, conv_b, patch: int):
    """Forward-only (frozen, no gradient ever reaches it under APLA).  [B,3,S,S] -> [B,Np,D] bf16."""
    _require_cuda(images, "patch_embed")
    if conv_w.requires_grad:
        raise NotImplementedError("trainable patch embedding is outside the APLA path")
    B, C, S, _ = images.shape
    D = conv_w.shape[0]
    K = C * patch * patch
    Kp = (K + 63) // 64 * 64

    def make():
        w = torch.zeros(D, Kp, device=conv_w.device, dtype=torch.float32)
        w[:, :K] = conv_w.detach().reshape(D, K).float()
        return w.to(_h())

    wp = CACHE.get(conv_w, f"pe{Kp}", make)
    cols = ops.patchify(images.float().contiguous(), patch, Kp)
    return ops.gemm_nt(cols, wp, b_f32(conv_b)).reshape(B, -1, D)
