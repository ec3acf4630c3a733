"""CPU oracle for the APLA training-step hot path.  TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``apla_amd``) never does and fails loudly when its HIP library is
missing rather than falling back to anything in here.

What it is: a from-the-math restatement, in plain PyTorch-CPU tensor ops with *explicit*
(hand-derived) backward formulas, of the reference's APLA ViT training step.  Every function
cites the reference file:line it restates (paths relative to /root/reference/src).  It is
dtype-generic (float32 or float64 follow the inputs) so the HIP kernels can be compared with an
fp64 evaluation of the same math.

Parity pin: the oracle is checked against golden vectors generated from the *actual reference
code* imported in the build container (tests/golden/make_golden.py -> tests/golden/*.npz);
see tests/test_oracle_golden.py.  Parity is therefore PINNED for the supervised path
(rows a1-a3, a5-a10 of SURVEY.md §8).  The xformers-based mem-eff variant (row a4) computes the
same dense softmax attention and is pinned only through that equivalence.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor

# ----------------------------------------------------------------------------------------------
# a1 / a5: index selection and weight split
# ----------------------------------------------------------------------------------------------


def sample_indices(dim: int) -> Tensor:
    """apla/appla_attn.py:26 — ``torch.randperm(dim)`` on the *global CPU generator*.

    Bit-exactness of the selection therefore depends on the caller reproducing the reference's
    RNG consumption order (ViT init draws first, then one randperm per block, block 0..L-1)."""
    return torch.randperm(dim)


def indices_from_trainable(trainable: Sequence[int], dim: int) -> Tensor:
    """apla/apla_vit.py:21-24 — pre-defined indices: trainable list followed by the *ascending*
    complement."""
    tset = set(int(t) for t in trainable)
    frozen = [i for i in range(dim) if i not in tset]
    return torch.tensor(list(int(t) for t in trainable) + frozen, dtype=torch.int64)


def split_proj(weight: Tensor, bias: Optional[Tensor], indices: Tensor, r: int):
    """apla/apla_vit.py:48-56 — W1 = W[idx[:r]], W2 = W[idx[r:]] (rows = output features)."""
    t, f = indices[:r], indices[r:]
    W1, W2 = weight[t, :].clone(), weight[f, :].clone()
    if bias is None:
        return W1, W2, None, None
    return W1, W2, bias[t].clone(), bias[f].clone()


def merge_proj(W1: Tensor, W2: Tensor, b1: Tensor, b2: Tensor, indices: Tensor):
    """Inverse of split_proj: natural-order [D,D] weight and [D] bias (the layout the HIP path
    keeps so that the scatter of appla_attn.py:70-79 becomes a no-op on activations)."""
    r = W1.shape[0]
    D = W1.shape[1]
    W = torch.empty(D, D, dtype=W1.dtype)
    b = torch.empty(D, dtype=W1.dtype)
    W[indices[:r]] = W1
    W[indices[r:]] = W2
    b[indices[:r]] = b1
    b[indices[r:]] = b2
    return W, b


# ----------------------------------------------------------------------------------------------
# elementary ops with explicit backward
# ----------------------------------------------------------------------------------------------


def layernorm_fwd(x: Tensor, g: Tensor, b: Tensor, eps: float = 1e-6):
    """nn.LayerNorm(eps=1e-6) — utils/transformers/vit.py:251,261,554.  Biased variance."""
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    xhat = (x - mean) * rstd
    return xhat * g + b, mean.squeeze(-1), rstd.squeeze(-1)


def layernorm_bwd_dx(dy: Tensor, x: Tensor, g: Tensor, mean: Tensor, rstd: Tensor) -> Tensor:
    """dX of LayerNorm (gamma/beta are frozen under APLA, apla_vit.py:80-81)."""
    xhat = (x - mean.unsqueeze(-1)) * rstd.unsqueeze(-1)
    wdy = dy * g
    c1 = wdy.mean(-1, keepdim=True)
    c2 = (wdy * xhat).mean(-1, keepdim=True)
    return (wdy - c1 - xhat * c2) * rstd.unsqueeze(-1)


_SQRT1_2 = 0.7071067811865476
_INV_SQRT_2PI = 0.3989422804014327


def gelu_fwd(a: Tensor) -> Tensor:
    """nn.GELU() exact erf form — utils/transformers/vit.py:153,157."""
    return 0.5 * a * (1.0 + torch.erf(a * _SQRT1_2))


def gelu_grad(a: Tensor) -> Tensor:
    return 0.5 * (1.0 + torch.erf(a * _SQRT1_2)) + a * torch.exp(-0.5 * a * a) * _INV_SQRT_2PI


def silu(a: Tensor) -> Tensor:
    return a * torch.sigmoid(a)


def silu_grad(a: Tensor) -> Tensor:
    s = torch.sigmoid(a)
    return s * (1.0 + a * (1.0 - s))


def linear_fwd(x: Tensor, W: Tensor, b: Optional[Tensor]) -> Tensor:
    y = x @ W.t()
    return y if b is None else y + b


# ----------------------------------------------------------------------------------------------
# a2: multi-head attention (appla_attn.py:52-60)
# ----------------------------------------------------------------------------------------------


def attention_fwd(qkv: Tensor, num_heads: int, scale: float, return_attn: bool = False):
    """qkv: [B,N,3*D] as produced by the qkv Linear.  Returns o [B,N,D], lse [B,H,N] (natural log of
    the softmax denominator *including* the max shift) and optionally attn [B,H,N,N].

    appla_attn.py:53-60: reshape [B,N,3,H,d] -> permute; softmax((q k^T) * scale) @ v; heads merged
    by transpose(1,2).reshape(B,N,C)."""
    B, N, D3 = qkv.shape
    D = D3 // 3
    d = D // num_heads
    t = qkv.reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]  # [B,H,N,d]
    s = (q @ k.transpose(-2, -1)) * scale
    m = s.max(-1, keepdim=True).values
    p = torch.exp(s - m)
    l = p.sum(-1, keepdim=True)
    attn = p / l
    o = (attn @ v).transpose(1, 2).reshape(B, N, D)
    lse = (m + torch.log(l)).squeeze(-1)
    return (o, lse, attn) if return_attn else (o, lse)


def attention_dropout_fwd_bwd(qkv: Tensor, num_heads: int, scale: float, keep: Tensor, p_drop: float, do: Optional[Tensor] = None):
    """appla_attn.py:56-60 WITH its dropout on the attention probabilities: attn = softmax(q k^T * scale); attn = attn_drop(attn), i.e.
    nn.Dropout: attn_d = keep ? attn / (1 - p) : 0; x = attn_d @ v.  ``keep`` [B,H,N,N] bool is the mask (philox_attn_keep_mask for the
    kernels' own).  Returns (o, lse) and, with ``do`` given, dqkv by autograd-free formulas: d attn = (dO V^T) o keep / (1 - p),
    dS = attn o (d attn - rowsum(attn o d attn)), dV = attn_d^T dO."""
    B, N, D3 = qkv.shape
    D = D3 // 3
    d = D // num_heads
    t = qkv.reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    s = (q @ k.transpose(-2, -1)) * scale
    m = s.max(-1, keepdim=True).values
    e = torch.exp(s - m)
    l = e.sum(-1, keepdim=True)
    attn = e / l
    attn_d = attn * keep.to(attn.dtype) / (1.0 - p_drop)
    o = (attn_d @ v).transpose(1, 2).reshape(B, N, D)
    lse = (m + torch.log(l)).squeeze(-1)
    if do is None:
        return o, lse
    doh = do.reshape(B, N, num_heads, d).permute(0, 2, 1, 3)
    dv = attn_d.transpose(-2, -1) @ doh
    dattn = (doh @ v.transpose(-2, -1)) * keep.to(attn.dtype) / (1.0 - p_drop)
    ds = attn * (dattn - (attn * dattn).sum(-1, keepdim=True)) * scale
    dq = ds @ k
    dk = ds.transpose(-2, -1) @ q
    return o, lse, torch.stack([dq, dk, dv], 0).permute(1, 3, 0, 2, 4).reshape(B, N, D3)


def attention_varlen_fwd(qkv: Tensor, seqlens, num_heads: int, scale: float):
    """Block-diagonal attention over a packed batch: qkv [total, 3*D]; sequence s owns tokens
    [sum(seqlens[:s]), sum(seqlens[:s+1])) and attends to itself only — what
    xformers.memory_efficient_attention(q, k, v, attn_bias=BlockDiagonalMask.from_seqlens(seqlens)) computes at
    appla_attn_mem_eff.py:40-42 (xformers 0.0.18 is not installed: pinned through this per-sequence equivalence with
    the dense softmax attention that the goldens do pin).  Returns o [total, D], lse [H, total]."""
    outs, lses, a = [], [], 0
    for n in seqlens:
        o, l = attention_fwd(qkv[a:a + n][None], num_heads, scale)
        outs.append(o[0])
        lses.append(l[0])  # [H, n]
        a += n
    return torch.cat(outs, 0), torch.cat(lses, 1)


def attention_varlen_bwd(do: Tensor, qkv: Tensor, o: Tensor, lse: Tensor, seqlens, num_heads: int, scale: float) -> Tensor:
    """Backward of attention_varlen_fwd, sequence by sequence.  lse is [H, total]."""
    outs, a = [], 0
    for n in seqlens:
        outs.append(attention_bwd(do[a:a + n][None], qkv[a:a + n][None], o[a:a + n][None], lse[:, a:a + n][None],
                                  num_heads, scale)[0])
        a += n
    return torch.cat(outs, 0)


def attention_bwd(do: Tensor, qkv: Tensor, o: Tensor, lse: Tensor, num_heads: int, scale: float) -> Tensor:
    """Flash-style backward from saved (qkv, o, lse): recompute P = exp(S*scale - lse);
    delta = rowsum(dO*O); dS = P*(dP - delta); dQ = dS K scale; dK = dS^T Q scale; dV = P^T dO."""
    B, N, D3 = qkv.shape
    D = D3 // 3
    d = D // num_heads
    t = qkv.reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    doh = do.reshape(B, N, num_heads, d).permute(0, 2, 1, 3)
    oh = o.reshape(B, N, num_heads, d).permute(0, 2, 1, 3)
    p = torch.exp((q @ k.transpose(-2, -1)) * scale - lse.unsqueeze(-1))
    dv = p.transpose(-2, -1) @ doh
    dp = doh @ v.transpose(-2, -1)
    delta = (doh * oh).sum(-1, keepdim=True)
    ds = p * (dp - delta) * scale
    dq = ds @ k
    dk = ds.transpose(-2, -1) @ q
    dqkv = torch.stack([dq, dk, dv], 0).permute(1, 3, 0, 2, 4).reshape(B, N, D3)
    return dqkv


# ----------------------------------------------------------------------------------------------
# a3: APLA projection (appla_attn.py:62-83) with explicit backward
# ----------------------------------------------------------------------------------------------


def apla_proj_fwd(x: Tensor, W1: Tensor, b1: Tensor, W2: Tensor, b2: Tensor, indices: Tensor) -> Tensor:
    """Two linears + two scatters along the feature dim (appla_attn.py:64-79):
    out[..., idx[:r]] = x W1^T + b1 ; out[..., idx[r:]] = x W2^T + b2."""
    r = W1.shape[0]
    out = torch.empty(x.shape[:-1] + (W1.shape[0] + W2.shape[0],), dtype=x.dtype)
    out[..., indices[:r]] = linear_fwd(x, W1, b1)
    out[..., indices[r:]] = linear_fwd(x, W2, b2)
    return out


def apla_proj_bwd(dy: Tensor, x: Tensor, W1: Tensor, W2: Tensor, indices: Tensor):
    """Autograd of the above: gathers are the adjoints of the scatters.  Only W1/b1 get a gradient
    (W2/b2 are requires_grad=False, appla_attn.py:43,45)."""
    r = W1.shape[0]
    dy1 = dy[..., indices[:r]]
    dy2 = dy[..., indices[r:]]
    dx = dy1 @ W1 + dy2 @ W2
    dW1 = dy1.reshape(-1, r).t() @ x.reshape(-1, x.shape[-1])
    db1 = dy1.reshape(-1, r).sum(0)
    return dx, dW1, db1


def apla_attention_fwd(x: Tensor, p: Dict[str, Tensor], prefix: str, num_heads: int, r: int,
                       return_attn: bool = False, attn_drop=None):
    """APLA_Attention.forward (appla_attn.py:50-83) on a parameter dict keyed like the reference
    state_dict (``<prefix>qkv.weight`` … ``<prefix>inds``).  ``attn_drop`` = (keep [B,H,N,N] bool, p): attn_drop of appla_attn.py:58."""
    D = x.shape[-1]
    scale = (D // num_heads) ** -0.5  # appla_attn.py:15
    qkv = linear_fwd(x, p[prefix + "qkv.weight"], p.get(prefix + "qkv.bias"))
    if attn_drop is not None:
        res = attention_dropout_fwd_bwd(qkv, num_heads, scale, attn_drop[0], attn_drop[1])
    else:
        res = attention_fwd(qkv, num_heads, scale, return_attn)
    o = res[0]
    y = apla_proj_fwd(o, p[prefix + "proj_weight1"], p[prefix + "proj_bias1"],
                      p[prefix + "proj_weight2"], p[prefix + "proj_bias2"], p[prefix + "inds"])
    ctx = (x, qkv, o, res[1], attn_drop)
    return (y, res[2], ctx) if (return_attn and attn_drop is None) else (y, None, ctx)


def apla_attention_bwd(dy: Tensor, ctx, p: Dict[str, Tensor], prefix: str, num_heads: int, need_dx: bool = True):
    x, qkv, o, lse, attn_drop = ctx
    D = x.shape[-1]
    scale = (D // num_heads) ** -0.5
    do, dW1, db1 = apla_proj_bwd(dy, o, p[prefix + "proj_weight1"], p[prefix + "proj_weight2"], p[prefix + "inds"])
    if not need_dx:
        return None, dW1, db1
    if attn_drop is not None:
        dqkv = attention_dropout_fwd_bwd(qkv, num_heads, scale, attn_drop[0], attn_drop[1], do)[2]
    else:
        dqkv = attention_bwd(do, qkv, o, lse, num_heads, scale)
    dx = dqkv @ p[prefix + "qkv.weight"]
    return dx, dW1, db1


# ----------------------------------------------------------------------------------------------
# a6: Block (vit.py:279-288), LayerScale (vit.py:232-244), Mlp (vit.py:152-168), SwiGLU (vit.py:108-149)
# ----------------------------------------------------------------------------------------------


def mlp_fwd(x: Tensor, p: Dict[str, Tensor], prefix: str, swiglu: bool, m_h=None):
    """``m_h`` [.., F]: the multiplicative mask keep / (1 - p) of Mlp.drop after the activation (vit.py:164-165); SwiGLUFFNFused has no dropout."""
    if swiglu:
        x12 = linear_fwd(x, p[prefix + "w12.weight"], p[prefix + "w12.bias"])
        x1, x2 = x12.chunk(2, dim=-1)
        h = silu(x1) * x2
        return linear_fwd(h, p[prefix + "w3.weight"], p[prefix + "w3.bias"]), (x12,)
    a = linear_fwd(x, p[prefix + "fc1.weight"], p[prefix + "fc1.bias"])
    h = gelu_fwd(a)
    if m_h is not None:
        h = h * m_h
    return linear_fwd(h, p[prefix + "fc2.weight"], p[prefix + "fc2.bias"]), (a, m_h)


def mlp_bwd_dx(dy: Tensor, ctx, p: Dict[str, Tensor], prefix: str, swiglu: bool) -> Tensor:
    if swiglu:
        (x12,) = ctx
        x1, x2 = x12.chunk(2, dim=-1)
        dh = dy @ p[prefix + "w3.weight"]
        dx12 = torch.cat([dh * x2 * silu_grad(x1), dh * silu(x1)], dim=-1)
        return dx12 @ p[prefix + "w12.weight"]
    a, m_h = ctx if len(ctx) == 2 else (ctx[0], None)
    dh = dy @ p[prefix + "fc2.weight"]
    if m_h is not None:
        dh = dh * m_h
    da = dh * gelu_grad(a)
    return da @ p[prefix + "fc1.weight"]


def block_fwd(x: Tensor, p: Dict[str, Tensor], i: int, num_heads: int, r: int, swiglu: bool = False,
              eps: float = 1e-6, dp=None, dm=None):
    """Block.forward, vit.py:279-288: x += drop_path(ls1(attn(norm1 x))); x += drop_path(ls2(mlp(norm2 x))).
    ``dp`` = (s1, s2): the per-sample factors floor(keep_prob + u) / keep_prob of the two DropPath calls (vit.py:74-82; `drop_path`
    below builds them from the uniform numbers), None = identity (p = 0 in every shipped config, SURVEY §5 hazard 14).  nn.Dropout
    ``dm``: the block's nn.Dropout sites as given masks — dict with optional "proj" [B,N,D] (proj_drop, appla_attn.py:82), "h" [B,N,F] and
    "fc2" [B,N,D] (Mlp.drop after the activation and after fc2, vit.py:164-167), all MULTIPLICATIVE (keep / (1 - p); `philox_keep_mask`
    gives the kernels' keep), and "attn" = (keep [B,H,N,N] bool, p) for attn_drop (appla_attn.py:58)."""
    pre = f"blocks.{i}."
    dm = dm or {}
    bc = (lambda s_: s_.reshape((-1,) + (1,) * (x.ndim - 1)).to(x.dtype))
    n1, mean1, rstd1 = layernorm_fwd(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"], eps)
    y, _, actx = apla_attention_fwd(n1, p, pre + "attn.", num_heads, r, attn_drop=dm.get("attn"))
    if dm.get("proj") is not None:
        y = y * dm["proj"]
    g1 = p.get(pre + "ls1.gamma")
    y = y * g1 if g1 is not None else y
    x1 = x + (y * bc(dp[0]) if dp is not None else y)
    n2, mean2, rstd2 = layernorm_fwd(x1, p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps)
    z, mctx = mlp_fwd(n2, p, pre + "mlp.", swiglu, m_h=dm.get("h"))
    if dm.get("fc2") is not None:
        z = z * dm["fc2"]
    g2 = p.get(pre + "ls2.gamma")
    z = z * g2 if g2 is not None else z
    x2 = x1 + (z * bc(dp[1]) if dp is not None else z)
    ctx = dict(x=x, mean1=mean1, rstd1=rstd1, actx=actx, x1=x1, mean2=mean2, rstd2=rstd2, mctx=mctx, dp=dp, dm=dm)
    return x2, ctx


def block_bwd(dx2: Tensor, ctx, p: Dict[str, Tensor], i: int, num_heads: int, swiglu: bool = False,
              need_dx: bool = True):
    """Backward of the block for the APLA trainable set: returns (dx_in | None, dW1, db1)."""
    pre = f"blocks.{i}."
    dp = ctx.get("dp")
    bc = (lambda s_: s_.reshape((-1,) + (1,) * (dx2.ndim - 1)).to(dx2.dtype))
    g2 = p.get(pre + "ls2.gamma")
    dz = dx2 * bc(dp[1]) if dp is not None else dx2          # d(drop_path(z)) / dz = the sample's factor
    dz = dz * g2 if g2 is not None else dz
    dm = ctx.get("dm") or {}
    if dm.get("fc2") is not None:
        dz = dz * dm["fc2"]
    dn2 = mlp_bwd_dx(dz, ctx["mctx"], p, pre + "mlp.", swiglu)
    dx1 = dx2 + layernorm_bwd_dx(dn2, ctx["x1"], p[pre + "norm2.weight"], ctx["mean2"], ctx["rstd2"])
    g1 = p.get(pre + "ls1.gamma")
    dy = dx1 * bc(dp[0]) if dp is not None else dx1
    dy = dy * g1 if g1 is not None else dy
    if dm.get("proj") is not None:
        dy = dy * dm["proj"]
    dn1, dW1, db1 = apla_attention_bwd(dy, ctx["actx"], p, pre + "attn.", num_heads, need_dx)
    if not need_dx:
        return None, dW1, db1
    dx = dx1 + layernorm_bwd_dx(dn1, ctx["x"], p[pre + "norm1.weight"], ctx["mean1"], ctx["rstd1"])
    return dx, dW1, db1


# ----------------------------------------------------------------------------------------------
# a7-a9: ViT forward, classifier head, CE loss, clip, AdamW
# ----------------------------------------------------------------------------------------------


def interpolate_pos_encoding(pos_embed: Tensor, npatch: int) -> Tensor:
    """vit.py:421-437 — bicubic resize of the patch position grid when the input grid differs."""
    N = pos_embed.shape[1] - 1
    if npatch == N:
        return pos_embed
    dim = pos_embed.shape[-1]
    cls_pe, patch_pe = pos_embed[:, 0], pos_embed[:, 1:]
    s = int(math.sqrt(N))
    patch_pe = torch.nn.functional.interpolate(
        patch_pe.reshape(1, s, s, dim).permute(0, 3, 1, 2), scale_factor=math.sqrt(npatch / N),
        mode="bicubic", align_corners=False, recompute_scale_factor=False)
    patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((cls_pe.unsqueeze(0), patch_pe), dim=1)


def patch_embed(images: Tensor, W: Tensor, b: Tensor, patch: int) -> Tensor:
    """PatchEmbed, vit.py:291-307: conv(k=p, stride=p) == unfold + GEMM.  [B,3,S,S] -> [B,Np,D]."""
    B, C, Hh, Ww = images.shape
    gh, gw = Hh // patch, Ww // patch
    cols = images[:, :, :gh * patch, :gw * patch].reshape(B, C, gh, patch, gw, patch)
    cols = cols.permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * patch * patch)
    return cols @ W.reshape(W.shape[0], -1).t() + b


def embed_tokens(images: Tensor, p: Dict[str, Tensor], patch: int) -> Tensor:
    """vit.py:387-396: patchify, prepend cls token, add (interpolated) position embedding."""
    x = patch_embed(images, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], patch)
    B = x.shape[0]
    x = torch.cat((p["cls_token"].expand(B, -1, -1), x), dim=1)
    return x + interpolate_pos_encoding(p["pos_embed"], x.shape[1] - 1)


def softmax_center(x: Tensor, center: Tensor, temp: float) -> Tensor:
    """Teacher centering + sharpening, dinov2/loss/dino_clstoken_loss.py:29-32 (and ibot_patch_loss.py:39-52)."""
    return torch.softmax((x - center) / temp, dim=-1)


def distill_ce(student: Tensor, teacher_probs: Tensor, temp: float, row_weight=None):
    """Per-row cross-entropy between a teacher distribution and the student's log-softmax at temperature `temp` — the summand of
    DINOLoss.forward (dino_clstoken_loss.py:65-77) and iBOTPatchLoss.forward_masked (ibot_patch_loss.py:103-121).
    Returns (sum_r w_r * ce_r, d/d student)."""
    logp = torch.log_softmax(student / temp, -1)
    ce = -(teacher_probs * logp).sum(-1)
    w = torch.ones_like(ce) if row_weight is None else row_weight
    grad = (torch.exp(logp) * teacher_probs.sum(-1, keepdim=True) - teacher_probs) * (w / temp).unsqueeze(-1)
    return (w * ce).sum(), grad


def center_ema(center: Tensor, teacher_output: Tensor, momentum: float) -> Tensor:
    """DINOLoss.apply_center_update for one process (dino_clstoken_loss.py:85-98): EMA of the teacher's batch mean."""
    return center * momentum + teacher_output.mean(0, keepdim=True) * (1 - momentum)


def augment_images(src_u8: Tensor, mean, std, flip=None, perm=None, lam=None, box=None) -> Tensor:
    """ToTensor + Normalize (+ horizontal flip) per sample (defaults/bases.py:69-231) and timm-style Mixup / CutMix against
    the partner sample perm[b] (utils/_utils.py:424-441): src uint8 [B,3,S,S] -> float64 [B,3,S,S].

    PARITY UNPINNED for the Mixup / CutMix part.  The algorithm lives in a third-party dependency that is neither vendored in the
    reference nor installed here: ``timm.data.mixup.Mixup`` (utils/_utils.py:426-428; timm is installed unpinned by docker/Dockerfile:137
    and docker/conda-dinov2.yaml:29), as are torchvision's ToTensor / Normalize / RandomHorizontalFlip.  Restated from the published
    algorithms: Mixup (Zhang et al., ICLR 2018) x = lam x_i + (1 - lam) x_j with the partner j = perm[i] (timm's batch mode pairs sample
    i with B-1-i: pass perm = arange(B).flip(0)); CutMix (Yun et al., ICCV 2019) pastes the partner's pixels inside a rectangle
    whose area fraction is 1 - lam.  The random draws (lam, the rectangle, the flips) are INPUTS here and in the kernel
    (apla_augment_images): what is checked is the mixing itself, against this definition only."""
    x = src_u8.double() / 255.0
    x = (x - torch.tensor(mean, dtype=torch.float64).view(1, 3, 1, 1)) / torch.tensor(std, dtype=torch.float64).view(1, 3, 1, 1)
    if flip is not None:
        x = torch.where(flip.bool().view(-1, 1, 1, 1), x.flip(-1), x)
    if perm is None:
        return x
    partner = x[perm.long()]
    if box is not None:
        out = x.clone()
        for b in range(x.shape[0]):
            y0, y1, x0, x1 = [int(v) for v in box[b]]
            out[b, :, y0:y1, x0:x1] = partner[b, :, y0:y1, x0:x1]
        return out
    l = lam.double().view(-1, 1, 1, 1)
    return l * x + (1 - l) * partner


def cross_entropy_fwd_bwd(logits: Tensor, labels: Tensor):
    """nn.CrossEntropyLoss (mean reduction), defaults/wrappers.py:312-316.  Returns loss, dlogits."""
    m = logits.max(-1, keepdim=True).values
    z = logits - m
    lse = torch.log(torch.exp(z).sum(-1, keepdim=True))
    logp = z - lse
    B = logits.shape[0]
    loss = -logp[torch.arange(B), labels].mean()
    dl = torch.exp(logp)
    dl[torch.arange(B), labels] -= 1.0
    return loss, dl / B


def cross_entropy_soft_fwd_bwd(logits: Tensor, targets: Tensor):
    """nn.CrossEntropyLoss with probability targets [B, C] (what timm's Mixup hands the criterion when the reference's
    advanced_aug is on, utils/_utils.py:424-441): loss = mean_b( -sum_c t_bc log softmax(logits_b)_c )."""
    logp = torch.log_softmax(logits, -1)
    B = logits.shape[0]
    return -(targets * logp).sum(-1).mean(), (torch.exp(logp) * targets.sum(-1, keepdim=True) - targets) / B


def vit_forward(images: Tensor, p: Dict[str, Tensor], cfg: Dict, keep_ctx: bool = True):
    """VisionTransformer.forward_features + Classifier head (vit.py:387-419, defaults/models.py:81-92).
    cfg: dict(patch, depth, heads, r, swiglu, eps).  ``p`` holds backbone keys as in the reference
    state_dict plus ``fc.weight``/``fc.bias`` for the classifier head."""
    x = embed_tokens(images, p, cfg["patch"])
    dms = cfg.get("drop_masks") or {}     # {"pos": multiplicative mask [B,N,D] of pos_drop (vit.py:395), "blocks": [block_fwd's dm per block]}
    if dms.get("pos") is not None:
        x = x * dms["pos"]
    ctxs = []
    dps = cfg.get("dp_scale")       # [2 * depth, B]: rows 2 i / 2 i + 1 = the factors of block i's attention / MLP branch (stochastic depth)
    for i in range(cfg["depth"]):
        x, c = block_fwd(x, p, i, cfg["heads"], cfg["r"], cfg.get("swiglu", False), cfg.get("eps", 1e-6),
                         dp=None if dps is None else (dps[2 * i], dps[2 * i + 1]), dm=(dms.get("blocks") or [None] * cfg["depth"])[i])
        ctxs.append(c if keep_ctx else None)
    cls_in = x[:, 0]
    xn, meanf, rstdf = layernorm_fwd(cls_in, p["norm.weight"], p["norm.bias"], cfg.get("eps", 1e-6))
    logits = linear_fwd(xn, p["fc.weight"], p["fc.bias"])
    return logits, dict(blocks=ctxs, x_last=x, cls_in=cls_in, xn=xn, meanf=meanf, rstdf=rstdf)


def vit_backward(dlogits: Tensor, ctx, p: Dict[str, Tensor], cfg: Dict) -> Dict[str, Tensor]:
    """Gradients of exactly the 2L+2 APLA-trainable tensors."""
    grads: Dict[str, Tensor] = {}
    grads["fc.weight"] = dlogits.t() @ ctx["xn"]
    grads["fc.bias"] = dlogits.sum(0)
    dxn = dlogits @ p["fc.weight"]
    dcls = layernorm_bwd_dx(dxn, ctx["cls_in"], p["norm.weight"], ctx["meanf"], ctx["rstdf"])
    dx = torch.zeros_like(ctx["x_last"])
    dx[:, 0] = dcls
    for i in reversed(range(cfg["depth"])):
        dx, dW1, db1 = block_bwd(dx, ctx["blocks"][i], p, i, cfg["heads"], cfg.get("swiglu", False), need_dx=(i > 0))
        grads[f"blocks.{i}.attn.proj_weight1"] = dW1
        grads[f"blocks.{i}.attn.proj_bias1"] = db1
    return grads


def trainable_names(depth: int) -> List[str]:
    """Order of ``model.named_parameters()`` restricted to requires_grad (backbone blocks then fc)."""
    names = []
    for i in range(depth):
        names += [f"blocks.{i}.attn.proj_weight1", f"blocks.{i}.attn.proj_bias1"]
    return names + ["fc.weight", "fc.bias"]


def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float) -> Tensor:
    """torch.nn.utils.clip_grad_norm_(…, max_norm) — defaults/trainer.py:130,136.  In place."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).to(next(iter(grads.values())).dtype)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return total


def adamw_step(p: Dict[str, Tensor], grads: Dict[str, Tensor], state: Dict, lr: float, wd: float,
               betas=(0.9, 0.999), eps: float = 1e-8):
    """torch.optim.AdamW with the two param groups of defaults/wrappers.py:205-221: weight decay on
    tensors with ndim >= 2 and not named *.bias, none on the rest."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    for name, g in grads.items():
        w = p[name]
        decay = wd if (w.ndim >= 2 and not name.endswith(".bias")) else 0.0
        m = state.setdefault("m." + name, torch.zeros_like(w))
        v = state.setdefault("v." + name, torch.zeros_like(w))
        w.mul_(1.0 - lr * decay)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        w.addcdiv_(m, denom, value=-lr / bc1)


def train_step(images: Tensor, labels: Tensor, p: Dict[str, Tensor], cfg: Dict, opt_state: Dict,
               lr: float = 1e-4, wd: float = 1e-5, clip: Optional[float] = 1.0):
    """Trainer.global_step (defaults/trainer.py:106-151) without AMP: fwd, CE, bwd, clip, AdamW."""
    logits, ctx = vit_forward(images, p, cfg)
    loss, dlogits = cross_entropy_fwd_bwd(logits, labels)
    grads = vit_backward(dlogits, ctx, p, cfg)
    gnorm = clip_grad_norm(grads, clip) if clip else None
    adamw_step(p, grads, opt_state, lr, wd)
    return logits, loss, grads, gnorm


# ---------------------------------------------------------------------------------------------- dropout / stochastic depth
def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123 reference implementation):
    counter [..., 4] and key [..., 2] uint32 arrays -> [..., 4] uint32.  Pinned by the Random123 known-answer vectors
    (tests/test_oracle_golden.py::test_philox_known_answers)."""
    import numpy as np
    c = np.array(counter, dtype=np.uint64) & 0xFFFFFFFF
    k = np.array(key, dtype=np.uint64) & 0xFFFFFFFF
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = k[..., 0].copy(), k[..., 1].copy()
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        n0 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF
        n2 = ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF
        c1, c3, c0, c2 = p1 & 0xFFFFFFFF, p0 & 0xFFFFFFFF, n0, n2
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def philox_keep_mask(n, p, seed, offset=0):
    """keep[i] of apla_dropout_fwd (include/apla_hip.h): word (i & 3) of Philox4x32-10(counter {i >> 2, offset}, key seed) >= p * 2^32.
    nn.Dropout semantics (utils/transformers/vit.py:152-168 Mlp.drop; appla_attn.py:82): y = keep ? x / (1 - p) : 0."""
    import numpy as np
    blk = np.arange((n + 3) // 4, dtype=np.uint64)
    ctr = np.stack([blk & 0xFFFFFFFF, blk >> 32, np.full_like(blk, offset & 0xFFFFFFFF), np.full_like(blk, (offset >> 32) & 0xFFFFFFFF)], axis=-1)
    key = np.stack([np.full_like(blk, seed & 0xFFFFFFFF), np.full_like(blk, (seed >> 32) & 0xFFFFFFFF)], axis=-1)
    words = philox4x32_10(ctr, key).reshape(-1)[:n].astype(np.uint64)
    t = float(np.float32(p)) * 4294967296.0
    threshold = 4294967295 if t >= 4294967295.0 else int(t)
    return words >= threshold


def philox_attn_keep_mask(B, H, N, p, seed, offset=0):
    """keep[b,h,q,k] of apla_attn_fwd_dropout / apla_attn_bwd_dropout (include/apla_hip.h): word (k & 3) of
    Philox4x32-10(counter {k >> 2, row & 0xffffffff, row >> 32, offset}, key seed) >= p * 2^32 with row = (b * H + h) * N + q."""
    import numpy as np
    rows = np.arange(B * H * N, dtype=np.uint64)
    kg = np.arange((N + 3) // 4, dtype=np.uint64)
    R, K = np.meshgrid(rows, kg, indexing="ij")
    ctr = np.stack([K & 0xFFFFFFFF, R & 0xFFFFFFFF, R >> 32, np.full_like(K, offset & 0xFFFFFFFF)], axis=-1)
    key = np.stack([np.full_like(K, seed & 0xFFFFFFFF), np.full_like(K, (seed >> 32) & 0xFFFFFFFF)], axis=-1)
    words = philox4x32_10(ctr, key).reshape(B * H * N, -1)[:, :N].astype(np.uint64)
    t = float(np.float32(p)) * 4294967296.0
    threshold = 4294967295 if t >= 4294967295.0 else int(t)
    return torch.from_numpy(words >= threshold).reshape(B, H, N, N)


def drop_path(x, u, drop_prob):
    """utils/transformers/vit.py:74-82 with the uniform numbers u [B] given: x / keep_prob * floor(keep_prob + u)."""
    keep = 1.0 - drop_prob
    return x / keep * torch.floor(keep + u).reshape((-1,) + (1,) * (x.ndim - 1))
