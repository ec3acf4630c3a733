"""Host-side ViT module tree for the MI355X APLA path.

This mirrors the *interface* of the reference backbone (utils/transformers/vit.py): same module/parameter names (so
reference checkpoints and ``state_dict`` layouts interchange: ``blocks.i.{norm1,attn.qkv,attn.proj,ls1.gamma,norm2,
mlp.fc1,mlp.fc2,ls2.gamma}``, ``patch_embed.proj``, ``cls_token``, ``pos_embed``, ``norm``, ``fc``), same factory
names and kwargs (``vit_small(pretrained=False, patch_size=…, pretrained_type=…, img_size=[…], block_conf=…)``,
vit.py:511-596) and the same construction order, so that under the same ``torch.manual_seed`` the synthetic
initialisation draws bit-identical weights (vit.py:310-350; checked against digests in tests/golden/g5_cfg1_digests.json).

The modules are *state holders plus a drop-in forward*: ``forward`` on HIP tensors runs the hand-written kernels of
libapla_hip.so through autograd Functions (apla_amd/functional.py).  The fused training step used by the trainer and
bench does not go through nn.Module.forward at all: apla_amd/engine.py reads the parameters of this tree, lays them
out for the kernels and runs forward+backward as one explicit launch sequence.  There is no CPU execution path.
"""
import math
from functools import partial

import torch
import torch.nn as nn

from . import functional as AF


def _trunc_normal_(t: torch.Tensor, std: float):
    # same draw sequence as the reference's helper (vit.py:34-71): uniform -> erfinv -> scale -> clamp to [-2, 2]
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


class LayerScale(nn.Module):
    """Per-channel scale, vit.py:232-244 (dinov2).  Frozen under APLA; the engine folds gamma into the weights."""

    def __init__(self, dim, init_values=1e-5, inplace=False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x.mul_(self.gamma) if self.inplace else x * self.gamma


class Mlp(nn.Module):
    """fc1 -> GELU(erf) -> fc2, vit.py:152-168."""

    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        if self.training and self.drop.p > 0.0:   # fc1 -> act -> drop -> fc2 -> drop (vit.py:162-168): the one-node MLP has no place for the first
            h = AF.dropout(AF.linear_gelu(x, self.fc1.weight, self.fc1.bias), self.drop, True)
            return AF.dropout(AF.linear(h, self.fc2.weight, self.fc2.bias), self.drop, True)
        return AF.mlp_gelu(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)


class SwiGLUFFNFused(nn.Module):
    """w12 -> silu(x1)*x2 -> w3 with hidden = (int(h*2/3)+7)//8*8, vit.py:108-149 (ViT-g)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0.0, bias=True):
        super().__init__()
        hidden = hidden_features or in_features
        hidden = (int(hidden * 2 / 3) + 7) // 8 * 8
        self.w12 = nn.Linear(in_features, 2 * hidden, bias=bias)
        self.w3 = nn.Linear(hidden, out_features or in_features, bias=bias)

    def forward(self, x):
        return AF.mlp_swiglu(x, self.w12.weight, self.w12.bias, self.w3.weight, self.w3.bias)


class Attention(nn.Module):
    """Plain MHSA, vit.py:171-196: the module APLA swaps out (exposes dim/num_heads/scale/qkv/proj/attn_drop/proj_drop,
    which is the host-model contract of apla_vit.py:17-37)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim = dim
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.return_attn_matrix = False

    def forward(self, x):
        y, attn = AF.attention_module_forward(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias,
                                              self.num_heads, self.scale, self.return_attn_matrix,
                                              AF.active_p(self.attn_drop, self.training))   # vit.py:190-192
        return AF.dropout(y, self.proj_drop, self.training), attn


class DropPath(nn.Module):
    """Stochastic depth per sample (utils/transformers/vit.py:85-93)."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return AF.drop_path(x, self.drop_prob, self.training)


class Block(nn.Module):
    """Pre-LN block, vit.py:247-288.  ``conf`` carries has_layerscale / layerscale_init_values."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, norm_layer=nn.LayerNorm, conf=None, use_swiglu=False):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()   # vit.py:257
        self.norm2 = norm_layer(dim)
        hidden = int(dim * mlp_ratio)
        self.mlp = SwiGLUFFNFused(dim, hidden, drop=drop) if use_swiglu else Mlp(dim, hidden, drop=drop)
        has_ls = bool(conf) and bool(_cfg_get(conf, "has_layerscale", False))
        if has_ls:
            init = _cfg_get(conf, "layerscale_init_values", 1e-5)
            self.ls1, self.ls2 = LayerScale(dim, init), LayerScale(dim, init)
        else:
            self.ls1, self.ls2 = nn.Identity(), nn.Identity()

    def forward(self, x, return_attention=False, return_intermediate=False):
        want_attn = return_attention or return_intermediate
        prev = getattr(self.attn, "return_attn_matrix", False)
        if want_attn:
            self.attn.return_attn_matrix = True
        try:
            res = self.attn(AF.layer_norm(x, self.norm1))
        finally:
            if want_attn:
                self.attn.return_attn_matrix = prev
        y, attn = res if isinstance(res, tuple) else (res, None)
        y = self.ls1(y)
        if return_attention and not return_intermediate:
            return attn
        x = x + self.drop_path(y)
        x = x + self.drop_path(self.ls2(self.mlp(AF.layer_norm(x, self.norm2))))
        return (x, attn) if return_intermediate else x


def _cfg_get(conf, key, default):
    if isinstance(conf, dict):
        return conf.get(key, default)
    return getattr(conf, key, default)


class PatchEmbed(nn.Module):
    """p x p stride-p conv patchifier, vit.py:291-307 (frozen and forward-only under APLA)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size, self.patch_size, self.embed_dim = img_size, patch_size, embed_dim
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return AF.patch_embed(x, self.proj.weight, self.proj.bias, self.patch_size)


class VisionTransformer(nn.Module):
    """vit.py:310-508 (forward/forward_features; the visualisation helpers are out of scope)."""

    def __init__(self, img_size=(224,), patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, norm_layer=nn.LayerNorm, use_swiglu=False, block_conf=None, **kwargs):
        super().__init__()
        self.is_memory_efficient = kwargs.get("is_memory_efficient", False)
        self.num_features = self.embed_dim = embed_dim
        self.depth, self.num_heads, self.patch_size, self.use_swiglu = depth, num_heads, patch_size, use_swiglu
        self.eps = getattr(norm_layer(embed_dim), "eps", 1e-5)
        self.patch_embed = PatchEmbed(img_size[0], patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, qk_scale, drop_rate, attn_drop_rate, dpr[i], norm_layer,
                  block_conf, use_swiglu) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.fc = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        _trunc_normal_(self.pos_embed, 0.02)
        _trunc_normal_(self.cls_token, 0.02)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, 0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def interpolate_pos_encoding(self, npatch: int) -> torch.Tensor:
        """Bicubic resize of the patch position grid when the input grid differs from the pretrain grid
        (vit.py:421-437).  Frozen parameter => computed with torch once per (grid) and cached by the engine."""
        pe = self.pos_embed
        N = pe.shape[1] - 1
        if npatch == N:
            return pe
        dim = pe.shape[-1]
        s = int(math.sqrt(N))
        grid = nn.functional.interpolate(pe[:, 1:].reshape(1, s, s, dim).permute(0, 3, 1, 2),
                                         scale_factor=math.sqrt(npatch / N), mode="bicubic", align_corners=False,
                                         recompute_scale_factor=False)
        return torch.cat((pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)

    def run_blocks(self, x, attn_bias=None, last_rows=None):
        """All blocks + the final norm on a token tensor ([B, N, D], or packed [1, total, D] with a BlockDiagonalMask):
        x += ls1(attn(norm1 x)); x += ls2(mlp(norm2 x)) (vit.py:279-288).  The residual stream is kept in fp32, every residual
        add is fused into the LayerNorm that follows it (AF.add_layer_norm: one pass forward, one pass backward) and frozen
        LayerScale vectors are folded into the GEMM that produces the branch (APLA projection / fc2) — what the fused engine
        does, here for the autograd path.  Blocks whose attention / MLP are not the modules of this package, or whose
        LayerScale is trainable, go through their own forward.  Returns (x_prenorm fp32, x_norm bf16).
        ``last_rows`` (packed batches): int64 row indices the caller will read of the result.  Everything behind the last block's
        attention product is token-wise, so that block projects, normalises and runs its MLP on those rows only and the result is
        [1, len(last_rows), D]; autograd scatters the two gradients (attention output, residual stream) back into zero rows."""
        if x.is_cuda and not x.requires_grad:
            return self._run_blocks_stream(x, attn_bias, last_rows)
        res, branch = x.float(), None
        n_blocks = len(self.blocks)
        for bi, blk in enumerate(self.blocks):
            if branch is None:
                h = AF.layer_norm(res, blk.norm1)
            else:
                res, h = AF.add_layer_norm(res, branch, blk.norm1)
            if last_rows is not None and bi == n_blocks - 1:
                if attn_bias is None:
                    raise ValueError("last_rows needs a packed batch")
                y, branch_of = self._block_branches(blk, h, attn_bias, last_rows)
                res = res.index_select(1, last_rows)
            else:
                y, branch_of = self._block_branches(blk, h, attn_bias, None)
            res, h = AF.add_layer_norm(res, y, blk.norm2)
            branch = branch_of(h)
        return AF.add_layer_norm(res, branch, self.norm)

    def _block_branches(self, blk, h, attn_bias, rows):
        """(attention branch of a block on the normalised input h, function computing its MLP branch from the second norm's output):
        LayerScale folded into the producing GEMM where it is frozen, stochastic depth applied to both branches (vit.py:279-288)."""
        g1, g2 = getattr(blk.ls1, "gamma", None), getattr(blk.ls2, "gamma", None)
        kw = {} if attn_bias is None else {"attn_bias": attn_bias}
        if rows is not None:
            kw["rows"] = rows
        if hasattr(blk.attn, "_project") and (g1 is None or not g1.requires_grad):   # APLA attention: scale folded into the projection
            y = blk.attn(h, ls_gamma=g1, **kw)
            y = y[0] if isinstance(y, tuple) else y
        else:
            y = blk.attn(h, **kw)
            y = blk.ls1(y[0] if isinstance(y, tuple) else y)
        stochastic = self.training and isinstance(blk.drop_path, DropPath) and blk.drop_path.drop_prob > 0.0
        if stochastic:
            if attn_bias is not None:
                raise NotImplementedError("stochastic depth on a packed batch (its samples are not the leading dimension) is not supported")
            y = blk.drop_path(y)

        def mlp_branch(h2):
            if isinstance(blk.mlp, Mlp) and (g2 is None or not g2.requires_grad) and not (self.training and blk.mlp.drop.p > 0.0):
                b = AF.mlp_gelu(h2, blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias, gamma=g2)
            else:
                b = blk.ls2(blk.mlp(h2))
            return blk.drop_path(b) if stochastic else b
        return y, mlp_branch

    def _run_blocks_stream(self, x, attn_bias, last_rows):
        """run_blocks with the gradient of the residual stream in 16 bits (AF.ResidualStream): the values of the stream stay fp32."""
        st, branch = AF.ResidualStream(x), None
        n_blocks = len(self.blocks)
        for bi, blk in enumerate(self.blocks):
            # block 0 reads the input tokens, which take no gradient on this path: a plain LayerNorm outside the graph
            h = AF.layer_norm(st.values, blk.norm1) if branch is None else AF.stream_add_layer_norm(st, branch, blk.norm1)
            last = last_rows is not None and bi == n_blocks - 1
            if last and attn_bias is None:
                raise ValueError("last_rows needs a packed batch")
            y, branch_of = self._block_branches(blk, h, attn_bias, last_rows if last else None)
            if last:
                st.select_rows(last_rows)
            h = AF.stream_add_layer_norm(st, y, blk.norm2)
            branch = branch_of(h)
        h = AF.stream_add_layer_norm(st, branch, self.norm)
        return st.exit(), h

    def forward_features(self, x):
        x = self.patch_embed(x)
        B = x.shape[0]
        x = torch.cat((self.cls_token.expand(B, -1, -1).to(x.dtype), x), dim=1)
        x = x + self.interpolate_pos_encoding(x.shape[1] - 1).to(x.dtype)
        x = AF.dropout(x, self.pos_drop, self.training)
        _, x = self.run_blocks(x)
        return x[:, 0]

    def forward(self, x):
        if isinstance(x, (list, tuple)):  # multi-resolution crops: one pass per resolution (vit.py:352-385)
            return torch.cat([self.fc(self.forward_features(xi)) for xi in x])
        return self.fc(self.forward_features(x))


def _factory(embed_dim, depth, num_heads, **fixed):
    def make(pretrained=False, **kwargs):
        if pretrained:
            raise RuntimeError("pretrained weights need network access (utils/transformers/transformers_utils.py:10-57); "
                               "load a checkpoint with load_state_dict instead")
        kwargs = dict(kwargs)
        ptype = kwargs.get("pretrained_type", "dinov2")
        kwargs.setdefault("patch_size", 16)
        qkv_bias = fixed.get("qkv_bias", True) if "qkv_bias_fn" not in fixed else fixed["qkv_bias_fn"](ptype)
        return VisionTransformer(embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=4, qkv_bias=qkv_bias,
                                 norm_layer=partial(nn.LayerNorm, eps=1e-6), use_swiglu=fixed.get("use_swiglu", False),
                                 **kwargs)
    return make


# vit.py:511-596: tiny/small/base/large/giant
vit_tiny = _factory(192, 12, 3)
vit_small = _factory(384, 12, 6)
vit_base = _factory(768, 12, 12, qkv_bias_fn=lambda ptype: ptype != "in21k")
vit_large = _factory(1024, 24, 16)
vit_giant = _factory(1536, 40, 24, use_swiglu=True)
