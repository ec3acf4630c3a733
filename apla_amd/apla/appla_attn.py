"""APLA_Attention — the drop-in operator (reference surface: apla/appla_attn.py:10-83).

Same constructor signature, same parameter/buffer names and shapes (``qkv.{weight,bias}``, ``proj_weight1 [r,D]``,
``proj_bias1 [r]`` trainable; ``proj_weight2 [D-r,D]``, ``proj_bias2 [D-r]`` frozen; buffer ``inds`` int64 [D]), same
index-selection rule (``torch.randperm(dim)`` on the global CPU generator *before* the qkv Linear is created, so the
RNG stream matches the reference draw for draw), same ``(x, attn)`` return convention.

What differs is the execution: forward/backward run on the MI355X kernels.  The two ``F.linear`` + two ``scatter_``
of the reference forward (:64-79) become ONE GEMM over a natural-order merged weight whose r trainable rows are
re-scattered from the fp32 masters each forward (weight-side scatter of r·D elements instead of an activation-side
scatter of B·N·D elements plus two host→device index copies per call); the weight gradient is computed only for the r
selected output features.  ``attn`` ([B,H,N,N]) is materialised only when ``return_attn_matrix`` is set (the reference
always builds it; ``Block.forward`` discards it unless ``return_attention=True``).
"""
import torch
import torch.nn as nn

from .. import functional as AF
from .. import ops


class APLA_Attention(nn.Module):
    def __init__(self, config, dim, indices=None, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.partial_size = config.partial_size
        self.dim = dim
        if not isinstance(self.partial_size, int) or not (0 < self.partial_size <= dim):
            # appla_attn.py:33 would fail with a TypeError on 'full'; say what is wrong instead
            raise TypeError(f"partial_size must be an int in (0, dim]; got {self.partial_size!r} "
                            "('full' is only valid through build_apla(is_multi_gpu=True))")
        # pre-defined indices, or sample ONCE at construction on the global CPU generator (appla_attn.py:22-27)
        self.indices = indices if indices is not None else torch.randperm(self.dim)
        if self.indices.numel() != dim:
            raise ValueError("indices must be a permutation of range(dim)")
        self.register_buffer("inds", self.indices)
        self.trainable_inds = self.indices[:self.partial_size]
        self.freezed_inds = self.indices[self.partial_size:]

        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)  # frozen
        for p in self.qkv.parameters():
            p.requires_grad = False
        r = self.partial_size
        self.proj_weight1 = nn.Parameter(torch.empty(r, dim), requires_grad=True)
        self.proj_weight2 = nn.Parameter(torch.empty(dim - r, dim), requires_grad=False)
        self.proj_bias1 = nn.Parameter(torch.empty(r), requires_grad=True)
        self.proj_bias2 = nn.Parameter(torch.empty(dim - r), requires_grad=False)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj_drop = nn.Dropout(proj_drop)
        self.return_attn_matrix = False
        self._proj_state = AF.AplaProjState()

    def _attend(self, x):
        B, N, _ = x.shape
        qkv = AF.linear(x, self.qkv.weight, self.qkv.bias)
        p = AF.active_p(self.attn_drop, self.training)
        seed = AF.draw_seed() if p > 0.0 else 0
        o, lse = AF.attention_core(qkv, B, N, self.num_heads, self.scale, p, seed)   # appla_attn.py:56-60
        return qkv, o, lse, (p, seed)

    def _project(self, o, gamma=None):
        return AF.apla_projection(o, self.proj_weight1, self.proj_bias1, self.proj_weight2, self.proj_bias2,
                                  self.inds, self._proj_state, gamma)

    def forward(self, x, ls_gamma=None):
        """``ls_gamma`` (extension, used by VisionTransformer.run_blocks): the block's frozen LayerScale vector; when given,
        x is ls1(attention(x)) with the scale folded into the projection GEMM."""
        B, N, _ = x.shape
        qkv, o, lse, (p, seed) = self._attend(x)
        # proj_drop (appla_attn.py:82) commutes with the LayerScale vector folded into the projection: both are element-wise
        y = AF.dropout(self._project(o, ls_gamma).to(x.dtype), self.proj_drop, self.training)
        attn = None
        if self.return_attn_matrix:
            attn = ops.attn_probs(qkv.detach().reshape(B * N, -1), lse, B, N, self.num_heads, self.scale, p, seed)   # after attn_drop (:58, :83)
        return y, attn
