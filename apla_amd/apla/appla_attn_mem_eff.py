"""APLA_MemEffAttention (reference surface: apla/appla_attn_mem_eff.py:22-67): ``forward(x, attn_bias=None) -> x``.

The reference delegates the attention product to ``xformers.ops.memory_efficient_attention``; on MI355X the same
fused (never-materialised) attention is what APLA_Attention already runs, so without a bias this subclass only changes
the return convention (tensor instead of ``(x, attn)`` — dinov2 Blocks call ``ls1(attn(norm1(x)))`` directly).

With ``attn_bias`` (the dinov2 nested-tensor path, block.py:254-288: crops of different sizes packed into one
``[1, total, C]`` tensor and a ``BlockDiagonalMask`` over their lengths) the block-diagonal restriction is handed to the
kernels as cumulative sequence offsets (``apla_attn_varlen_fwd/bwd``); the qkv Linear and the APLA projection are
token-wise and run on the packed tensor unchanged (appla_attn_mem_eff.py:37-65).
"""
from .appla_attn import APLA_Attention
from .. import functional as AF
from ..nested import BlockDiagonalMask


class APLA_MemEffAttention(APLA_Attention):
    def forward(self, x, attn_bias=None, ls_gamma=None, rows=None):
        """``ls_gamma`` (extension): the block's frozen LayerScale vector; when given the result is ls1(attention(x)), the
        scale being folded into the projection GEMM (apla_amd/ssl/backbone.py uses it on the packed path).
        ``rows`` (extension, packed path): int64 row indices; only those token rows of the attention output are projected and
        returned ([1, len(rows), C]) — the last block of a backbone whose consumers read a subset of its tokens."""
        if attn_bias is None:
            y, _ = super().forward(x, ls_gamma)
            return y
        if not isinstance(attn_bias, BlockDiagonalMask):
            raise TypeError("attn_bias must be an apla_amd.nested.BlockDiagonalMask (the xformers mask class the reference "
                            f"uses is not a dependency of this package); got {type(attn_bias).__name__}")
        AF.require_no_dropout(self.attn_drop, self.training)
        if x.ndim != 3 or x.shape[0] != 1 or x.shape[1] != attn_bias.total:
            raise ValueError(f"a packed batch must be [1, {attn_bias.total}, C]; got {tuple(x.shape)}")
        qkv = AF.linear(x, self.qkv.weight, self.qkv.bias)
        o = AF.attention_core_varlen(qkv, attn_bias.cu_seqlens(x.device), attn_bias.max_seqlen, self.num_heads, self.scale,
                                        runs=attn_bias.runs())
        if rows is not None:
            o = o.index_select(1, rows)
        return AF.dropout(self._project(o, ls_gamma).to(x.dtype), self.proj_drop, self.training)
