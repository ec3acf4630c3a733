"""APLA_MemEffAttention (reference surface: apla/appla_attn_mem_eff.py:22-67): ``forward(x, attn_bias=None) -> x``.

The reference delegates the attention product to ``xformers.ops.memory_efficient_attention``; on MI355X the same
fused (never-materialised) attention is what APLA_Attention already runs, so this subclass only changes the return
convention (tensor instead of ``(x, attn)`` — dinov2 Blocks call ``ls1(attn(norm1(x)))`` directly).  Block-diagonal
``attn_bias`` for packed multi-crop sequences (dinov2 nested-tensor path) is a SURVEY §8f "next" row and raises.
"""
from .appla_attn import APLA_Attention


class APLA_MemEffAttention(APLA_Attention):
    def forward(self, x, attn_bias=None):
        if attn_bias is not None:
            raise NotImplementedError("block-diagonal attn_bias (packed crops) is not implemented on the HIP path yet")
        y, _ = super().forward(x)
        return y
