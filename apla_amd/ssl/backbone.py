"""Multi-crop ("nested") forward of the ViT backbone for the DINOv2-APLA step: self_supervised/dinov2/dinov2_vits.py:210-267
(prepare_tokens_with_masks, forward_features_list) over dinov2/layers/block.py:188-288 (NestedTensorBlock.forward_nested).

The reference concatenates the token sequences of all crops of all resolutions into one [1, total, C] tensor, runs every
block once on it with a block-diagonal attention mask and splits the result again.  Here the packed tensor goes through
the same modules as the dense path — LayerNorm / qkv / projection / MLP are token-wise, and APLA_MemEffAttention hands the
sequence offsets to the block-diagonal attention kernels — so a list forward is ONE pass per block, not one per
resolution.
"""
from typing import List, Optional

import torch
from torch import nn

from .. import functional as AF
from ..apla.appla_attn_mem_eff import APLA_MemEffAttention
from ..nested import BlockDiagonalMask
from ..vit import VisionTransformer


class DinoVisionTransformer(VisionTransformer):
    """apla_amd.vit.VisionTransformer + the learnable mask token and the list forward of the dinov2 backbone."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.mask_token = nn.Parameter(torch.zeros(1, self.embed_dim))

    def prepare_tokens_with_masks(self, x, masks: Optional[torch.Tensor] = None):
        x = self.patch_embed(x)
        if masks is not None:  # iBOT: masked patches are replaced by the mask token before the blocks (dinov2_vits.py:213-214)
            x = torch.where(masks.unsqueeze(-1), self.mask_token.to(x.dtype).unsqueeze(0), x)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1).to(x.dtype), x), dim=1)
        return x + self.interpolate_pos_encoding(x.shape[1] - 1).to(x.dtype)

    def _block_packed(self, blk, x, attn_bias):
        if not isinstance(blk.attn, APLA_MemEffAttention):
            raise NotImplementedError("the packed forward needs APLA_MemEffAttention blocks (build_apla(..., 'apla_attn_mem_eff'))")
        x = x + blk.ls1(blk.attn(AF.layer_norm(x, blk.norm1), attn_bias=attn_bias))
        return x + blk.ls2(blk.mlp(AF.layer_norm(x, blk.norm2)))

    def forward_features_list(self, x_list: List[torch.Tensor], masks_list: List[Optional[torch.Tensor]]):
        toks = [self.prepare_tokens_with_masks(x, m) for x, m in zip(x_list, masks_list)]
        attn_bias, x = BlockDiagonalMask.from_tensor_list(toks)
        for blk in self.blocks:
            x = self._block_packed(blk, x, attn_bias)
        outs = []
        for xi, masks in zip(attn_bias.split(x), masks_list):
            x_norm = AF.layer_norm(xi, self.norm)
            outs.append({"x_norm_clstoken": x_norm[:, 0], "x_norm_regtokens": x_norm[:, 1:1], "x_norm_patchtokens": x_norm[:, 1:],
                         "x_prenorm": xi, "masks": masks})
        return outs

    def forward_features_dict(self, x, masks=None):
        if isinstance(x, (list, tuple)):
            return self.forward_features_list(list(x), list(masks) if masks is not None else [None] * len(x))
        return self.forward_features_list([x], [masks])[0]
