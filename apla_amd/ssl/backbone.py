"""Multi-crop ("nested") forward of the ViT backbone for the DINOv2-APLA step: self_supervised/dinov2/dinov2_vits.py:210-267
(prepare_tokens_with_masks, forward_features_list) over dinov2/layers/block.py:188-288 (NestedTensorBlock.forward_nested).

The reference concatenates the token sequences of all crops of all resolutions into one [1, total, C] tensor, runs every
block once on it with a block-diagonal attention mask and splits the result again.  Here the packed tensor goes through
the same modules as the dense path — LayerNorm / qkv / projection / MLP are token-wise, and APLA_MemEffAttention hands the
sequence offsets to the block-diagonal attention kernels — so a list forward is ONE pass per block, not one per
resolution.
"""
import math
from typing import List, Optional

import torch
from torch import nn

from .. import functional as AF
from .. import ops
from ..apla.appla_attn_mem_eff import APLA_MemEffAttention
from ..nested import BlockDiagonalMask
from ..vit import Attention, VisionTransformer


class MemEffAttention(Attention):
    """dinov2/layers/attention.py:66-89: plain attention with the ``forward(x, attn_bias=None) -> x`` convention.  This is
    the module the backbone keeps when ``partial_size: full`` leaves the projection un-split (apla_vit.py:66-75: the whole
    ``attn.proj`` Linear is trainable; its dW runs on the same TN MFMA kernel with r = D)."""

    def forward(self, x, attn_bias=None, rows=None):
        if attn_bias is None:
            return super().forward(x)[0]
        if not isinstance(attn_bias, BlockDiagonalMask):
            raise TypeError(f"attn_bias must be an apla_amd.nested.BlockDiagonalMask, got {type(attn_bias).__name__}")
        AF.require_no_dropout(self.attn_drop, self.training)
        if x.ndim != 3 or x.shape[0] != 1 or x.shape[1] != attn_bias.total:
            raise ValueError(f"a packed batch must be [1, {attn_bias.total}, C]; got {tuple(x.shape)}")
        qkv = AF.linear(x, self.qkv.weight, self.qkv.bias)
        o = AF.attention_core_varlen(qkv, attn_bias.cu_seqlens(x.device), attn_bias.max_seqlen, self.num_heads, self.scale,
                                        runs=attn_bias.runs())
        if rows is not None:
            o = o.index_select(1, rows)
        return AF.dropout(AF.linear(o, self.proj.weight, self.proj.bias).to(x.dtype), self.proj_drop, self.training)


class DinoVisionTransformer(VisionTransformer):
    """apla_amd.vit.VisionTransformer + what the dinov2 backbone adds (self_supervised/dinov2/dinov2_vits.py:41-350): the
    learnable mask token, MemEffAttention blocks, the position-embedding resize with the 0.1 offset, the list forward and
    the ``forward(x, masks=None, is_training=False)`` convention.  Parameter names are those of the dinov2 checkpoint."""

    def __init__(self, *args, interpolate_offset=0.1, interpolate_antialias=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.mask_token = nn.Parameter(torch.zeros(1, self.embed_dim))
        self.interpolate_offset, self.interpolate_antialias = interpolate_offset, interpolate_antialias
        for blk in self.blocks:
            blk.attn.__class__ = MemEffAttention   # same state, dinov2 calling convention
        self._pos_cache = {}

    def interpolate_pos_encoding(self, npatch: int) -> torch.Tensor:
        """dinov2_vits.py:176-208 for square inputs: bicubic resize of the pretraining grid by the scale factor
        (side + 0.1) / M — the "historical kludge" the checkpoints were trained with; it is NOT the size-based resize of
        utils/transformers/vit.py.  The position table is frozen, so the result is cached per grid size."""
        pe = self.pos_embed
        N = pe.shape[1] - 1
        if npatch == N:
            return pe
        key = (npatch, pe._version, pe.device)
        if key not in self._pos_cache:
            if any(k[1:] != key[1:] for k in self._pos_cache):   # the table changed (checkpoint load, .to(device)): start over
                self._pos_cache = {}
            dim, M, side = pe.shape[-1], int(math.sqrt(N)), int(math.sqrt(npatch))
            assert M * M == N and side * side == npatch
            kw = {"scale_factor": (float(side + self.interpolate_offset) / M,) * 2} if self.interpolate_offset else {"size": (side, side)}
            grid = nn.functional.interpolate(pe.detach().float()[:, 1:].reshape(1, M, M, dim).permute(0, 3, 1, 2), mode="bicubic",
                                             antialias=self.interpolate_antialias, **kw)
            assert grid.shape[-2:] == (side, side)
            self._pos_cache[key] = torch.cat((pe.detach().float()[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)
        return self._pos_cache[key]

    def prepare_tokens_with_masks(self, x, masks: Optional[torch.Tensor] = None):
        x = self.patch_embed(x)
        if masks is not None:  # iBOT: masked patches are replaced by the mask token before the blocks (dinov2_vits.py:213-214)
            x = torch.where(masks.unsqueeze(-1), self.mask_token.to(x.dtype).unsqueeze(0), x)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1).to(x.dtype), x), dim=1)
        return x + self.interpolate_pos_encoding(x.shape[1] - 1).to(x.dtype)

    def pack_tokens(self, x_list: List[torch.Tensor], masks_list: List[Optional[torch.Tensor]]):
        """prepare_tokens_with_masks of every crop group + the packing of dinov2 block.py:204-217 in one pass per group: the patch
        GEMM's 16-bit rows, the class token, the mask token and the resized position table are combined in fp32 straight into the
        group's rows of the packed residual stream (apla_assemble_tokens_masked) — the torch route is where + cat + add + cat, four
        passes and a 16-bit position add.  Needs the class / mask tokens and the table frozen (they are under APLA)."""
        groups = []
        for x in x_list:
            if x.shape[-1] % self.patch_size or x.shape[-2] != x.shape[-1]:
                raise ValueError(f"square crops with sides divisible by {self.patch_size} expected, got {tuple(x.shape)}")
            groups.append((x.shape[0], (x.shape[-1] // self.patch_size) ** 2))
        mask = BlockDiagonalMask([n + 1 for b, n in groups for _ in range(b)])
        mask._batch_sizes = [b for b, _ in groups]
        D = self.embed_dim
        packed = torch.empty(1, mask.total, D, device=x_list[0].device, dtype=torch.float32)
        cls = self.cls_token.detach().reshape(-1).float().contiguous()
        row = 0
        for x, m, (b, n) in zip(x_list, masks_list, groups):
            patches = self.patch_embed(x).reshape(b * n, D)
            pos = self.interpolate_pos_encoding(n).detach().reshape(n + 1, D).float().contiguous()
            ops.assemble_tokens(patches, cls, pos, b, n, out=packed[0, row:row + b * (n + 1)],
                                masked=None if m is None else m.reshape(b, n).contiguous(),
                                mask_token=None if m is None else self.mask_token.detach().reshape(-1).float().contiguous())
            row += b * (n + 1)
        return mask, packed

    def _packed_input(self, x_list, masks_list):
        """(mask, packed tokens [1, total, D]) of a list of crop groups: one assemble kernel per group when the token parameters are
        frozen (they are under APLA), the reference's torch route (prepare_tokens_with_masks + concatenation) otherwise."""
        for blk in self.blocks:
            if not isinstance(blk.attn, (APLA_MemEffAttention, MemEffAttention)):   # block.py:249
                raise NotImplementedError("the packed forward needs (APLA_)MemEffAttention blocks (build_apla(..., 'apla_attn_mem_eff'))")
        frozen = not (self.cls_token.requires_grad or self.pos_embed.requires_grad or self.mask_token.requires_grad
                      or any(p.requires_grad for p in self.patch_embed.parameters()))
        if x_list[0].is_cuda and frozen:
            return self.pack_tokens(x_list, masks_list)
        return BlockDiagonalMask.from_tensor_list([self.prepare_tokens_with_masks(x, m) for x, m in zip(x_list, masks_list)])

    def forward_features_list(self, x_list: List[torch.Tensor], masks_list: List[Optional[torch.Tensor]]):
        attn_bias, x = self._packed_input(x_list, masks_list)
        x_pre, x_nrm = self.run_blocks(x, attn_bias)     # LayerNorm is token-wise: normalise packed, split afterwards
        outs = []
        for xi, x_norm, masks in zip(attn_bias.split(x_pre), attn_bias.split(x_nrm), masks_list):
            outs.append({"x_norm_clstoken": x_norm[:, 0], "x_norm_regtokens": x_norm[:, 1:1], "x_norm_patchtokens": x_norm[:, 1:],
                         "x_prenorm": xi, "masks": masks})
        return outs

    def forward_cls_and_masked(self, x_list: List[torch.Tensor], masks_list: List[Optional[torch.Tensor]],
                               masked_idx: Optional[torch.Tensor] = None):
        """What the DINOv2 meta-architecture reads of ``forward_features_list`` (models.py:231-347): the normalised class token of
        every crop, and the normalised patch tokens of the FIRST crop group at ``masked_idx`` (int64 indices into its
        [crops * patches] patch tokens: the collate's ``mask_indices_list``).  Same values; the last block's projection, MLP and
        both LayerNorms around it run on those rows only (every token still attends and is attended to: the block's K / V are
        dense), forward and backward — 5 519 of the student's 58 496 and 5 007 of the teacher's 32 896 rows at config 4.
        Returns ([cls of group 0, cls of group 1, ...], masked patch tokens or None)."""
        attn_bias, x = self._packed_input(x_list, masks_list)
        rows, counts = [attn_bias.seq_starts(x.device)], list(attn_bias._batch_sizes)
        if masked_idx is not None and masked_idx.numel():
            n0 = attn_bias.seqlens[0] - 1                                   # patches per crop of group 0
            rows.append(masked_idx + torch.div(masked_idx, n0, rounding_mode="floor") + 1)    # crop s, patch t -> row s (n0 + 1) + 1 + t
        rows = torch.cat(rows)
        _, x_norm = self.run_blocks(x, attn_bias, last_rows=rows)
        x_norm = x_norm[0]
        n_cls = sum(counts)
        cls = list(x_norm[:n_cls].split(counts))
        return cls, (x_norm[n_cls:] if len(rows) > n_cls else None)

    def forward_features_dict(self, x, masks=None):
        if isinstance(x, (list, tuple)):
            return self.forward_features_list(list(x), list(masks) if masks is not None else [None] * len(x))
        return self.forward_features_list([x], [masks])[0]

    def forward(self, *args, is_training=False, **kwargs):
        """dinov2_vits.py:342-349: the feature dictionary while training, the CLS feature otherwise."""
        x = args[0] if args else kwargs.pop("x")
        ret = self.forward_features_dict(x, kwargs.get("masks"))
        if is_training:
            return ret
        return ret["x_norm_clstoken"]
