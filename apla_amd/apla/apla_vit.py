"""APLA plugin (reference surface: apla/apla_vit.py:11-101): ``replace_attn_with_apla`` / ``build_apla``.

Host-model contract (unchanged): ``model.blocks`` is iterable and every ``block.attn`` exposes ``num_heads``, ``qkv``
(Linear, optional bias), ``scale``, ``attn_drop.p``, ``proj_drop.p`` and ``proj`` (Linear).  ``attn.dim`` is used when
present; stock timm ``Attention`` lacks it, so we fall back to ``qkv.in_features``.
"""
import json
import os

import torch

from .appla_attn import APLA_Attention
from .appla_attn_mem_eff import APLA_MemEffAttention


def _cfg_has(config, key):
    if isinstance(config, dict):
        return key in config
    return hasattr(config, key)


def _cfg_get(config, key):
    return config[key] if isinstance(config, dict) and not hasattr(config, key) else getattr(config, key)


def load_inds(path):
    """``{"block_i": [r ints]}`` JSON (params/**/inds-*.json; utils/helpfuns.py:69-73)."""
    with open(os.path.abspath(path), "r") as f:
        return json.load(f)


def indices_from_trainable(trainable, dim):
    """Trainable list followed by the ascending complement (apla_vit.py:21-24)."""
    chosen = set(int(t) for t in trainable)
    if len(chosen) != len(trainable) or any(not (0 <= t < dim) for t in chosen):
        raise ValueError("trainable indices must be distinct and in [0, dim)")
    return torch.tensor([int(t) for t in trainable] + [i for i in range(dim) if i not in chosen], dtype=torch.int64)


def replace_attn_with_apla(model, config, attn_module):
    inds_dict = load_inds(_cfg_get(config, "inds_path")) if _cfg_has(config, "inds_path") else None
    for i, block in enumerate(model.blocks):
        attn = block.attn
        dim = getattr(attn, "dim", None) or attn.qkv.in_features
        indices = indices_from_trainable(inds_dict[f"block_{i}"], dim) if inds_dict is not None else None
        new = attn_module(config=config, dim=dim, indices=indices, num_heads=attn.num_heads,
                          qkv_bias=attn.qkv.bias is not None, qk_scale=attn.scale, attn_drop=attn.attn_drop.p,
                          proj_drop=attn.proj_drop.p)
        with torch.no_grad():
            new.qkv.weight.data = attn.qkv.weight.data.clone()
            if attn.qkv.bias is not None:
                new.qkv.bias.data = attn.qkv.bias.data.clone()
            W = attn.proj.weight.data
            new.proj_weight1.data = W[new.trainable_inds, :].clone()   # rows = output features (apla_vit.py:51-52)
            new.proj_weight2.data = W[new.freezed_inds, :].clone()
            if attn.proj.bias is not None:
                b = attn.proj.bias.data
                new.proj_bias1.data = b[new.trainable_inds].clone()
                new.proj_bias2.data = b[new.freezed_inds].clone()
            else:  # reference leaves torch.empty garbage here (apla_vit.py:54-56); a missing bias is a zero bias
                new.proj_bias1.data.zero_()
                new.proj_bias2.data.zero_()
        new.to(attn.qkv.weight.device)
        block.attn = new


def build_apla(config, model, attn_class, is_multi_gpu=False):
    partial_size = _cfg_get(config, "partial_size")
    if is_multi_gpu:
        if partial_size == "full":  # full-rank projection tuning, no module swap (apla_vit.py:66-75)
            for name, p in model.named_parameters():
                p.requires_grad = "attn.proj" in name
            return model
        assert _cfg_has(config, "inds_path"), \
            '"inds_path" should be present with multi-gpu training with random sampling'
    for p in model.parameters():
        p.requires_grad = False
    if attn_class == "apla_attn":
        attn_module = APLA_Attention
    elif attn_class == "apla_attn_mem_eff":
        attn_module = APLA_MemEffAttention
    else:
        raise NotImplementedError(attn_class)
    replace_attn_with_apla(model=model, config=config, attn_module=attn_module)
    return model
