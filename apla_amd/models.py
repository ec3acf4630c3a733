"""Classifier = ViT backbone (+APLA) + trainable linear head (reference surface: defaults/models.py:19-92).

``model_params`` / ``system_params`` follow the reference's YAML schema (params/**/__common__.yml):
``backbone_type`` ("vit_small" …), ``transformers_params`` (img_size, patch_size, pretrained_type, block_conf …),
``pretrained``, ``n_classes``, ``freeze_backbone``, ``adaptation: {mode: "apla", params: {partial_size, inds_path?}}``;
``system_params.which_GPUs`` decides ``is_multi_gpu`` exactly as defaults/models.py:53 does.
"""
import torch.nn as nn

from . import vit
from .apla import build_apla


class AttrDict(dict):
    """EasyDict stand-in (attribute access, AttributeError on missing keys, nested dicts converted)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


class Classifier(nn.Module):
    def __init__(self, model_params, system_params, backbone_state_dict=None):
        """``backbone_state_dict``: an UNSPLIT pretrained backbone (dinov2 / timm key names, ``attn.proj.*`` still whole) to load
        into the freshly built ViT BEFORE ``build_apla`` splits the projection — what the reference's ``pretrained: true`` does
        inside the ViT factory (utils/transformers/vit.py:545-560 -> transformers_utils.download_weights) before
        defaults/models.py:49-54 calls build_apla.  Loaded strictly: a key mismatch raises."""
        super().__init__()
        mp = AttrDict(model_params)
        sp = AttrDict(system_params)
        self.backbone_type = mp.backbone_type
        self.n_classes = mp.n_classes
        self.freeze_backbone = mp.get("freeze_backbone", False)
        self.use_mixed_precision = False
        tp = dict(mp.get("transformers_params", {}))
        if "adaptation" not in mp:
            raise NotImplementedError("only adaptation.mode == 'apla' is implemented on the MI355X path")
        assert mp.adaptation.mode == "apla", "Only adaptation with APLA is enabled"
        assert "vit" in self.backbone_type, "Only supports ViT with multi-gpu training"
        if not hasattr(vit, self.backbone_type):
            raise ValueError(f"unknown backbone_type {self.backbone_type}")
        model = getattr(vit, self.backbone_type)(pretrained=mp.get("pretrained", False), **tp)
        if backbone_state_dict is not None:
            from .checkpoint import load_pretrained_backbone
            load_pretrained_backbone(model, backbone_state_dict, tp.get("pretrained_type", "dinov2"), strict=True)
        self.backbone = build_apla(config=mp.adaptation.params, model=model, attn_class="apla_attn",
                                   is_multi_gpu=len(str(sp.get("which_GPUs", "0")).split(",")) > 1)
        self.backbone.fc = nn.Identity()
        self.fc = nn.Linear(self.backbone.num_features, self.n_classes)  # created after the freeze -> trainable
        if self.freeze_backbone:
            for p in self.backbone.parameters():
                p.requires_grad = False

    def forward(self, x, return_embedding=False):
        emb = self.backbone(x)
        out = self.fc(emb.float())
        return (out, emb) if return_embedding else out


def get_params_groups(model):
    """Two AdamW groups (defaults/wrappers.py:205-221): weight decay on >=2-D non-bias tensors, none on the rest."""
    reg, no_reg = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_reg if name.endswith(".bias") or p.ndim == 1 else reg).append(p)
    return [{"params": reg}, {"params": no_reg, "weight_decay": 0.0}]
