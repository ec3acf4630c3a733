"""Validation / test loop and kNN evaluation of the reference trainer (defaults/trainer.py:162-455) on the fused engine.

* ``Evaluator.evaluate(batches)`` — forward-only passes (``AplaTrainEngine.forward_only``: the step's own launch sequence
  without the backward), mean validation loss, the device-resident confusion-matrix meter (no per-batch D2H of logits:
  the reference copies every batch's logits to the host, utils/metrics.py:52-65), optionally kNN metrics.
* ``build_feature_bank`` / ``knn_predict`` — L2-normalised CLS features of the training set, cosine similarity against the
  bank, temperature-weighted vote of the k nearest neighbours (trainer.py:346-455).  The similarity product is a plain
  [B, D] x [D, N_bank] library GEMM + ``topk``; it stays in fp32 like the reference's ``torch.mm`` on normalised features.

Batches are ``(images [B,3,S,S], labels [B])`` on the GPU with the engine's batch size; a shorter last batch is padded by
the caller or dropped (``drop_last`` is true in every shipped loader configuration).
"""
from typing import Dict, Iterable, Optional, Tuple

import torch
import torch.nn.functional as F

from .data import ClassificationMeter


def knn_predict(feature: torch.Tensor, feature_bank: torch.Tensor, feature_labels: torch.Tensor, knn_k: int, knn_t: float,
                classes: int = 10, multi_label: bool = False) -> torch.Tensor:
    """trainer.py:392-455.  feature [B, D] (normalised), feature_bank [D, N], feature_labels [N] (class ids) or [C, N]
    (multi-label indicator rows); returns class scores [B, C]."""
    sim = torch.mm(feature, feature_bank)
    w, idx = sim.topk(k=knn_k, dim=-1)                                   # [B, k]
    w = (w / knn_t).exp()
    if multi_label:
        lab = feature_labels.to(w.dtype)                                 # [C, N]
        gathered = lab[:, idx]                                           # [C, B, k]
        wn = w / w.abs().sum(dim=-1, keepdim=True).clamp_min(1e-12)      # F.normalize(p=1) over the neighbours
        return (gathered * wn.unsqueeze(0)).sum(-1).t()
    lab = feature_labels.long()[idx]                                     # [B, k]
    scores = torch.zeros(feature.shape[0], classes, device=feature.device, dtype=w.dtype)
    scores.scatter_add_(1, lab, w)
    return scores / scores.sum(1, keepdim=True)


class Evaluator:
    def __init__(self, engine, n_classes: int, knn_nhood: int = 200, knn_t: float = 0.1):
        self.eng, self.C, self.k, self.t = engine, n_classes, knn_nhood, knn_t
        self.feature_bank: Optional[torch.Tensor] = None    # [D, N] as in the reference (features transposed)
        self.targets_bank: Optional[torch.Tensor] = None

    @torch.no_grad()
    def build_feature_bank(self, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]], process_group=None):
        feats, labs = [], []
        for images, labels in batches:
            _, f, _ = self.eng.forward_only(images)
            feats.append(F.normalize(f, dim=1).clone())
            labs.append(labels.clone())
        bank, tgt = torch.cat(feats).t().contiguous(), torch.cat(labs).t().contiguous()
        if process_group is not None and torch.distributed.get_world_size(process_group) > 1:   # dist_gather(cat_dim=-1)
            world = torch.distributed.get_world_size(process_group)
            fb, tb = [torch.empty_like(bank) for _ in range(world)], [torch.empty_like(tgt) for _ in range(world)]
            torch.distributed.all_gather(fb, bank, group=process_group)
            torch.distributed.all_gather(tb, tgt, group=process_group)
            bank, tgt = torch.cat(fb, dim=-1), torch.cat(tb, dim=-1)
        self.feature_bank, self.targets_bank = bank, tgt
        return bank.shape[1]

    @torch.no_grad()
    def evaluate(self, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]], mode: str = "val", knn: bool = False) -> Dict[str, float]:
        if knn and self.feature_bank is None:
            raise RuntimeError("kNN evaluation needs build_feature_bank() first (trainer.py:171-172)")
        dev = self.eng.device
        meter = ClassificationMeter(self.C, dev, keep_probs=True)      # (ROC-AUC up to 64 classes: data.py)
        knn_meter = ClassificationMeter(self.C, dev) if knn else None
        loss_sum, n = torch.zeros((), device=dev), 0
        for images, labels in batches:
            logits, f, loss = self.eng.forward_only(images, labels)
            loss_sum += loss.reshape(())
            n += 1
            meter.add_preds(logits, labels)
            if knn:
                k = min(self.k, self.feature_bank.shape[1])
                knn_meter.add_preds(knn_predict(F.normalize(f, dim=1), self.feature_bank, self.targets_bank, k, self.t, self.C), labels)
        out = {f"{mode}_{k}": v for k, v in meter.get_values().items()}
        out[f"{mode}_loss"] = float(loss_sum / max(n, 1))
        if knn:
            out.update({f"knn_{mode}_{k}": v for k, v in knn_meter.get_values().items()})
        return out
