"""Step-adjacent input-side work on the device.

* ``DeviceAugment`` — what the reference does on CPU workers per sample (ToTensor, Normalize, horizontal flip: defaults/
  bases.py:69-231) and per batch in the collate function (timm ``Mixup``: Mixup / CutMix / label smoothing,
  utils/_utils.py:424-441), applied to a batch of decoded uint8 images that already sits in device memory: ONE kernel
  (``apla_augment_images``) writes the normalised fp32 batch the patch embedding reads, and the probability targets go to
  the soft-target cross-entropy.  The sampling follows timm's ``Mixup(mode='batch')``: with probability ``prob`` the batch
  is mixed; CutMix is chosen with probability ``switch_prob`` when both alphas are positive; one lambda ~ Beta(alpha, alpha)
  per batch; the partner of sample i is sample B-1-i (``x.flip(0)``); CutMix pastes a box of area (1 - lambda) at a uniform
  centre and corrects lambda to the pasted area.  timm is not a dependency and its generator cannot be reproduced, so this
  is pinned semantically (tests against the numpy restatement in oracle/), not bit for bit.
* ``ClassificationMeter`` — the reference's ``ClassificationMetrics`` (utils/metrics.py:38-110) keeps python lists and does a
  device-to-host copy of the logits every step; here the confusion matrix is accumulated on the device and only read
  when a value is asked for.
"""
import ctypes
import math
from typing import Optional, Tuple

import torch

from . import ops
from ._lib import check, lib

IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def augment_images(src: torch.Tensor, *, mean=IMAGENET_MEAN, std=IMAGENET_STD, flip: Optional[torch.Tensor] = None,
                   perm: Optional[torch.Tensor] = None, lam: Optional[torch.Tensor] = None, box: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """src: uint8 [B,3,S,S] or [B,S,S,3] on the GPU -> fp32 [B,3,S,S]; see include/apla_hip.h:apla_augment_images."""
    ops._req(src, torch.uint8, "src", 4)
    if not src.is_contiguous():
        raise ValueError("augment_images: src must be contiguous")
    hwc = src.shape[-1] == 3 and src.shape[1] != 3
    B, S = src.shape[0], src.shape[2] if not hwc else src.shape[1]
    if (src.shape[1:] != (3, S, S) and not hwc) or (hwc and src.shape[1:] != (S, S, 3)) or S % 4 != 0:
        raise ValueError(f"augment_images: expected [B,3,S,S] or [B,S,S,3] with S % 4 == 0, got {tuple(src.shape)}")
    if out is None:
        out = torch.empty(B, 3, S, S, device=src.device, dtype=torch.float32)
    ops._req(out, torch.float32, "out", 4)
    if tuple(out.shape) != (B, 3, S, S) or not out.is_contiguous():
        raise ValueError("augment_images: bad out buffer")
    for t_, dt, nm, shape in ((flip, torch.uint8, "flip", (B,)), (perm, torch.int32, "perm", (B,)), (lam, torch.float32, "lam", (B,)),
                              (box, torch.int32, "box", (B, 4))):
        if t_ is not None:
            ops._req(t_, dt, nm)
            if tuple(t_.shape) != shape or not t_.is_contiguous():
                raise ValueError(f"augment_images: {nm} must be contiguous {shape}")
    if (lam is not None or box is not None) and perm is None:
        raise ValueError("augment_images: lam / box need perm")
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    check(lib().apla_augment_images(src.data_ptr(), out.data_ptr(), m3, s3, ops._ptr(flip), ops._ptr(perm), ops._ptr(lam),
                                    ops._ptr(box), B, S, int(hwc), ops._stream()), "apla_augment_images")
    return out


class DeviceAugment:
    def __init__(self, n_classes: int, *, hflip_p: float = 0.0, mixup_alpha: float = 0.0, cutmix_alpha: float = 0.0, prob: float = 1.0,
                 switch_prob: float = 0.5, label_smoothing: float = 0.0, mean=IMAGENET_MEAN, std=IMAGENET_STD, seed: int = 0):
        self.C, self.hflip_p = n_classes, hflip_p
        self.mixup_alpha, self.cutmix_alpha, self.prob, self.switch_prob = mixup_alpha, cutmix_alpha, prob, switch_prob
        self.smoothing, self.mean, self.std = label_smoothing, mean, std
        self.gen = torch.Generator().manual_seed(seed)  # host generator: a handful of scalars per batch

    @property
    def soft_targets(self) -> bool:
        return self.mixup_alpha > 0 or self.cutmix_alpha > 0 or self.smoothing > 0

    def _beta(self, a: float) -> float:
        return float(torch._sample_dirichlet(torch.tensor([a, a], dtype=torch.float64), generator=self.gen)[0])  # Beta(a, a)

    def sample(self, B: int, S: int):
        """Host-side draw of the batch's parameters -> (flip [B] uint8 | None, lam float, box (y0,y1,x0,x1) | None)."""
        flip = (torch.rand(B, generator=self.gen) < self.hflip_p).to(torch.uint8) if self.hflip_p > 0 else None
        lam, box = 1.0, None
        mix_on = (self.mixup_alpha > 0 or self.cutmix_alpha > 0) and float(torch.rand((), generator=self.gen)) < self.prob
        if mix_on:
            use_cutmix = self.cutmix_alpha > 0 and (self.mixup_alpha <= 0 or float(torch.rand((), generator=self.gen)) < self.switch_prob)
            lam = self._beta(self.cutmix_alpha if use_cutmix else self.mixup_alpha)
            if use_cutmix:  # timm rand_bbox: cut ratio sqrt(1 - lam), uniform centre, clipped; lambda corrected to the pasted area
                ratio = math.sqrt(1.0 - lam)
                ch, cw = int(S * ratio), int(S * ratio)
                cy, cx = int(torch.randint(0, S, (), generator=self.gen)), int(torch.randint(0, S, (), generator=self.gen))
                y0, y1 = max(cy - ch // 2, 0), min(cy + ch // 2, S)
                x0, x1 = max(cx - cw // 2, 0), min(cx + cw // 2, S)
                box = (y0, y1, x0, x1)
                lam = 1.0 - (y1 - y0) * (x1 - x0) / float(S * S)
        return flip, lam, box

    def targets(self, labels: torch.Tensor, lam: float) -> torch.Tensor:
        """timm mixup_target: smoothed one-hot of the labels mixed with that of the reversed batch."""
        off, on = self.smoothing / self.C, 1.0 - self.smoothing + self.smoothing / self.C
        y = torch.full((labels.numel(), self.C), off, device=labels.device, dtype=torch.float32)
        y.scatter_(1, labels.long().view(-1, 1), on)
        return y if lam == 1.0 else y * lam + y.flip(0) * (1.0 - lam)

    def __call__(self, images_u8: torch.Tensor, labels: torch.Tensor, out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """uint8 batch + class ids on the device -> (fp32 normalised batch, targets): class ids unchanged when nothing
        needs probability targets, else [B, C] probabilities."""
        B = images_u8.shape[0]
        S = images_u8.shape[2] if images_u8.shape[1] == 3 else images_u8.shape[1]
        flip, lam, box = self.sample(B, S)
        dev = images_u8.device
        mixed = lam != 1.0 or box is not None
        perm = torch.arange(B - 1, -1, -1, device=dev, dtype=torch.int32) if mixed else None
        x = augment_images(images_u8, mean=self.mean, std=self.std, flip=None if flip is None else flip.to(dev), perm=perm,
                           lam=torch.full((B,), lam, device=dev) if mixed and box is None else None,
                           box=torch.tensor(box, dtype=torch.int32, device=dev).repeat(B, 1).contiguous() if box is not None else None,
                           out=out)
        return x, (self.targets(labels, lam) if self.soft_targets else labels)


def _auc(pos: torch.Tensor, score: torch.Tensor) -> torch.Tensor:
    """Area under the ROC curve of `score` for the boolean truths `pos` (ties count one half), on the tensors' device."""
    s = score.double()
    srt = torch.sort(s).values
    rank = 0.5 * (torch.searchsorted(srt, s, right=False) + torch.searchsorted(srt, s, right=True) + 1).double()   # average rank
    n1 = pos.sum().double()
    n0 = pos.numel() - n1
    return (rank[pos].sum() - n1 * (n1 + 1) / 2) / (n1 * n0)


def _average_precision(truth: torch.Tensor, score: torch.Tensor) -> torch.Tensor:
    """sum over thresholds of (R_n - R_{n-1}) P_n, one threshold per distinct score from the highest down (sklearn's definition)."""
    order = torch.argsort(score.double(), descending=True, stable=True)
    ys, ss = truth.double()[order], score.double()[order]
    last = torch.ones_like(ss, dtype=torch.bool)
    last[:-1] = ss[1:] != ss[:-1]                                   # last element of every run of equal scores
    tp = torch.cumsum(ys, 0)[last]
    n = (torch.nonzero(last).squeeze(1) + 1).double()
    rec = tp / ys.sum()
    return ((rec - torch.cat([rec.new_zeros(1), rec[:-1]])) * (tp / n)).sum()


class ClassificationMeter:
    """The numbers of the reference's ``ClassificationMetrics`` (utils/metrics.py:38-112) from state kept on the device: a
    confusion matrix (row: truth, column: prediction) and, with ``keep_probs``, the softmax rows for the ROC-AUC — the reference
    copies every batch's logits to the host (:52-65); here nothing leaves the device before ``get_values``.
    accuracy, mean_per_class_accuracy (:67-72), quadratic_kappa (Cohen's kappa, quadratic weights, 0 for two classes: :87-90),
    recall (macro over the labels that occur, :92) and roc_auc (one-vs-one macro, 0.5 when a class has no sample: :93-98; computed
    for up to 64 classes, else not reported).  Values are not rounded (the reference rounds to three decimals)."""

    def __init__(self, n_classes: int, device="cuda", keep_probs: bool = False):
        self.C = n_classes
        self.cm = torch.zeros(n_classes, n_classes, device=device, dtype=torch.int64)
        self.keep_probs = keep_probs and n_classes <= 64
        self._probs, self._truths = [], []

    def reset(self):
        self.cm.zero_()
        self._probs, self._truths = [], []

    def add_preds(self, logits: torch.Tensor, truths: torch.Tensor):
        preds = logits.argmax(1)
        self.cm.view(-1).index_add_(0, truths.long() * self.C + preds, torch.ones_like(preds, dtype=torch.int64))
        if self.keep_probs:
            self._probs.append(torch.softmax(logits.float(), dim=1))
            self._truths.append(truths.long().clone())

    def _roc_auc(self, process_group=None) -> float:
        prob, truth = torch.cat(self._probs), torch.cat(self._truths)
        if process_group is not None and torch.distributed.get_world_size(process_group) > 1:
            world = torch.distributed.get_world_size(process_group)
            pl, tl = [torch.empty_like(prob) for _ in range(world)], [torch.empty_like(truth) for _ in range(world)]
            torch.distributed.all_gather(pl, prob, group=process_group)
            torch.distributed.all_gather(tl, truth, group=process_group)
            prob, truth = torch.cat(pl), torch.cat(tl)
        if int(torch.unique(truth).numel()) < self.C:
            return 0.5
        if self.C == 2:
            return float(_auc(truth == 1, prob[:, 1]))
        tot, pairs = prob.new_zeros((), dtype=torch.float64), 0
        for a in range(self.C):
            for b in range(a + 1, self.C):
                m = (truth == a) | (truth == b)
                tot += 0.5 * (_auc(truth[m] == a, prob[m, a]) + _auc(truth[m] == b, prob[m, b]))
                pairs += 1
        return float(tot / pairs)

    def get_values(self, process_group=None, do_reset: bool = True) -> dict:
        cm = self.cm.clone()
        if process_group is not None:
            torch.distributed.all_reduce(cm, group=process_group)
        cmf = cm.double()
        total = cmf.sum().clamp_min(1)
        per_class = cmf.diagonal() / cmf.sum(1)                     # nan for absent classes
        per_class = torch.nan_to_num(per_class, nan=0.0, posinf=0.0)  # utils/metrics.py:67-72 fills invalid entries with 0
        out = {"accuracy": float(cmf.diagonal().sum() / total), "mean_per_class_accuracy": float(per_class.mean())}
        present = (cmf.sum(0) + cmf.sum(1)) > 0                     # sklearn works on the labels that occur in truths or predictions
        sub = cmf[present][:, present]
        n = int(present.sum())
        if self.C > 2 and n > 1:
            idx = torch.arange(n, device=cm.device, dtype=torch.float64)
            w = (idx[:, None] - idx[None, :]) ** 2
            expected = torch.outer(sub.sum(1), sub.sum(0)) / sub.sum().clamp_min(1)
            out["quadratic_kappa"] = float(1.0 - (w * sub).sum() / (w * expected).sum())
        else:
            out["quadratic_kappa"] = 0.0
        rec = torch.nan_to_num(sub.diagonal() / sub.sum(1), nan=0.0, posinf=0.0)
        out["recall"] = float(rec.mean()) if n else 0.0
        if self.keep_probs and self._probs:
            out["roc_auc"] = self._roc_auc(process_group)
        if do_reset:
            self.reset()
        return out


class MultiLabelMeter:
    """The numbers of the reference's ``MultiLabelClassificationMetrics`` (utils/metrics.py:115-189) from scores kept on the device:
    sigmoid scores and indicator truths of every batch; ``get_values``: subset accuracy, macro precision / recall / f1 of the
    thresholded predictions (zero_division = 0), mAP and the mean per-class ROC-AUC of the scores (``mean_roc_auc`` :17-35: a class
    without positives counts 0.5).  Unrounded."""

    def __init__(self, n_classes: int, device="cuda", act_threshold: float = 0.5):
        self.C, self.thr = n_classes, act_threshold
        self._scores, self._truths = [], []

    def reset(self):
        self._scores, self._truths = [], []

    def add_preds(self, logits: torch.Tensor, truths: torch.Tensor, using_knn: bool = False):
        self._scores.append(torch.sigmoid(logits.float()))          # (the reference applies the sigmoid to kNN scores too: :137-146)
        self._truths.append(truths.float().clone())

    def get_values(self, process_group=None, do_reset: bool = True) -> dict:
        score, truth = torch.cat(self._scores), torch.cat(self._truths)
        if process_group is not None and torch.distributed.get_world_size(process_group) > 1:
            world = torch.distributed.get_world_size(process_group)
            sl, tl = [torch.empty_like(score) for _ in range(world)], [torch.empty_like(truth) for _ in range(world)]
            torch.distributed.all_gather(sl, score, group=process_group)
            torch.distributed.all_gather(tl, truth, group=process_group)
            score, truth = torch.cat(sl), torch.cat(tl)
        C = truth.shape[1]
        ap = torch.stack([_average_precision(truth[:, c], score[:, c]) for c in range(C)])
        auc = torch.stack([_auc(truth[:, c] > 0, score[:, c]) if bool(truth[:, c].sum() > 0) else score.new_tensor(0.5, dtype=torch.float64)
                           for c in range(C)])
        pred = (score > self.thr).double()
        t = truth.double()
        tp = (pred * t).sum(0)
        z = lambda v: torch.nan_to_num(v, nan=0.0, posinf=0.0)
        out = {"accuracy": float((pred == t).all(1).double().mean()), "mAP": float(ap.mean()), "roc_auc": float(auc.mean()),
               "precision": float(z(tp / pred.sum(0)).mean()), "recall": float(z(tp / t.sum(0)).mean()),
               "f1": float(z(2 * tp / (pred.sum(0) + t.sum(0))).mean())}
        if do_reset:
            self.reset()
        return out
