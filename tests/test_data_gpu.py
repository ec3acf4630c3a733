"""Input-side kernel and device metrics: apla_augment_images against the oracle's restatement of ToTensor +
Normalize + flip + Mixup / CutMix; the device-side confusion matrix against the reference's numpy definition."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import apla_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hwc", [False, True])
@pytest.mark.parametrize("mode", ["plain", "flip", "mixup", "cutmix"])
def test_augment_images_matches_oracle(mode, hwc):
    from apla_amd import data
    B, S = 6, 32
    g = torch.Generator().manual_seed(3)
    src = torch.randint(0, 256, (B, 3, S, S), generator=g, dtype=torch.uint8)
    flip = (torch.rand(B, generator=g) < 0.5).to(torch.uint8) if mode != "plain" else None
    perm = torch.arange(B - 1, -1, -1, dtype=torch.int32) if mode in ("mixup", "cutmix") else None
    lam = torch.rand(B, generator=g) if mode == "mixup" else None
    box = torch.tensor([[2 + b, 20, 5, 31 - b] for b in range(B)], dtype=torch.int32) if mode == "cutmix" else None
    ref = O.augment_images(src, data.IMAGENET_MEAN, data.IMAGENET_STD, flip, perm, lam, box)
    dev_src = (src.permute(0, 2, 3, 1).contiguous() if hwc else src).cuda()
    c = lambda t: None if t is None else t.cuda()  # noqa: E731
    out = data.augment_images(dev_src, flip=c(flip), perm=c(perm), lam=c(lam), box=c(box))
    assert out.shape == (B, 3, S, S) and rel_err(out.cpu(), ref) < 1e-6


def test_device_augment_pipeline_and_targets():
    """DeviceAugment end to end: targets are proper probability rows, the mixed batch equals the oracle for the parameters
    that were drawn, and the result feeds the engine's soft-target step."""
    from apla_amd import data
    B, S, C = 8, 32, 10
    g = torch.Generator().manual_seed(5)
    src = torch.randint(0, 256, (B, 3, S, S), generator=g, dtype=torch.uint8)
    labels = torch.randint(0, C, (B,), generator=g)
    aug = data.DeviceAugment(C, hflip_p=0.5, mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, label_smoothing=0.1, seed=1)
    ref_aug = data.DeviceAugment(C, hflip_p=0.5, mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, label_smoothing=0.1, seed=1)
    for _ in range(4):  # a few draws: both Mixup and CutMix batches occur
        x, tgt = aug(src.cuda(), labels.cuda())
        flip, lam, box = ref_aug.sample(B, S)
        perm = torch.arange(B - 1, -1, -1, dtype=torch.int32)
        ref = O.augment_images(src, data.IMAGENET_MEAN, data.IMAGENET_STD, flip, perm,
                               None if box is not None else torch.full((B,), lam),
                               None if box is None else torch.tensor(box, dtype=torch.int32).repeat(B, 1))
        assert rel_err(x.cpu(), ref) < 1e-6
        assert tgt.shape == (B, C) and torch.allclose(tgt.sum(1).cpu(), torch.ones(B), atol=1e-6)
        onehot = torch.nn.functional.one_hot(labels, C).float() * 0.9 + 0.01
        assert torch.allclose(tgt.cpu(), onehot * lam + onehot.flip(0) * (1 - lam), atol=1e-6)
    assert not data.DeviceAugment(C).soft_targets


def test_classification_meter_matches_numpy_definition():
    from apla_amd.data import ClassificationMeter
    C = 7
    g = torch.Generator().manual_seed(2)
    meter = ClassificationMeter(C)
    cm = np.zeros((C, C))
    for _ in range(5):
        logits = torch.randn(33, C, generator=g)
        truths = torch.randint(0, C - 1, (33,), generator=g)   # class C-1 never occurs: its per-class accuracy counts as 0
        meter.add_preds(logits.cuda(), truths.cuda())
        np.add.at(cm, (truths.numpy(), logits.argmax(1).numpy()), 1)   # utils/metrics.py:65
    vals = meter.get_values()
    with np.errstate(divide="ignore", invalid="ignore"):
        per = cm.diagonal() / cm.sum(axis=1)
    per = np.nan_to_num(per, nan=0.0, posinf=0.0)
    assert abs(vals["accuracy"] - cm.diagonal().sum() / cm.sum()) < 1e-12
    assert abs(vals["mean_per_class_accuracy"] - per.mean()) < 1e-12
    assert int(meter.cm.sum()) == 0   # reset


@pytest.mark.gpu
def test_knn_and_meters_on_device_against_oracle_and_golden():
    """evaluate.knn_predict, ClassificationMeter and MultiLabelMeter with their state on the GPU: against golden G13 (the reference's
    own code) and, on fresh random inputs with ties and an absent class, against the oracle restatement (oracle/eval_oracle.py)."""
    import numpy as np
    from conftest import GOLDEN
    from oracle import eval_oracle as E
    from apla_amd.data import ClassificationMeter, MultiLabelMeter
    from apla_amd.evaluate import knn_predict
    dev = "cuda"
    d = np.load(os.path.join(GOLDEN, "g13_knn_metrics.npz"))
    k, temp, C = int(d["knn_k"]), float(d["knn_t"]), int(d["knn_classes"])
    f, bank = torch.tensor(d["knn_feature"], device=dev), torch.tensor(d["knn_bank"], device=dev)
    got = knn_predict(f, bank, torch.tensor(d["knn_labels"], device=dev), k, temp, C)
    assert np.abs(got.cpu().numpy() - d["knn_scores"]).max() < 2e-6
    lm = torch.tensor(d["knn_labels_multi"], device=dev)
    assert np.abs(knn_predict(f, bank, lm, k, temp, lm.shape[0], multi_label=True).cpu().numpy() - d["knn_scores_multi"]).max() < 2e-6
    m = ClassificationMeter(7, dev, keep_probs=True)
    lg, tr = torch.tensor(d["mc_logits"], device=dev), torch.tensor(d["mc_truths"], device=dev)
    m.add_preds(lg[:150], tr[:150]); m.add_preds(lg[150:], tr[150:])
    r = m.get_values()
    for key in ("accuracy", "mean_per_class_accuracy", "quadratic_kappa", "roc_auc", "recall"):
        assert abs(r[key] - float(d["mc_" + key])) <= 5.01e-4, (key, r[key])
    # fresh inputs: 9 classes of which one never occurs, coarse logits (ties), two batches
    g = torch.Generator().manual_seed(77)
    lg = (torch.randn(500, 9, generator=g) * 2).round() / 2
    tr = torch.randint(0, 8, (500,), generator=g)
    lg[torch.arange(500), tr] += 1.0
    m = ClassificationMeter(9, dev, keep_probs=True)
    m.add_preds(lg[:200].to(dev), tr[:200].to(dev)); m.add_preds(lg[200:].to(dev), tr[200:].to(dev))
    r, o = m.get_values(), E.classification_metrics(lg.numpy(), tr.numpy(), 9)
    for key in ("accuracy", "mean_per_class_accuracy", "quadratic_kappa", "roc_auc", "recall"):
        assert abs(r[key] - o[key]) < 1e-6, (key, r[key], o[key])
    lg = (torch.randn(400, 6, generator=g) * 2).round() / 2
    tr = (torch.rand(400, 6, generator=g) < 0.3).float()
    tr[:, 5] = 0                                           # a class without positives: ROC-AUC 0.5, precision / recall / f1 0
    mm = MultiLabelMeter(6, dev)
    mm.add_preds(lg[:100].to(dev), tr[:100].to(dev)); mm.add_preds(lg[100:].to(dev), tr[100:].to(dev))
    r, o = mm.get_values(), E.multilabel_metrics(lg.numpy(), tr.numpy())
    for key in ("accuracy", "precision", "recall", "f1", "roc_auc"):
        assert abs(r[key] - o[key]) < 1e-6, (key, r[key], o[key])
