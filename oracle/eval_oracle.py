"""CPU oracle for the evaluation side of the trainer.  TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Restates, in plain numpy, the kNN vote of ``Trainer.knn_predict`` (defaults/trainer.py:392-455) and the numbers the reference's
metric objects report (utils/metrics.py:38-189: ``ClassificationMetrics.get_values``, ``MultiLabelClassificationMetrics.get_value``,
``mean_roc_auc``), from the definitions of the sklearn functions the reference calls — sklearn itself is not imported here.

Parity pin: golden G13 (tests/golden/g13_knn_metrics.npz) holds the outputs of the reference's own ``knn_predict`` source and of
its metric classes (loaded in place, tests/golden/make_golden.py:g13_knn_metrics) on seeded inputs; tests/test_oracle_golden.py
checks every function below against it.  The reference rounds its metric values to three decimals: the metrics are pinned to
+-5e-4, the kNN scores to 1e-6.
"""
from __future__ import annotations

from typing import Dict

import numpy as np


# ---------------------------------------------------------------------------------------------- kNN vote (trainer.py:392-455)
def knn_predict(feature: np.ndarray, feature_bank: np.ndarray, feature_labels: np.ndarray, knn_k: int, knn_t: float,
                classes: int = 10, multi_label: bool = False) -> np.ndarray:
    """feature [B, D] (L2-normalised), feature_bank [D, N], feature_labels [N] class ids or [C, N] indicator rows.
    Single-label (:437-455): scores[b, c] = sum over the k nearest neighbours with label c of exp(sim / t), rows normalised to sum 1.
    Multi-label (:409-435): scores[b, c] = sum_k label[c, nb_k] * exp(sim_k / t) / sum_k exp(sim_k / t)."""
    feature, feature_bank = np.asarray(feature, np.float64), np.asarray(feature_bank, np.float64)
    sim = feature @ feature_bank                                   # :411 / :438
    idx = np.argsort(-sim, axis=1, kind="stable")[:, :knn_k]       # topk (:413 / :440)
    w = np.exp(np.take_along_axis(sim, idx, axis=1) / knn_t)       # :428 / :445
    if multi_label:
        lab = np.asarray(feature_labels, np.float64)               # [C, N]
        wn = w / np.abs(w).sum(axis=1, keepdims=True)              # F.normalize(p=1, dim=2) (:431)
        return np.einsum("cbk,bk->bc", lab[:, idx], wn)            # gather + weighted sum (:425, :431-432)
    lab = np.asarray(feature_labels, np.int64)[idx]                # :442
    scores = np.zeros((feature.shape[0], classes))
    for b in range(feature.shape[0]):
        np.add.at(scores[b], lab[b], w[b])                         # one-hot * weight, summed over neighbours (:448-452)
    return scores / scores.sum(axis=1, keepdims=True)              # :454


# ---------------------------------------------------------------------------------------------- ranking metrics
def _auc(y: np.ndarray, s: np.ndarray) -> float:
    """Area under the ROC curve of scores s for binary truths y; ties count one half (the trapezoid rule sklearn applies)."""
    y = np.asarray(y).astype(bool)
    s = np.asarray(s, np.float64)
    n1, n0 = int(y.sum()), int((~y).sum())
    order = np.argsort(s, kind="stable")
    ranks = np.empty(len(s))
    sv = s[order]
    i = 0
    while i < len(sv):                                             # average ranks over ties
        j = i
        while j + 1 < len(sv) and sv[j + 1] == sv[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[y].sum() - n1 * (n1 + 1) / 2.0) / (n1 * n0))


def _average_precision(y: np.ndarray, s: np.ndarray) -> float:
    """sklearn.metrics.average_precision_score for one class: sum over thresholds of (R_n - R_{n-1}) P_n, one threshold per
    distinct score, taken from the highest score down."""
    y = np.asarray(y).astype(np.float64)
    order = np.argsort(-np.asarray(s, np.float64), kind="stable")
    ys, ss = y[order], np.asarray(s, np.float64)[order]
    last = np.r_[np.nonzero(np.diff(ss))[0], len(ss) - 1]          # last index of every run of equal scores
    tp = np.cumsum(ys)[last]
    prec = tp / (last + 1.0)
    rec = tp / ys.sum()
    return float(np.sum(np.diff(np.r_[0.0, rec]) * prec))


def softmax(x: np.ndarray) -> np.ndarray:
    e = np.exp(x - x.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


# ---------------------------------------------------------------------------------------------- ClassificationMetrics (metrics.py:38-112)
def classification_metrics(logits: np.ndarray, truths: np.ndarray, n_classes: int) -> Dict[str, float]:
    """What ``ClassificationMetrics(raw=True).get_values()`` reports for all the predictions added (:74-112), unrounded."""
    logits, truths = np.asarray(logits, np.float64), np.asarray(truths, np.int64)
    prob = softmax(logits)                                          # act_fn (:43)
    preds = prob.argmax(axis=1)                                     # :60
    cm = np.zeros((n_classes, n_classes))
    np.add.at(cm, (truths, preds), 1)                               # :64, row: truth, column: prediction
    out = {"accuracy": float((preds == truths).mean())}             # accuracy_score (:85)
    with np.errstate(divide="ignore", invalid="ignore"):
        per = cm.diagonal() / cm.sum(axis=1)
    out["mean_per_class_accuracy"] = float(np.mean(np.where(np.isfinite(per), per, 0.0)))   # :67-72
    labels = np.union1d(truths, preds)                              # sklearn works on the labels that occur
    sub = cm[np.ix_(labels, labels)]
    if n_classes > 2:                                               # cohen_kappa_score(weights='quadratic') (:87-88)
        n = len(labels)
        w = (np.arange(n)[:, None] - np.arange(n)[None, :]) ** 2.0
        expected = np.outer(sub.sum(axis=1), sub.sum(axis=0)) / sub.sum()
        out["quadratic_kappa"] = float(1.0 - (w * sub).sum() / (w * expected).sum())
    else:
        out["quadratic_kappa"] = 0.0                                # :89-90
    with np.errstate(divide="ignore", invalid="ignore"):           # recall_score(average='macro', zero_division=0) (:92)
        rec = sub.diagonal() / sub.sum(axis=1)
    out["recall"] = float(np.mean(np.where(np.isfinite(rec), rec, 0.0)))
    if len(np.unique(truths)) < n_classes:                          # roc_auc_score raises -> 0.5 (:93-98)
        out["roc_auc"] = 0.5
    elif n_classes == 2:                                            # the positive-class probability (:54-55)
        out["roc_auc"] = _auc(truths == 1, prob[:, 1])
    else:                                                           # multi_class='ovo', average='macro' (Hand & Till)
        tot, pairs = 0.0, 0
        for a in range(n_classes):
            for b in range(a + 1, n_classes):
                m = (truths == a) | (truths == b)
                tot += 0.5 * (_auc(truths[m] == a, prob[m, a]) + _auc(truths[m] == b, prob[m, b]))
                pairs += 1
        out["roc_auc"] = tot / pairs
    out["confusion_matrix"] = cm
    return out


# ---------------------------------------------------------------------------------------------- MultiLabelClassificationMetrics (metrics.py:115-189)
def multilabel_metrics(logits: np.ndarray, truths: np.ndarray, threshold: float = 0.5) -> Dict[str, float]:
    """What ``MultiLabelClassificationMetrics.get_value()`` reports (:156-189), unrounded: scores = sigmoid(logits) (:144-146);
    mAP and mean ROC-AUC on the scores (``mean_roc_auc`` :17-35: its sample weights are constant inside each of the two classes
    and cancel; a class without positives counts 0.5), the rest on the thresholded predictions (:148-154)."""
    truths = np.asarray(truths, np.float64)
    score = 1.0 / (1.0 + np.exp(-np.asarray(logits, np.float64)))
    C = truths.shape[1]
    out = {"mAP": float(np.mean([_average_precision(truths[:, c], score[:, c]) for c in range(C)])),
           "roc_auc": float(np.mean([_auc(truths[:, c] > 0, score[:, c]) if truths[:, c].sum() > 0 else 0.5 for c in range(C)]))}
    pred = (score > threshold).astype(np.float64)
    out["accuracy"] = float(np.mean((pred == truths).all(axis=1)))   # accuracy_score on indicator rows: exact match
    tp = (pred * truths).sum(axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        p = tp / pred.sum(axis=0)
        r = tp / truths.sum(axis=0)
        f = 2 * tp / (pred.sum(axis=0) + truths.sum(axis=0))
    z = lambda v: np.where(np.isfinite(v), v, 0.0)                   # zero_division=0
    out["precision"], out["recall"], out["f1"] = float(z(p).mean()), float(z(r).mean()), float(z(f).mean())
    return out
