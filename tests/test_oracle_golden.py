"""Pin the CPU oracle (oracle/apla_oracle.py) to vectors produced by the ACTUAL reference code
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, rel_err, t
from oracle import apla_oracle as O

TOL = 2e-5  # fp32 oracle vs fp32 reference: differences are summation order only


def test_g1_index_selection_bit_exact():
    g = load_golden("g1_indices.npz")
    for seed in (0, 7, 123):
        for D in (384, 768, 1024, 1536):
            torch.manual_seed(seed)
            got = O.sample_indices(D)
            assert torch.equal(got, t(g[f"seed{seed}_D{D}"])), (seed, D)
    # SURVEY §4 item 4 known answers
    torch.manual_seed(7)
    assert O.sample_indices(768)[:8].tolist() == [687, 650, 63, 284, 415, 268, 77, 275]


def test_g2_weight_split_random_path():
    g = load_golden("g2_g8_split.npz")
    for i in range(2):
        inds = t(g[f"rand_inds{i}"])
        W1, W2, b1, b2 = O.split_proj(t(g[f"rand_full_w{i}"]), t(g[f"rand_full_b{i}"]), inds, 8)
        assert torch.equal(W1, t(g[f"rand_w1_{i}"])) and torch.equal(W2, t(g[f"rand_w2_{i}"]))
        assert torch.equal(b1, t(g[f"rand_b1_{i}"])) and torch.equal(b2, t(g[f"rand_b2_{i}"]))
        Wm, bm = O.merge_proj(W1, W2, b1, b2, inds)
        assert torch.equal(Wm, t(g[f"rand_full_w{i}"])) and torch.equal(bm, t(g[f"rand_full_b{i}"]))


def test_g8_indices_from_json_ascending_complement():
    g = load_golden("g2_g8_split.npz")
    for i in range(2):
        inds = O.indices_from_trainable(g[f"json_trainable{i}"].tolist(), 64)
        assert torch.equal(inds, t(g[f"json_inds{i}"]))
        W1, _, _, _ = O.split_proj(t(g[f"json_full_w{i}"]), None, inds, 8)
        assert torch.equal(W1, t(g[f"json_w1_{i}"]))


def _params(g, tag, dtype):
    pre = tag + "."
    return {k[len(pre):]: t(g[k], dtype) for k in g.files if k.startswith(pre)}


@pytest.mark.parametrize("tag", ["tiny", "mid"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_g3_module_fwd_bwd(tag, dtype):
    g = load_golden("g3_module.npz")
    p = _params(g, tag, dtype)
    B, N, D, H, r = [int(v) for v in g[f"{tag}.meta"]]
    x = p["x"]
    y, attn, ctx = O.apla_attention_fwd(x, p, "", H, r, return_attn=True)
    assert rel_err(y, p["y"]) < TOL
    assert rel_err(attn, p["attn"]) < TOL
    dy = 2.0 * y / y.numel()  # d/dy of y.square().mean()
    dx, dW1, db1 = O.apla_attention_bwd(dy, ctx, p, "", H)
    assert rel_err(dx, p["dx"]) < 5 * TOL
    assert rel_err(dW1, p["dW1"]) < 5 * TOL
    assert rel_err(db1, p["db1"]) < 5 * TOL


@pytest.mark.parametrize("tag", ["gelu", "swiglu"])
def test_g4_block_fwd_bwd(tag):
    g = load_golden("g4_block.npz")
    p = _params(g, tag, torch.float64)
    B, N, D, H, r, swiglu = [int(v) for v in g[f"{tag}.meta"]]
    out, ctx = O.block_fwd(p["x"], p, 0, H, r, swiglu=bool(swiglu))
    assert rel_err(out, p["out"]) < TOL
    dout = 2.0 * out / out.numel()
    dx, dW1, db1 = O.block_bwd(dout, ctx, p, 0, H, swiglu=bool(swiglu))
    assert rel_err(dx, p["dx"]) < 5 * TOL
    assert rel_err(dW1, p["dW1"]) < 5 * TOL
    assert rel_err(db1, p["db1"]) < 5 * TOL


def test_g5_tiny_model_full_step():
    g = load_golden("g5_tiny_model.npz")
    D, L, H, r, C, patch = [int(v) for v in g["meta"]]
    p = {k[2:]: t(g[k]) for k in g.files if k.startswith("p.")}
    cfg = dict(patch=patch, depth=L, heads=H, r=r)
    images, labels = t(g["images"]), t(g["labels"])
    logits, ctx = O.vit_forward(images, p, cfg)
    assert rel_err(logits, g["logits"]) < TOL
    loss, dlogits = O.cross_entropy_fwd_bwd(logits, labels)
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    grads = O.vit_backward(dlogits, ctx, p, cfg)
    assert sorted(grads) == sorted(O.trainable_names(L))
    for n, gr in grads.items():
        assert rel_err(gr, g["g." + n]) < 1e-4, n
    gnorm = O.clip_grad_norm(grads, 1.0)
    assert abs(float(gnorm) - float(g["gnorm"])) < 1e-5 * max(1.0, float(g["gnorm"]))
    O.adamw_step(p, grads, {}, lr=1e-4, wd=1e-5)
    for n in grads:
        assert rel_err(p[n], g["after." + n]) < 1e-6, n


def test_swap_invariance_and_grad_slice_identity():
    """SURVEY §4 items 1-2 as properties of the oracle (no fixture needed)."""
    torch.manual_seed(3)
    D, H, r, B, N = 64, 2, 8, 2, 9
    W, b = torch.randn(D, D, dtype=torch.float64) * 0.1, torch.randn(D, dtype=torch.float64) * 0.1
    inds = O.sample_indices(D)
    W1, W2, b1, b2 = O.split_proj(W, b, inds, r)
    x = torch.randn(B, N, D, dtype=torch.float64)
    y = O.apla_proj_fwd(x, W1, b1, W2, b2, inds)
    assert torch.equal(y, O.apla_proj_fwd(x, W1, b1, W2, b2, inds))
    assert rel_err(y, x @ W.t() + b) < 1e-14
    dy = torch.randn_like(y)
    dx, dW1, db1 = O.apla_proj_bwd(dy, x, W1, W2, inds)
    full_dW = dy.reshape(-1, D).t() @ x.reshape(-1, D)
    assert rel_err(dW1, full_dW[inds[:r]]) < 1e-14
    assert rel_err(db1, dy.reshape(-1, D).sum(0)[inds[:r]]) < 1e-14
    assert rel_err(dx, dy @ W) < 1e-13


def test_g7_data_parallel_mean_semantics():
    """DDP averaging (defaults/wrappers.py:183): mean of per-shard grads == full-batch grad."""
    g = load_golden("g5_tiny_model.npz")
    D, L, H, r, C, patch = [int(v) for v in g["meta"]]
    p = {k[2:]: t(g[k], torch.float64) for k in g.files if k.startswith("p.")}
    cfg = dict(patch=patch, depth=L, heads=H, r=r)
    gen = torch.Generator().manual_seed(0)
    images = torch.randn(4, 3, 48, 48, generator=gen, dtype=torch.float64)
    labels = torch.randint(0, C, (4,), generator=gen)

    def grads_of(im, lb):
        logits, ctx = O.vit_forward(im, p, cfg)
        _, dl = O.cross_entropy_fwd_bwd(logits, lb)
        return O.vit_backward(dl, ctx, p, cfg)

    full = grads_of(images, labels)
    a, b = grads_of(images[:2], labels[:2]), grads_of(images[2:], labels[2:])
    for n in full:
        assert rel_err((a[n] + b[n]) / 2, full[n]) < 1e-12, n


def test_oracle_soft_cross_entropy_matches_torch():
    """nn.CrossEntropyLoss with probability targets (what the reference's criterion computes under advanced_aug)."""
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(9, 17, generator=g, dtype=torch.float64, requires_grad=True)
    t = torch.softmax(torch.randn(9, 17, generator=g, dtype=torch.float64), -1)
    ref = torch.nn.CrossEntropyLoss()(logits, t)
    ref.backward()
    loss, dl = O.cross_entropy_soft_fwd_bwd(logits.detach(), t)
    assert torch.allclose(loss, ref.detach(), atol=1e-12) and torch.allclose(dl, logits.grad, atol=1e-12)


def test_g10_ssl_losses_oracle():
    """G10: DINO / iBOT losses of the reference classes (two iterations, centre EMA) reproduced by the oracle's restatement."""
    g = load_golden("g10_ssl_losses.npz")
    K, n, n_local = [int(v) for v in g["meta"]]
    center = torch.zeros(1, K, dtype=torch.float64)
    for it in range(2):
        teacher = t(g[f"dino{it}.teacher"], torch.float64)
        tp = O.softmax_center(teacher, center, 0.05)
        assert rel_err(tp, g[f"dino{it}.tprobs"].reshape(2 * n, K)) < 2e-6
        center = O.center_ema(center, teacher, 0.9)          # applied lazily by the reference at the next call
        tp2 = tp.view(2, n, K)
        s_glob, s_loc = t(g[f"dino{it}.s_glob"], torch.float64), t(g[f"dino{it}.s_loc"], torch.float64)
        lg, dg = O.distill_ce(s_glob, tp, 0.1, torch.full((2 * n,), 1.0 / (2 * n), dtype=torch.float64))
        assert abs(float(lg) - float(g[f"dino{it}.loss_g"])) < 2e-6 * abs(float(lg)) and rel_err(dg, g[f"dino{it}.ds_glob"]) < 2e-6
        tsum = tp2[0] + tp2[1]
        ll, dl = 0.0, []
        for chunk in s_loc.chunk(n_local):
            l, d = O.distill_ce(chunk, tsum, 0.1, torch.full((n,), 1.0 / n, dtype=torch.float64))
            ll, dl = ll + l, dl + [d]
        assert abs(float(ll) - float(g[f"dino{it}.loss_l"])) < 2e-6 * abs(float(ll)) and rel_err(torch.cat(dl), g[f"dino{it}.ds_loc"]) < 2e-6
    assert rel_err(center, g["dino.center"]) < 2e-6
    s3, t3, m3 = t(g["ibotd.s"], torch.float64), t(g["ibotd.t"], torch.float64), t(g["ibotd.m"]).bool()
    w = m3.double() / m3.double().sum(-1, keepdim=True).clamp(min=1.0) / m3.shape[0]
    l3, d3 = O.distill_ce(s3.reshape(-1, K), t3.reshape(-1, K), 0.1, w.reshape(-1))
    assert abs(float(l3) - float(g["ibotd.loss"])) < 2e-6 * abs(float(l3)) and rel_err(d3.reshape(s3.shape), g["ibotd.ds"]) < 2e-6


# ------------------------------------------------------------------------------------------- G13: kNN vote and metric objects
def test_g13_knn_vote_and_metrics_oracle():
    """oracle/eval_oracle.py against the reference's own knn_predict source and metric classes (golden G13): the kNN scores to 1e-6,
    the metric values to the three decimals the reference rounds to, the confusion matrices exactly."""
    from oracle import eval_oracle as E
    d = np.load(os.path.join(GOLDEN, "g13_knn_metrics.npz"))
    k, t, C = int(d["knn_k"]), float(d["knn_t"]), int(d["knn_classes"])
    s = E.knn_predict(d["knn_feature"], d["knn_bank"], d["knn_labels"], k, t, C)
    assert np.abs(s - d["knn_scores"]).max() < 1e-6 and np.allclose(s.sum(1), 1.0)
    sm = E.knn_predict(d["knn_feature"], d["knn_bank"], d["knn_labels_multi"], k, t, d["knn_labels_multi"].shape[0], True)
    assert np.abs(sm - d["knn_scores_multi"]).max() < 1e-6
    for tag, n_cls in (("mc", 7), ("bin", 2)):
        r = E.classification_metrics(d[tag + "_logits"], d[tag + "_truths"], n_cls)
        assert np.array_equal(r["confusion_matrix"], d[tag + "_confusion"])
        for key in ("accuracy", "mean_per_class_accuracy", "quadratic_kappa", "roc_auc", "recall"):
            assert abs(r[key] - float(d[f"{tag}_{key}"])) <= 5.01e-4, (tag, key, r[key], float(d[f"{tag}_{key}"]))
    r = E.multilabel_metrics(d["ml_logits"], d["ml_truths"])
    for key in ("accuracy", "mAP", "precision", "recall", "f1", "roc_auc"):
        assert abs(r[key] - float(d["ml_" + key])) <= 5.01e-4, (key, r[key], float(d["ml_" + key]))


def test_philox_known_answers():
    """The oracle's Philox4x32-10 (the generator behind apla_dropout_fwd's keep mask) against the Random123 known-answer vectors
    (kat_vectors: philox4x32 10)."""
    import numpy as np
    from oracle import apla_oracle as O
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = O.philox4x32_10(np.array(ctr, dtype=np.uint64), np.array(key, dtype=np.uint64))
        assert tuple(int(v) for v in got) == want, (ctr, [hex(int(v)) for v in got])
    m = O.philox_keep_mask(100000, 0.25, seed=12345, offset=7)
    assert abs(m.mean() - 0.75) < 5e-3 and m[:8].tolist() != m[8:16].tolist()
