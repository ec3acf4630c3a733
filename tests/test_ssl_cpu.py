"""Host side of the DINOv2-APLA step on CPU: the masking collate and the schedules are bit-exact against what the REFERENCE's
own collate / MaskingGenerator / CosineScheduler produced (golden G12, tests/golden/make_golden.py:g12_ssl_step), and the
SSL oracle (oracle/ssl_oracle.py) reproduces the reference step's losses, gradients and updated parameters."""
import random

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, t


@pytest.fixture(scope="module")
def g12():
    return load_golden("g12_ssl_step_apla.npz")


def _replay_collate(g):
    """Same draws as the generator: crops from torch.Generator(32) (after the index / head-noise draws), masks from random.seed(5)."""
    from apla_amd.ssl import MaskingGenerator, collate_data_and_cast
    D, depth, heads, patch, pre, gsz, lsz, K, hid, bott, B, n_local = [int(v) for v in g["meta"]]
    mg = MaskingGenerator(input_size=(gsz // patch, gsz // patch), max_num_patches=0.5 * gsz // patch * gsz // patch)
    random.seed(5)
    out = []
    for it in (1, 2):
        glob, loc = t(g[f"it{it}.glob"]), t(g[f"it{it}.loc"])
        samples = [([glob[b], glob[B + b]] + [loc[c * B + b] for c in range(n_local)], torch.tensor(0)) for b in range(B)]
        out.append(collate_data_and_cast(samples, n_global_crops=2, n_local_crops=n_local, mask_ratio_tuple=(0.1, 0.5),
                                         mask_probability=0.5, dtype=torch.float32, n_tokens=(gsz // patch) ** 2, mask_generator=mg)["images"])
    return out


def test_collate_and_masks_bit_exact(g12):
    for it, data in zip((1, 2), _replay_collate(g12)):
        assert torch.equal(data["collated_global_crops"], t(g12[f"it{it}.glob"])) and torch.equal(data["collated_local_crops"], t(g12[f"it{it}.loc"]))
        assert np.array_equal(data["collated_masks"].numpy(), g12[f"it{it}.masks"])
        assert np.array_equal(data["mask_indices_list"].numpy(), g12[f"it{it}.mask_indices"])
        assert np.array_equal(data["masks_weight"].numpy(), g12[f"it{it}.masks_weight"])
        assert data["upperbound"] == int(g12[f"it{it}.upperbound"])
        assert int(data["n_masked_patches"]) == len(g12[f"it{it}.mask_indices"])


def test_masking_generator_properties():
    from apla_amd.ssl import MaskingGenerator
    random.seed(0)
    mg = MaskingGenerator(input_size=(16, 16), max_num_patches=0.5 * 16 * 16)
    assert mg(0).sum() == 0
    for target in (4, 30, 77, 128):
        m = mg(target)
        assert m.shape == (16, 16) and m.dtype == bool and 0 < m.sum() <= target


def test_schedules_match_reference(g12):
    from apla_amd.ssl import CosineScheduler
    lr = CosineScheduler(base_value=1e-3, final_value=1e-6, total_iters=6, warmup_iters=2, start_warmup_value=0)
    wd = CosineScheduler(base_value=0.04, final_value=1e-4, total_iters=6)
    mom = CosineScheduler(base_value=0.9, final_value=1.0, total_iters=6)
    tt = CosineScheduler(base_value=0.07, final_value=0.07, total_iters=3, warmup_iters=3, start_warmup_value=0.04)
    assert np.array_equal(lr.schedule, g12["sched.lr"]) and np.array_equal(wd.schedule, g12["sched.wd"])
    assert np.array_equal(mom.schedule, g12["sched.mom"])
    assert np.array_equal(np.array([tt[i] for i in range(6)]), g12["sched.tt"])
    assert lr[100] == 1e-6


def test_build_schedulers_freezes_last_layer_lr():
    from apla_amd.ssl import build_schedulers
    lr, wd, mom, tt, last = build_schedulers(lr=1e-3, eta_min=1e-6, warmup_epochs=1, weight_decay=0.04, momentum_teacher=0.994,
                                             final_momentum_teacher=1.0, warmup_teacher_temp=0.04, teacher_temp=0.07,
                                             warmup_teacher_temp_epochs=2, freeze_last_layer_epochs=1, iters_per_epoch=5, total_iters=20)
    assert len(lr.schedule) == 20 and np.all(last.schedule[:5] == 0) and np.array_equal(last.schedule[5:], lr.schedule[5:])
    assert wd[19] > 1e-4 and wd[20] == 1e-4 and mom[0] == 0.994 and tt[0] == 0.04 and tt[10] == 0.07 and tt[50] == 0.07


@pytest.mark.parametrize("tag", ["apla", "full"])
def test_ssl_oracle_reproduces_reference_step(tag):
    from oracle import ssl_oracle as SO
    g = load_golden(f"g12_ssl_step_{tag}.npz")
    st = SO.state_from_golden(g)
    for it in (1, 2):
        out = SO.train_iteration(st, SO.batch_from_golden(g, it), hyper=[float(v) for v in g[f"it{it}.hyper"]], clip=3.0,
                                 freeze_last=(it == 1))
        assert abs(float(out["loss"]) - float(g[f"it{it}.loss"])) < 2e-5
        for k, v in out["loss_dict"].items():
            assert abs(float(v) - float(g[f"it{it}.ld.{k}"])) < 2e-5, k
        assert abs(float(out["gnorm"]) - float(g[f"it{it}.gnorm"])) < 1e-4 * float(g[f"it{it}.gnorm"])
        for name in [str(n) for n in g["trainable"]]:
            assert rel_err(out["grads"][name], g[f"it{it}.g.{name}"]) < 5e-5, name
            # Adam's early steps are ~lr * sign(g): elements whose fp32 reference gradient is rounding noise may differ by a
            # fraction of one lr-sized step (lr = 1e-3 against parameters of size ~0.1)
            assert rel_err(st["student"][name], g[f"it{it}.student.{name}"]) < 1e-3, name
            assert rel_err(st["teacher"][name], g[f"it{it}.teacher.{name}"]) < 1e-3, name
    c = SO.centers(st)
    assert rel_err(c["dino"], g["dino.center"]) < 1e-5 and rel_err(c["ibot"], g["ibot.center"]) < 1e-5


def test_chunked_exchange_launch_order_is_rank_invariant():
    """the chunked gradient exchange of Dinov2Trainer must issue the same sequence of collectives on every rank, whatever
    order the autograd engine of a rank completes the gradients in and whichever tensors receive none there.  The launch rule
    (descending chunk index, a complete chunk waits for its successors) on shuffled completion orders, with tensors left out."""
    import random
    from apla_amd.ssl.trainer import Dinov2Trainer

    class Recorder:
        active = True
        def __init__(self, n): self.chunks, self.order, self.waited = [None] * n, [], 0
        def launch_chunk(self, k): self.order.append(k)
        def wait(self): self.waited += 1

    sizes = [3, 1, 4, 2, 5]
    tensors = [k for k, n in enumerate(sizes) for _ in range(n)]
    rng = random.Random(0)
    for trial in range(50):
        tr = object.__new__(Dinov2Trainer)
        tr.exchanger, tr._chunk_size = Recorder(len(sizes)), list(sizes)
        tr._reset_exchange()
        done = list(tensors)
        rng.shuffle(done)
        for k in done[:rng.randrange(len(done) + 1)]:     # some tensors never get a gradient on this "rank"
            tr._chunk_seen[k] += 1
            tr._launch_ready()
            assert tr.exchanger.order == list(range(len(sizes) - 1, len(sizes) - 1 - len(tr.exchanger.order), -1))
        n_hook = len(tr.exchanger.order)
        tr._finish_exchange()
        assert tr.exchanger.order == [4, 3, 2, 1, 0] and tr.exchanger.waited == 1
        assert tr._chunk_seen == [0] * 5 and tr._chunk_next == 4
        # how many chunks left from a hook (overlapped with backward) and how many only after it is on record
        assert sum(tr.exchange_counts.values()) == 5 and tr.exchange_counts["from_hooks"] == n_hook
