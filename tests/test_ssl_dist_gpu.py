"""Data-parallel DINOv2-APLA iteration: two processes share cuda:0 (gloo on CUDA tensors, as in tests/test_dist_gpu.py —
everything of the N > 1 path except the RCCL transport).  (a) Both ranks fed the SAME batch must reproduce the
single-process iteration: the flat-gradient all-reduce is a SUM with 1/world folded into the optimizer, the centre
all-reduces divide by rows * world (dino_clstoken_loss.py:90-101) — and the reference's own two iterations (golden G12).
(b) Fed DIFFERENT batches, the replicas' students,
teachers and centres stay bit-identical."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def _trainer(g, pg=None):
    from apla_amd.ssl import CosineScheduler, Dinov2Trainer
    from test_ssl_step_gpu import build_from_golden
    model = build_from_golden(g, "apla")
    sched = (CosineScheduler(base_value=1e-3, final_value=1e-6, total_iters=6, warmup_iters=2, start_warmup_value=0),
             CosineScheduler(base_value=0.04, final_value=1e-4, total_iters=6), CosineScheduler(base_value=0.9, final_value=1.0, total_iters=6),
             CosineScheduler(base_value=0.07, final_value=0.07, total_iters=3, warmup_iters=3, start_warmup_value=0.04), None)
    return Dinov2Trainer(model, iters_per_epoch=1, epochs=6, grad_clipping=3.0, freeze_last_layer_epochs=1, schedules=sched, process_group=pg,
                         exchange_chunk_mb=0.25)    # several chunks at this geometry: the hook-driven exchange is what runs


def _batch(g, it):
    from conftest import t
    return {"images": {"collated_global_crops": t(g[f"it{it}.glob"]), "collated_local_crops": t(g[f"it{it}.loc"]),
                       "collated_masks": t(g[f"it{it}.masks"]), "mask_indices_list": t(g[f"it{it}.mask_indices"]),
                       "masks_weight": t(g[f"it{it}.masks_weight"]), "upperbound": int(g[f"it{it}.upperbound"]),
                       "n_masked_patches": torch.tensor([len(g[f"it{it}.mask_indices"])])}}


def _worker(outdir, same_data):
    import torch.distributed as dist
    from conftest import load_golden
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    g = load_golden("g12_ssl_step_apla.npz")
    tr = _trainer(g, dist.group.WORLD)
    assert tr.world == 2 and tr.exchanger.active and len(tr.exchanger.chunks) >= 3
    for step in (1, 2):
        it = step if same_data else (1 + (step + rank) % 2)     # different batches per rank, swapped at the second step
        tr.global_step(_batch(g, it))
    tr.model.dino_loss.apply_center_update()
    torch.cuda.synchronize()
    teacher = torch.cat([p.detach().reshape(-1) for n, p in tr.model.teacher.named_parameters()
                         if dict(tr.model.student.named_parameters())[n].requires_grad])
    np.save(os.path.join(outdir, f"student_{rank}.npy"), tr.optimizer.flat.cpu().numpy())
    np.save(os.path.join(outdir, f"teacher_{rank}.npy"), teacher.cpu().numpy())
    np.save(os.path.join(outdir, f"center_{rank}.npy"), tr.model.dino_loss.center.cpu().numpy())
    if rank == 0:
        np.savez(os.path.join(outdir, "student_named.npz"),
                 **{n: p.detach().cpu().numpy() for n, p in tr.model.student.named_parameters() if p.requires_grad})
    dist.barrier()


@pytest.mark.parametrize("same_data", [True, False])
def test_two_rank_ssl_iteration(tmp_path, same_data):
    from apla_amd.dist import launch
    from conftest import load_golden
    launch(_worker, (str(tmp_path), same_data), n_procs=2, backend="gloo")
    for what in ("student", "teacher", "center"):
        assert np.array_equal(np.load(tmp_path / f"{what}_0.npy"), np.load(tmp_path / f"{what}_1.npy")), what
    if not same_data:
        return
    g = load_golden("g12_ssl_step_apla.npz")
    tr = _trainer(g)
    for it in (1, 2):
        tr.global_step(_batch(g, it))
    tr.model.dino_loss.apply_center_update()
    torch.cuda.synchronize()
    ref = tr.optimizer.flat.cpu().numpy()
    got = np.load(tmp_path / "student_0.npy")
    # same batch on both ranks: the summed gradient is exactly twice the local one and 1/world halves it again
    assert float(np.abs(got - ref).max()) < 1e-6, float(np.abs(got - ref).max())
    assert np.allclose(np.load(tmp_path / "center_0.npy"), tr.model.dino_loss.center.cpu().numpy(), rtol=1e-5, atol=1e-7)
    # ... and the two-rank result against what the REFERENCE's own classes produced for these two iterations (golden G12:
    # tests/golden/make_golden.py): the student's accumulated update of every trainable tensor and the DINO centre
    from conftest import rel_err, t
    named = np.load(tmp_path / "student_named.npz")
    assert sorted(named.files) == sorted(str(n) for n in g["trainable"])
    for n in named.files:
        ref_upd = t(g[f"it2.student.{n}"]) - t(g["init." + n])
        upd = torch.from_numpy(named[n]) - t(g["init." + n])
        assert float((upd - ref_upd).abs().mean() / (ref_upd.abs().mean() + 1e-12)) < 0.2, n
    assert rel_err(torch.from_numpy(np.load(tmp_path / "center_0.npy")), g["dino.center"]) < 2e-2


def _rccl_worker(outdir):
    """One rank, backend nccl (= RCCL): the trainer's hook-driven chunked exchange with real RCCL calls on the side stream."""
    import torch.distributed as dist
    from apla_amd.ssl import CosineScheduler, Dinov2Trainer
    from conftest import load_golden
    from test_ssl_step_gpu import build_from_golden
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 400))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        g = load_golden("g12_ssl_step_apla.npz")
        sched = (CosineScheduler(base_value=1e-3, final_value=1e-6, total_iters=6, warmup_iters=2, start_warmup_value=0),
                 CosineScheduler(base_value=0.04, final_value=1e-4, total_iters=6), CosineScheduler(base_value=0.9, final_value=1.0, total_iters=6),
                 CosineScheduler(base_value=0.07, final_value=0.07, total_iters=3, warmup_iters=3, start_warmup_value=0.04), None)
        tr = Dinov2Trainer(build_from_golden(g, "apla"), iters_per_epoch=1, epochs=6, grad_clipping=3.0, freeze_last_layer_epochs=1,
                           schedules=sched, process_group=dist.group.WORLD, exchange_chunk_mb=0.25, force_exchange=True)
        assert tr.world == 1 and tr.exchanger.active and len(tr.exchanger.chunks) >= 3
        for it in (1, 2):
            tr.global_step(_batch(g, it))
        torch.cuda.synchronize()
        np.save(os.path.join(outdir, "rccl_student.npy"), tr.optimizer.flat.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_rccl_chunked_exchange_on_one_rank(tmp_path):
    """The RCCL transport under the self-supervised trainer's exchange (chunks started by post-accumulate hooks while backward runs, an
    all-reduce per chunk on the side stream, the wait before the norm pass), which the two-rank gloo test cannot cover on a one-GPU
    box: a `nccl` group of ONE rank with the exchange forced on must leave the iteration bit-identical to the plain one."""
    import torch.multiprocessing as mp
    from conftest import load_golden
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rccl_worker, args=(str(tmp_path),))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    g = load_golden("g12_ssl_step_apla.npz")
    tr = _trainer(g)
    assert not tr.exchanger.active
    for it in (1, 2):
        tr.global_step(_batch(g, it))
    torch.cuda.synchronize()
    assert np.array_equal(np.load(tmp_path / "rccl_student.npy"), tr.optimizer.flat.cpu().numpy())
