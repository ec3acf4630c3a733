"""The data-parallel TRAINING STEP end to end on the GPU (SURVEY §8e / golden G7 semantics): two processes share cuda:0,
each owns half of the global batch and runs the fused engine with a process group; their updated parameters must equal
those of ONE process stepping on the whole batch (DDP mean = SUM all-reduce, 1/world folded into the optimizer).

The collective here is gloo on CUDA tensors — RCCL cannot put two ranks on one device, and the test boxes have a single
GPU — so this covers everything of the N > 1 path except the RCCL transport itself: chunked exchange on the side stream
between the two backward segments, hipGraph replay around it, grad_scale = 1/world in the fused clip + AdamW."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def _worker(outdir, use_graphs):
    import torch.distributed as dist
    from apla_amd.engine import AplaTrainEngine, OptimConfig
    from test_engine_gpu import small_vit
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    gen = torch.Generator().manual_seed(0)
    images = torch.randn(8, 3, 32, 32, generator=gen)
    labels = torch.randint(0, 10, (8,), generator=gen)
    sl = slice(rank * 4, rank * 4 + 4)
    eng = AplaTrainEngine(small_vit(depth=4, r=64), 4, 32, optim=OptimConfig(lr=1e-3, weight_decay=1e-2, grad_clipping=1.0),
                          process_group=dist.group.WORLD, use_graphs=use_graphs)
    assert eng.world == 2 and len(eng.chunks) == 4 and len(eng.seg_cuts) == 4
    for _ in range(3):
        eng.train_step(images[sl].cuda(), labels[sl].cuda())
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, f"params_{rank}.npy"), eng.flat_params.cpu().numpy())
    np.save(os.path.join(outdir, f"gnorm_{rank}.npy"), np.array([float(eng.grad_norm)]))
    dist.barrier()


@pytest.mark.parametrize("use_graphs", [False, True])
def test_two_rank_step_equals_full_batch_step(tmp_path, use_graphs):
    from apla_amd.dist import launch
    from apla_amd.engine import AplaTrainEngine, OptimConfig
    from test_engine_gpu import small_vit
    launch(_worker, (str(tmp_path), use_graphs), n_procs=2, backend="gloo")
    p0, p1 = np.load(tmp_path / "params_0.npy"), np.load(tmp_path / "params_1.npy")
    assert np.array_equal(p0, p1)                                   # replicas stay bit-identical
    gen = torch.Generator().manual_seed(0)
    images = torch.randn(8, 3, 32, 32, generator=gen)
    labels = torch.randint(0, 10, (8,), generator=gen)
    eng = AplaTrainEngine(small_vit(depth=4, r=64), 8, 32, optim=OptimConfig(lr=1e-3, weight_decay=1e-2, grad_clipping=1.0),
                          use_graphs=use_graphs)
    for _ in range(3):
        eng.train_step(images.cuda(), labels.cuda())
    torch.cuda.synchronize()
    ref = eng.flat_params.cpu().numpy()
    # same math, different batching of the bf16 kernels (per-rank row sets, two partial sums): agreement to a few 1e-4 of the
    # three accumulated AdamW updates (lr 1e-3)
    assert float(np.abs(p0 - ref).max()) < 3e-4, float(np.abs(p0 - ref).max())
    assert abs(float(np.load(tmp_path / "gnorm_0.npy")[0]) - float(eng.grad_norm)) < 2e-2 * float(eng.grad_norm)


def _rccl_worker(outdir):
    """One rank, backend nccl (= RCCL): the engine takes the world > 1 path (APLA_FORCE_EXCHANGE)."""
    import torch.distributed as dist
    from apla_amd.engine import AplaTrainEngine, OptimConfig
    from test_engine_gpu import small_vit
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 400),
                      APLA_FORCE_EXCHANGE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        gen = torch.Generator().manual_seed(0)
        images, labels = torch.randn(4, 3, 32, 32, generator=gen).cuda(), torch.randint(0, 10, (4,), generator=gen).cuda()
        eng = AplaTrainEngine(small_vit(depth=4, r=64), 4, 32, optim=OptimConfig(lr=1e-3, weight_decay=1e-2, grad_clipping=1.0),
                              process_group=dist.group.WORLD, use_graphs=True)
        assert eng.world == 1 and eng.exchanger.active and len(eng.seg_cuts) == 4
        for _ in range(3):
            eng.train_step(images, labels)
        torch.cuda.synchronize()
        np.save(os.path.join(outdir, "rccl_params.npy"), eng.flat_params.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_rccl_exchange_path_on_one_rank(tmp_path):
    """The RCCL transport itself, which the two-rank gloo tests cannot cover on a one-GPU box: a process group of ONE rank on
    backend nccl, the engine forced onto its world > 1 path (four hipGraph segments, an RCCL all-reduce per chunk on the side
    stream, the wait before the optimizer).  The result must be bit-identical to the plain single-process step."""
    import torch.multiprocessing as mp
    from apla_amd.engine import AplaTrainEngine, OptimConfig
    from test_engine_gpu import small_vit
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rccl_worker, args=(str(tmp_path),))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    gen = torch.Generator().manual_seed(0)
    images, labels = torch.randn(4, 3, 32, 32, generator=gen).cuda(), torch.randint(0, 10, (4,), generator=gen).cuda()
    eng = AplaTrainEngine(small_vit(depth=4, r=64), 4, 32, optim=OptimConfig(lr=1e-3, weight_decay=1e-2, grad_clipping=1.0), use_graphs=True)
    for _ in range(3):
        eng.train_step(images, labels)
    torch.cuda.synchronize()
    assert np.array_equal(np.load(tmp_path / "rccl_params.npy"), eng.flat_params.cpu().numpy())
