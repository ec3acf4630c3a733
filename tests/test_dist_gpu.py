"""The data-parallel TRAINING STEP end to end on the GPU (SURVEY §8e / golden G7 semantics): two processes share cuda:0,
each owns half of the global batch and runs the fused engine with a process group; their updated parameters must equal
those of ONE process stepping on the whole batch (DDP mean = SUM all-reduce, 1/world folded into the optimizer) AND the CPU
oracle's step on the whole batch (exchanged gradients, gradient norm, parameters after three updates).

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
    images = torch.randn(4 * world, 3, 32, 32, generator=gen)
    labels = torch.randint(0, 10, (4 * world,), generator=gen)
    sl = slice(rank * 4, rank * 4 + 4)
    eng = AplaTrainEngine(small_vit(depth=4, r=64), 4, 32, optim=OptimConfig(lr=1e-3, weight_decay=1e-2, grad_clipping=1.0),
                          process_group=dist.group.WORLD, use_graphs=use_graphs)
    assert eng.world == world and len(eng.chunks) == 4 and len(eng.seg_cuts) == 4 and eng.exchanger.grad_scale == 1.0 / world
    assert sorted(eng.chunks)[0][0] == 0 and sorted(eng.chunks)[-1][1] == eng.flat_grads.numel()      # the chunks tile the buffer
    # step 1 in two halves, so that the exchanged gradient (SUM over ranks; the optimizer applies 1/world) can be stored
    eng.set_batch(images[sl].cuda(), labels[sl].cuda())
    eng.forward_backward()
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, f"grads_{rank}.npy"), (eng.flat_grads / world).cpu().numpy())
    eng.optimizer_step()
    for _ in range(2):
        eng.train_step(images[sl].cuda(), labels[sl].cuda())
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, f"params_{rank}.npy"), eng.flat_params.cpu().numpy())
    np.save(os.path.join(outdir, f"gnorm_{rank}.npy"), np.array([float(eng.grad_norm)]))
    dist.barrier()


@pytest.mark.parametrize("world,use_graphs", [(2, False), (2, True), (3, True), (5, True)])
def test_two_rank_step_equals_full_batch_step(tmp_path, world, use_graphs):
    """World sizes 3 and 5 as well (sizes the code had never executed; 5 ranks + this process = the six processes a
    box lets one user put on its GPU — the 8-rank plumbing runs on the CPU, tests/test_dist_cpu.py): 1 / world in the optimizer,
    chunk tiling, replicas bit-identical, the update equal to the one-process full-batch step and to the oracle's."""
    from apla_amd.dist import launch
    from apla_amd.engine import AplaTrainEngine, OptimConfig
    from test_engine_gpu import small_vit
    launch(_worker, (str(tmp_path), use_graphs), n_procs=world, backend="gloo")
    p0 = np.load(tmp_path / "params_0.npy")
    for r in range(1, world):
        assert np.array_equal(p0, np.load(tmp_path / f"params_{r}.npy")), r       # replicas stay bit-identical
    gen = torch.Generator().manual_seed(0)
    images = torch.randn(4 * world, 3, 32, 32, generator=gen)
    labels = torch.randint(0, 10, (4 * world,), generator=gen)
    eng = AplaTrainEngine(small_vit(depth=4, r=64), 4 * world, 32, optim=OptimConfig(lr=1e-3, weight_decay=1e-2, grad_clipping=1.0),
                          use_graphs=use_graphs)
    for _ in range(3):
        eng.train_step(images.cuda(), labels.cuda())
    torch.cuda.synchronize()
    ref = eng.flat_params.cpu().numpy()
    # same math, different batching of the bf16 kernels (per-rank row sets, two partial sums): agreement to a few 1e-4 of the
    # three accumulated AdamW updates (lr 1e-3)
    if world == 2:
        assert float(np.abs(p0 - ref).max()) < 3e-4, float(np.abs(p0 - ref).max())
    # more ranks = more partial sums per gradient element: AdamW moves every element by ~lr per step whatever its size, so an element
    # whose gradient sits at the bf16 noise floor can end up to 3 x lr away — the UPDATE as a whole must agree (the gradients
    # themselves are compared with the oracle below)
    p_init = small_vit(depth=4, r=64)
    init = torch.cat([q.detach().reshape(-1).float() for _, q in p_init.named_parameters() if q.requires_grad]).numpy()
    upd, upd_ref = p0 - init, ref - init
    assert float(np.linalg.norm(upd - upd_ref) / np.linalg.norm(upd_ref)) < 0.1, float(np.linalg.norm(upd - upd_ref) / np.linalg.norm(upd_ref))
    assert abs(float(np.load(tmp_path / "gnorm_0.npy")[0]) - float(eng.grad_norm)) < 2e-2 * float(eng.grad_norm)
    # ... and against the ORACLE's step on the whole batch (the reference's DDP semantics, defaults/wrappers.py:182-183: the
    # mean over the global batch of the per-sample gradients; golden G7 pins that the two-half-batch average equals it):
    # gradients after the exchange, global gradient norm of the third step, parameters after three clip + AdamW updates.
    from oracle import apla_oracle as O
    from test_engine_gpu import oracle_params, rel_l2
    model = small_vit(depth=4, r=64)
    p = oracle_params(model)
    p_init = {k: v.clone() for k, v in p.items()}
    cfg = dict(patch=16, depth=4, heads=2, r=64)
    state = {}
    for _ in range(3):
        _, _, _, gnorm = O.train_step(images.double(), labels, p, cfg, state, lr=1e-3, wd=1e-2, clip=1.0)
    g0 = np.load(tmp_path / "grads_0.npy")
    for r in range(1, world):
        assert np.array_equal(g0, np.load(tmp_path / f"grads_{r}.npy")), r     # every rank holds the same exchanged buffer
    lf, cf = O.vit_forward(images.double(), oracle_params(model), cfg)
    _, dlf = O.cross_entropy_fwd_bwd(lf, labels)
    gref = O.vit_backward(dlf, cf, oracle_params(model), cfg)        # first-step gradients before the clip (train_step clips in place)
    strip = lambda n: n[len("backbone."):] if n.startswith("backbone.") else n  # noqa: E731
    for n in eng.names:
        off, k, shape = eng.slices[n]
        assert rel_l2(g0[off:off + k], gref[strip(n)].reshape(-1)) < 3e-2, n
    assert abs(float(np.load(tmp_path / "gnorm_0.npy")[0]) - float(gnorm)) < 3e-2 * float(gnorm)
    upd, upd_ref = [], []
    for n in eng.names:
        off, k, shape = eng.slices[n]
        upd.append(torch.from_numpy(p0[off:off + k]).double() - p_init[strip(n)].reshape(-1))
        upd_ref.append(p[strip(n)].reshape(-1) - p_init[strip(n)].reshape(-1))
    # AdamW normalises every element's update to ~lr, so elements whose gradient is at the bf16 noise floor move differently;
    # the update as a whole must agree
    assert rel_l2(torch.cat(upd), torch.cat(upd_ref)) < 0.15


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


def _module_worker(outdir):
    import torch.distributed as dist
    from apla_amd.module_trainer import ModulePathTrainer
    from test_engine_gpu import small_vit
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    gen = torch.Generator().manual_seed(0)
    images, labels = torch.randn(8, 3, 32, 32, generator=gen), torch.randint(0, 10, (8,), generator=gen)
    sl = slice(rank * 4, rank * 4 + 4)
    tr = ModulePathTrainer(small_vit(depth=3, r=64), lr=1e-3, weight_decay=1e-2, grad_clipping=1.0, process_group=dist.group.WORLD)
    assert tr.world == 2 and tr.exchanger.active
    for _ in range(3):
        tr.train_step(images[sl].cuda(), labels[sl].cuda())
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, f"mp_params_{rank}.npy"), tr.optimizer.flat.cpu().numpy())
    dist.barrier()


def test_two_rank_module_path_trainer_equals_the_fused_step(tmp_path):
    """ModulePathTrainer (main.py --dr / --dpr; here without dropout so that it is deterministic) on two ranks: replicas bit-identical,
    and — the mean of the half-batch gradients being the full-batch gradient — the parameters after three clip + AdamW steps agree with
    the FUSED engine's single-process steps on the whole batch up to the 16-bit roundings the two paths place differently."""
    from apla_amd.dist import launch
    from apla_amd.engine import AplaTrainEngine, OptimConfig
    from test_engine_gpu import small_vit
    launch(_module_worker, (str(tmp_path),), n_procs=2, backend="gloo")
    p0, p1 = np.load(tmp_path / "mp_params_0.npy"), np.load(tmp_path / "mp_params_1.npy")
    assert np.array_equal(p0, p1)
    gen = torch.Generator().manual_seed(0)
    images, labels = torch.randn(8, 3, 32, 32, generator=gen).cuda(), torch.randint(0, 10, (8,), generator=gen).cuda()
    model = small_vit(depth=3, r=64)
    init = torch.cat([p.detach().reshape(-1).float().cpu() for p in model.parameters() if p.requires_grad])
    eng = AplaTrainEngine(model, 8, 32, optim=OptimConfig(lr=1e-3, weight_decay=1e-2, grad_clipping=1.0), use_graphs=False)
    for _ in range(3):
        eng.train_step(images, labels)
    torch.cuda.synchronize()
    ref = torch.cat([p.detach().reshape(-1).float().cpu() for p in eng.model.parameters() if p.requires_grad])
    got = torch.from_numpy(p0)
    assert got.numel() == ref.numel()
    upd, upd_ref = got - init, ref - init
    assert float((upd - upd_ref).norm() / upd_ref.norm()) < 0.15     # Adam normalises every element's step to ~lr (see the fused two-rank test)
