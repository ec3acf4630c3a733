"""Data-parallel host logic on CPU with the gloo backend, world_size 2: chunked all-reduce of the flat
APLA-trainable gradient buffer reproduces DDP mean semantics (defaults/wrappers.py:183); launcher rendezvous works."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _dp_worker(outdir, n_chunks=2):
    import torch.distributed as dist
    from conftest import load_golden, t
    from oracle import apla_oracle as O
    from apla_amd.dist import GradExchanger, backward_order_chunks, dist_average_tensor, is_rank0, synchronize
    rank, world = dist.get_rank(), dist.get_world_size()
    g = load_golden("g5_tiny_model.npz")
    D, L, H, r, C, patch = [int(v) for v in g["meta"]]
    p = {k[2:]: t(g[k], torch.float64) for k in g.files if k.startswith("p.")}
    cfg = dict(patch=patch, depth=L, heads=H, r=r)
    gen = torch.Generator().manual_seed(0)
    per = 2 if world <= 3 else 1        # images per rank
    images = torch.randn(per * world, 3, 48, 48, generator=gen, dtype=torch.float64)
    labels = torch.randint(0, C, (per * world,), generator=gen)
    sl = slice(rank * per, rank * per + per)  # this rank's shard of the global batch
    logits, ctx = O.vit_forward(images[sl], p, cfg)
    loss, dl = O.cross_entropy_fwd_bwd(logits, labels[sl])
    grads = O.vit_backward(dl, ctx, p, cfg)
    names = O.trainable_names(L)
    flat = torch.cat([grads[n].reshape(-1) for n in names])
    offs = np.cumsum([0] + [grads[n].numel() for n in names])
    block_offs = [int(offs[2 * i]) for i in range(L)]
    chunks = backward_order_chunks(block_offs, flat.numel(), n_chunks)
    assert chunks[0][1] == flat.numel() and chunks[-1][0] == 0   # backward order: tail first
    ex = GradExchanger(flat, chunks)
    for k in range(len(chunks)):
        ex.launch_chunk(k)
    ex.wait()
    flat *= ex.grad_scale
    mean_loss = dist_average_tensor(loss)
    synchronize()
    if is_rank0():
        lf, cf = O.vit_forward(images, p, cfg)
        loss_f, dlf = O.cross_entropy_fwd_bwd(lf, labels)
        gf = O.vit_backward(dlf, cf, p, cfg)
        ref = torch.cat([gf[n].reshape(-1) for n in names])
        err = float((flat - ref).abs().max() / ref.abs().max())
        np.save(os.path.join(outdir, "result.npy"), np.array([err, float(mean_loss - loss_f), world]))


@pytest.mark.parametrize("world,n_chunks", [(2, 2), (3, 4), (8, 4)])
def test_chunked_allreduce_matches_full_batch(tmp_path, world, n_chunks):
    """World sizes beyond two (8 = the driver's scaling run): the chunked SUM all-reduce and the 1 / world of the
    optimizer reproduce the full-batch gradient of the oracle, the averaged loss the full-batch loss."""
    from apla_amd.dist import launch
    launch(_dp_worker, (str(tmp_path), n_chunks), n_procs=world, backend="gloo")
    err, dloss, w = np.load(tmp_path / "result.npy")
    assert w == world and err < 1e-12 and abs(dloss) < 1e-12


def test_backward_order_chunks_properties():
    from apla_amd.dist import backward_order_chunks
    offs = [i * 1000 for i in range(12)]
    total = 12 * 1000 + 777
    for n in (1, 2, 3, 4):
        ch = backward_order_chunks(offs, total, n)
        assert sorted(ch)[0][0] == 0 and sorted(ch)[-1][1] == total
        assert all(a[0] == b[1] for a, b in zip(ch[:-1], ch[1:]))      # contiguous, descending
        assert all(lo in offs or lo == 0 for lo, _ in ch)               # cuts only at block boundaries
    assert backward_order_chunks([], 10, 2) == [(0, 10)]


def test_single_process_exchanger_is_identity():
    from apla_amd.dist import GradExchanger
    flat = torch.arange(10, dtype=torch.float32)
    ex = GradExchanger(flat, [(5, 10), (0, 5)])
    ex.launch_chunk(0), ex.launch_chunk(1), ex.wait()
    assert ex.grad_scale == 1.0 and torch.equal(flat, torch.arange(10, dtype=torch.float32))


@pytest.mark.parametrize("world", [2, 8])
def test_bench_self_launches_its_ranks(world):
    """`python bench.py --gpus N` with no launcher environment must start the N ranks itself and print ONE JSON line with
    n_gpus = N.  Here (no GPU) the children run --launcher-check: the same spawn / rendezvous / MAX-over-ranks
    plumbing over gloo without the GPU step — at N = 8 too, the driver's scaling run: the per-rank lists of the line
    (`ranks`, built by the helpers the real line uses) have one entry per rank, in rank order."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--launcher-check"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["ranks_seen"] == world and d["steps"] == 3
    rk = d["ranks"]
    assert len(rk["ms_per_step"]) == world and len(rk["exchange_wait_ms"]) == world and len(rk["socket_w"]) == world
    assert rk["ms_per_step"][0] == rk["ms_per_step_min"] and rk["ms_per_step"][-1] == rk["ms_per_step_max"]      # rank r sleeps (1 + r) ms per step
    assert all(a <= b + 0.5 for a, b in zip(rk["ms_per_step"][:-1], rk["ms_per_step"][1:])), rk["ms_per_step"]
    assert rk["env"]["NCCL_MAX_NCHANNELS"] == "8" and rk["reserved_cus"] == 8          # the RCCL channel budget travels to every rank


def test_bench_parent_fails_when_a_rank_fails():
    """A failed child means the parent exits non-zero (and takes the other ranks down with it)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    # without --launcher-check the ranks need a GPU: in this container every child exits with the "needs an MI355X" message
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-peak-probe", "--no-parity"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
