"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) / xGMI.

Replaces, for the APLA step, DistributedDataParallel (defaults/wrappers.py:182-183) and the helpers of
utils/dist_utills.py + utils/launch.py:

* ``GradExchanger`` — the only data-path collective: SUM all-reduce of the flat fp32 buffer that holds exactly the
  APLA-trainable gradients (L*(r*D+r) + D*C + C elements: 10 MB for ViT-B, not the 86 M-parameter model), issued in
  chunks that follow backward order on a side HIP stream so the first chunk overlaps the remaining backward.  The DDP
  mean is applied later as ``grad_scale = 1/world`` inside the fused clip+AdamW kernel.  No buffer broadcast per
  forward (indices are immutable and identical on all ranks), no per-iteration barrier (trainer.py:92).
* ``launch`` / ``init_from_env`` — single-node spawn (utils/launch.py:27-94) or torchrun-style env rendezvous.

Works on CPU tensors with the gloo backend (tests/test_dist_cpu.py) — the host logic is device-agnostic.
"""
import os
import socket
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def ddp_is_on() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def is_rank0() -> bool:
    return (not ddp_is_on()) or dist.get_rank() == 0


def print_ddp(*a, **k):
    if is_rank0():
        print(*a, **k)


def synchronize():
    """dist.barrier when DDP is on (utils/dist_utills.py:42-45).  NOT called inside the step loop."""
    if ddp_is_on():
        dist.barrier()


def dist_average_tensor(t: torch.Tensor) -> torch.Tensor:
    """Mean over ranks of a (scalar) tensor, e.g. the logged loss (utils/dist_utills.py:48-58)."""
    if not ddp_is_on():
        return t
    out = t.detach().clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out / dist.get_world_size()


def backward_order_chunks(offsets: Sequence[int], total: int, n_chunks: int = 2) -> List[Tuple[int, int]]:
    """Split [0,total) at parameter boundaries into ``n_chunks`` contiguous ranges, returned in BACKWARD order (the tail
    of the buffer — head + last blocks — becomes ready first).  ``offsets`` are the start offsets of the per-block groups."""
    offsets = sorted(set(int(o) for o in offsets if 0 < o < total))
    if n_chunks <= 1 or not offsets:
        return [(0, total)]
    cuts = []
    for k in range(1, n_chunks):
        target = total * k / n_chunks
        cuts.append(min(offsets, key=lambda o: abs(o - target)))
    cuts = sorted(set(cuts))
    bounds = [0] + cuts + [total]
    ranges = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1) if bounds[i + 1] > bounds[i]]
    return ranges[::-1]


class GradExchanger:
    def __init__(self, flat_grads: torch.Tensor, chunks: Sequence[Tuple[int, int]], process_group=None, always: bool = False,
                 local: bool = False):
        """``always``: issue the collectives even in a group of one process (diagnostic: lets a single-GPU box run the real
        RCCL calls on the side stream; an all-reduce over one rank leaves the buffer unchanged).  ``local``: never exchange, whatever
        torch.distributed's state is (an engine that ONE rank of a job builds for itself, e.g. bench.py's parity check on rank 0:
        without it `process_group=None` means the default group, and that rank would wait for collectives nobody else issues)."""
        self.flat = flat_grads
        self.chunks = list(chunks)
        self.pg = None if local else process_group
        self.world = 1 if local else self.world_of(process_group)
        always = always and not local
        self.active = self.world > 1 or (always and dist.is_initialized() and process_group is not None)
        self._comm = torch.cuda.Stream() if (flat_grads.is_cuda and self.active) else None
        self._wait_events = None    # timing of wait(): see time_waits()
        covered = sorted(self.chunks)
        assert covered[0][0] == 0 and covered[-1][1] == flat_grads.numel() and all(
            covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1)), "chunks must tile the flat buffer"

    @staticmethod
    def world_of(process_group=None) -> int:
        """Ranks an exchanger built with this group sums over: the group's size, or the default group's when none is given and
        torch.distributed is initialised, else 1."""
        if dist.is_available() and dist.is_initialized() and (process_group is not None or ddp_is_on()):
            return dist.get_world_size(process_group)
        return 1

    def launch_chunk(self, k: int):
        """Start the all-reduce of chunk k (call right after the backward segment that produced it was enqueued)."""
        if not self.active or k >= len(self.chunks):
            return
        lo, hi = self.chunks[k]
        view = self.flat[lo:hi]
        if self._comm is None:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._comm.wait_event(ev)
        with torch.cuda.stream(self._comm):
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)

    def wait(self):
        """Make the compute stream wait for all outstanding chunks (before clip/AdamW)."""
        if self._comm is not None:
            if self._wait_events is not None:       # two events on the compute stream around the wait: what the exchange costs the step
                cur = torch.cuda.current_stream()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self._comm)
                e1.record(cur)
                self._wait_events.append((e0, e1))
                return
            torch.cuda.current_stream().wait_stream(self._comm)

    def time_waits(self, on: bool = True):
        """Start (or stop) recording how long the compute stream is held at wait(): the part of the gradient exchange that the
        backward did not cover.  bench.py reports it per rank so that a scaling curve explains itself (exchange_wait_ms)."""
        self._wait_events = [] if (on and self._comm is not None) else None

    def mean_wait_ms(self):
        """Mean duration of the recorded waits in ms (after a synchronize); None when nothing was recorded."""
        if not self._wait_events:
            return None
        ts = [a.elapsed_time(b) for a, b in self._wait_events]
        return sum(ts) / len(ts)

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_from_env(backend: Optional[str] = None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def _worker(local_rank, fn, args, world, port, backend):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    init_from_env(backend)
    try:
        fn(*args)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def launch(fn, args=(), n_procs: Optional[int] = None, backend: Optional[str] = None):
    """Single-node launcher, one process per GPU (utils/launch.py:27-75): spawn when n_procs > 1, else call inline."""
    if n_procs is None:
        n_procs = max(1, torch.cuda.device_count())
    if n_procs <= 1:
        return fn(*args)
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(fn, args, n_procs, _free_port(), backend), nprocs=n_procs, daemon=False)
