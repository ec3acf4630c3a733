"""Packed ("nested") batches for the dinov2-style path.

The reference concatenates crops of different sizes along the token axis and restricts attention to each crop with
``xformers.ops.fmha.BlockDiagonalMask.from_seqlens`` (dinov2/layers/block.py:188-217, used by
``APLA_MemEffAttention.forward(x, attn_bias)`` — apla/appla_attn_mem_eff.py:27-42).  xformers is not a dependency here:
this class carries the same information (the sequence lengths) and mirrors the part of the xformers API that the
reference touches — ``from_seqlens``, ``from_tensor_list``, ``split`` — so dinov2 ``NestedTensorBlock`` code keeps working.
The mask itself is never materialised: the HIP kernels take the cumulative offsets (``apla_attn_varlen_fwd/bwd``).
"""
from typing import List, Sequence, Tuple

import torch


# Offsets of a layout as device tensors, shared by every mask object of that layout: a multi-crop training loop builds a new mask with
# the same sequence lengths every iteration, and a host -> device copy from pageable memory makes the host wait for the stream — the
# launch path of the next kernels then starts from zero lead (seen as 40-80 us idle gaps behind every such copy in the kernel trace).
_INDEX_CACHE = {}


def _device_index(kind, values, dtype, device):
    key = (kind, values, dtype, str(device))
    t = _INDEX_CACHE.get(key)
    if t is None:
        if len(_INDEX_CACHE) >= 64:
            _INDEX_CACHE.clear()
        t = _INDEX_CACHE[key] = torch.tensor(values, dtype=dtype, device=device)
    return t


class BlockDiagonalMask:
    def __init__(self, seqlens: Sequence[int]):
        seqlens = [int(n) for n in seqlens]
        if not seqlens or min(seqlens) <= 0:
            raise ValueError("BlockDiagonalMask needs at least one sequence and positive lengths")
        self.seqlens: List[int] = seqlens
        self.total = sum(seqlens)
        self.max_seqlen = max(seqlens)
        offs = [0]
        for n in seqlens:
            offs.append(offs[-1] + n)
        self.seqstart_py: List[int] = offs
        self._cu = {}            # device -> int32 offsets
        self._batch_sizes = None  # dinov2 stores the per-crop-size batch sizes here (block.py:207)

    # -- xformers-compatible constructors / helpers -----------------------------------------------------------------
    @classmethod
    def from_seqlens(cls, q_seqlen: Sequence[int], kv_seqlen=None) -> "BlockDiagonalMask":
        if kv_seqlen is not None and list(kv_seqlen) != list(q_seqlen):
            raise NotImplementedError("only self-attention masks (q_seqlen == kv_seqlen) are supported")
        return cls(q_seqlen)

    @classmethod
    def from_tensor_list(cls, tensors: Sequence[torch.Tensor]) -> Tuple["BlockDiagonalMask", torch.Tensor]:
        """Tensors [B_i, N_i, C] -> (mask over sum_i B_i sequences, packed tensor [1, sum_i B_i*N_i, C])."""
        seqlens = []
        for t in tensors:
            seqlens += [t.shape[1]] * t.shape[0]
        mask = cls(seqlens)
        mask._batch_sizes = [t.shape[0] for t in tensors]
        packed = torch.cat([t.reshape(1, -1, t.shape[-1]) for t in tensors], dim=1)
        return mask, packed

    def split(self, x: torch.Tensor, batch_sizes=None) -> List[torch.Tensor]:
        """Inverse of from_tensor_list: [1, total, C] -> list of [B_i, N_i, C] (consecutive equal-length sequences are
        grouped; ``batch_sizes`` overrides the grouping like xformers' argument of the same name)."""
        if x.shape[0] != 1 or x.shape[1] != self.total:
            raise ValueError(f"expected a packed tensor [1, {self.total}, C], got {tuple(x.shape)}")
        batch_sizes = batch_sizes if batch_sizes is not None else self._batch_sizes
        if batch_sizes is None:
            batch_sizes, prev = [], None
            for n in self.seqlens:
                if n == prev:
                    batch_sizes[-1] += 1
                else:
                    batch_sizes.append(1)
                prev = n
        out, seq, tok = [], 0, 0
        for bs in batch_sizes:
            n = self.seqlens[seq]
            if any(m != n for m in self.seqlens[seq:seq + bs]):
                raise ValueError("batch_sizes groups sequences of different lengths")
            out.append(x[:, tok:tok + bs * n].reshape(bs, n, x.shape[-1]))
            seq, tok = seq + bs, tok + bs * n
        return out

    # -- what the kernels need ----------------------------------------------------------------------------------------
    def runs(self) -> List[Tuple[int, int]]:
        """Consecutive sequences of equal length as (count, length) pairs: a multi-crop batch is two such runs (global crops,
        local crops), each a uniform batch the dense attention kernels take at their own block size."""
        if getattr(self, "_runs", None) is None:
            out: List[Tuple[int, int]] = []
            for n in self.seqlens:
                if out and out[-1][1] == n:
                    out[-1] = (out[-1][0] + 1, n)
                else:
                    out.append((1, n))
            self._runs = out
        return self._runs

    def cu_seqlens(self, device) -> torch.Tensor:
        device = torch.device(device)
        if device not in self._cu:
            self._cu[device] = _device_index("cu", tuple(self.seqstart_py), torch.int32, device)
        return self._cu[device]

    def seq_starts(self, device) -> torch.Tensor:
        """int64 [S]: first packed row of every sequence (the row of its class token), on `device`."""
        key = ("starts", torch.device(device))
        if key not in self._cu:
            self._cu[key] = _device_index("starts", tuple(self.seqstart_py[:-1]), torch.long, torch.device(device))
        return self._cu[key]

    def materialize(self, dtype=torch.float32, device="cpu") -> torch.Tensor:
        """Dense additive bias [total, total] (0 inside a block, -inf outside): for tests and the CPU oracle only."""
        m = torch.full((self.total, self.total), float("-inf"), dtype=dtype, device=device)
        for a, b in zip(self.seqstart_py[:-1], self.seqstart_py[1:]):
            m[a:b, a:b] = 0
        return m
