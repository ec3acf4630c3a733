"""torch.optim.AdamW + clip_grad_norm_ over one flat fp32 buffer, on the fused HIP optimizer kernels.

The fused classification engine owns its flat buffer itself (apla_amd/engine.py).  This class gives the same thing to
module-path training loops (the DINOv2-APLA step, apla_amd/ssl/trainer.py): every trainable Parameter becomes a view
into one flat buffer, its ``.grad`` a view into a second one (autograd accumulates into it in place), and a step is
``apla_grad_sumsq`` (one deterministic global norm, defaults/trainer.py:127-131 ``clip_grad_norm_``) followed by
``apla_adamw_apply`` per run of tensors that share a step count.  Semantics kept from the reference's optimizer
(defaults/wrappers.py:187-221): two groups — weight decay on tensors that are neither ``*.bias`` nor 1-D, none on the
rest —, one learning rate for all, per-tensor ``step`` that only advances when the tensor is updated (torch skips
parameters whose ``.grad`` is None: ``skip=`` here).
"""
from typing import Iterable, List, Sequence, Tuple

import torch

from . import functional as AF
from ._lib import check, lib
from .ops import _stream


class DynamicLossScale:
    """torch.cuda.amp.GradScaler's scale bookkeeping (scaler.update(), defaults/trainer.py:129-138; self_supervised/dinov2/
    trainer.py:124-135) for loops whose optimizer reports whether the step was applied (``FlatAdamW.step(check_finite=True)``):
    a skipped step multiplies the scale by ``backoff_factor`` and resets the growth counter, ``growth_interval`` applied steps in a row
    multiply it by ``growth_factor``.  ``enabled=False`` is the constant 1 (bf16)."""

    def __init__(self, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000,
                 enabled: bool = True):
        self.enabled = enabled
        self.scale = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.growth_tracker, self.skipped_steps = 0, 0

    def update(self, applied: bool):
        if not self.enabled:
            return
        if not applied:
            self.scale *= self.backoff_factor
            self.growth_tracker, self.skipped_steps = 0, self.skipped_steps + 1
            return
        self.growth_tracker += 1
        if self.growth_tracker >= self.growth_interval:
            self.scale *= self.growth_factor
            self.growth_tracker = 0

    def state_dict(self):   # GradScaler.state_dict() layout (defaults/bases.py:465-466)
        return {"scale": self.scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": self.growth_tracker}

    def load_state_dict(self, sd):
        self.scale, self.growth_tracker = float(sd.get("scale", self.scale)), int(sd.get("_growth_tracker", 0))


class FlatAdamW:
    def __init__(self, named_params: Iterable[Tuple[str, torch.nn.Parameter]], lr=1e-4, weight_decay=1e-5, betas=(0.9, 0.999),
                 eps=1e-8):
        items = [(n, p) for n, p in named_params if p.requires_grad]
        if not items:
            raise ValueError("FlatAdamW: no trainable parameters")
        dev = items[0][1].device
        if dev.type != "cuda":
            raise AF.AplaHipError("FlatAdamW runs on the GPU (no CPU fallback)")
        self.names: List[str] = [n for n, _ in items]
        self.params: List[torch.nn.Parameter] = [p for _, p in items]
        sizes = [p.numel() for p in self.params]
        self.offsets = [0]
        for s in sizes:
            self.offsets.append(self.offsets[-1] + s)
        n = self.offsets[-1]
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        self.decay_mask = torch.zeros(n, device=dev, dtype=torch.uint8)
        self.norm_ws = torch.zeros(512, device=dev, dtype=torch.float32)
        self.steps = [0] * len(items)
        for (name, p), a, b in zip(items, self.offsets[:-1], self.offsets[1:]):
            if p.dtype != torch.float32:
                raise TypeError(f"FlatAdamW keeps fp32 masters; {name} is {p.dtype}")
            self.flat[a:b].copy_(p.detach().reshape(-1))
            p.data = self.flat[a:b].view(p.shape)
            p.grad = self.grads[a:b].view(p.shape)
            if not (name.endswith(".bias") or p.ndim == 1):   # get_params_groups, defaults/wrappers.py:205-221
                self.decay_mask[a:b] = 1
        self.lr, self.weight_decay, self.betas, self.eps = lr, weight_decay, betas, eps

    def zero_grad(self):
        self.grads.zero_()
        for p, a, b in zip(self.params, self.offsets[:-1], self.offsets[1:]):
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * a:   # someone set .grad = None: re-attach the view
                p.grad = self.grads[a:b].view(p.shape)

    def grad_norm(self) -> torch.Tensor:
        """Global gradient norm before clipping, as left on the device by the last step (a 0-d tensor; no host sync)."""
        return self.norm_ws[1]

    def step(self, max_norm: float = 0.0, grad_scale: float = 1.0, skip: Sequence[str] = (), check_finite: bool = False) -> bool:
        """clip_grad_norm_(all trainable tensors, max_norm) then AdamW on every tensor whose name contains none of `skip`
        (utils/_utils.py:418-421 ``cancel_gradients`` matches by substring).  With ``check_finite`` (fp16 under a loss scale,
        whose inverse the caller folds into ``grad_scale``) the step is GradScaler.step: if the norm of the unscaled gradients
        is not finite nothing is updated, no step count advances and False is returned (one host read, as GradScaler does)."""
        L = lib()
        n = self.flat.numel()
        s = _stream()
        check(L.apla_grad_sumsq(self.grads.data_ptr(), n, float(grad_scale), self.norm_ws.data_ptr(), s), "apla_grad_sumsq")
        if check_finite and not bool(torch.isfinite(self.norm_ws[2:258].sum()).item()):
            self.norm_ws[1] = float("inf")
            return False
        active = [not any(k in name for k in skip) for name in self.names]
        for i, on in enumerate(active):
            if on:
                self.steps[i] += 1
        i = 0
        while i < len(self.names):   # runs of consecutive active tensors with the same step count share one launch
            if not active[i]:
                i += 1
                continue
            j = i
            while j + 1 < len(self.names) and active[j + 1] and self.steps[j + 1] == self.steps[i]:
                j += 1
            a, b = self.offsets[i], self.offsets[j + 1]
            check(L.apla_adamw_apply(self.flat.data_ptr() + 4 * a, self.grads.data_ptr() + 4 * a, self.exp_avg.data_ptr() + 4 * a,
                                     self.exp_avg_sq.data_ptr() + 4 * a, self.decay_mask.data_ptr() + a, b - a, float(self.lr),
                                     float(self.weight_decay), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                     int(self.steps[i]), float(max_norm), float(grad_scale), self.norm_ws.data_ptr(), s),
                  "apla_adamw_apply")
            i = j + 1
        if not all(active) and (max_norm > 0.0 or grad_scale != 1.0):
            # clip_grad_norm_ / unscale_ rescale EVERY gradient, also those of the tensors the update leaves out (the reference drops
            # them afterwards): the same coefficient the kernel derived, from the norm it left in norm_ws[1]
            coef = float(grad_scale) if max_norm <= 0.0 else (max_norm / (self.norm_ws[1] + 1e-6)).clamp(max=1.0) * float(grad_scale)
            for i, on in enumerate(active):
                if not on:
                    self.grads[self.offsets[i]:self.offsets[i + 1]].mul_(coef)
        # the kernels wrote through raw pointers: tell the 16-bit weight cache (keyed by tensor version) that values changed
        for p, on in zip(self.params, active):
            if on:
                torch.autograd.graph.increment_version(p)
        return True

    # -- torch.optim-compatible state, for the reference's session layout (bases.py:456-467) ---------------------------
    def state_dict(self):
        return {"state": {i: {"step": torch.tensor(float(self.steps[i])), "exp_avg": self.exp_avg[a:b].view(p.shape).clone(),
                              "exp_avg_sq": self.exp_avg_sq[a:b].view(p.shape).clone()}
                          for i, (p, a, b) in enumerate(zip(self.params, self.offsets[:-1], self.offsets[1:])) if self.steps[i] > 0},
                "param_names": list(self.names), "lr": self.lr, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        for i, st in sd["state"].items():
            a, b = self.offsets[int(i)], self.offsets[int(i) + 1]
            self.steps[int(i)] = int(st["step"])
            self.exp_avg[a:b].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[a:b].copy_(st["exp_avg_sq"].reshape(-1))
        self.lr, self.weight_decay = sd.get("lr", self.lr), sd.get("weight_decay", self.weight_decay)
