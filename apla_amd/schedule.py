"""Per-iteration learning-rate schedule of the reference trainer, as plain host-side arithmetic.

The reference drives ``torch.optim.lr_scheduler`` objects through ``MixedLRScheduler.step()`` once per training iteration
(defaults/trainer.py:137; utils/_utils.py:369-415): a ``LinearWarmup`` (utils/_utils.py:123-156) that adds a constant
increment for ``warmup_iters`` iterations, optionally followed by ``CosineAnnealingLR`` (stepped only once the warm-up is
over, T_max = steps_per_epoch*epochs - warmup_iters, defaults/wrappers.py:279-289).  The fused optimizer kernel takes the
learning rate as a scalar argument (``AplaTrainEngine.train_step(lr=...)``), so the schedule is just a sequence of floats;
this class reproduces the reference's sequence **including its quirks** (tests/golden/g9_lr_schedule.json holds the
sequences produced by the reference classes themselves):

* ``LinearWarmup`` starts from ``eta_min + delta`` (the scheduler constructor performs one step) and keeps adding
  ``delta = (max_lr - eta_min) / warmup_iters`` while its step count is ``<= warmup_iters``, so the plateau is
  ``max_lr + delta``, one increment above ``max_lr``;
* the cosine phase uses PyTorch's recursive (chainable) form, so it anneals from that plateau, not from ``max_lr``.
"""
import math
from typing import List, Optional


class LRSchedule:
    def __init__(self, max_lr: float, *, warmup_iters: int = 0, warmup_epochs: int = 0, steps_per_epoch: Optional[int] = None,
                 warmup_eta_min: float = 1e-8, cosine: bool = False, epochs: Optional[int] = None,
                 cosine_eta_min: float = 0.0, use_warmup: bool = True):
        self.max_lr = float(max_lr)
        self.use_warmup = use_warmup
        self.cosine = cosine
        self.warmup_iters = 0
        if use_warmup:
            if warmup_epochs:  # utils/_utils.py:127-139: epochs win over iterations
                if steps_per_epoch is None:
                    raise TypeError("LinearWarmup with warmup_epochs settings must include steps_per_epoch")
                warmup_iters = steps_per_epoch * warmup_epochs
            if not warmup_iters:
                warmup_iters = 1
            self.warmup_iters = int(warmup_iters)
            self.warmup_eta_min = float(warmup_eta_min)
            self.delta = (self.max_lr - self.warmup_eta_min) / self.warmup_iters
        if cosine:
            if steps_per_epoch is None or epochs is None:
                raise TypeError("CosineAnnealingLR needs steps_per_epoch and epochs (T_max = steps_per_epoch * epochs)")
            self.T_max = steps_per_epoch * epochs - self.warmup_iters  # defaults/wrappers.py:283-286
            self.cosine_eta_min = float(cosine_eta_min)
        self.reset()

    def reset(self):
        self.iter = 0            # MixedLRScheduler.iter
        self._warm_epoch = 0     # LinearWarmup.last_epoch after its constructor's initial step
        self._cos_epoch = 0
        # LinearWarmup.__init__ sets lr = eta_min, then _LRScheduler.__init__ performs one step (last_epoch 0 -> +delta)
        self.lr = self.warmup_eta_min + self.delta if self.use_warmup else self.max_lr

    def step(self) -> float:
        """One MixedLRScheduler.step(): call it after every optimizer step; returns the lr of the NEXT iteration."""
        self.iter += 1
        if self.use_warmup:
            self._warm_epoch += 1
            if self._warm_epoch <= self.warmup_iters:  # utils/_utils.py:150-154
                self.lr += self.delta
        if self.cosine and self.iter > self.warmup_iters:  # utils/_utils.py:407-409
            self._cos_epoch += 1
            e, T, eta = self._cos_epoch, self.T_max, self.cosine_eta_min
            if (e - 1 - T) % (2 * T) == 0:  # torch CosineAnnealingLR.get_lr restart branch (base lr = max_lr + delta plateau)
                base = self.warmup_eta_min if self.use_warmup else self.max_lr  # 'initial_lr' captured at construction
                self.lr = self.lr + (base - eta) * (1 - math.cos(math.pi / T)) / 2
            else:
                self.lr = (1 + math.cos(math.pi * e / T)) / (1 + math.cos(math.pi * (e - 1) / T)) * (self.lr - eta) + eta
        return self.lr

    def sequence(self, n: int) -> List[float]:
        """lr used at iterations 0 .. n-1 (from a fresh schedule)."""
        self.reset()
        out = []
        for _ in range(n):
            out.append(self.lr)
            self.step()
        self.reset()
        return out
