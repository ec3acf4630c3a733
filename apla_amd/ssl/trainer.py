"""The DINOv2-APLA training iteration (self_supervised/dinov2/trainer.py:57-162, ``Dinov2Trainer.global_step``).

One iteration = schedules -> zero_grad -> teacher + student forward and the four losses (``DINOv2.forward``) -> backward
through the HIP kernels -> gradient exchange (world > 1) -> global-norm clip -> last-layer freeze -> AdamW -> EMA teacher.
What changes against the reference's loop is where things run, not what they compute:
* clip + AdamW: ``FlatAdamW`` (apla_grad_sumsq / apla_adamw_apply over one flat buffer) instead of ``clip_grad_norm_`` +
  ``torch.optim.AdamW`` walking a dozen tensors; ``possibly_cancel_last_layer_grads`` (trainer.py:84-90) becomes the
  optimizer's ``skip=`` list — same effect: the prototype layer is left out of the update but not out of the norm;
* data parallel: the reference wraps the model in DistributedDataParallel (wrappers.py:73-78), whose buckets are reduced while
  backward still runs.  The trainable set here is one flat buffer (97 MB at config 4, 92 MB of it the head); it is exchanged in
  chunks of at most ``exchange_chunk_mb`` of consecutive tensors, each all-reduce (RCCL, sum; the mean is folded into
  ``grad_scale``) issued on a side stream by the hook of the chunk's last gradient — the prototype layer's 67 MB leave as soon
  as the head's backward has produced them and travel under the backbone's backward — and joined before the norm pass; plus the
  centre all-reduces the losses already issue (losses.py).  No per-step barrier;
* mixed precision: 16-bit operands with fp32 accumulation inside the kernels.  ``compute_dtype=torch.bfloat16`` (default) needs
  no loss scale.  ``compute_dtype=torch.float16`` is the reference's ``use_mixed_precision`` branch (trainer.py:124-135,
  autocast + GradScaler): the loss is multiplied by the scale before backward, the optimizer unscales inside its norm pass,
  an iteration whose gradients are not finite leaves parameters, moments and step counts untouched and halves the scale, and
  ``growth_interval`` finite iterations in a row double it — torch.cuda.amp.GradScaler's defaults and, like it, one host read
  of the found-inf flag per iteration.
"""
from typing import Dict, Optional

import torch
import torch.distributed as dist

from .. import ops
from ..dist import GradExchanger
from ..optim import DynamicLossScale, FlatAdamW
from .collate import build_schedulers
from .losses import grad_prescale
from .models import DINOv2


class Dinov2Trainer:
    def __init__(self, model: DINOv2, *, iters_per_epoch: int, epochs: int, lr: float = 1e-4, weight_decay: float = 1e-5,
                 eta_min: float = 1e-6, warmup_epochs: int = 0, grad_clipping: float = 0.0, freeze_last_layer_epochs: int = 0,
                 momentum_teacher: float = 0.994, final_momentum_teacher: float = 1.0, warmup_teacher_temp: float = 0.04,
                 teacher_temp: float = 0.07, warmup_teacher_temp_epochs: int = 30, schedules=None, process_group=None,
                 compute_dtype=torch.bfloat16, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000, exchange_chunk_mb: float = 32.0, force_exchange: bool = False):
        if compute_dtype not in (torch.bfloat16, torch.float16):
            raise TypeError("compute_dtype must be torch.bfloat16 or torch.float16")
        self.compute_dtype = compute_dtype
        # GradScaler state (fp16 only): torch.cuda.amp.GradScaler defaults, bases.py / defaults/trainer.py:50-51
        self.scaler = DynamicLossScale(init_scale, growth_factor, backoff_factor, growth_interval, enabled=compute_dtype == torch.float16)
        self.model = model
        self.iters_per_epoch, self.total_iters = iters_per_epoch, iters_per_epoch * epochs
        self.grad_clipping, self.freeze_last_for = grad_clipping, freeze_last_layer_epochs
        (self.lr_schedule, self.wd_schedule, self.momentum_schedule, self.teacher_temp_schedule,
         self.last_layer_lr_schedule) = schedules if schedules is not None else build_schedulers(
            lr=lr, eta_min=eta_min, warmup_epochs=warmup_epochs, weight_decay=weight_decay, momentum_teacher=momentum_teacher,
            final_momentum_teacher=final_momentum_teacher, warmup_teacher_temp=warmup_teacher_temp, teacher_temp=teacher_temp,
            warmup_teacher_temp_epochs=warmup_teacher_temp_epochs, freeze_last_layer_epochs=freeze_last_layer_epochs,
            iters_per_epoch=iters_per_epoch, total_iters=self.total_iters)
        self.optimizer = FlatAdamW(model.student.named_parameters(), lr=lr, weight_decay=weight_decay)
        model.flatten_teacher(self.optimizer)    # the EMA as one launch over two flat buffers (models.update_teacher)
        self.pg = process_group
        self.world = GradExchanger.world_of(process_group)
        self._setup_exchange(exchange_chunk_mb, force_exchange)
        self.iters, self.epoch = 1, 1    # the reference counts from 1 (bases.py; trainer.py:96-99 indexes the schedules with it)
        self.loss: Optional[torch.Tensor] = None
        self.loss_dict: Dict[str, torch.Tensor] = {}

    def _setup_exchange(self, chunk_mb: float, force: bool):
        """Chunks of consecutive tensors of the flat gradient buffer (a tensor larger than the budget is a chunk of its own) and one
        post-accumulate hook per tensor: the hook of the last tensor of a chunk to receive its gradient starts the chunk's all-reduce."""
        opt = self.optimizer
        budget, chunks, owner = int(chunk_mb * 2 ** 20 / 4), [], []
        lo = 0
        for i, (a, b) in enumerate(zip(opt.offsets[:-1], opt.offsets[1:])):
            if b - lo > budget and a > lo:
                chunks.append((lo, a))
                lo = a
            owner.append(len(chunks))
        chunks.append((lo, opt.offsets[-1]))
        self.exchanger = GradExchanger(opt.grads, chunks, self.pg, always=force)
        # exchange_counts (set up by _reset_exchange): chunks started by a gradient hook (their all-reduce runs under the rest of backward)
        # vs after backward (no overlap left): a tensor that gets no gradient keeps its chunk — and, by the descending launch order,
        # every lower one — open until the end
        self._chunk_size = [owner.count(k) for k in range(len(chunks))]
        self._reset_exchange()
        if not self.exchanger.active:
            return

        # Every rank must issue the SAME sequence of collectives on the communicator, whatever order its autograd engine completes the
        # gradients in (and whichever tensors receive none on that rank): chunks are launched strictly in descending index order —
        # the order backward fills the flat buffer in, heads first.  A chunk that completes early waits for its successors' launch;
        # what is still open after backward goes in the same order from _finish_exchange.
        def make_hook(k):
            def hook(_param):
                self._chunk_seen[k] += 1
                self._launch_ready()
            return hook
        for p, k in zip(opt.params, owner):
            p.register_post_accumulate_grad_hook(make_hook(k))

    def _reset_exchange(self):
        n = len(self.exchanger.chunks)
        self._chunk_seen, self._chunk_next = [0] * n, n - 1     # _chunk_next: the one chunk that may be launched now
        if not hasattr(self, "exchange_counts"):                # cumulative over the run (not reset per iteration)
            self.exchange_counts = {"from_hooks": 0, "after_backward": 0}

    def _launch_ready(self):
        while self._chunk_next >= 0 and self._chunk_seen[self._chunk_next] >= self._chunk_size[self._chunk_next]:
            self.exchanger.launch_chunk(self._chunk_next)
            self._chunk_next -= 1
            self.exchange_counts["from_hooks"] += 1      # overlapped with the rest of backward

    def _finish_exchange(self):
        """After backward: the chunks not launched yet (tensors that received no gradient this iteration keep theirs open), in the
        same descending order; then the compute stream waits for the side stream."""
        ex = self.exchanger
        if not ex.active:
            return
        while self._chunk_next >= 0:
            ex.launch_chunk(self._chunk_next)
            self._chunk_next -= 1
            self.exchange_counts["after_backward"] += 1  # nothing left to overlap with: a tensor of this (or a later) chunk got no gradient
        ex.wait()
        self._reset_exchange()

    # the scaler's state under the names the trainer has always exposed
    loss_scale = property(lambda self: self.scaler.scale)
    growth_tracker = property(lambda self: self.scaler.growth_tracker)
    skipped_steps = property(lambda self: self.scaler.skipped_steps)

    def global_step(self, batch) -> torch.Tensor:
        """``batch`` is the collate's dictionary (``batch['images']`` holds the crops and the mask bookkeeping)."""
        it = self.iters
        lr, wd = float(self.lr_schedule[it]), float(self.wd_schedule[it])
        teacher_temp, mom = float(self.teacher_temp_schedule[it]), float(self.momentum_schedule[it])
        opt = self.optimizer
        opt.lr, opt.weight_decay = lr, wd     # apply_optim_scheduler: one lr for both groups, wd on the regularised group
        opt.zero_grad()
        self._reset_exchange()      # (a backward that raised leaves its counts behind)
        fp16 = self.compute_dtype == torch.float16
        with ops.use_half(self.compute_dtype), grad_prescale(self.loss_scale):
            loss, loss_dict = self.model(images=batch["images"], teacher_temp=teacher_temp)
            (loss * self.loss_scale if fp16 else loss).backward()      # scaler.scale(loss).backward() (:125)
        self._finish_exchange()
        skip = ("dino_head.last_layer", "ibot_head.last_layer") if (self.freeze_last_for and self.epoch <= self.freeze_last_for) else ()
        # unscale_ + clip_grad_norm_ + cancel + scaler.step (:126-134): one norm pass with 1/(world * scale) folded in
        applied = opt.step(max_norm=self.grad_clipping or 0.0, grad_scale=1.0 / (self.world * self.loss_scale), skip=skip,
                           check_finite=fp16)
        self.scaler.update(applied)                                     # scaler.update() (:135)
        self.model.update_teacher(mom)
        self.loss, self.loss_dict = loss.detach(), {k: v.detach() for k, v in loss_dict.items()}
        self.iters += 1
        if (self.iters - 1) % self.iters_per_epoch == 0:
            self.epoch += 1
        return self.loss

    @property
    def feature_extractor(self):
        return self.model.teacher.backbone
