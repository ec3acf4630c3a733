"""The classification training step on the DROP-IN module path: ``Trainer.global_step`` (defaults/trainer.py:106-151) as the reference
runs it — model forward, criterion, ``loss.backward()``, gradient exchange, ``clip_grad_norm_``, AdamW — over the modules of this
package (``apla_amd.vit`` + ``build_apla``: every tensor op of the backbone is a HIP kernel behind a ``torch.autograd.Function``) and
``FlatAdamW``.

``AplaTrainEngine`` (the fused, captured launch sequence) is what ``main.py`` and ``bench.py`` use.  Until round 5 it had no dropout /
stochastic depth and ``main.py --dr / --dpr`` (main.py:101-111 of the reference) trained here; since round 6 the engine takes both and
this trainer is what ``main.py --module_path`` selects, or what a model the engine refuses falls back to: same parameters, same optimizer
semantics (two groups, one flat fp32 buffer, global-norm clip, 1/world folded into the optimizer), the same ``train_step(images,
labels, lr=)`` / ``grad_norm`` surface, 0.89 of the fused step's speed without regularisation, 0.77 with ``--dr 0.1``
(profiles/r06_d_dropout_bench.md).
"""
from typing import Optional

import torch
import torch.distributed as dist

from . import ops
from .dist import GradExchanger
from .optim import DynamicLossScale, FlatAdamW


class ModulePathTrainer:
    def __init__(self, model, *, lr: float = 1e-4, weight_decay: float = 1e-5, grad_clipping: float = 0.0, process_group=None,
                 compute_dtype=torch.bfloat16, loss_scale=1.0, soft_targets: bool = False, label_smoothing: float = 0.0):
        """``loss_scale``: a float (static) or "dynamic" (GradScaler semantics, fp16 only: defaults/trainer.py:129-138)."""
        if compute_dtype not in (torch.bfloat16, torch.float16):
            raise TypeError("compute_dtype must be torch.bfloat16 or torch.float16")
        self.model = model.cuda().train()
        self.device = next(self.model.parameters()).device      # (apla_amd.evaluate.Evaluator reads it, as from the fused engine)
        self.compute_dtype = compute_dtype
        dyn = loss_scale == "dynamic"
        self.scaler = DynamicLossScale(enabled=dyn) if dyn else DynamicLossScale(init_scale=float(loss_scale), growth_interval=1 << 62)
        if not dyn:
            self.scaler.backoff_factor = 1.0     # a static scale stays what it is; a non-finite step is still skipped and counted
        self.grad_clipping, self.soft_targets = float(grad_clipping or 0.0), soft_targets
        self.optimizer = FlatAdamW(self.model.named_parameters(), lr=lr, weight_decay=weight_decay)
        self.exchanger = GradExchanger(self.optimizer.grads, [(0, self.optimizer.grads.numel())], process_group)
        self.world = self.exchanger.world
        self.criterion = torch.nn.CrossEntropyLoss(label_smoothing=0.0 if soft_targets else label_smoothing)
        self.step_count = 0

    @property
    def grad_norm(self) -> torch.Tensor:
        return self.optimizer.grad_norm()

    def train_step(self, images, labels, lr: Optional[float] = None) -> torch.Tensor:
        opt = self.optimizer
        if lr is not None:
            opt.lr = float(lr)
        opt.zero_grad()
        with ops.use_half(self.compute_dtype):
            logits = self.model(images).float()
            loss = self.criterion(logits, labels if self.soft_targets else labels.long())
            scale = self.scaler.scale
            (loss * scale if scale != 1.0 else loss).backward()
        if self.exchanger.active:
            self.exchanger.launch_chunk(0)
            self.exchanger.wait()
        applied = opt.step(max_norm=self.grad_clipping, grad_scale=1.0 / (self.world * scale),
                           check_finite=self.compute_dtype == torch.float16)
        self.scaler.update(applied)
        self.step_count += 1
        return loss.detach()

    loss_scale = property(lambda self: self.scaler.scale)
    skipped_steps = property(lambda self: self.scaler.skipped_steps)

    @torch.no_grad()
    def forward_only(self, images, labels=None):
        """(logits, features, loss | None) in evaluation mode (every dropout is the identity there)."""
        was = self.model.training
        self.model.eval()
        try:
            with ops.use_half(self.compute_dtype):
                logits, emb = self.model(images, return_embedding=True)
            logits = logits.float()
            loss = torch.nn.functional.cross_entropy(logits, labels.long()) if labels is not None and not self.soft_targets else None
        finally:
            self.model.train(was)
        return logits, emb.float(), loss


def wants_dropout(model) -> bool:
    """Does any module of `model` ask for element-wise dropout (nn.Dropout with p > 0: --dr / --adr)?  That is what AplaTrainEngine
    refuses; stochastic depth alone (DropPath, --dpr) stays on the fused step since round 6 (fused into its LayerNorm kernels)."""
    return any(isinstance(mod, torch.nn.Dropout) and mod.p > 0.0 for mod in model.modules())
