"""Host side of the DINOv2-APLA step's input: iBOT block masking, the multi-crop collate and the per-iteration schedules
(self_supervised/dinov2/dinov2_utils.py:21-163, trainer.py:7-54).

This is integer / bookkeeping work on a handful of small arrays per batch; it stays on the host as in the reference and is
held bit-exact to it: under the same ``random.seed`` the masks, ``mask_indices_list``, ``masks_weight`` and ``upperbound``
are identical (tests/golden/g12_ssl_step_*.npz were produced by the reference's collate; tests/test_ssl_cpu.py).  The
draw ORDER from Python's ``random`` module is therefore part of the contract: per masked sample one ``uniform`` for the
target count, then per attempt ``uniform`` (area), ``uniform`` (log aspect), ``randint`` (top), ``randint`` (left), and
one ``shuffle`` of the mask list at the end.
"""
import math
import random
from typing import Dict, List, Sequence

import numpy as np
import torch


class MaskingGenerator:
    """Block-wise masking of a patch grid (BEiT style), dinov2_utils.py:65-140."""

    def __init__(self, input_size, num_masking_patches=None, min_num_patches=4, max_num_patches=None, min_aspect=0.3,
                 max_aspect=None):
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 2
        self.height, self.width = input_size
        self.num_patches = self.height * self.width
        self.num_masking_patches = num_masking_patches
        self.min_num_patches = min_num_patches
        self.max_num_patches = num_masking_patches if max_num_patches is None else max_num_patches
        max_aspect = max_aspect or 1 / min_aspect
        self.log_aspect_ratio = (math.log(min_aspect), math.log(max_aspect))

    def get_shape(self):
        return self.height, self.width

    def _mask(self, mask: np.ndarray, max_mask_patches) -> int:
        """Up to 10 attempts to place one rectangle that adds between 1 and max_mask_patches new masked cells."""
        added = 0
        for _ in range(10):
            area = random.uniform(self.min_num_patches, max_mask_patches)
            aspect = math.exp(random.uniform(*self.log_aspect_ratio))
            h = int(round(math.sqrt(area * aspect)))
            w = int(round(math.sqrt(area / aspect)))
            if w < self.width and h < self.height:
                top = random.randint(0, self.height - h)
                left = random.randint(0, self.width - w)
                window = mask[top:top + h, left:left + w]
                fresh = h * w - int(window.sum())
                if 0 < fresh <= max_mask_patches:
                    window[...] = True          # in-place on the view: same cells the reference sets one by one
                    added += fresh
                if added > 0:
                    break
        return added

    def __call__(self, num_masking_patches=0) -> np.ndarray:
        mask = np.zeros(shape=self.get_shape(), dtype=bool)
        count = 0
        while count < num_masking_patches:
            budget = min(num_masking_patches - count, self.max_num_patches)
            added = self._mask(mask, budget)
            if added == 0:
                break
            count += added
        return mask


def collate_data_and_cast(samples_list, n_global_crops, n_local_crops, mask_ratio_tuple, mask_probability, dtype, n_tokens=None,
                          mask_generator=None) -> Dict:
    """List of ``([global crops..., local crops...], label)`` -> the batch dictionary of dinov2_utils.py:21-62 (CPU tensors;
    ``DINOv2.forward`` moves them to the GPU).  Crops are stacked crop-major: all first global crops, then all second."""
    glob = torch.stack([s[0][i] for i in range(n_global_crops) for s in samples_list])
    loc = torch.stack([s[0][i] for i in range(n_global_crops, n_global_crops + n_local_crops) for s in samples_list])
    labels = torch.cat([s[1].unsqueeze(0) for s in samples_list], dim=0)
    B, N = len(glob), n_tokens
    n_masked_samples = int(B * mask_probability)
    probs = torch.linspace(*mask_ratio_tuple, n_masked_samples + 1)
    upperbound = 0
    masks: List[torch.Tensor] = []
    for i in range(n_masked_samples):
        lo, hi = probs[i], probs[i + 1]
        masks.append(torch.BoolTensor(mask_generator(int(N * random.uniform(lo, hi)))))
        upperbound += int(N * hi)
    for _ in range(n_masked_samples, B):
        masks.append(torch.BoolTensor(mask_generator(0)))
    random.shuffle(masks)
    collated_masks = torch.stack(masks).flatten(1)
    mask_indices_list = collated_masks.flatten().nonzero().flatten()
    masks_weight = (1 / collated_masks.sum(-1).clamp(min=1.0)).unsqueeze(-1).expand_as(collated_masks)[collated_masks]
    return {
        "images": {
            "collated_global_crops": glob.to(dtype),
            "collated_local_crops": loc.to(dtype),
            "collated_masks": collated_masks,
            "mask_indices_list": mask_indices_list,
            "masks_weight": masks_weight,
            "upperbound": upperbound,
            "n_masked_patches": torch.full((1,), fill_value=mask_indices_list.shape[0], dtype=torch.long),
        },
        "labels": labels,
    }


class CosineScheduler:
    """freeze (zeros) -> linear warm-up -> cosine to final_value, one value per iteration (dinov2_utils.py:143-163)."""

    def __init__(self, base_value, final_value, total_iters, warmup_iters=0, start_warmup_value=0, freeze_iters=0):
        self.final_value = final_value
        self.total_iters = total_iters
        freeze = np.zeros((freeze_iters))
        warm = np.linspace(start_warmup_value, base_value, warmup_iters)
        it = np.arange(total_iters - warmup_iters - freeze_iters)
        cos = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * it / len(it)))
        self.schedule = np.concatenate((freeze, warm, cos))
        assert len(self.schedule) == self.total_iters

    def __getitem__(self, it):
        return self.final_value if it >= self.total_iters else self.schedule[it]


def build_schedulers(*, lr, eta_min, warmup_epochs, weight_decay, momentum_teacher, final_momentum_teacher, warmup_teacher_temp,
                     teacher_temp, warmup_teacher_temp_epochs, freeze_last_layer_epochs, iters_per_epoch, total_iters):
    """The five schedules of trainer.py:7-54 (lr, wd, teacher momentum, teacher temperature, last-layer lr); the weight
    decay anneals to the hard-coded 1e-4 of trainer.py:20."""
    lr_kw = dict(start_warmup_value=0, base_value=lr, final_value=eta_min, total_iters=total_iters,
                 warmup_iters=warmup_epochs * iters_per_epoch)
    lr_s = CosineScheduler(**lr_kw)
    wd_s = CosineScheduler(base_value=weight_decay, final_value=1e-4, total_iters=total_iters, warmup_iters=0)
    mom_s = CosineScheduler(base_value=momentum_teacher, final_value=final_momentum_teacher, total_iters=total_iters, warmup_iters=0)
    tt_iters = warmup_teacher_temp_epochs * iters_per_epoch
    tt_s = CosineScheduler(start_warmup_value=warmup_teacher_temp, base_value=teacher_temp, final_value=teacher_temp,
                           total_iters=tt_iters, warmup_iters=tt_iters)
    last_s = CosineScheduler(**lr_kw)
    last_s.schedule[: freeze_last_layer_epochs * iters_per_epoch] = 0
    return lr_s, wd_s, mom_s, tt_s, last_s


def synthetic_samples(batch: int, global_size: int, local_size: int, n_local: int, generator: torch.Generator) -> Sequence:
    """N(0,1) crops in the shape the dataset's __getitem__ returns (post-Normalize statistics): for benches and tests."""
    return [([torch.randn(3, global_size, global_size, generator=generator) for _ in range(2)]
             + [torch.randn(3, local_size, local_size, generator=generator) for _ in range(n_local)], torch.tensor(0))
            for _ in range(batch)]
