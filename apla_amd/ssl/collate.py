"""Host side of the DINOv2-APLA step's input: iBOT block masking, the multi-crop collate and the per-iteration schedules.

Written from the CONTRACT of the reference's host code (self_supervised/dinov2/dinov2_utils.py:21-163, trainer.py:7-54), not from
its text.  The contract has three parts, and tests/test_ssl_cpu.py holds this file to all of them bit for bit against golden G12
(produced by the reference's own collate under ``random.seed``):

1. **The draw order from Python's ``random`` module.**  Per masked sample: one ``uniform`` between two neighbouring mask-ratio
   edges (the target count); then, per rectangle attempt, ``uniform`` (area), ``uniform`` (log aspect ratio), and — only if the
   rectangle fits strictly inside the grid — ``randint`` (top) and ``randint`` (left).  Unmasked samples draw nothing.  One
   ``shuffle`` of a B-element list ends the batch.
2. **The arithmetic that turns draws into integers.**  The ratio edges are a float32 ``torch.linspace``; target counts and the
   ``upperbound`` are truncations of float32 products with the patch count.
3. **The batch dictionary**: crop-major stacks of the global and local crops, the flattened boolean masks, the flat indices of the
   masked patches, one weight 1 / (masked patches of that image) per masked patch, ``upperbound``, ``n_masked_patches``, labels.

Layout here: all masks of a batch live in ONE pre-allocated ``[B, H, W]`` boolean array that the generator fills in place; the
shuffle is applied to row indices (``random.shuffle`` permutes positions, whatever the list holds); indices and weights are derived
from that array.  The five schedules come from one piecewise helper.
"""
import math
import random
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

_ATTEMPTS_PER_RECTANGLE = 10


class MaskingGenerator:
    """Block-wise masking of a patch grid: rectangles of random area and aspect ratio are added until a target number of
    masked patches is reached (or no further rectangle can be placed).  Constructor and call signature are the reference's
    (dinov2_utils.py:65-140) so that the two classes are interchangeable; ``fill`` is this implementation's in-place form."""

    def __init__(self, input_size, num_masking_patches=None, min_num_patches=4, max_num_patches=None, min_aspect=0.3,
                 max_aspect=None):
        self.height, self.width = input_size if isinstance(input_size, tuple) else (input_size, input_size)
        self.num_patches = self.height * self.width
        self.num_masking_patches = num_masking_patches
        self.min_num_patches = min_num_patches
        self.max_num_patches = max_num_patches if max_num_patches is not None else num_masking_patches
        if not max_aspect:
            max_aspect = 1 / min_aspect
        self.log_aspect_ratio = (math.log(min_aspect), math.log(max_aspect))

    def get_shape(self) -> Tuple[int, int]:
        return self.height, self.width

    def _draw_rectangle(self, budget) -> Optional[Tuple[int, int, int, int]]:
        """One attempt: (top, left, h, w) of a rectangle strictly smaller than the grid, or None (then only two draws were made)."""
        area = random.uniform(self.min_num_patches, budget)
        aspect = math.exp(random.uniform(*self.log_aspect_ratio))
        h, w = int(round(math.sqrt(area * aspect))), int(round(math.sqrt(area / aspect)))
        if h >= self.height or w >= self.width:
            return None
        top = random.randint(0, self.height - h)
        return top, random.randint(0, self.width - w), h, w

    def _add_rectangle(self, grid: np.ndarray, budget) -> int:
        """Mask one rectangle that contributes between 1 and ``budget`` new patches; 0 if ten attempts found none."""
        for _ in range(_ATTEMPTS_PER_RECTANGLE):
            rect = self._draw_rectangle(budget)
            if rect is None:
                continue
            top, left, h, w = rect
            cells = grid[top:top + h, left:left + w]
            new = cells.size - np.count_nonzero(cells)
            if 1 <= new <= budget:
                cells.fill(True)
                return new
        return 0

    def fill(self, grid: np.ndarray, target: int) -> int:
        """Add rectangles to the boolean ``[H, W]`` array ``grid`` (in place) until ``target`` patches are masked; returns the count."""
        masked = 0
        while masked < target:
            new = self._add_rectangle(grid, min(target - masked, self.max_num_patches))
            if new == 0:
                break
            masked += new
        return masked

    def __call__(self, num_masking_patches=0) -> np.ndarray:
        grid = np.zeros(self.get_shape(), dtype=bool)
        self.fill(grid, num_masking_patches)
        return grid


def _draw_masks(n_images: int, n_tokens: int, mask_ratio_tuple, mask_probability: float, mask_generator) -> Tuple[np.ndarray, int]:
    """The batch's masks as one ``[B, H, W]`` array (already in shuffled order) and the ``upperbound`` of masked patches."""
    grid = np.zeros((n_images,) + tuple(mask_generator.get_shape()), dtype=bool)
    n_masked = int(n_images * mask_probability)
    edges = torch.linspace(mask_ratio_tuple[0], mask_ratio_tuple[1], n_masked + 1)     # float32, as the contract's arithmetic
    upperbound = 0
    in_place = hasattr(mask_generator, "fill")
    for i in range(n_masked):
        target = int(n_tokens * random.uniform(edges[i], edges[i + 1]))
        if in_place:
            mask_generator.fill(grid[i], target)
        else:                       # any callable with the reference's interface
            grid[i] = mask_generator(target)
        upperbound += int(n_tokens * edges[i + 1])
    order = list(range(n_images))
    random.shuffle(order)
    return grid[order], upperbound


def collate_data_and_cast(samples_list, n_global_crops, n_local_crops, mask_ratio_tuple, mask_probability, dtype, n_tokens=None,
                          mask_generator=None) -> Dict:
    """List of ``([global crops..., local crops...], label)`` -> the batch dictionary (CPU tensors; ``DINOv2.forward`` moves them
    to the GPU).  Crops are stacked crop-major: every sample's first crop, then every sample's second crop, ..."""
    def crop_major(first: int, count: int) -> torch.Tensor:
        return torch.cat([torch.stack([crops[k] for crops, _ in samples_list]) for k in range(first, first + count)])

    global_crops = crop_major(0, n_global_crops)
    local_crops = crop_major(n_global_crops, n_local_crops)
    grid, upperbound = _draw_masks(len(global_crops), n_tokens, mask_ratio_tuple, mask_probability, mask_generator)
    masks = torch.from_numpy(np.ascontiguousarray(grid)).flatten(1)
    per_image = masks.sum(dim=1)
    weight_of_image = 1.0 / per_image.clamp(min=1).to(torch.float32)
    masked_at = torch.from_numpy(np.flatnonzero(grid))
    return {
        "images": {
            "collated_global_crops": global_crops.to(dtype),
            "collated_local_crops": local_crops.to(dtype),
            "collated_masks": masks,
            "mask_indices_list": masked_at,
            "masks_weight": torch.repeat_interleave(weight_of_image, per_image),
            "upperbound": upperbound,
            "n_masked_patches": torch.tensor([masked_at.numel()], dtype=torch.long),
        },
        "labels": torch.stack([label for _, label in samples_list]),
    }


def piecewise_cosine(base_value, final_value, total_iters, warmup_iters=0, start_warmup_value=0, freeze_iters=0) -> np.ndarray:
    """One value per iteration: ``freeze_iters`` zeros, a linear ramp ``start_warmup_value -> base_value`` over ``warmup_iters``,
    then half a cosine period from ``base_value`` down (or up) to ``final_value`` over the rest (float64)."""
    values = np.zeros(total_iters)
    ramp_end = freeze_iters + warmup_iters
    if ramp_end > total_iters:
        raise ValueError(f"freeze ({freeze_iters}) + warm-up ({warmup_iters}) exceed the schedule's {total_iters} iterations")
    values[freeze_iters:ramp_end] = np.linspace(start_warmup_value, base_value, warmup_iters)
    n = total_iters - ramp_end
    if n > 0:
        phase = np.pi * np.arange(n) / n
        values[ramp_end:] = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(phase))
    return values


class CosineScheduler:
    """Indexable schedule (``sched[it]``): ``piecewise_cosine`` inside its range, ``final_value`` beyond it.  ``schedule`` is the
    array itself (the trainer zeroes a prefix of the last layer's copy)."""

    def __init__(self, base_value, final_value, total_iters, warmup_iters=0, start_warmup_value=0, freeze_iters=0):
        self.final_value = final_value
        self.total_iters = total_iters
        self.schedule = piecewise_cosine(base_value, final_value, total_iters, warmup_iters, start_warmup_value, freeze_iters)

    def __getitem__(self, it):
        return self.schedule[it] if it < self.total_iters else self.final_value


def build_schedulers(*, lr, eta_min, warmup_epochs, weight_decay, momentum_teacher, final_momentum_teacher, warmup_teacher_temp,
                     teacher_temp, warmup_teacher_temp_epochs, freeze_last_layer_epochs, iters_per_epoch, total_iters):
    """(lr, weight decay, teacher momentum, teacher temperature, last-layer lr) as the trainer of the reference sets them up
    (trainer.py:7-54): weight decay anneals to 1e-4 whatever the configuration says; the teacher temperature ramps linearly
    and then stays; the last layer's learning rate is the backbone's with its first epochs zeroed."""
    def learning_rate():
        return CosineScheduler(base_value=lr, final_value=eta_min, total_iters=total_iters,
                               warmup_iters=warmup_epochs * iters_per_epoch, start_warmup_value=0)

    ramp = warmup_teacher_temp_epochs * iters_per_epoch
    last_layer = learning_rate()
    last_layer.schedule[:freeze_last_layer_epochs * iters_per_epoch] = 0
    return (learning_rate(),
            CosineScheduler(base_value=weight_decay, final_value=1e-4, total_iters=total_iters),
            CosineScheduler(base_value=momentum_teacher, final_value=final_momentum_teacher, total_iters=total_iters),
            CosineScheduler(base_value=teacher_temp, final_value=teacher_temp, total_iters=ramp, warmup_iters=ramp,
                            start_warmup_value=warmup_teacher_temp),
            last_layer)


def synthetic_samples(batch: int, global_size: int, local_size: int, n_local: int, generator: torch.Generator) -> Sequence:
    """N(0,1) crops in the shape the dataset's __getitem__ returns (post-Normalize statistics): for benches and tests."""
    return [([torch.randn(3, global_size, global_size, generator=generator) for _ in range(2)]
             + [torch.randn(3, local_size, local_size, generator=generator) for _ in range(n_local)], torch.tensor(0))
            for _ in range(batch)]
