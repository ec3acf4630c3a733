"""DINO CLS-token loss and iBOT patch loss on the HIP kernels (reference: self_supervised/dinov2/loss/
dino_clstoken_loss.py, ibot_patch_loss.py — same class names, methods, buffers and call conventions).

Both are cross-entropies between a centred / sharpened teacher distribution and the student's log-softmax over K
prototypes (65 536 in the shipped config): one fused pass per row (``apla_distill_ce``) gives the loss and the gradient
with respect to the student logits, so the [rows, K] log-softmax is never materialised; the teacher side is one
``apla_softmax_center`` pass.  The centre is an EMA of the teacher's mean output; with several processes its batch sum is
all-reduced exactly where the reference does it (``reduce_center_update``).  The Sinkhorn-Knopp alternative of the
reference is a handful of [rows, K] normalisations; it is kept in torch ops here (used only with `centering:
sinkhorn_knopp`).
"""
import torch
import torch.distributed as dist
from torch import nn

from .. import ops
from .._lib import check, lib


def softmax_center(x: torch.Tensor, center: torch.Tensor, temp: float) -> torch.Tensor:
    """softmax((x - center) / temp) over the last dim; x [..., K] fp32 / bf16 on the GPU, center broadcast [K]; fp32 out."""
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    ops._req(x2, None, "x", 2)
    c = center.reshape(-1).float().contiguous()
    if c.numel() != K:
        raise ValueError("softmax_center: center must have K entries")
    out = torch.empty(x2.shape[0], K, device=x.device, dtype=torch.float32)
    check(lib().apla_softmax_center(x2.data_ptr(), ops._DT[x2.dtype], x2.stride(0), c.data_ptr(), 1.0 / float(temp), out.data_ptr(),
                                    K, x2.shape[0], K, ops._stream()), "apla_softmax_center")
    return out.reshape(x.shape)


# fp16 only: the factor the trainer will multiply the loss by before backward (its loss scale).  The cross-entropy kernels write the
# gradient with respect to the student logits in the student's 16-bit type during FORWARD; entries of that gradient are weight / rows
# times a probability difference — most of them below fp16's smallest normal number — so they are written pre-multiplied by this factor
# (what GradScaler's scaled backward would have produced) and backward multiplies by (upstream gradient / factor), exactly 1 when the
# trainer's scale is the hint.  1.0 (bf16, fp32) changes nothing.
_PRESCALE = 1.0


class grad_prescale:
    def __init__(self, factor: float):
        self.factor = float(factor)

    def __enter__(self):
        global _PRESCALE
        self._old, _PRESCALE = _PRESCALE, self.factor
        return self

    def __exit__(self, *exc):
        global _PRESCALE
        _PRESCALE = self._old
        return False


class CenteredTeacher:
    """softmax((logits - center) / temp) of the teacher, NOT materialised: the cross-entropy kernel computes the targets on the fly
    (apla_distill_ce_centered).  What `softmax_center_teacher(..., lazy=True)` returns; `.probs()` gives the tensor the eager call
    would have returned.  `center` is the tensor the loss held at call time (the EMA update replaces it, never writes into it)."""

    def __init__(self, logits, center, temp):
        self.logits, self.center, self.temp = logits, center, float(temp)
        self.shape = logits.shape

    def probs(self):
        return softmax_center(self.logits, self.center, self.temp)

    def __getitem__(self, idx):   # row slices only (forward_masked: t[:n_masked_patches])
        return CenteredTeacher(self.logits[idx], self.center, self.temp)

    def squeeze(self, dim):
        return CenteredTeacher(self.logits.squeeze(dim), self.center, self.temp)

    @staticmethod
    def usable(logits):
        K = logits.shape[-1]
        return logits.is_cuda and logits.dtype in (ops.half(), torch.float32) and K % 8 == 0 and logits.stride(-1) == 1


def launch_distill_ce(s, target, temp, row_weight, weight, want_grad, ds=None, row_loss=None):
    """One launch of the fused cross-entropy over the rows of `s` ([..., K], 16-bit or fp32): `target` is a probability tensor of the
    same shape — or of R / n rows, which then repeat (row r reads target row r % (R / n): the local crops of DINOLoss.forward share
    their targets) — or a CenteredTeacher.  Returns (row_loss fp32 [R], ds or None); ds[r, :] = d(sum_r row_loss) / d s[r, :], in the
    student's 16-bit type when it has one (written into `ds` if given: a [R, K] row slice of a larger buffer), else fp32; `row_loss`:
    the caller's fp32 [R] slice to write the row losses into."""
    K = s.shape[-1]
    s2 = s.reshape(-1, K)
    s2 = s2 if s2.stride(-1) == 1 else s2.contiguous()
    ops._req(s2, None, "student", 2)
    R = s2.shape[0]
    rw = None
    if row_weight is not None:
        rw = row_weight.reshape(-1).float().contiguous()
        if rw.numel() != R:
            raise ValueError("distill_ce: one weight per row expected")
    centered = isinstance(target, CenteredTeacher)
    if centered:
        t2 = target.logits.reshape(-1, K)
        ops._req(t2, None, "teacher logits", 2)
        g16 = s2.dtype == ops.half()
    else:
        t2 = target.reshape(-1, K)
        t2 = t2 if t2.dtype == torch.float32 else t2.float()
        t2 = t2 if t2.stride(-1) == 1 else t2.contiguous()
        ops._req(t2, torch.float32, "teacher", 2)
        # the gradient in the student's own dtype (what backward returns anyway): a 16-bit student saves the fp32 round trip of a
        # [rows, 65 536] tensor (iBOT: 1.3 GB written, re-read, scaled and cast per iteration)
        g16 = s2.dtype == ops.half() and K % 8 == 0 and s2.stride(0) % 8 == 0 and t2.stride(0) % 4 == 0
    t_rows = t2.shape[0]
    if t2.shape[1] != K or t_rows == 0 or R % t_rows or (centered and t_rows != R):
        raise ValueError(f"distill_ce: student {tuple(s2.shape)} vs teacher {tuple(t2.shape)}")
    if ds is not None:
        if not g16 or ds.dtype != s2.dtype or tuple(ds.shape) != (R, K) or ds.stride(1) != 1:
            raise ValueError("distill_ce: the caller's gradient rows need the student's 16-bit type and shape")
    elif want_grad:
        ds = torch.empty(R, K, device=s.device, dtype=s2.dtype if g16 else torch.float32)
    if row_loss is None:
        row_loss = torch.empty(R, device=s.device, dtype=torch.float32)
    elif row_loss.dtype != torch.float32 or tuple(row_loss.shape) != (R,) or not row_loss.is_contiguous():
        raise ValueError("distill_ce: the caller's row losses must be a contiguous fp32 [rows] slice")
    ds_dt = ops._DT[ds.dtype] if ds is not None else ops._DT[torch.float32]
    ld_ds = ds.stride(0) if ds is not None else K
    if centered:
        c = target.center.reshape(-1).float().contiguous()
        check(lib().apla_distill_ce_centered(s2.data_ptr(), ops._DT[s2.dtype], s2.stride(0), t2.data_ptr(), ops._DT[t2.dtype], t2.stride(0),
                                             c.data_ptr(), 1.0 / float(temp), 1.0 / target.temp, ops._ptr(rw), float(weight), ops._ptr(ds),
                                             ds_dt, ld_ds, row_loss.data_ptr(), R, K, ops._stream()), "apla_distill_ce_centered")
    else:
        check(lib().apla_distill_ce_bcast(s2.data_ptr(), ops._DT[s2.dtype], s2.stride(0), t2.data_ptr(), t2.stride(0), t_rows,
                                          1.0 / float(temp), ops._ptr(rw), float(weight), ops._ptr(ds), ds_dt, ld_ds, 0, row_loss.data_ptr(),
                                          R, K, ops._stream()), "apla_distill_ce")
    return row_loss, ds


class _DistillCE(torch.autograd.Function):
    """sum_r w_r * CE(t_r, softmax(s_r / temp)) with the gradient wrt s produced in the same pass; `t` a probability tensor or a
    CenteredTeacher (then one kernel, no [rows, K] probability tensor)."""

    @staticmethod
    def forward(ctx, s, t, temp, row_weight, weight):
        P = _PRESCALE
        row_loss, ds = launch_distill_ce(s, t, temp, row_weight, weight * P, s.requires_grad)
        ctx.save_for_backward(ds)
        ctx.meta = (s.shape, s.dtype, P)
        total = row_loss.sum()
        return total if P == 1.0 else total / P

    @staticmethod
    def backward(ctx, g):
        (ds,) = ctx.saved_tensors
        shape, dtype, P = ctx.meta
        if P != 1.0:
            g = g / P
        if ds.dtype == dtype:   # same-dtype product (a bf16 tensor times an fp32 0-dim tensor takes torch's slow mixed-type kernel)
            return (ds * g.to(dtype)).reshape(shape), None, None, None, None
        return (ds * g).reshape(shape).to(dtype), None, None, None, None


def distill_ce(student, teacher_probs, temp, row_weight=None, weight=1.0):
    return _DistillCE.apply(student, teacher_probs, temp, row_weight, weight)


class _CenteredLoss(nn.Module):
    def _init_center_state(self, center_momentum):
        self.center_momentum = center_momentum
        self.updated = True
        self.reduce_handle = None
        self.async_batch_center = None

    @torch.no_grad()
    def _apply_center(self, n_rows):
        if self.updated is False:
            world = dist.get_world_size() if dist.is_initialized() else 1
            if self.reduce_handle is not None:
                self.reduce_handle.wait()
            _t = self.async_batch_center / (n_rows * world)
            self.center = self.center * self.center_momentum + _t * (1 - self.center_momentum)
            self.updated = True

    @torch.no_grad()
    def sinkhorn_knopp_teacher(self, teacher_output, teacher_temp, n_iterations=3, n_masked_patches_tensor=None):
        """dino_clstoken_loss.py:34-63 / ibot_patch_loss.py:57-86 (torch ops; normalisations of a [K, rows] matrix)."""
        Q = torch.exp(teacher_output.float() / teacher_temp).t()
        world = dist.get_world_size() if dist.is_initialized() else 1
        if n_masked_patches_tensor is not None:
            B = n_masked_patches_tensor.clone()
            if dist.is_initialized():
                dist.all_reduce(B)
        else:
            B = Q.shape[1] * world
        K = Q.shape[0]
        sum_Q = torch.sum(Q)
        if dist.is_initialized():
            dist.all_reduce(sum_Q)
        Q /= sum_Q
        for _ in range(n_iterations):
            rows = torch.sum(Q, dim=1, keepdim=True)
            if dist.is_initialized():
                dist.all_reduce(rows)
            Q /= rows
            Q /= K
            Q /= torch.sum(Q, dim=0, keepdim=True)
            Q /= B
        Q *= B
        return Q.t()


class DINOLoss(_CenteredLoss):
    def __init__(self, out_dim, student_temp=0.1, center_momentum=0.9):
        super().__init__()
        self.student_temp = student_temp
        self.register_buffer("center", torch.zeros(1, out_dim))
        self.len_teacher_output = None
        self._init_center_state(center_momentum)

    @torch.no_grad()
    def softmax_center_teacher(self, teacher_output, teacher_temp):
        self.apply_center_update()
        return softmax_center(teacher_output, self.center, teacher_temp)

    def forward(self, student_output_list, teacher_out_softmaxed_centered_list):
        """-sum_s sum_t mean_rows( sum_k t * log_softmax(s / T_s) ) (dino_clstoken_loss.py:65-77): the inner sum over teacher
        views is a sum of targets, so each student view costs one fused pass."""
        tsum = teacher_out_softmaxed_centered_list[0].float()
        for t in teacher_out_softmaxed_centered_list[1:]:
            tsum = tsum + t.float()
        total = 0
        for s in student_output_list:
            total = total + distill_ce(s, tsum, self.student_temp, None, 1.0 / s.shape[0])
        return total

    @torch.no_grad()
    def update_center(self, teacher_output):
        self.reduce_center_update(teacher_output)

    @torch.no_grad()
    def reduce_center_update(self, teacher_output):
        self.updated = False
        self.len_teacher_output = len(teacher_output)
        self.async_batch_center = torch.sum(teacher_output.float(), dim=0, keepdim=True)
        if dist.is_initialized():
            self.reduce_handle = dist.all_reduce(self.async_batch_center, async_op=True)

    @torch.no_grad()
    def apply_center_update(self):
        self._apply_center(self.len_teacher_output)


class iBOTPatchLoss(_CenteredLoss):
    def __init__(self, patch_out_dim, student_temp=0.1, center_momentum=0.9):
        super().__init__()
        self.student_temp = student_temp
        self.register_buffer("center", torch.zeros(1, 1, patch_out_dim))
        self.len_teacher_patch_tokens = None
        self._init_center_state(center_momentum)

    @torch.no_grad()
    def softmax_center_teacher(self, teacher_patch_tokens, teacher_temp, lazy=False):
        """ibot_patch_loss.py:46-55.  lazy=True: the targets stay (logits, centre, temperature) and forward / forward_masked compute
        them inside the cross-entropy kernel (same values; the [rows, 65 536] fp32 tensor is never written or re-read)."""
        self.apply_center_update()
        if lazy and CenteredTeacher.usable(teacher_patch_tokens):
            return CenteredTeacher(teacher_patch_tokens, self.center, teacher_temp)
        return softmax_center(teacher_patch_tokens, self.center, teacher_temp)

    def forward(self, student_patch_tokens, teacher_patch_tokens, student_masks_flat):
        """(B, N, K) tokens, (B, N) mask: masked mean of the per-token cross-entropies, then batch mean (:88-101)."""
        m = student_masks_flat.float()
        w = m / m.sum(dim=-1, keepdim=True).clamp(min=1.0)
        return distill_ce(student_patch_tokens, teacher_patch_tokens, self.student_temp, w, 1.0 / student_masks_flat.shape[0])

    def forward_masked(self, student_patch_tokens_masked, teacher_patch_tokens_masked, student_masks_flat, n_masked_patches=None,
                       masks_weight=None):
        """Only the masked tokens, already gathered: rows weighted by masks_weight, divided by the batch size (:103-121)."""
        s, t = student_patch_tokens_masked, teacher_patch_tokens_masked
        if masks_weight is None:
            masks_weight = (1 / student_masks_flat.sum(-1).clamp(min=1.0)).unsqueeze(-1).expand_as(student_masks_flat)[student_masks_flat]
        if n_masked_patches is not None:
            s, t = s[:n_masked_patches], t[:n_masked_patches]
        return distill_ce(s, t, self.student_temp, masks_weight, 1.0 / student_masks_flat.shape[0])

    @torch.no_grad()
    def update_center(self, teacher_patch_tokens):
        self.reduce_center_update(teacher_patch_tokens)

    @torch.no_grad()
    def reduce_center_update(self, teacher_patch_tokens):
        self.updated = False
        self.len_teacher_patch_tokens = len(teacher_patch_tokens)
        x = teacher_patch_tokens
        if x.dim() == 3 and x.shape[0] == 1 and x.dtype == ops.half() and x.is_cuda and x.shape[-1] % 4 == 0 and x.stride(-1) == 1 and x.stride(1) % 4 == 0:
            # one fp32-accumulating pass over the 16-bit rows (torch: a float copy of the [~5000, 65536] tensor, then a mean kernel)
            self.async_batch_center = (ops.colsum(x[0]) / x.shape[1]).unsqueeze(0)
        else:
            self.async_batch_center = torch.sum(x.float().mean(1), dim=0, keepdim=True)
        if dist.is_initialized():
            self.reduce_handle = dist.all_reduce(self.async_batch_center, async_op=True)

    @torch.no_grad()
    def apply_center_update(self):
        self._apply_center(self.len_teacher_patch_tokens)
