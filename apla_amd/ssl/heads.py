"""DINO projection head, KoLeo regulariser and the EMA teacher update of the DINOv2-APLA step.

* ``DINOHead`` — self_supervised/dinov2/layers/dino_head.py:12-40: 3-layer MLP (GELU), L2-normalised bottleneck,
  weight-normalised prototype layer with the norm frozen at 1.  Same parameter names (``mlp.{0,2,4}.{weight,bias}``,
  ``last_layer.weight_g / weight_v``).  The products run on the HIP GEMM (`apla_gemm_nt`); the weight gradients of the
  three MLP layers and the [K, 256] prototype gradient (the few thousand token rows are its reduction axis) on the TN MFMA kernel
  (`apla_proj_dw`: dW = dy^T x with fp32 accumulation; round 1 ran the prototype gradient as an fp32 vendor GEMM, 1.3 ms per
  iteration).
* ``KoLeoLoss`` — dinov2/loss/koleo_loss.py:17-45 (a nearest-neighbour search among the CLS tokens of one crop, fp32): one HIP launch
  forward, one backward (`apla_koleo_fwd` / `apla_koleo_bwd`), both crops of models.py:410-413 in the same launch.
* ``update_teacher`` — dinov2/models.py:443-453: teacher = m * teacher + (1 - m) * student over the trainable tensors.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .. import functional as AF
from .. import ops


def _normed_weight(v, g, transposed=False):
    """(W 16-bit, row norms | None, W^T 16-bit | None) of the weight-normalised prototype layer: one kernel (apla_weight_norm_fwd[_t])
    where it applies.  W^T is what the dX GEMM reads; without it the backward transposes W itself."""
    if v.is_cuda and v.dtype == torch.float32 and v.ndim == 2 and v.shape[1] % 4 == 0 and v.is_contiguous():
        if transposed:
            return ops.weight_norm_fwd(v.detach(), g.detach().reshape(-1).contiguous(), transposed=True)
        return (*ops.weight_norm_fwd(v.detach(), g.detach().reshape(-1).contiguous()), None)
    W = v.detach() * (g.detach() / v.detach().norm(dim=1, keepdim=True))
    return W.to(ops.half()).contiguous(), None, None


def _normed_weight_grads(dW, v, g, norm, need_v, need_g):
    """(dv, dg) from dW = d(loss) / dW (fp32): apla_weight_norm_bwd, or torch's formulas where the kernel does not apply."""
    if norm is not None:
        dv, dg = ops.weight_norm_bwd(dW.contiguous(), v.detach(), g.detach().reshape(-1).contiguous(), norm, want_dg=need_g)
        return (dv if need_v else None), (dg.reshape(g.shape) if need_g else None)
    vd, gd = v.detach(), g.detach()
    n = vd.norm(dim=1, keepdim=True)
    dot = (dW * vd).sum(dim=1, keepdim=True)
    return ((gd / n) * dW - vd * (gd * dot / n ** 3)) if need_v else None, (dot / n).reshape(g.shape) if need_g else None


class _ProtoLinear(torch.autograd.Function):
    """y = x @ W^T for the weight-normalised [K, 256] prototype matrix W = v g / ||v||: W built in 16 bits by one kernel, forward and
    dX on the MFMA GEMM, dW = dy^T x on the TN MFMA kernel, (dv, dg) from dW by one kernel."""

    @staticmethod
    def forward(ctx, x, v, g):
        x2 = AF._as2d_bf16(x)
        Wh, norm, WhT = _normed_weight(v, g, transposed=x.requires_grad)
        y = ops.gemm_nt(x2, Wh)
        ctx.save_for_backward(x2, Wh, v, g, norm, WhT)
        ctx.shape = x.shape
        return y.reshape(*x.shape[:-1], Wh.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, Wh, v, g, norm, WhT = ctx.saved_tensors
        dy2 = AF._as2d_bf16(dy)
        dx = dv = dg = None
        if ctx.needs_input_grad[0]:   # [rows, 256] = dy [rows, K] W [K, 256]: 35 tiles with a 65 536-long reduction -> split along K
            if WhT is None:
                WhT = Wh.t().contiguous()
            gemm = ops.gemm_nt_splitk if ops.gemm_splitk_wanted(dy2.shape[0], WhT.shape[0], WhT.shape[1]) else ops.gemm_nt
            dx = gemm(dy2, WhT).reshape(ctx.shape)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            K, Din = Wh.shape
            if K % 64 == 0 and Din % 128 == 0:
                dW = torch.empty(K, Din, device=dy2.device, dtype=torch.float32)
                ops.proj_dw(dy2, x2, dW, torch.empty(K, device=dy2.device, dtype=torch.float32))
            else:   # shapes the TN kernel does not take
                dW = torch.mm(dy2.float().t(), x2.float())
            dv, dg = _normed_weight_grads(dW, v, g, norm, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dv, dg


class _ProtoLosses(torch.autograd.Function):
    """Prototype layer + every cross-entropy term that reads its output, as ONE autograd node: y = x W^T, then per term one fused
    cross-entropy launch over a row range of y (``losses.launch_distill_ce``) that writes d(term)/dy straight into that range of ONE
    [rows, K] gradient buffer.  Backward is two GEMMs on that buffer (dx = dy W, dW = dy^T x) with the upstream scalars applied to
    the ROWS of the small operands (x and dx, [rows, 256]).  The module route (``_ProtoLinear`` + one ``_DistillCE`` per term) passes
    the [rows, 65 536] gradient through torch four more times per iteration — the scalar multiply of each term, the zero-padded
    copy behind the row slice, the concatenation behind ``split`` — 1.3 ms at config 4.

    terms: tuples (out_index, row_begin, row_end, target, temp, row_weight, weight); rows no term covers get a zero gradient.
    Returns a [n_out] fp32 tensor: out[i] = sum of the terms with out_index i."""

    @staticmethod
    def forward(ctx, x, v, g, n_out, terms):
        from .losses import _PRESCALE, launch_distill_ce
        P = _PRESCALE
        x2 = AF._as2d_bf16(x)
        Wh, norm, WhT = _normed_weight(v, g, transposed=x.requires_grad)
        y = ops.gemm_nt(x2, Wh)
        rows, K = y.shape
        need = x.requires_grad or v.requires_grad or g.requires_grad
        dy = torch.empty_like(y) if need else None
        row_loss = torch.zeros(rows, device=y.device, dtype=torch.float32)    # every term writes its rows; rows no term covers stay 0
        spans, covered = [], 0
        for (oi, a, b, target, temp, rw, weight) in sorted(terms, key=lambda t: t[1]):
            if a < covered or b > rows or a >= b or not 0 <= oi < n_out:
                raise ValueError("proto_losses: terms must cover disjoint row ranges")
            if need and a > covered:
                dy[covered:a].zero_()
            launch_distill_ce(y[a:b], target, temp, rw, weight * P, need, ds=dy[a:b] if need else None, row_loss=row_loss[a:b])
            spans.append((oi, a, b))
            covered = b
        if need and covered < rows:
            dy[covered:].zero_()
        idx, member = _row_term_index(rows, tuple(spans), n_out, y.device)
        ctx.save_for_backward(x2, Wh, dy, idx, v, g, norm, WhT)
        ctx.meta = (x.shape, P)
        res = (member * row_loss).sum(dim=1)     # [n_out]: one masked row sum per output, in a fixed order
        return res if P == 1.0 else res / P

    @staticmethod
    def backward(ctx, gout):
        x2, Wh, dy, idx, v, g, norm, WhT = ctx.saved_tensors
        shape, P = ctx.meta
        r = (gout if P == 1.0 else gout / P).float().index_select(0, idx).unsqueeze(1)   # [rows, 1]: the upstream scalar of each row's term
        dx = dv = dg = None
        if ctx.needs_input_grad[0]:   # [rows, 256] = dy [rows, K] W [K, 256]: a 65 536-long reduction -> split along K
            if WhT is None:
                WhT = Wh.t().contiguous()
            gemm = ops.gemm_nt_splitk if ops.gemm_splitk_wanted(dy.shape[0], WhT.shape[0], WhT.shape[1]) else ops.gemm_nt
            dx = (gemm(dy, WhT).float() * r).reshape(shape)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            K, Din = Wh.shape
            xr = (x2.float() * r).to(x2.dtype)
            if K % 64 == 0 and Din % 128 == 0:
                dW = torch.empty(K, Din, device=dy.device, dtype=torch.float32)
                ops.proj_dw(dy, xr, dW, torch.empty(K, device=dy.device, dtype=torch.float32))
            else:
                dW = torch.mm(dy.float().t(), xr.float())
            dv, dg = _normed_weight_grads(dW, v, g, norm, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, dv, dg, None, None


_ROW_INDEX_CACHE = {}


def _row_term_index(rows, spans, n_out, device):
    """(int64 [rows]: the output index of the term covering each row — 0 for uncovered rows: their gradient is zero anyway —,
    fp32 [n_out, rows]: 1 where the row belongs to the output, else 0).  Cached on the device per layout (the number of masked patches
    varies from batch to batch; the cache is small and bounded)."""
    key = (rows, spans, n_out, str(device))
    ent = _ROW_INDEX_CACHE.get(key)
    if ent is None:
        if len(_ROW_INDEX_CACHE) >= 256:
            _ROW_INDEX_CACHE.clear()
        host = torch.zeros(rows, dtype=torch.long)
        member = torch.zeros(n_out, rows, dtype=torch.float32)
        for oi, a, b in spans:
            host[a:b] = oi
            member[oi, a:b] = 1.0
        ent = _ROW_INDEX_CACHE[key] = (host.to(device), member.to(device))
    return ent


def proto_losses(x, v, g, n_out, terms):
    """``v``, ``g``: the weight-norm parameters of the prototype layer (``last_layer.weight_v`` / ``weight_g``)."""
    return _ProtoLosses.apply(x, v, g, n_out, tuple(terms))


class DINOHead(nn.Module):
    def __init__(self, in_dim, out_dim, use_bn=False, nlayers=3, hidden_dim=2048, bottleneck_dim=256, mlp_bias=True):
        super().__init__()
        if use_bn:
            raise NotImplementedError("use_bn is false in the shipped configs")
        nlayers = max(nlayers, 1)
        if nlayers == 1:
            self.mlp = nn.Linear(in_dim, bottleneck_dim, bias=mlp_bias)
        else:
            layers = [nn.Linear(in_dim, hidden_dim, bias=mlp_bias), nn.GELU()]
            for _ in range(nlayers - 2):
                layers += [nn.Linear(hidden_dim, hidden_dim, bias=mlp_bias), nn.GELU()]
            layers.append(nn.Linear(hidden_dim, bottleneck_dim, bias=mlp_bias))
            self.mlp = nn.Sequential(*layers)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        self.last_layer = nn.utils.weight_norm(nn.Linear(bottleneck_dim, out_dim, bias=False))
        self.last_layer.weight_g.data.fill_(1)

    def bottleneck(self, x):
        """The MLP and the L2 normalisation (dino_head.py:33-37): what the prototype layer reads."""
        mods = [self.mlp] if isinstance(self.mlp, nn.Linear) else list(self.mlp)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.GELU):
                x = AF.linear_gelu(x, m.weight, m.bias)        # the activation rides in the GEMM epilogue
                i += 2
                continue
            x = AF.linear(x, m.weight, m.bias) if isinstance(m, nn.Linear) else F.gelu(x.float()).to(x.dtype)
            i += 1
        return F.normalize(x.float(), dim=-1, p=2, eps=1e-12)

    def prototype_weight(self):
        v, g = self.last_layer.weight_v, self.last_layer.weight_g
        return v * (g / v.norm(dim=1, keepdim=True))           # torch.nn.utils.weight_norm, dim=0

    def forward(self, x):
        return _ProtoLinear.apply(self.bottleneck(x), self.last_layer.weight_v, self.last_layer.weight_g)


class _KoLeoFn(torch.autograd.Function):
    """Sum over `groups` equal row groups of KoLeoLoss: apla_koleo_fwd / apla_koleo_bwd, one launch each."""

    @staticmethod
    def forward(ctx, x, groups, eps):
        x = x.contiguous()
        out, saved = ops.koleo_fwd(x, groups, eps)
        ctx.save_for_backward(x, *saved)
        ctx.groups, ctx.eps = groups, eps
        return out[groups]

    @staticmethod
    def backward(ctx, g):
        x, *saved = ctx.saved_tensors
        return ops.koleo_bwd(x, ctx.groups, saved, g.float().reshape(1), ctx.eps), None, None


class KoLeoLoss(nn.Module):
    """Kozachenko-Leonenko entropic regulariser (Sablayrolles et al. 2018), koleo_loss.py:17-45: normalise, nearest other row by inner
    product, -mean log of the distance to it.  On the GPU one HIP launch forward and one backward (apla_koleo_fwd: torch runs ~60 small
    kernels for the two crops of an iteration); ``grouped(x, n)`` = sum(self(c) for c in x.chunk(n)) as models.py:410-413 calls it."""

    def grouped(self, student_output, groups: int, eps=1e-8):
        if student_output.shape[0] % groups:
            raise ValueError("KoLeoLoss.grouped: the rows do not split into equal groups")
        x = student_output if student_output.dtype in (torch.float32, ops.half()) else student_output.float()
        return _KoLeoFn.apply(x, groups, eps)

    def forward(self, student_output, eps=1e-8):
        return self.grouped(student_output, 1, eps)


@torch.no_grad()
def update_teacher(student: nn.Module, teacher: nn.Module, m: float):
    """EMA of the student's TRAINABLE tensors into the teacher (dinov2/models.py:443-453 walks the same-named parameter
    lists and uses torch._foreach_mul_/_foreach_add_); frozen tensors are identical in both and stay untouched."""
    sp = dict(student.named_parameters())
    t_list, s_list = [], []
    for name, tp in teacher.named_parameters():
        if name in sp and sp[name].requires_grad:
            t_list.append(tp)
            s_list.append(sp[name].detach())
    if t_list:
        torch._foreach_mul_(t_list, m)
        torch._foreach_add_(t_list, s_list, alpha=1 - m)
    return len(t_list)
