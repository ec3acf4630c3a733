"""CPU oracle of the DINOv2-APLA training iteration — TEST INFRASTRUCTURE ONLY (imported by tests/ alone; the product never
touches it).  A from-the-formulas restatement in plain PyTorch-CPU float64 with functional ops and torch.autograd, each
piece citing the reference lines it follows.  PINNED: tests/test_ssl_cpu.py::test_ssl_oracle_reproduces_reference_step
checks it against golden G12 (two iterations run by the reference's own classes, tests/golden/make_golden.py:g12_ssl_step)
— losses, clipped gradients, updated student / teacher parameters and centres, for both adaptation modes.

Parameters are plain dicts ``name -> tensor`` in the reference's state_dict naming (``backbone.blocks.i.attn.proj_weight1``,
``dino_head.last_layer.weight_v`` …).  Attention is dense per crop: the reference's packed block-diagonal pass computes
exactly this (attention never crosses crops), and xformers, which it needs for the packed form, is not installed
 — the nested boundary itself stays unpinned, as recorded there.
"""
import math

import torch
import torch.nn.functional as F

DT = torch.float64


# ------------------------------------------------------------------------------------------------ backbone
def _pos_embed(p, npatch, offset=0.1):
    """dinov2_vits.py:176-208 (square inputs, interpolate_offset 0.1, no antialias)."""
    pe = p["backbone.pos_embed"]
    N = pe.shape[1] - 1
    if npatch == N:
        return pe
    dim, M, side = pe.shape[-1], int(math.sqrt(N)), int(math.sqrt(npatch))
    sf = float(side + offset) / M
    grid = F.interpolate(pe[:, 1:].reshape(1, M, M, dim).permute(0, 3, 1, 2), mode="bicubic", antialias=False, scale_factor=(sf, sf))
    assert grid.shape[-1] == side
    return torch.cat((pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)


def _proj(p, pre, o, D):
    """APLA projection (appla_attn_mem_eff.py:44-63): two linears scattered to their output columns; or the plain Linear
    when partial_size is 'full' (apla_vit.py:66-75)."""
    if pre + "proj.weight" in p:
        return o @ p[pre + "proj.weight"].t() + p[pre + "proj.bias"]
    inds = p[pre + "inds"].long()
    r = p[pre + "proj_weight1"].shape[0]
    y = torch.empty(o.shape, dtype=o.dtype)
    y[..., inds[:r]] = o @ p[pre + "proj_weight1"].t() + p[pre + "proj_bias1"]
    y[..., inds[r:]] = o @ p[pre + "proj_weight2"].t() + p[pre + "proj_bias2"]
    return y


def backbone(p, images, masks, cfg):
    """prepare_tokens_with_masks + blocks + final norm (dinov2_vits.py:210-288; block.py:104-140; attention.py:53-64)."""
    D, depth, H, patch = cfg["D"], cfg["depth"], cfg["heads"], cfg["patch"]
    B = images.shape[0]
    cols = F.unfold(images, kernel_size=patch, stride=patch).transpose(1, 2)                 # [B, Np, 3*p*p]
    x = cols @ p["backbone.patch_embed.proj.weight"].reshape(D, -1).t() + p["backbone.patch_embed.proj.bias"]
    if masks is not None:
        x = torch.where(masks.unsqueeze(-1), p["backbone.mask_token"].unsqueeze(0), x)
    x = torch.cat((p["backbone.cls_token"].expand(B, -1, -1), x), dim=1)
    x = x + _pos_embed(p, x.shape[1] - 1)
    N, d = x.shape[1], D // H
    for i in range(depth):
        pre = f"backbone.blocks.{i}."
        h = F.layer_norm(x, (D,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], 1e-6)
        qkv = (h @ p[pre + "attn.qkv.weight"].t() + p[pre + "attn.qkv.bias"]).reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4)
        att = torch.softmax((qkv[0] * d ** -0.5) @ qkv[1].transpose(-2, -1), dim=-1)
        o = (att @ qkv[2]).transpose(1, 2).reshape(B, N, D)
        x = x + _proj(p, pre + "attn.", o, D) * p[pre + "ls1.gamma"]
        h = F.layer_norm(x, (D,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], 1e-6)
        h = F.gelu(h @ p[pre + "mlp.fc1.weight"].t() + p[pre + "mlp.fc1.bias"]) @ p[pre + "mlp.fc2.weight"].t() + p[pre + "mlp.fc2.bias"]
        x = x + h * p[pre + "ls2.gamma"]
    xn = F.layer_norm(x, (D,), p["backbone.norm.weight"], p["backbone.norm.bias"], 1e-6)
    return xn[:, 0], xn[:, 1:]


def dino_head(p, x):
    """dino_head.py:12-40: MLP (GELU), L2 normalisation (eps 1e-12), weight-normalised prototypes (torch weight_norm, dim 0)."""
    for k in (0, 2):
        x = F.gelu(x @ p[f"dino_head.mlp.{k}.weight"].t() + p[f"dino_head.mlp.{k}.bias"])
    x = x @ p["dino_head.mlp.4.weight"].t() + p["dino_head.mlp.4.bias"]
    x = F.normalize(x, dim=-1, p=2, eps=1e-12)
    v, g = p["dino_head.last_layer.weight_v"], p["dino_head.last_layer.weight_g"]
    return x @ (v * (g / v.norm(dim=1, keepdim=True))).t()


# ------------------------------------------------------------------------------------------------ losses
def _dino(student_list, teacher_list, temp=0.1):
    """dino_clstoken_loss.py:65-77."""
    total = 0
    for s in student_list:
        lsm = F.log_softmax(s / temp, dim=-1)
        for t in teacher_list:
            total = total - torch.sum(t * lsm, dim=-1).mean()
    return total


def _koleo(x, eps=1e-8):
    """koleo_loss.py:17-45."""
    x = F.normalize(x, eps=eps, p=2, dim=-1)
    dots = (x @ x.t()).detach().clone()
    dots.fill_diagonal_(-1)
    nn_idx = dots.argmax(dim=1)
    return -torch.log(F.pairwise_distance(x, x[nn_idx], 2, eps=1e-8) + eps).mean()


def _center_apply(c):
    if c["pending"] is not None:    # apply_center_update: dino_clstoken_loss.py:90-101 / ibot_patch_loss.py:140-152
        c["value"] = c["value"] * 0.9 + c["pending"] * 0.1
        c["pending"] = None


# ------------------------------------------------------------------------------------------------ state / batches
def state_from_golden(g):
    D, depth, heads, patch, pre, gsz, lsz, K, hid, bott, B, n_local = [int(v) for v in g["meta"]]
    init = {k[len("init."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("init.")}
    conv = lambda v: v.to(DT) if v.is_floating_point() else v.clone()   # noqa: E731
    trainable = [str(n) for n in g["trainable"]]
    student = {k: conv(v) for k, v in init.items()}
    teacher = {k: conv(v) for k, v in init.items()}
    return dict(cfg=dict(D=D, depth=depth, heads=heads, patch=patch, K=K, n_local=n_local), student=student, teacher=teacher,
                trainable=trainable, adam={n: dict(step=0, m=torch.zeros_like(student[n]), v=torch.zeros_like(student[n])) for n in trainable},
                dino_center=dict(value=torch.zeros(1, K, dtype=DT), pending=None),
                ibot_center=dict(value=torch.zeros(1, 1, K, dtype=DT), pending=None))


def state_from_state_dict(sd, trainable, cfg):
    """Oracle state from a student state_dict in the reference's naming (student.backbone.* / dino_head.* without the
    'student.' prefix); the teacher starts as a copy (models.py:175-177)."""
    conv = lambda v: v.detach().cpu().to(DT) if v.is_floating_point() else v.detach().cpu().clone()   # noqa: E731
    student = {k: conv(v) for k, v in sd.items()}
    teacher = {k: v.clone() for k, v in student.items()}
    K = cfg["K"]
    return dict(cfg=cfg, student=student, teacher=teacher, trainable=list(trainable),
                adam={n: dict(step=0, m=torch.zeros_like(student[n]), v=torch.zeros_like(student[n])) for n in trainable},
                dino_center=dict(value=torch.zeros(1, K, dtype=DT), pending=None),
                ibot_center=dict(value=torch.zeros(1, 1, K, dtype=DT), pending=None))


def batch_from_collate(images):
    """The collate's dictionary (apla_amd.ssl.collate_data_and_cast(...)["images"]) in the oracle's form."""
    return dict(glob=images["collated_global_crops"].to(DT), loc=images["collated_local_crops"].to(DT), masks=images["collated_masks"],
                idx=images["mask_indices_list"].long(), masks_weight=images["masks_weight"].to(DT), upperbound=int(images["upperbound"]))


def batch_from_golden(g, it):
    f = lambda k: torch.from_numpy(g[f"it{it}.{k}"])   # noqa: E731
    return dict(glob=f("glob").to(DT), loc=f("loc").to(DT), masks=f("masks"), idx=f("mask_indices").long(),
                masks_weight=f("masks_weight").to(DT), upperbound=int(g[f"it{it}.upperbound"]))


def centers(st):
    _center_apply(st["dino_center"]), _center_apply(st["ibot_center"])
    return dict(dino=st["dino_center"]["value"], ibot=st["ibot_center"]["value"])


# ------------------------------------------------------------------------------------------------ one iteration
def train_iteration(st, batch, hyper, clip, freeze_last, koleo_w=0.1, dino_w=1.0, ibot_w=1.0, betas=(0.9, 0.999), eps=1e-8):
    """models.py:207-441 (forward), trainer.py:106-141 (clip, cancel last layer, AdamW with the two groups of
    defaults/wrappers.py:205-221), models.py:443-453 (EMA).  hyper = [lr, wd, teacher_temp, momentum]."""
    lr, wd, ttemp, mom = hyper
    cfg, S, T = st["cfg"], st["student"], st["teacher"]
    K, n_local, n_global = cfg["K"], cfg["n_local"], 2
    glob, loc, masks, idx, mw, upper = batch["glob"], batch["loc"], batch["masks"], batch["idx"], batch["masks_weight"], batch["upperbound"]
    n_masked = idx.shape[0]
    with torch.no_grad():
        tcls, tpatch = backbone(T, glob, None, cfg)
        a, b = tcls.chunk(2)
        tcls = torch.cat((b, a))
        n_cls = tcls.shape[0]
        buf = torch.zeros(upper + n_cls, cfg["D"], dtype=DT)
        buf[:n_cls] = tcls
        buf[n_cls:n_cls + n_masked] = tpatch.flatten(0, 1)[idx]
        after = dino_head(T, buf)
        tcls_h, tpatch_h = after[:n_cls], after[n_cls:n_cls + n_masked]
        _center_apply(st["dino_center"])
        t_dino = torch.softmax((tcls_h - st["dino_center"]["value"]) / ttemp, dim=-1).view(2, -1, K)
        st["dino_center"]["pending"] = tcls_h.sum(0, keepdim=True) / len(tcls_h)
        _center_apply(st["ibot_center"])
        t_ibot = torch.softmax((tpatch_h.unsqueeze(0) - st["ibot_center"]["value"]) / ttemp, dim=-1).squeeze(0)
        st["ibot_center"]["pending"] = tpatch_h.unsqueeze(0).mean(1).sum(0, keepdim=True) / 1
    for n in st["trainable"]:
        S[n] = S[n].detach().requires_grad_(True)
    g_cls, g_patch = backbone(S, glob, masks, cfg)
    l_cls, _ = backbone(S, loc, None, cfg)
    pbuf = torch.zeros(upper, cfg["D"], dtype=DT)
    pbuf = torch.cat((g_patch.flatten(0, 1)[idx], pbuf[n_masked:]))
    outs = dino_head(S, torch.cat((l_cls, g_cls, pbuf)))
    o_loc, o_glob, o_patch = outs[:len(l_cls)], outs[len(l_cls):len(l_cls) + len(g_cls)], outs[len(l_cls) + len(g_cls):][:n_masked]
    terms = (n_global - 1) * n_global + max(n_local * n_global, 1)
    ld = {"dino_local_crops_loss": _dino(o_loc.chunk(n_local), list(t_dino)) / terms}
    ld["dino_global_crops_loss"] = _dino([o_glob], [t_dino.flatten(0, 1)]) * 2 / terms
    kl = koleo_w * sum(_koleo(c) for c in g_cls.chunk(2))
    ld["koleo_loss"] = kl / 2
    ib = -(torch.sum(t_ibot * F.log_softmax(o_patch / 0.1, dim=-1), dim=-1) * mw).sum() / masks.shape[0] * 2 * (1.0 / n_global)
    ld["ibot_loss"] = ib / 2
    total = dino_w * (ld["dino_local_crops_loss"] + ld["dino_global_crops_loss"]) + kl + ibot_w * ib
    grads = dict(zip(st["trainable"], torch.autograd.grad(total, [S[n] for n in st["trainable"]])))
    gnorm = torch.sqrt(sum((g_ ** 2).sum() for g_ in grads.values()))
    coef = min(1.0, clip / (float(gnorm) + 1e-6)) if clip else 1.0
    grads = {n: g_ * coef for n, g_ in grads.items()}
    with torch.no_grad():
        for n in st["trainable"]:
            if freeze_last and "dino_head.last_layer" in n:
                S[n] = S[n].detach()
                continue
            a_ = st["adam"][n]
            a_["step"] += 1
            pw = S[n].detach()
            if not (n.endswith(".bias") or pw.ndim == 1):
                pw = pw * (1 - lr * wd)
            a_["m"] = a_["m"] * betas[0] + grads[n] * (1 - betas[0])
            a_["v"] = a_["v"] * betas[1] + grads[n] ** 2 * (1 - betas[1])
            bc1, bc2 = 1 - betas[0] ** a_["step"], 1 - betas[1] ** a_["step"]
            S[n] = pw - (lr / bc1) * a_["m"] / (a_["v"].sqrt() / math.sqrt(bc2) + eps)
        for n in st["trainable"]:
            T[n] = T[n] * mom + S[n] * (1 - mom)
    return dict(loss=total.detach(), loss_dict={k: v.detach() for k, v in ld.items()}, gnorm=gnorm.detach(), grads=grads)
