"""DINOv2-APLA meta-architecture: student / teacher pair, heads, losses and the step's forward
(self_supervised/dinov2/models.py:37-453 — same class name, constructor argument, ``forward(images, teacher_temp=…) ->
(loss, loss_dict)``, ``update_teacher(m)`` and ``state_dict`` layout ``student.backbone.* / student.dino_head.* /
teacher.*``).

What runs where (everything on the MI355X; nothing here has a CPU path):
* teacher: one dense pass over the global crops (no grad), DINO head on [CLS tokens | masked patch tokens], centred
  softmax on ``apla_softmax_center``;
* student: ONE packed pass over global + local crops (block-diagonal attention kernels, apla_amd/ssl/backbone.py) — the
  reference's ``NestedTensorBlock`` path —, one DINO-head pass over [local CLS | global CLS | masked patches];
* losses: DINO (local→global, global→global), iBOT on the masked patches (``apla_distill_ce``: loss and d/dlogits in one
  pass over the 65 536 prototypes), KoLeo on the global CLS tokens.

``params`` is the reference's nested configuration (attribute access; plain dicts are wrapped): ``model_params.{backbone_type,
transformers_params.student, dinov2.{dino,ibot,centering}, adaptation}``, ``crops_params``, ``system_params.which_GPUs``.
"""
from copy import deepcopy
from functools import partial

import torch
from torch import nn

from .. import ops
from ..apla import build_apla
from ..models import AttrDict
from .backbone import DinoVisionTransformer
from .heads import DINOHead, KoLeoLoss, proto_losses
from .losses import DINOLoss, iBOTPatchLoss

_GEOMETRY = {"vit_small": (384, 12, 6, False), "vit_base": (768, 12, 12, False), "vit_large": (1024, 24, 16, False),
             "vit_giant": (1536, 40, 24, True)}


def build_model(backbone, backbone_args, only_teacher=False, img_size=224):
    """models.py:37-58: a randomly initialised (student, teacher, embed_dim) triple of dinov2-layout backbones."""
    if only_teacher:
        raise NotImplementedError
    if getattr(backbone_args, "drop_path_rate", 0) or getattr(backbone_args, "num_register_tokens", 0):
        raise NotImplementedError("stochastic depth / register tokens are 0 in the shipped DINOv2-APLA config; not on the HIP path")
    dim, depth, heads, swiglu = _GEOMETRY[backbone]
    init = getattr(backbone_args, "layerscale", None)
    def make():
        return DinoVisionTransformer(img_size=[img_size], patch_size=backbone_args.patch_size, embed_dim=dim, depth=depth, num_heads=heads,
                                     qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), use_swiglu=swiglu,
                                     block_conf=dict(has_layerscale=bool(init), layerscale_init_values=init or 1.0),
                                     interpolate_offset=getattr(backbone_args, "interpolate_offset", 0.1),
                                     interpolate_antialias=getattr(backbone_args, "interpolate_antialias", False))
    teacher = make()
    student = make()
    return student, teacher, dim


def _wrap(cfg):
    if isinstance(cfg, dict) and not isinstance(cfg, AttrDict):
        return AttrDict({k: _wrap(v) for k, v in cfg.items()})
    return cfg


_COEF_CACHE = {}


def _loss_coefficients(shown, weights, device):
    """(fp32 [n] shown * weights, fp32 [n] shown) on the device, cached per value set (schedules do not touch them)."""
    key = (tuple(shown), tuple(weights), str(device))
    ent = _COEF_CACHE.get(key)
    if ent is None:
        if len(_COEF_CACHE) >= 64:
            _COEF_CACHE.clear()
        ent = _COEF_CACHE[key] = (torch.tensor([a * b for a, b in zip(shown, weights)], dtype=torch.float32, device=device),
                                  torch.tensor(list(shown), dtype=torch.float32, device=device))
    return ent


class DINOv2(nn.Module):
    def __init__(self, params, backbones=None, backbone_state_dict=None):
        """``backbones`` = (student, teacher, embed_dim) overrides build_model (tests use small geometries).
        ``backbone_state_dict``: pretrained weights for BOTH backbones (the reference loads the torch.hub dinov2 weights into
        student and teacher before ``build_apla``, models.py:78-114).  An unsplit backbone (``attn.proj.*`` whole) is loaded
        strictly before the split; a state dict saved after the split (``attn.proj_weight1`` …, optionally under
        ``student.backbone.``) is loaded strictly after it.  A key mismatch raises."""
        super().__init__()
        params = _wrap(params)
        self.wrapper_params = params
        self.model_params = mp = params.model_params
        dino_p, ibot_p = mp.dinov2.dino, mp.dinov2.ibot
        student_p = mp.transformers_params.student
        assert "vit" in mp.backbone_type, "Only supports ViT"
        student_bb, teacher_bb, embed_dim = backbones if backbones is not None else build_model(
            mp.backbone_type, student_p, img_size=student_p.pre_img_size)
        if backbones is None and getattr(mp, "pretrained", False):
            raise RuntimeError("no network here: load LVD-142M weights with apla_amd.checkpoint.load_pretrained and pass "
                               "pretrained: false")
        split_sd = None
        if backbone_state_dict is not None:
            from ..checkpoint import is_apla_state_dict, load_pretrained_backbone
            sd = dict(backbone_state_dict)
            if any(k.startswith("student.backbone.") for k in sd):
                sd = {k[len("student.backbone."):]: v for k, v in sd.items() if k.startswith("student.backbone.")}
            elif any(k.startswith("backbone.") for k in sd):
                sd = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
            if is_apla_state_dict(sd):
                split_sd = sd
            else:
                load_pretrained_backbone(student_bb, sd, "dinov2", strict=True)
        teacher_bb.load_state_dict(deepcopy(student_bb.state_dict()))
        if split_sd is not None and "adaptation" not in mp:
            raise KeyError("the checkpoint holds a split APLA projection but the configuration has no adaptation section")
        if "adaptation" in mp:
            ad = mp.adaptation
            assert ad.mode == "apla", "Only supports adaptation with APLA"
            multi = len(str(params.system_params.which_GPUs).split(",")) > 1
            student_bb = build_apla(config=ad.params, model=student_bb, attn_class="apla_attn_mem_eff", is_multi_gpu=multi)
            teacher_bb = build_apla(config=ad.params, model=teacher_bb, attn_class="apla_attn_mem_eff", is_multi_gpu=multi)
            if split_sd is not None:
                from ..checkpoint import load_apla_state_dict
                load_apla_state_dict(student_bb, split_sd)
                load_apla_state_dict(teacher_bb, split_sd)
        self.do_dino = dino_p.loss_weight > 0
        self.do_koleo = dino_p.koleo_loss_weight > 0
        self.do_ibot = ibot_p.loss_weight > 0
        self.ibot_separate_head = ibot_p.separate_head
        self.sparse_last_block = True  # False: every block on every token, as the reference runs it (forward_features_list)
        self.unfused_losses = False   # True: the module-by-module route of the reference (head, split, one loss object per term)
        if self.ibot_separate_head:
            raise NotImplementedError("ibot.separate_head is false in the shipped config (models.py:160-171 reads an undefined "
                                      "`params.ibot` there)")
        student, teacher = {"backbone": student_bb}, {"backbone": teacher_bb}
        head = partial(DINOHead, in_dim=embed_dim, out_dim=dino_p.head_n_prototypes, hidden_dim=dino_p.head_hidden_dim,
                       bottleneck_dim=dino_p.head_bottleneck_dim, nlayers=dino_p.head_nlayers)
        if self.do_dino:
            self.dino_loss_weight = dino_p.loss_weight
            self.dino_loss = DINOLoss(out_dim=dino_p.head_n_prototypes)
            if self.do_koleo:
                self.koleo_loss = KoLeoLoss()
        if self.do_dino or self.do_ibot:
            student["dino_head"], teacher["dino_head"] = head(), head()
        if self.do_ibot:
            self.ibot_loss_weight = ibot_p.loss_weight
            assert max(ibot_p.mask_ratio_min_max) > 0 and ibot_p.mask_sample_probability > 0
            self.ibot_patch_loss = iBOTPatchLoss(patch_out_dim=dino_p.head_n_prototypes)
        self.student, self.teacher = nn.ModuleDict(student), nn.ModuleDict(teacher)
        for k, v in self.student.items():
            self.teacher[k].load_state_dict(v.state_dict())
        for p in self.teacher.parameters():
            p.requires_grad = False

    # ------------------------------------------------------------------------------------------------------------------
    def forward(self, images, return_embedding=False, teacher_temp=None):
        if return_embedding:   # evaluation: CLS feature of the teacher backbone (models.py:208-210)
            return [None, None], self.teacher.backbone(images.cuda(non_blocking=True))
        crops = self.wrapper_params.crops_params
        n_global, n_local = crops.n_global_crops, crops.n_local_crops
        assert n_global == 2
        dev = next(self.student.parameters()).device
        glob = images["collated_global_crops"].to(dev, non_blocking=True)
        loc = images["collated_local_crops"].to(dev, non_blocking=True)
        masks = images["collated_masks"].to(dev, non_blocking=True)
        idx = images["mask_indices_list"].to(dev, non_blocking=True)
        masks_weight = images["masks_weight"].to(dev, non_blocking=True)
        n_masked, upper = idx.shape[0], int(images["upperbound"])
        n_local_terms, n_global_terms = max(n_local * n_global, 1), (n_global - 1) * n_global
        centering = self.model_params.dinov2.centering
        do_dino, do_ibot = self.do_dino, self.do_ibot

        sparse = self.sparse_last_block and hasattr(self.teacher.backbone, "forward_cls_and_masked")
        with torch.no_grad():   # ---- teacher (models.py:231-300)
            if sparse:   # the same tokens, the last block's token-wise part on the rows read here only (backbone.forward_cls_and_masked)
                (t_cls_tok,), t_masked = self.teacher.backbone.forward_cls_and_masked([glob], [None], idx if do_ibot else None)
            else:
                tout = self.teacher.backbone(glob, is_training=True)
                t_cls_tok = tout["x_norm_clstoken"]
                t_masked = tout["x_norm_patchtokens"].flatten(0, 1).index_select(0, idx) if do_ibot else None
            a, b = t_cls_tok.chunk(n_global)
            tcls = torch.cat((b, a))     # swapped halves: crop A's student is matched with crop B's teacher
            n_cls = tcls.shape[0]
            if do_ibot:
                buf = tcls.new_zeros(upper + n_cls, tcls.shape[-1])
                buf[:n_cls] = tcls
                buf[n_cls:n_cls + n_masked] = t_masked
                after = self.teacher.dino_head(buf)
                tcls_h, tpatch_h = after[:n_cls], after[n_cls:n_cls + n_masked]
            else:
                tcls_h, tpatch_h = self.teacher.dino_head(tcls), None
            t_dino = t_ibot = None
            if centering == "centering":
                if do_dino:
                    t_dino = self.dino_loss.softmax_center_teacher(tcls_h, teacher_temp=teacher_temp).view(n_global, -1, tcls_h.shape[-1])
                    self.dino_loss.update_center(tcls_h)
                if do_ibot:
                    t_ibot = self.ibot_patch_loss.softmax_center_teacher(tpatch_h.unsqueeze(0), teacher_temp=teacher_temp, lazy=True).squeeze(0)
                    self.ibot_patch_loss.update_center(tpatch_h.unsqueeze(0))
            elif centering == "sinkhorn_knopp":
                if do_dino:
                    t_dino = self.dino_loss.sinkhorn_knopp_teacher(tcls_h, teacher_temp=teacher_temp).view(n_global, -1, tcls_h.shape[-1])
                if do_ibot:
                    t_ibot = self.ibot_patch_loss.sinkhorn_knopp_teacher(
                        tpatch_h, teacher_temp=teacher_temp, n_masked_patches_tensor=images["n_masked_patches"].to(dev))
            else:
                raise NotImplementedError(centering)

        # ---- student: one packed pass over both resolutions (models.py:306-308), one head pass (models.py:326-347)
        if sparse:
            (s_glob_cls, s_loc_cls), s_masked = self.student.backbone.forward_cls_and_masked([glob, loc], [masks, None], idx if do_ibot else None)
        else:
            sg, sl = self.student.backbone([glob, loc], masks=[masks, None], is_training=True)
            s_loc_cls, s_glob_cls = sl["x_norm_clstoken"], sg["x_norm_clstoken"]
            s_masked = sg["x_norm_patchtokens"].flatten(0, 1).index_select(0, idx) if do_ibot else None
        head_in = [s_loc_cls, s_glob_cls]
        if do_ibot:
            pbuf = s_glob_cls.new_zeros(upper, s_glob_cls.shape[-1])
            pbuf[:n_masked] = s_masked
            head_in.append(pbuf)
        loss_dict, total = {}, 0
        loss_scales = 2   # both global crops go through together
        head = self.student.dino_head
        if (do_dino and do_ibot and n_local > 0 and centering == "centering" and isinstance(head, DINOHead) and not self.unfused_losses):
            # one autograd node from the bottleneck features to the three cross-entropy sums (heads._ProtoLosses): same kernels, same
            # values as the module route below, the [rows, K] gradient written once and read by the two GEMMs only
            Bc, off = s_loc_cls.shape[0] // n_local, s_loc_cls.shape[0]
            tsum = t_dino[0].float() + t_dino[1].float()                     # DINOLoss.forward: the targets of one student view add up
            # every local crop against the same Bc target rows: ONE term whose targets repeat (launch_distill_ce), not n_local launches
            terms = [(0, 0, off, tsum, self.dino_loss.student_temp, None, 1.0 / Bc)]
            terms.append((1, off, off + s_glob_cls.shape[0], t_dino.flatten(0, 1), self.dino_loss.student_temp, None, 1.0 / s_glob_cls.shape[0]))
            off += s_glob_cls.shape[0]
            terms.append((2, off, off + n_masked, t_ibot[:n_masked], self.ibot_patch_loss.student_temp, masks_weight, 1.0 / masks.shape[0]))
            sums = proto_losses(head.bottleneck(torch.cat(head_in)), head.last_layer.weight_v, head.last_layer.weight_g, 3, terms)
            # the reference's scalar bookkeeping (models.py:380-432: l = sum / (n_g + n_l); total += weight * l; ...) on the three sums at
            # once: two small launches forward and two backward instead of a dozen each
            n_terms = n_global_terms + n_local_terms
            shown = (1.0 / n_terms, loss_scales / n_terms, loss_scales * (1.0 / n_global) / 2)       # what loss_dict displays
            weights = (self.dino_loss_weight, self.dino_loss_weight, self.ibot_loss_weight * 2)       # total = sum(weights * shown * sums)
            c_total, c_shown = _loss_coefficients(shown, weights, sums.device)
            total = (sums * c_total).sum()
            disp = sums.detach() * c_shown
            loss_dict["dino_local_crops_loss"], loss_dict["dino_global_crops_loss"], loss_dict["ibot_loss"] = disp[0], disp[1], disp[2]
            if self.do_koleo:
                kl = self.model_params.dinov2.dino.koleo_loss_weight * self.koleo_loss.grouped(s_glob_cls, 2)
                total = total + kl
                loss_dict["koleo_loss"] = kl.detach() / loss_scales
            return total, loss_dict
        outs = head(torch.cat(head_in)).split([t.shape[0] for t in head_in])
        o_loc, o_glob = outs[0], outs[1]
        if do_dino and n_local > 0:
            l = self.dino_loss(student_output_list=o_loc.chunk(n_local), teacher_out_softmaxed_centered_list=t_dino) \
                / (n_global_terms + n_local_terms)
            loss_dict["dino_local_crops_loss"] = l
            total = total + self.dino_loss_weight * l
        loss_scales = 2   # both global crops go through together
        if do_dino:
            l = self.dino_loss(student_output_list=[o_glob], teacher_out_softmaxed_centered_list=[t_dino.flatten(0, 1)]) \
                * loss_scales / (n_global_terms + n_local_terms)
            loss_dict["dino_global_crops_loss"] = l
            total = total + self.dino_loss_weight * l
            if self.do_koleo:   # per global crop: never between two views of one image
                kl = self.model_params.dinov2.dino.koleo_loss_weight * self.koleo_loss.grouped(s_glob_cls, 2)
                total = total + kl
                loss_dict["koleo_loss"] = kl / loss_scales
        if do_ibot:
            l = self.ibot_patch_loss.forward_masked(outs[2][:n_masked], t_ibot, student_masks_flat=masks, n_masked_patches=n_masked,
                                                    masks_weight=masks_weight) * loss_scales * (1.0 / n_global)
            loss_dict["ibot_loss"] = l / 2
            total = total + self.ibot_loss_weight * l
        return total, loss_dict

    def _ema_pairs(self):
        return [(ps, pt) for k in self.student.keys() for ps, pt in zip(self.student[k].parameters(), self.teacher[k].parameters())
                if ps.requires_grad]

    @torch.no_grad()
    def flatten_teacher(self, optimizer) -> bool:
        """Move the teacher's copies of the tensors ``optimizer`` (a FlatAdamW over the student) updates into ONE flat fp32 buffer laid
        out like ``optimizer.flat``; ``update_teacher`` is then one launch over the two buffers (apla_ema_update) instead of two
        torch._foreach passes over ~100 tensors.  False (nothing changed) if the two parameter lists do not line up."""
        pairs = self._ema_pairs()
        if (len(pairs) != len(optimizer.params) or any(a is not b for (a, _), b in zip(pairs, optimizer.params))
                or any(pt.dtype != torch.float32 or pt.shape != ps.shape or pt.device != ps.device for ps, pt in pairs)):
            return False
        flat = torch.empty_like(optimizer.flat)
        for (_, pt), a, b in zip(pairs, optimizer.offsets[:-1], optimizer.offsets[1:]):
            flat[a:b].copy_(pt.detach().reshape(-1))
            pt.data = flat[a:b].view(pt.shape)
        self._ema_flat = (flat, optimizer.flat, list(optimizer.offsets), [pt for _, pt in pairs], [ps for ps, _ in pairs])
        return True

    @torch.no_grad()
    def update_teacher(self, m):
        """teacher = m * teacher + (1 - m) * student (models.py:443-453).  The reference walks every parameter; frozen
        tensors are identical in both networks, so only the trainable ones are touched here (same result, and the frozen
        copies stay bit-identical instead of collecting rounding noise).  After ``flatten_teacher`` one kernel does it."""
        bound = getattr(self, "_ema_flat", None)
        if bound is not None:
            t_flat, s_flat, offs, t_params, s_params = bound
            tb, sb = t_flat.data_ptr(), s_flat.data_ptr()
            if all(pt.data_ptr() == tb + 4 * a and ps.data_ptr() == sb + 4 * a for pt, ps, a in zip(t_params, s_params, offs)):
                ops.ema_update(t_flat, s_flat, m)
                for pt in t_params:      # written through a raw pointer: the 16-bit weight caches key on the tensor version
                    torch.autograd.graph.increment_version(pt)
                return len(t_params)
            self._ema_flat = None        # someone re-homed a tensor (load_state_dict(assign=True), .to(...)): back to the lists
        pairs = self._ema_pairs()
        t_list, s_list = [pt for _, pt in pairs], [ps.detach() for ps, _ in pairs]
        torch._foreach_mul_(t_list, m)
        torch._foreach_add_(t_list, s_list, alpha=1 - m)
        return len(t_list)

    def train(self, train_mode=True):
        super().train(train_mode)
        self.teacher.eval()
        return self
