"""Checkpoint interchange with the reference.

* Backbone weights: the module tree uses the reference's / timm's parameter names (``blocks.i.{norm1,attn.qkv,attn.proj,
  ls1.gamma,norm2,mlp.fc1,mlp.fc2,ls2.gamma}``, utils/transformers/vit.py), so a dinov2 checkpoint loads with
  ``load_state_dict`` once the keys the reference drops are dropped too (``mask_token``; transformers_utils.py:45-47).
  Load it BEFORE ``build_apla`` — exactly like the reference, which splits ``attn.proj`` afterwards (apla_vit.py:31-49).
* Session files: ``Trainer.save_session`` writes ``{'iters','state_dict','original_state','optimizer','epoch','parameters',
  'best_val_target'[, 'scaler']}`` (defaults/bases.py:447-467) where ``optimizer`` is a ``torch.optim.AdamW.state_dict()``
  over the two parameter groups of ``get_params_groups`` (defaults/wrappers.py:205-221: decayed = 2-D weights, then
  non-decayed = biases, each in ``named_parameters`` order).  ``session_dict`` / ``load_session`` translate between that
  layout and the engine's flat fp32 buffers (params, exp_avg, exp_avg_sq), so a run can move between the two stacks.
"""
from typing import Dict, Optional

import torch

DINOV2_DROPPED_KEYS = ("mask_token",)  # transformers_utils.py:45-47


def clean_pretrained_state_dict(sd: Dict[str, torch.Tensor], pretrained_type: str = "dinov2") -> Dict[str, torch.Tensor]:
    if pretrained_type == "dinov2":
        return {k: v for k, v in sd.items() if not any(sub in k for sub in DINOV2_DROPPED_KEYS)}
    return dict(sd)


def load_pretrained_backbone(backbone: torch.nn.Module, sd: Dict[str, torch.Tensor], pretrained_type: str = "dinov2",
                             strict: bool = True):
    """Load a (dinov2 / timm-named) checkpoint into a freshly built ViT — before ``build_apla``."""
    if any(hasattr(b.attn, "proj_weight1") for b in getattr(backbone, "blocks", [])):
        raise RuntimeError("load the pretrained backbone before build_apla(): the checkpoint holds the unsplit attn.proj")
    return backbone.load_state_dict(clean_pretrained_state_dict(sd, pretrained_type), strict=strict)


def _strip_prefix(sd, prefix):
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in sd.items()}


def read_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """A checkpoint file's weights: the ``state_dict`` entry of a session file (defaults/bases.py:456-467) or the file itself
    (a bare dinov2 / timm state dict); DDP's ``module.`` prefix removed."""
    try:   # plain tensors first; session files that carry pickled objects (parameters / original_state) need the full loader
        sd = torch.load(path, map_location="cpu", weights_only=True)
    except Exception:  # noqa: BLE001  (pickle.UnpicklingError and friends)
        sd = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return _strip_prefix(sd, "module.")


def is_apla_state_dict(sd: Dict[str, torch.Tensor]) -> bool:
    """Does the state dict come from a model AFTER the APLA split (``attn.proj_weight1`` …, appla_attn.py:37-45)?"""
    return any(k.endswith("attn.proj_weight1") for k in sd)


def load_apla_state_dict(model: torch.nn.Module, sd: Dict[str, torch.Tensor]):
    """The reference's rule for APLA / session checkpoints (utils/pretrained_loader.py:27-30): no key of the model may be
    missing, and the only unexpected keys allowed are ``partial_size`` entries.  Raises otherwise."""
    res = model.load_state_dict(sd, strict=False)
    if res.missing_keys:
        raise KeyError(f"checkpoint lacks {len(res.missing_keys)} keys of the model, e.g. {res.missing_keys[:4]}")
    bad = [k for k in res.unexpected_keys if "partial_size" not in k]
    if bad:
        raise KeyError(f"checkpoint has {len(bad)} keys the model does not know, e.g. {bad[:4]}")
    return res


def build_classifier_from_checkpoint(path: str, model_params, system_params):
    """Classifier (apla_amd.models) initialised from a checkpoint file of either kind:
    * an APLA / session checkpoint (split projection, ``backbone.`` prefix): build, then load with the reference's strict rule;
    * an unsplit backbone (dinov2 / timm names, with or without a ``backbone.`` prefix; a classification head, if any, is
      ignored): loaded into the ViT before ``build_apla`` splits ``attn.proj`` — so the frozen weights and the trainable rows
      both start from the pretrained projection, as in the reference.
    Returns (model, kind) with kind in {"apla", "backbone"}."""
    from .models import Classifier
    sd = read_state_dict(path)
    if is_apla_state_dict(sd):
        model = Classifier(model_params, system_params)
        load_apla_state_dict(model, sd)
        return model, "apla"
    head = {k: v for k, v in sd.items() if k.startswith("fc.")}
    if any(k.startswith("backbone.") for k in sd):
        sd = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
    dropped = sorted(k for k in sd if k.startswith("fc.") or k.startswith("head."))
    sd = {k: v for k, v in sd.items() if not (k.startswith("fc.") or k.startswith("head."))}
    model = Classifier(model_params, system_params, backbone_state_dict=sd)
    # a full unsplit Classifier (``backbone.`` + ``fc.``): the reference's non-APLA branch loads the head too
    # (utils/pretrained_loader.py:33, strict=True); a head of another shape (other class count) cannot be taken over
    own = {"fc." + k: v for k, v in model.fc.state_dict().items()}
    if head and set(head) == set(own) and all(tuple(head[k].shape) == tuple(own[k].shape) for k in own):
        model.fc.load_state_dict({k[len("fc."):]: v for k, v in head.items()})
    elif head or dropped:
        from .dist import print_ddp
        print_ddp(f"checkpoint: classification head keys not loaded (shape / name mismatch): {(sorted(head) or dropped)[:4]}")
    return model, "backbone"


def _group_order(model: torch.nn.Module):
    """Names of the trainable parameters in torch-optimizer index order: group 0 (decayed) then group 1."""
    reg, no_reg = [], []
    for name, p in model.named_parameters():
        if p.requires_grad:
            (no_reg if name.endswith(".bias") or p.ndim == 1 else reg).append(name)
    return reg, no_reg


def optimizer_state_dict(engine) -> dict:
    """The engine's AdamW state as a ``torch.optim.AdamW(get_params_groups(model)).state_dict()`` (CPU tensors)."""
    reg, no_reg = _group_order(engine.model)
    oc, state, idx = engine.optim, {}, 0
    if getattr(engine, "dynamic_scale", False):  # steps actually taken (skipped overflow steps do not count)
        steps = float(engine.scaler[3 * (engine._scaler_calls & 1) + 2])
    else:
        steps = float(getattr(engine, "applied_steps", engine.step_count))
    for name in reg + no_reg:
        off, k, shape = engine.slices[name]
        state[idx] = {"step": torch.tensor(steps),
                      "exp_avg": engine.exp_avg[off:off + k].view(shape).detach().cpu().clone(),
                      "exp_avg_sq": engine.exp_avg_sq[off:off + k].view(shape).detach().cpu().clone()}
        idx += 1
    common = dict(lr=oc.lr, betas=tuple(oc.betas), eps=oc.eps, amsgrad=False, maximize=False, foreach=None, capturable=False,
                  differentiable=False, fused=None)
    groups = [dict(common, weight_decay=oc.weight_decay, params=list(range(len(reg)))),
              dict(common, weight_decay=0.0, params=list(range(len(reg), len(reg) + len(no_reg))))]
    return {"state": state, "param_groups": groups}


def load_optimizer_state_dict(engine, osd: dict):
    reg, no_reg = _group_order(engine.model)
    names = reg + no_reg
    if sorted(osd["state"].keys()) not in ([], list(range(len(names)))):
        raise ValueError(f"optimizer state has {len(osd['state'])} entries, the model has {len(names)} trainable tensors")
    steps = set()
    for idx, name in enumerate(names):
        if idx not in osd["state"]:
            continue
        st = osd["state"][idx]
        off, k, shape = engine.slices[name]
        if tuple(st["exp_avg"].shape) != tuple(shape):
            raise ValueError(f"optimizer state {idx} ({name}): shape {tuple(st['exp_avg'].shape)} != {tuple(shape)}")
        engine.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1).float())
        engine.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1).float())
        steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError(f"per-tensor step counts differ ({sorted(steps)}): the fused optimizer keeps one step count")
    engine.step_count = steps.pop() if steps else 0
    if hasattr(engine, "norm_ws"):
        engine.norm_ws[260:262] = 0.0   # skipped-update counters (apla_adamw_step): the restored step count is "applied steps"


def session_dict(engine, *, iters: int = 0, epoch: int = 0, parameters: Optional[dict] = None,
                 best_val_target: float = 0.0, original_state: Optional[dict] = None) -> dict:
    """What ``Trainer.save_session`` stores (defaults/bases.py:456-464); ``torch.save`` it to interchange."""
    out = {"iters": iters,
           "state_dict": {k: v.detach().cpu().clone() for k, v in engine.model.state_dict().items()},
           "original_state": original_state,
           "optimizer": optimizer_state_dict(engine),
           "epoch": epoch,
           "parameters": parameters,
           "best_val_target": best_val_target}
    if getattr(engine, "dynamic_scale", False):  # GradScaler.state_dict() layout (defaults/bases.py:465-466)
        slot = 3 * (engine._scaler_calls & 1)
        out["scaler"] = {"scale": float(engine.scaler[slot]), "growth_factor": 2.0, "backoff_factor": 0.5,
                         "growth_interval": 2000, "_growth_tracker": int(engine.scaler[slot + 1])}
    return out


def load_session(engine, session: dict, load_optimizer: bool = True):
    """Restore model (trainable rows live in the flat buffer: copied in place) and optimizer state from a session dict."""
    sd = session["state_dict"]
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}  # DDP-wrapped saves
    own = engine.model.state_dict()
    missing = [k for k in own if k not in sd]
    if missing:
        raise KeyError(f"session state_dict lacks {missing[:4]}…")
    with torch.no_grad():
        for k, v in own.items():
            v.copy_(sd[k].to(v.dtype))  # parameters are views of engine.flat_params: in-place copy keeps the aliasing
    engine.refresh_frozen_copies()
    if load_optimizer and session.get("optimizer") is not None:
        load_optimizer_state_dict(engine, session["optimizer"])
    if getattr(engine, "dynamic_scale", False):
        sc = session.get("scaler") or {}
        scale, tracker = float(sc.get("scale", 65536.0)), float(sc.get("_growth_tracker", 0))
        steps = float(engine.step_count)
        engine.scaler.copy_(torch.tensor([scale, tracker, steps, scale, tracker, steps, scale, 0.0]))
        engine._scaler_calls = 0
        engine._scaler_step0 = steps   # skipped_steps counts from here (engine.skipped_steps)
    return session.get("iters", 0), session.get("epoch", 0)


def load_trainer_session(trainer, session: dict, load_optimizer: bool = True):
    """Resume a ``ModulePathTrainer`` or a ``Dinov2Trainer`` from the session dictionary ``main.py`` writes for them (the reference's
    layout, bases.py:456-467): model ``state_dict``, ``FlatAdamW`` state, and — written under fp16 — the loss scaler's
    ``GradScaler.state_dict()`` (bases.py:430-433 restores it the same way), so that an fp16 run resumes at the scale and growth count
    it stopped with instead of 65 536 / 0.  ``iters`` / ``epoch`` go back into a trainer that counts them."""
    trainer.model.load_state_dict(session["state_dict"])
    if load_optimizer and session.get("optimizer") is not None:
        trainer.optimizer.load_state_dict(session["optimizer"])
    if session.get("scaler") is not None and getattr(trainer, "scaler", None) is not None:
        trainer.scaler.load_state_dict(session["scaler"])
    for k in ("iters", "epoch"):
        if hasattr(trainer, k) and k in session:
            setattr(trainer, k, int(session[k]))
    return trainer
