#!/usr/bin/env python
"""`python main.py --params_path params/.../apla.yml [--batch_size ..] [--lr ..] [--gpu 0,1,..]` — the reference's entry
point (src/main.py:18-55, 241-264) for the supervised APLA fine-tuning path, on the MI355X-native stack.

Kept from the reference: the CLI flags that touch this path, the YAML schema (`__common__.yml` next to the given file is
loaded first and the given file overrides it key by key — utils/helpfuns.py:114-134), the argument -> parameter overrides
(main.py:58-158), model construction from `model_params` (defaults/models.py), the two AdamW groups, the per-iteration
LinearWarmup / CosineAnnealingLR schedule, gradient clipping, one process per GPU.  The training step itself is
`AplaTrainEngine.train_step` (fused forward + loss + backward + gradient exchange + clip + AdamW).

Also here: `--dinov2` (the DINOv2-APLA self-supervised step, src/main.py:169-171 -> apla_amd/ssl), `--test` / `--knn`
(validation + kNN evaluation on the engine's forward, defaults/trainer.py:162-345 -> apla_amd/evaluate.py), and
`--pretrained_path`: an APLA / session checkpoint is loaded strictly (utils/pretrained_loader.py:27-30), an unsplit dinov2 /
timm backbone is loaded before `build_apla` splits the projection (apla_amd/checkpoint.py).

NOT rebuilt (out of scope): the dataset zoo and torchvision transforms, wandb logging, the BYOL / SimSiam /
DINO-v1 trainers.  Data therefore comes from one of two sources:
  * `dataset_params.dataset: "TensorFile"` + `data_location: file.pt` — a dict {"images": uint8|float [N,3,S,S],
    "labels": int [N]} that is normalised (ImageNet mean/std) and served in shuffled batches from device memory;
  * anything else — synthetic N(0,1) images with random labels of the shapes the YAML describes (a plumbing /
    throughput run; announced loudly).
"""
import argparse
import copy
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KNOWN_CLASSES = {"NABirds": 555, "ImageNet": 1000, "CIFAR10": 10, "CIFAR100": 100, "ISIC2019": 8, "Flowers102": 102,
                 "Food101": 101, "OxfordPets": 37, "StanfordCars": 196, "DTD": 47, "SUN397": 397}
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def parse_arguments(argv=None):
    p = argparse.ArgumentParser(description="APLA fine-tuning on MI355X (reference CLI surface, src/main.py:18-55)", allow_abbrev=False)
    p.add_argument("--params_path", type=str, required=True)
    p.add_argument("--gpu", type=str, help="comma-separated GPU ids: one process per GPU")
    p.add_argument("--batch_size", type=int)
    p.add_argument("--val_every", type=float)
    p.add_argument("--log_every", type=int)
    p.add_argument("--mixed_precision", action="store_true", default=False)
    p.add_argument("--num_workers", type=str)
    p.add_argument("--prefetch_factor", type=str)
    p.add_argument("--lr", type=float)
    p.add_argument("--warmup", type=int)
    p.add_argument("--epochs", type=int)
    p.add_argument("--wd", type=float)
    p.add_argument("--dpr", type=float)
    p.add_argument("--dr", type=float)
    p.add_argument("--adr", type=float)
    p.add_argument("--model_name", type=str)
    p.add_argument("--pretrained_path", type=str)
    p.add_argument("--save_dir", type=str)
    p.add_argument("--debug", action="store_true", default=False)
    p.add_argument("--dry", action="store_true", default=False, help="build everything, run a handful of iterations, save nothing")
    p.add_argument("--job_id", type=str)
    p.add_argument("--offline", action="store_true", default=False)
    p.add_argument("--test", action="store_true", default=False)
    p.add_argument("--knn", action="store_true", default=False)
    for flag in ("byol", "simsiam", "dino", "dinov2"):
        p.add_argument("--" + flag, action="store_true", default=False)
    # additions of this implementation
    p.add_argument("--steps_per_epoch", type=int, help="synthetic data only: iterations that make one epoch (default 100)")
    p.add_argument("--module_path", action="store_true", default=False,
                   help="train on the drop-in module path (autograd over the kernels) instead of the fused step")
    p.add_argument("--n_classes", type=int, help="classes when the dataset name is not a known one")
    p.add_argument("--dtype", choices=["bf16", "fp16"], default=None, help="16-bit operand type (default bf16; --mixed_precision alone keeps bf16)")
    return p.parse_args(argv)


def update_nested_values(dst: dict, src: dict):
    """utils/helpfuns.py:114-134: dict values recurse (new sub-dicts are added whole), everything else overwrites."""
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            update_nested_values(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def load_parameters(params_path: str) -> dict:
    """__common__.yml of the same directory (one level up for '_others', main.py:244-247) overridden by the given file."""
    d = os.path.dirname(os.path.abspath(params_path))
    common = os.path.join(d, "..", "__common__.yml") if "_others" in params_path else os.path.join(d, "__common__.yml")
    params = {}
    if os.path.exists(common):
        with open(common) as f:
            params = yaml.safe_load(f) or {}
    with open(params_path) as f:
        update_nested_values(params, yaml.safe_load(f) or {})
    return params


def _set(params, path, value):
    d = params
    for k in path[:-1]:
        d = d.setdefault(k, {})
    d[path[-1]] = value


def update_params_from_args(params: dict, args) -> dict:
    """main.py:58-158 (the overrides that exist on this path)."""
    opt = ("optimization_params", "default")
    if args.warmup:
        _set(params, opt + ("scheduler", "params", "LinearWarmup", "warmup_iters"), args.warmup)
    if args.epochs:
        _set(params, ("training_params", "epochs"), args.epochs)
    if args.pretrained_path:
        _set(params, ("transfer_learning_params", "pretrained_path"), args.pretrained_path)
    if args.lr:
        _set(params, opt + ("optimizer", "params", "lr"), args.lr)
    if args.wd is not None:
        _set(params, opt + ("optimizer", "params", "weight_decay"), args.wd)
    for name, key in (("dpr", "drop_path_rate"), ("dr", "drop_rate"), ("adr", "attn_drop_rate")):
        if getattr(args, name) is not None:
            _set(params, ("model_params", "transformers_params", key), getattr(args, name))
    if args.gpu:
        _set(params, ("system_params", "which_GPUs"), args.gpu)
    if args.model_name:
        _set(params, ("training_params", "model_name"), args.model_name)
    if args.save_dir:
        _set(params, ("training_params", "save_dir"), args.save_dir)
    if args.batch_size:
        for loader in ("trainloader", "valloader", "testloader"):
            _set(params, ("dataloader_params", loader, "batch_size"), args.batch_size)
    if args.val_every is not None:
        _set(params, ("training_params", "val_every"), args.val_every)
    if args.log_every is not None:
        _set(params, ("training_params", "log_every"), args.log_every)
    if args.job_id is not None:
        _set(params, ("training_params", "job_id"), args.job_id)
    if args.mixed_precision:
        _set(params, ("training_params", "use_mixed_precision"), True)
    return params


def resolve_run(params: dict, args) -> dict:
    """Everything the training loop needs, derived from the merged parameters (no GPU involved: unit-testable)."""
    if args.byol or args.simsiam or args.dino:
        raise NotImplementedError("BYOL / SimSiam / DINO trainers are out of scope here (SURVEY §2); --dinov2 and the supervised APLA path run")
    mp = params["model_params"]
    ad = mp.get("adaptation") or {}
    if ad.get("mode") != "apla":
        raise AssertionError("model_params.adaptation.mode must be 'apla' (defaults/models.py:34)")
    ds = params.get("dataset_params", {})
    tt = ds.get("train_transforms", {})
    if tt.get("RandomResizedCrop", {}).get("apply"):
        img = int(tt["RandomResizedCrop"]["size"])
    elif tt.get("CenterCrop", {}).get("apply"):
        img = int(tt["CenterCrop"]["height"])
    elif tt.get("Resize", {}).get("apply"):
        img = int(tt["Resize"]["height"])
    else:
        img = int(mp.get("transformers_params", {}).get("img_size", [224])[0])
    n_classes = args.n_classes or mp.get("n_classes") or KNOWN_CLASSES.get(ds.get("dataset"))
    if not n_classes:
        raise ValueError(f"cannot tell the number of classes of dataset {ds.get('dataset')!r}: pass --n_classes")
    opt = params["optimization_params"]["default"]
    if opt["optimizer"]["type"] != "AdamW":
        raise NotImplementedError("the fused optimizer implements AdamW (the type every shipped APLA config uses)")
    sched = opt.get("scheduler", {})
    types = sched.get("type") or []
    types = [types] if not isinstance(types, list) else [t for t in types if t]
    for t in types:
        if t not in ("LinearWarmup", "CosineAnnealingLR"):
            raise NotImplementedError(f"scheduler {t!r}: only LinearWarmup and CosineAnnealingLR are implemented")
    tp = params.get("training_params", {})
    gpus = str(params.get("system_params", {}).get("which_GPUs", "0"))
    # inds_path in the shipped files is relative to the reference's src/ working directory ("../params/…"): accept it as
    # given, else look for the file next to the parameter file
    ap = ad.setdefault("params", {})
    if ap.get("inds_path") and not os.path.exists(ap["inds_path"]):
        cand = os.path.join(os.path.dirname(os.path.abspath(args.params_path)), os.path.basename(ap["inds_path"]))
        if os.path.exists(cand):
            ap["inds_path"] = cand
    adv = bool(tt.get("advanced_aug"))
    adv_params = tt.get("advanced_aug_params", {}) if adv else {}
    return dict(img=img, n_classes=int(n_classes), batch=int(params["dataloader_params"]["trainloader"]["batch_size"]),
                lr=float(opt["optimizer"]["params"]["lr"]), wd=float(opt["optimizer"]["params"].get("weight_decay", 0.0)),
                sched_types=types, sched_params=sched.get("params", {}), epochs=int(tp.get("epochs", 1)),
                grad_clipping=float(tp.get("grad_clipping") or 0.0), log_every=int(tp.get("log_every", 25)),
                model_name=tp.get("model_name", "model"), save_dir=tp.get("save_dir"), gpus=[g for g in gpus.split(",") if g != ""],
                dataset=ds.get("dataset"), data_location=ds.get("data_location"),
                pretrained_path=params.get("transfer_learning_params", {}).get("pretrained_path"),
                soft_targets=adv, label_smoothing=float(adv_params.get("label_smoothing", 0.0)))


def make_schedule(run, steps_per_epoch):
    from apla_amd.schedule import LRSchedule
    sp = run["sched_params"]
    warm = sp.get("LinearWarmup", {}) if "LinearWarmup" in run["sched_types"] else {}
    return LRSchedule(run["lr"], use_warmup="LinearWarmup" in run["sched_types"], cosine="CosineAnnealingLR" in run["sched_types"],
                      warmup_iters=int(warm.get("warmup_iters", 0) or 0), warmup_epochs=int(warm.get("warmup_epochs", 0) or 0),
                      steps_per_epoch=steps_per_epoch, epochs=run["epochs"],
                      cosine_eta_min=float(sp.get("CosineAnnealingLR", {}).get("eta_min", 0.0)))


class TensorBatches:
    """Shuffled, drop_last batches of a device-resident tensor dataset (or synthetic data), sharded over ranks."""

    def __init__(self, run, rank, world, device, steps_per_epoch=None):
        self.B, self.rank, self.world, self.device = run["batch"], rank, world, device
        self.images = self.labels = self.val_images = self.val_labels = None
        if run["dataset"] == "TensorFile":
            blob = torch.load(run["data_location"], map_location="cpu")
            imgs, self.labels = blob["images"], blob["labels"].to(device).int()
            self._check_labels(self.labels, run["n_classes"], "labels")
            imgs = imgs.float().div_(255.0) if imgs.dtype == torch.uint8 else imgs.float()
            mean, std = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1), torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
            self.images = ((imgs - mean) / std).to(device)
            self.steps = self.images.shape[0] // (self.B * world)
            if "val_images" in blob:
                v = blob["val_images"]
                v = v.float().div_(255.0) if v.dtype == torch.uint8 else v.float()
                self.val_images, self.val_labels = ((v - mean) / std).to(device), blob["val_labels"].to(device).int()
                self._check_labels(self.val_labels, run["n_classes"], "val_labels")
        else:
            print(f"\033[93m[main] no loader for dataset {run['dataset']!r} at {run['data_location']!r}: SYNTHETIC N(0,1) images / random "
                  f"labels [{self.B},3,{run['img']},{run['img']}], {run['n_classes']} classes\033[0m", flush=True)
            self.steps = steps_per_epoch or 100
            self.shape, self.C = (self.B, 3, run["img"], run["img"]), run["n_classes"]

    @staticmethod
    def _check_labels(labels, n_classes, what):
        """torch's CrossEntropyLoss raises on a class id outside [0, C); the fused CE kernel can only poison the loss with NaN, so
        the file is validated once here (a wrong --n_classes, a dataset missing from KNOWN_CLASSES, an ignore_index of -1 …)."""
        lo, hi = int(labels.min()), int(labels.max())
        if lo < 0 or hi >= n_classes:
            raise ValueError(f"TensorFile {what}: class ids span [{lo}, {hi}] but the model has n_classes = {n_classes}")

    def eval_batches(self):
        """Evaluation batches of the engine's batch size (drop_last): `val_images` / `val_labels` of the tensor file when it
        has them, else the training tensors; three synthetic batches otherwise."""
        if self.images is None:
            g = torch.Generator(device=self.device).manual_seed(999)
            for _ in range(3):
                yield torch.randn(self.shape, device=self.device, generator=g), torch.randint(0, self.C, (self.B,), device=self.device, generator=g)
            return
        imgs, labs = (self.val_images, self.val_labels) if self.val_images is not None else (self.images, self.labels)
        for s in range(imgs.shape[0] // self.B):
            yield imgs[s * self.B:(s + 1) * self.B], labs[s * self.B:(s + 1) * self.B]

    def epoch(self, epoch):
        g = torch.Generator(device=self.device).manual_seed(1000 * epoch + (0 if self.images is not None else self.rank))
        if self.images is None:
            for _ in range(self.steps):
                yield torch.randn(self.shape, device=self.device, generator=g), torch.randint(0, self.C, (self.B,), device=self.device, generator=g)
            return
        perm = torch.randperm(self.images.shape[0], device=self.device, generator=g)  # same permutation on every rank (DistributedSampler)
        for s in range(self.steps):
            idx = perm[(s * self.world + self.rank) * self.B:(s * self.world + self.rank + 1) * self.B]
            yield self.images[idx], self.labels[idx]


# self_supervised/dinov2/augmentation_strategy.json: repetition_strategy.n_augmentations = [1, 1, 8]; RandomResizedCrop sizes
DINOV2_CROPS = dict(n_global_crops=2, n_local_crops=8, global_crops_size=224, local_crops_size=98, img_size=224)


def resolve_dinov2_run(params: dict, args) -> dict:
    """What Dinov2Trainer / DINOv2Wrapper read from the merged parameters (self_supervised/dinov2/trainer.py:7-80,
    wrappers.py:36-71); no GPU involved."""
    mp = params["model_params"]
    opt = params["optimization_params"]["default"]
    if opt["optimizer"]["type"] != "AdamW" or opt["scheduler"]["type"] != ["LinearWarmup", "CosineAnnealingLR"]:
        raise NotImplementedError("the DINOv2 trainer implements AdamW with [LinearWarmup, CosineAnnealingLR] (trainer.py:8)")
    tp, teacher = params.get("training_params", {}), mp["transformers_params"]["teacher"]
    ap = (mp.get("adaptation") or {}).get("params", {})
    if ap.get("inds_path") and not os.path.exists(ap["inds_path"]):
        cand = os.path.join(os.path.dirname(os.path.abspath(args.params_path)), os.path.basename(ap["inds_path"]))
        if os.path.exists(cand):
            ap["inds_path"] = cand
    gpus = [g for g in str(params.get("system_params", {}).get("which_GPUs", "0")).split(",") if g != ""]
    return dict(batch=int(params["dataloader_params"]["trainloader"]["batch_size"]), epochs=int(tp.get("epochs", 1)),
                lr=float(opt["optimizer"]["params"]["lr"]), wd=float(opt["optimizer"]["params"].get("weight_decay", 0.0)),
                eta_min=float(opt["scheduler"]["params"]["CosineAnnealingLR"]["eta_min"]),
                warmup_epochs=int(opt["scheduler"]["params"]["LinearWarmup"].get("warmup_epochs", 0) or 0),
                grad_clipping=float(tp.get("grad_clipping") or 0.0), freeze_last=int(tp.get("freeze_last_layer_epochs", 0) or 0),
                log_every=int(tp.get("log_every", 25)), teacher=dict(teacher), gpus=gpus, model_name=tp.get("model_name", "model"),
                save_dir=tp.get("save_dir"), patch=int(mp["transformers_params"]["student"]["patch_size"]))


def main_dinov2(params, args):
    """--dinov2: the self-supervised DINOv2-APLA pretraining loop (main.py:166-207 -> DINOv2Wrapper + Dinov2Trainer) on
    synthetic crops (the dataset zoo / PIL augmentations are out of scope): collate with iBOT masks on the host,
    everything else on the GPU."""
    import random
    import torch.distributed as dist
    from apla_amd.dist import dist_average_tensor, init_from_env, is_rank0, synchronize
    from apla_amd.ssl import DINOv2, Dinov2Trainer, MaskingGenerator, collate_data_and_cast
    from apla_amd.ssl.collate import synthetic_samples
    from apla_amd.checkpoint import read_state_dict as ckpt_read
    run = resolve_dinov2_run(params, args)
    rank, world, local = init_from_env()
    dev = torch.device("cuda", local if world > 1 else int(run["gpus"][0]) if run["gpus"] else 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)   # identical student / teacher / head initialisation on every rank
    p = copy.deepcopy(params)
    p["crops_params"] = dict(DINOV2_CROPS)
    pretrained = bool(p["model_params"].get("pretrained"))
    p["model_params"]["pretrained"] = False
    pp = params.get("transfer_learning_params", {}).get("pretrained_path") or args.pretrained_path
    model = DINOv2(p, backbone_state_dict=ckpt_read(pp) if pp else None)
    if pp and is_rank0():
        print(f"[main] loaded {pp} into student and teacher")
    elif pretrained and is_rank0():
        print("\033[93m[main] model_params.pretrained is true but there is no network and no --pretrained_path: random initialisation\033[0m")
    model = model.to(dev).train()
    steps = args.steps_per_epoch or 100
    t = run["teacher"]
    trainer = Dinov2Trainer(model, iters_per_epoch=steps, epochs=run["epochs"], lr=run["lr"], weight_decay=run["wd"], eta_min=run["eta_min"],
                            warmup_epochs=run["warmup_epochs"], grad_clipping=run["grad_clipping"], freeze_last_layer_epochs=run["freeze_last"],
                            momentum_teacher=t["momentum_teacher"], final_momentum_teacher=t["final_momentum_teacher"],
                            warmup_teacher_temp=t["warmup_teacher_temp"], teacher_temp=t["teacher_temp"],
                            warmup_teacher_temp_epochs=t["warmup_teacher_temp_epochs"], process_group=dist.group.WORLD if world > 1 else None,
                            compute_dtype=torch.float16 if args.dtype == "fp16" else torch.bfloat16)
    c, ib = DINOV2_CROPS, p["model_params"]["dinov2"]["ibot"]
    side = c["img_size"] // run["patch"]
    mask_gen = MaskingGenerator(input_size=(side, side), max_num_patches=0.5 * c["img_size"] // run["patch"] * c["img_size"] // run["patch"])
    random.seed(1000 + rank)
    gen = torch.Generator().manual_seed(1000 + rank)
    if is_rank0():
        print(f"\033[93m[main] SYNTHETIC crops: {run['batch']} x (2 x {c['global_crops_size']} + 8 x {c['local_crops_size']}) per GPU\033[0m\n"
              f"[main] DINOv2-APLA {p['model_params']['backbone_type']}/{run['patch']}  {world} GPU(s) x bs {run['batch']}  "
              f"{len(trainer.optimizer.names)} trainable tensors, {trainer.optimizer.flat.numel():,} parameters, operands {args.dtype}"
              + (f", dynamic loss scale from {trainer.loss_scale:g}" if args.dtype == "fp16" else ""), flush=True)
    epochs = 1 if args.dry else run["epochs"]
    t0, seen, loss = time.perf_counter(), 0, None
    for epoch in range(epochs):
        for _ in range(steps):
            batch = collate_data_and_cast(synthetic_samples(run["batch"], c["global_crops_size"], c["local_crops_size"], c["n_local_crops"], gen),
                                          n_global_crops=2, n_local_crops=c["n_local_crops"], mask_ratio_tuple=tuple(ib["mask_ratio_min_max"]),
                                          mask_probability=ib["mask_sample_probability"], dtype=torch.float32, n_tokens=side * side,
                                          mask_generator=mask_gen)
            for k_, v_ in batch["images"].items():   # page-locked like the reference's DataLoader(pin_memory=True): the model's
                if torch.is_tensor(v_) and not v_.is_cuda:   # .to(dev, non_blocking=True) then does not make the host wait for the stream
                    batch["images"][k_] = v_.pin_memory()
            loss = trainer.global_step(batch)
            seen += run["batch"] * world
            it = trainer.iters - 1
            if it % run["log_every"] == 0 or it == 1:
                avg = dist_average_tensor(loss)
                if is_rank0():
                    torch.cuda.synchronize()
                    terms = "  ".join(f"{k} {float(v):.4f}" for k, v in trainer.loss_dict.items())
                    print(f"[main] epoch {trainer.epoch} it {it}: train_loss {float(avg):.4f}  {terms}  lr {trainer.optimizer.lr:.3e}  "
                          f"{seen / (time.perf_counter() - t0):.0f} images/s", flush=True)
            if args.dry and it >= 3:
                break
    synchronize()
    if is_rank0() and run["save_dir"] and not (args.dry or args.debug):
        os.makedirs(run["save_dir"], exist_ok=True)
        path = os.path.join(run["save_dir"], run["model_name"] + ".pth")
        sess = {"state_dict": model.state_dict(), "optimizer": trainer.optimizer.state_dict(), "iters": trainer.iters,
                "epoch": trainer.epoch, "parameters": params}   # bases.py:456-467 session layout
        if args.dtype == "fp16":
            sess["scaler"] = trainer.scaler.state_dict()        # bases.py:465-466
        torch.save(sess, path)
        print(f"[main] saved {path}")
    return float(loss)


def main(params, args):
    if args.dinov2:
        return main_dinov2(params, args)
    import torch.distributed as dist
    from apla_amd import checkpoint as ckpt
    from apla_amd.dist import dist_average_tensor, init_from_env, is_rank0, synchronize
    from apla_amd.engine import AplaTrainEngine, OptimConfig
    from apla_amd.models import Classifier
    run = resolve_run(params, args)
    rank, world, local = init_from_env()
    dev = torch.device("cuda", local if world > 1 else int(run["gpus"][0]) if run["gpus"] else 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)  # identical construction (weights, indices) on every rank; multi-GPU runs should still pass inds_path
    mp = copy.deepcopy(params["model_params"])
    mp["n_classes"] = run["n_classes"]
    # transformers_params.img_size stays what the YAML says (518 for dinov2 weights: their pos_embed has 37x37+1 entries);
    # the engine interpolates it once to the training resolution (vit.py:421-437)
    pretrained = bool(mp.get("pretrained"))
    mp["pretrained"] = False  # no network here: weights come from --pretrained_path or stay at their initialisation
    sp = params.get("system_params", {"which_GPUs": "0"})
    if run["pretrained_path"]:
        # an APLA / session checkpoint is loaded strictly after the split (utils/pretrained_loader.py:27-30); an unsplit dinov2 /
        # timm backbone is loaded BEFORE build_apla, as the reference's ViT factory does — a mismatch raises either way
        model, kind = ckpt.build_classifier_from_checkpoint(run["pretrained_path"], mp, sp)
        if is_rank0():
            print(f"[main] loaded {run['pretrained_path']} ({'APLA / session checkpoint' if kind == 'apla' else 'unsplit backbone, before build_apla'})")
    else:
        model = Classifier(mp, sp)
        if pretrained and is_rank0():
            print("\033[93m[main] model_params.pretrained is true but there is no network and no --pretrained_path: random initialisation\033[0m")
    hdt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    from apla_amd.module_trainer import ModulePathTrainer
    # --dr / --dpr / --adr (main.py:101-111) keep the fused step since round 6: stochastic depth inside its LayerNorm kernels (free), the
    # nn.Dropout sites as mask passes around its launches (profiles/r06_c_dropout_bench.md: 15.7 ms against 18.2 ms on the module path at
    # config 2 with --dr 0.1).  The drop-in module path (autograd over the same kernels) takes what the engine refuses, or --module_path.
    module_path = bool(getattr(args, "module_path", False))
    eng = None
    if not module_path:
        try:
            eng = AplaTrainEngine(model, run["batch"], run["img"], device=dev, process_group=dist.group.WORLD if world > 1 else None,
                                  optim=OptimConfig(lr=run["lr"], weight_decay=run["wd"], grad_clipping=run["grad_clipping"]),
                                  compute_dtype=hdt, loss_scale="dynamic" if hdt == torch.float16 else 1.0,
                                  soft_targets=run["soft_targets"])
        except NotImplementedError as e:
            module_path = True
            if is_rank0():
                print(f"[main] the fused step does not take this model ({e}); training on the module path (apla_amd.module_trainer)", flush=True)
    if module_path:
        eng = ModulePathTrainer(model.to(dev), lr=run["lr"], weight_decay=run["wd"], grad_clipping=run["grad_clipping"],
                                process_group=dist.group.WORLD if world > 1 else None, compute_dtype=hdt,
                                loss_scale="dynamic" if hdt == torch.float16 else 1.0, soft_targets=run["soft_targets"])
    data = TensorBatches(run, rank, world, dev, args.steps_per_epoch)
    sched = make_schedule(run, data.steps)
    epochs = 0 if args.test else (1 if args.dry else run["epochs"])   # --test: evaluate the loaded weights only (main.py:219-222)
    iters, t0, seen, loss = 0, time.perf_counter(), 0, None
    tp_knn = bool(params.get("training_params", {}).get("knn_eval")) and not args.dry
    soft_engine = run["soft_targets"]   # an engine built for probability targets has no class-id loss to evaluate with
    if is_rank0():
        print(f"[main] {mp['backbone_type']} APLA r={mp['adaptation']['params']['partial_size']}  {world} GPU(s) x bs {run['batch']}  "
              f"{epochs} epoch(s) x {data.steps} it  lr {run['lr']} wd {run['wd']} schedule {run['sched_types'] or 'constant'}", flush=True)
    for epoch in range(epochs):
        for images, labels in data.epoch(epoch):
            if run["soft_targets"]:  # advanced_aug: the criterion receives probability targets (here: label smoothing only;
                # Mixup / CutMix are data-side augmentations of the out-of-scope input pipeline)
                eps_ = run["label_smoothing"]
                labels = torch.nn.functional.one_hot(labels.long(), run["n_classes"]).float() * (1 - eps_) + eps_ / run["n_classes"]
            loss = eng.train_step(images, labels, lr=sched.lr)
            sched.step()
            iters += 1
            seen += run["batch"] * world
            if iters % run["log_every"] == 0 or iters == 1:
                loss = dist_average_tensor(loss)
                if is_rank0():
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    print(f"[main] epoch {epoch} it {iters}: train_loss {float(loss):.4f}  lr {sched.lr:.3e}  "
                          f"grad_norm {float(eng.grad_norm):.3f}  {seen / dt:.0f} images/s", flush=True)
            if args.dry and iters >= 5:
                break
    synchronize()
    if is_rank0() and run["save_dir"] and not (args.dry or args.debug or args.test):
        os.makedirs(run["save_dir"], exist_ok=True)
        path = os.path.join(run["save_dir"], run["model_name"] + ".pth")
        if module_path:   # the reference's session layout (bases.py:456-464) with FlatAdamW's torch.optim-shaped state
            sess = {"iters": iters, "state_dict": {k: v.detach().cpu().clone() for k, v in eng.model.state_dict().items()},
                    "original_state": None, "optimizer": eng.optimizer.state_dict(), "epoch": epochs, "parameters": params,
                    "best_val_target": 0.0}
            if hdt == torch.float16:
                sess["scaler"] = eng.scaler.state_dict()     # bases.py:465-466: saved under mixed precision
            torch.save(sess, path)
        else:
            torch.save(ckpt.session_dict(eng, iters=iters, epoch=epochs, parameters=params), path)
        print(f"[main] saved {path}")
    if (args.test or args.knn or tp_knn) and is_rank0() and soft_engine:
        print("\033[93m[main] --test / --knn: an engine built for probability targets has no class-id loss to evaluate with; skipped\033[0m", flush=True)
    if (args.test or args.knn or tp_knn) and is_rank0() and not soft_engine:
        # (on the module path too: ModulePathTrainer.forward_only runs the model in eval mode, where every dropout is the identity —
        # the reference's Trainer.test / evaluate run whatever the dropout rates are)
        # Trainer.test / evaluate (defaults/trainer.py:162-345) on rank 0: loss + accuracy on the evaluation batches, optionally
        # kNN metrics against a feature bank of the training batches (the evaluation split of a TensorFile is its
        # `val_images` / `val_labels` entries when present, else the training tensors; synthetic data otherwise)
        from apla_amd.evaluate import Evaluator
        ev = Evaluator(eng, run["n_classes"], knn_nhood=int(params.get("dataset_params", {}).get("knn_nhood", 200)))
        knn = bool(args.knn or tp_knn)
        if knn:
            ev.build_feature_bank(data.epoch(0))
        metrics = ev.evaluate(data.eval_batches(), mode="test" if args.test else "val", knn=knn)
        print("[main] " + "  ".join(f"{k} {v:.4f}" for k, v in metrics.items()), flush=True)
        main.last_metrics = metrics
    return float(loss) if loss is not None else float("nan")


if __name__ == "__main__":
    _args = parse_arguments()
    print(f"\nUSING PARAMS FROM PATH: {os.path.abspath(_args.params_path)}\n")
    _params = update_params_from_args(load_parameters(_args.params_path), _args)
    _gpus = [g for g in str(_params.get("system_params", {}).get("which_GPUs", "0")).split(",") if g != ""]
    if len(_gpus) > 1 and "RANK" not in os.environ:
        from apla_amd.dist import launch
        os.environ.setdefault("HIP_VISIBLE_DEVICES", ",".join(_gpus))
        launch(main, (_params, _args), n_procs=len(_gpus))
    else:
        main(_params, _args)
