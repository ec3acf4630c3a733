#!/usr/bin/env python3
"""Side measurement: the DINOv2-APLA training iteration at the shape of BASELINE config 4 (ViT-B/14, 2 x 224 global + 8 x 98
local crops, DINO head 768 -> 2048 -> 2048 -> 256 -> 65 536 prototypes, iBOT masking, KoLeo, EMA teacher) on one MI355X,
synthetic crops, random-init weights.  Prints one JSON line (images/s = source images per second).

    python tools/ssl_bench.py [--batch 64] [--steps 10] [--warmup 3] [--partial-size 128|full] [--backbone vit_base]
"""
import argparse
import json
import os
import random
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_cfg4(batch=64, backbone="vit_base", partial_size="128", prototypes=65536, seed=0):
    """The trainer and ONE collated batch (on the GPU) at the shape of BASELINE config 4; deterministic in `seed`."""
    from apla_amd.ssl import DINOv2, Dinov2Trainer, MaskingGenerator, collate_data_and_cast
    from apla_amd.ssl.collate import synthetic_samples
    from apla_amd.ssl.models import _GEOMETRY
    torch.manual_seed(seed)
    random.seed(seed)
    D, depth = _GEOMETRY[backbone][:2]
    if partial_size == "full":
        adaptation, gpus = dict(mode="apla", params=dict(partial_size="full")), "0,1"
    else:
        r = int(partial_size)
        g = torch.Generator().manual_seed(seed)
        f = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
        json.dump({f"block_{i}": torch.randperm(D, generator=g)[:r].tolist() for i in range(depth)}, f)
        f.close()
        adaptation, gpus = dict(mode="apla", params=dict(partial_size=r, inds_path=f.name)), "0"
    params = dict(
        model_params=dict(backbone_type=backbone, pretrained=False, adaptation=adaptation,
                          transformers_params=dict(student=dict(patch_size=14, pre_img_size=518, layerscale=1e-5, interpolate_offset=0.1,
                                                                interpolate_antialias=False, drop_path_rate=0, num_register_tokens=0)),
                          dinov2=dict(centering="centering",
                                      dino=dict(loss_weight=1.0, head_n_prototypes=prototypes, head_bottleneck_dim=256, head_nlayers=3,
                                                head_hidden_dim=2048, koleo_loss_weight=0.1),
                                      ibot=dict(loss_weight=1.0, mask_sample_probability=0.5, mask_ratio_min_max=[0.1, 0.5], separate_head=False))),
        crops_params=dict(n_global_crops=2, n_local_crops=8), system_params=dict(which_GPUs=gpus))
    model = DINOv2(params).cuda().train()
    tr = Dinov2Trainer(model, iters_per_epoch=1000, epochs=10, lr=1e-3, weight_decay=1e-5, grad_clipping=3.0, freeze_last_layer_epochs=1,
                       warmup_teacher_temp_epochs=1)
    mg = MaskingGenerator(input_size=(16, 16), max_num_patches=0.5 * 16 * 16)
    gen = torch.Generator().manual_seed(seed + 1)
    batch_ = collate_data_and_cast(synthetic_samples(batch, 224, 98, 8, gen), n_global_crops=2, n_local_crops=8,
                                   mask_ratio_tuple=(0.1, 0.5), mask_probability=0.5, dtype=torch.float32, n_tokens=256, mask_generator=mg)
    for k, v in batch_["images"].items():   # inputs resident in HBM before the timed region
        if torch.is_tensor(v):
            batch_["images"][k] = v.cuda()
    return tr, batch_


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--backbone", default="vit_base")
    ap.add_argument("--partial-size", default="128")
    ap.add_argument("--prototypes", type=int, default=65536)
    args = ap.parse_args()
    tr, batch = build_cfg4(args.batch, args.backbone, args.partial_size, args.prototypes)
    model = tr.model
    for _ in range(args.warmup):
        tr.global_step(batch)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.global_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    n_train = sum(p.numel() for p in model.student.parameters() if p.requires_grad)
    print(json.dumps({"metric": "images/sec, DINOv2-APLA self-supervised iteration (side measurement)", "value": round(args.batch / dt, 1),
                      "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 2),
                      "dtype": "bf16", "data": "synthetic",
                      "config": {"workload": f"{args.backbone}/14 student+teacher, 2x224 + 8x98 crops, bs={args.batch}, partial_size={args.partial_size}, "
                                             f"{args.prototypes} prototypes, masked patches {int(batch['images']['n_masked_patches'])} "
                                             f"(upperbound {batch['images']['upperbound']})", "trainable_params": n_train},
                      "loss": round(float(tr.loss), 4), "loss_terms": {k: round(float(v), 4) for k, v in tr.loss_dict.items()},
                      "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)


if __name__ == "__main__":
    main()
