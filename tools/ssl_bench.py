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


MFMA_PEAK_TFLOPS = 2500.0   # dense 16-bit MFMA peak of one MI355X (/opt/skills/guides/MI355X_MICROARCH.md), bf16 and fp16 alike


def ssl_flops(batch, D, depth, heads, n_glob_tok, n_loc_tok, n_local, rank, n_masked, upper, prototypes, hidden=2048, bott=256,
              nlayers=3, patch=14):
    """ALGORITHMIC matrix FLOPs of one DINOv2-APLA iteration (2 x multiply-add count of every product the iteration needs), by
    part.  Needs = what the mathematics of the step requires, not what any implementation launches:
      teacher   forward of the backbone over the 2B global crops, head over [2B CLS | masked patches] (upperbound-padded rows are
                implementation padding: not counted);
      student   forward over 2B global + n_local*B local crops; backward dX through every frozen GEMM (24 D^2 per token and block,
                like forward) except qkv/attention of block 0 (nothing trainable lies below its projection), dW only for the
                `rank` trainable projection rows of each block (2 T r D); attention forward 4 N^2 D per sequence and block,
                backward 2.5 x that (S recomputed, dP, dV, dQ, dK: five products for the forward's two);
      head      768 -> hidden -> hidden -> bott -> prototypes, fully trainable in the student: backward = dX + dW = 2 x forward.
    Returns {part: FLOPs, ..., 'total': FLOPs}."""
    Tt = 2 * batch * n_glob_tok
    Ts = Tt + n_local * batch * n_loc_tok
    gemm_tok = 24 * D * D                                     # qkv 6 D^2 + proj 2 D^2 + fc1 8 D^2 + fc2 8 D^2
    attn_g, attn_l = 4 * n_glob_tok ** 2 * D, 4 * n_loc_tok ** 2 * D
    pe = 2 * (3 * patch * patch) * D                           # patch embedding, per patch token (frozen: forward only)
    head_row = 2 * (D * hidden + (nlayers - 2) * hidden * hidden + hidden * bott + bott * prototypes)
    f = {}
    f["teacher.backbone_gemm"] = Tt * depth * gemm_tok + 2 * batch * (n_glob_tok - 1) * pe
    f["teacher.attention"] = 2 * batch * depth * attn_g
    f["teacher.head"] = (2 * batch + n_masked) * head_row
    f["student.fwd.backbone_gemm"] = Ts * depth * gemm_tok + (2 * batch * (n_glob_tok - 1) + n_local * batch * (n_loc_tok - 1)) * pe
    f["student.fwd.attention"] = depth * (2 * batch * attn_g + n_local * batch * attn_l)
    f["student.bwd.backbone_dx"] = Ts * (depth * gemm_tok - 8 * D * D)          # minus qkv (6 D^2) and proj (2 D^2) dX of block 0
    r = D if rank == "full" else int(rank)
    f["student.bwd.proj_dw"] = Ts * depth * 2 * r * D
    f["student.bwd.attention"] = 2.5 * (depth - 1) * (2 * batch * attn_g + n_local * batch * attn_l)
    rows_s = (2 + n_local) * batch + n_masked
    f["student.fwd.head"] = rows_s * head_row
    f["student.bwd.head"] = 2 * rows_s * head_row
    f["total"] = sum(f.values())
    return f


def build_cfg4(batch=64, backbone="vit_base", partial_size="128", prototypes=65536, seed=0, dtype=torch.bfloat16, process_group=None,
               force_exchange=False):
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
                       warmup_teacher_temp_epochs=1, compute_dtype=dtype, process_group=process_group, force_exchange=force_exchange)
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
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"], help="fp16 runs under the dynamic loss scale (GradScaler semantics)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="one-rank nccl group with the chunked gradient exchange forced on: the RCCL calls of the world > 1 path on one GPU")
    args = ap.parse_args()
    from apla_amd.ssl.models import _GEOMETRY
    pg = None
    if args.force_exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        pg = dist.group.WORLD
    tr, batch = build_cfg4(args.batch, args.backbone, args.partial_size, args.prototypes,
                           dtype=torch.float16 if args.dtype == "fp16" else torch.bfloat16, process_group=pg, force_exchange=args.force_exchange)
    model = tr.model
    for _ in range(args.warmup):
        tr.global_step(batch)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    from apla_amd import telemetry
    hwmon = telemetry.find_hwmon()
    sampler = telemetry.Sampler(hwmon, period=0.01) if hwmon else None
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.global_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    power = None
    if sampler:
        sampler.stop()
        power = sampler.summary(since=t0)
        if power is not None:
            power["cap_w"] = telemetry.power_cap_w(hwmon)
    n_train = sum(p.numel() for p in model.student.parameters() if p.requires_grad)
    D, depth, heads = _GEOMETRY[args.backbone][:3]
    fl = ssl_flops(args.batch, D, depth, heads, 257, 50, 8, args.partial_size, int(batch["images"]["n_masked_patches"]),
                   int(batch["images"]["upperbound"]), args.prototypes)
    achieved = fl["total"] / dt / 1e12
    roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                "algorithmic_tflop_per_iteration": round(fl["total"] / 1e12, 3),
                "floor_ms_at_peak": round(fl["total"] / (MFMA_PEAK_TFLOPS * 1e12) * 1e3, 2),
                "tflop_by_part": {k: round(v / 1e12, 3) for k, v in fl.items() if k != "total"},
                "note": "whole-iteration figure: algorithmic matrix FLOPs (tools/ssl_bench.py:ssl_flops) / wall time of one iteration"}
    print(json.dumps({"metric": "images/sec, DINOv2-APLA self-supervised iteration (side measurement)", "value": round(args.batch / dt, 1),
                      "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 2),
                      "dtype": args.dtype, "data": "synthetic", "roofline": roofline, "power": power,
                      **({"loss_scale": tr.loss_scale, "skipped_steps": tr.skipped_steps} if args.dtype == "fp16" else {}),
                      **({"exchange": {"chunks_mb": [round((b - a) * 4 / 2 ** 20, 1) for a, b in tr.exchanger.chunks], "backend": "nccl, one rank", "chunk_launches": dict(tr.exchange_counts)}}
                         if args.force_exchange else {}),
                      "config": {"workload": f"{args.backbone}/14 student+teacher, 2x224 + 8x98 crops, bs={args.batch}, partial_size={args.partial_size}, "
                                             f"{args.prototypes} prototypes, masked patches {int(batch['images']['n_masked_patches'])} "
                                             f"(upperbound {batch['images']['upperbound']})", "trainable_params": n_train},
                      "loss": round(float(tr.loss), 4), "loss_terms": {k: round(float(v), 4) for k, v in tr.loss_dict.items()},
                      "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
    if args.force_exchange:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
