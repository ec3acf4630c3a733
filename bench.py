#!/usr/bin/env python3
"""Benchmark of the MI355X-native APLA training step (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full APLA training step of BASELINE config 2 — ViT-B/16 (dinov2-shaped: qkv bias, LayerScale, eps 1e-6),
partial_size 192, 1000 classes, 224x224, batch 128 per GPU, bf16 MFMA / fp32 accumulate — forward + cross-entropy +
backward + gradient all-reduce (world > 1) + global-norm clip + AdamW, on a synthetic batch already resident in HBM.
Rank 0 prints ONE JSON line.  `value` is whole-job images/s.  See DESIGN.md §Measurement for the roofline accounting.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per image for the configs of BASELINE.md §3 (forward + APLA backward, GFLOP)
STEP_GF_PER_IMG = {("vit_small", 224, 16): 18.76, ("vit_base", 224, 16): 70.99, ("vit_base", 224, 14): 94.14,
                   ("vit_large", 224, 14): 330.78, ("vit_giant", 518, 14): 7629.5}
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")  # written by tools/measure_round.sh (separate --pmc passes)


def dominant_kernel_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes: FETCH_SIZE and WRITE_SIZE are
    in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read stream (MI355X_MICROARCH.md §HBM), so
    it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  None when no PMC summary has been committed."""
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
        return (2.0 * d["FETCH_SIZE_KiB"] + d["WRITE_SIZE_KiB"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


def build_model(backbone, r, n_classes, img, patch, seed=0):
    from apla_amd.models import Classifier
    torch.manual_seed(seed)
    tp = dict(img_size=[img], patch_size=patch, pretrained_type="dinov2", is_memory_efficient=True,
              block_conf=dict(has_layerscale=True, layerscale_init_values=1.0))
    mp = dict(backbone_type=backbone, n_classes=n_classes, pretrained=False, transformers_params=tp,
              adaptation=dict(mode="apla", params=dict(partial_size=r)))
    return Classifier(mp, dict(which_GPUs="0"))


def time_dominant_kernel(M, D, F, hdt=torch.bfloat16, iters=20):
    """Live HIP-event timing of the dominant kernel: the fc1 GEMM+GELU launch (largest single share of step FLOPs)."""
    from apla_amd import ops
    dev = "cuda"
    a = torch.randn(M, D, device=dev).to(hdt)
    w = (torch.randn(F, D, device=dev) * D ** -0.5).to(hdt)
    b = torch.zeros(F, device=dev)
    h = torch.empty(M, F, device=dev, dtype=hdt)
    g = torch.empty_like(h)
    for _ in range(3):
        ops.gemm_nt(a, w, b, epilogue=ops.EPI_GELU, aux_out=g, out=h)
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters):
        ops.gemm_nt(a, w, b, epilogue=ops.EPI_GELU, aux_out=g, out=h)
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * D * F / (ms * 1e-3) / 1e12


def executed_gflop_per_image(bb, eng, n_classes):
    """SURVEY §8a formula minus what the CLS-only last block skips: forward proj + ffn + att + the Q third of qkv, backward
    proj dX + ffn dX + 2*att (the rows that never reach x[:, 0])."""
    N, D, L, F, r = eng.N, bb.embed_dim, bb.depth, eng.blocks[0].F, eng.blocks[0].r
    qkv, proj, att = 2 * N * D * 3 * D, 2 * N * D * D, 4 * N * N * D
    ffn = (2 * N * D * 2 * F + 2 * N * F * D) if eng.swiglu else 4 * N * D * F
    fwd = L * (qkv + proj + att + ffn) + 2 * eng.Np * 3 * eng.patch ** 2 * D + 2 * D * n_classes
    bwd = (L - 1) * (qkv + proj + ffn + 2 * att) + ffn + L * 2 * N * r * D + 4 * D * n_classes
    dead = (proj + ffn + att + qkv // 3) + (proj + ffn + 2 * att) if L > 1 else 0
    return round((fwd + bwd - dead) / 1e9, 2)


def cpu_baseline(backbone, r, n_classes, img, patch, sample_bs, threads):
    """The CPU oracle (a port of the reference step) timed on this box's host cores on a bounded sample."""
    from oracle import apla_oracle as O
    torch.set_num_threads(threads)
    model = build_model(backbone, r, n_classes, img, patch)
    p = {(k[len("backbone."):] if k.startswith("backbone.") else k): v.detach().clone() for k, v in model.state_dict().items()}
    bb = model.backbone
    cfg = dict(patch=patch, depth=bb.depth, heads=bb.num_heads, r=r)
    g = torch.Generator().manual_seed(0)
    images = torch.randn(sample_bs, 3, img, img, generator=g)
    labels = torch.randint(0, n_classes, (sample_bs,), generator=g)
    state = {}
    O.train_step(images, labels, p, cfg, state)  # warm-up
    t0, n = time.perf_counter(), 0
    while n < 3 or (time.perf_counter() - t0 < 12 and n < 20):
        O.train_step(images, labels, p, cfg, state)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return sample_bs / dt, n, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--backbone", default="vit_base")
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--img", type=int, default=224, help="side measurements at the other BASELINE geometries (cfg 3: "
                    "--backbone vit_large --patch 14 --batch 256 --partial-size 256; cfg 5: --backbone vit_giant --img 518 "
                    "--patch 14 --batch 32 --partial-size 512 --dtype fp16); the default run is config 2")
    ap.add_argument("--patch", type=int, default=16)
    ap.add_argument("--partial-size", type=int, default=192)
    ap.add_argument("--classes", type=int, default=1000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"],
                    help="16-bit operand type (bf16 = BASELINE config 2; fp16 = the reference's autocast dtype, needs --loss-scale)")
    ap.add_argument("--loss-scale", type=float, default=None, help="static loss scale (default 1 for bf16, 1024 for fp16)")
    ap.add_argument("--res-dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--grad-dtype", default="bf16", choices=["fp32", "bf16"])
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-bs", type=int, default=16)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"--gpus {args.gpus} needs a launcher: python -m torch.distributed.run --nproc-per-node {args.gpus} "
                 f"--master-addr 127.0.0.1 bench.py --gpus {args.gpus} …  (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    pg = None
    force_pg = world == 1 and os.environ.get("APLA_FORCE_EXCHANGE") == "1"   # diagnostic: the N > 1 code path on one rank
    if world > 1 or force_pg:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_pg:
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        pg = dist.group.WORLD

    from apla_amd.engine import AplaTrainEngine, OptimConfig
    img, patch = args.img, args.patch
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
    hdt = dt[args.dtype]
    loss_scale = args.loss_scale if args.loss_scale is not None else (1024.0 if args.dtype == "fp16" else 1.0)
    if args.res_dtype == "bf16" or args.grad_dtype == "bf16":  # "bf16" on these switches means "the 16-bit operand type"
        dt["bf16"] = hdt
    model = build_model(args.backbone, args.partial_size, args.classes, img, patch, seed=0)  # same seed => same indices on all ranks
    eng = AplaTrainEngine(model, args.batch, img, res_dtype=dt[args.res_dtype], grad_dtype=dt[args.grad_dtype],
                          optim=OptimConfig(lr=1e-4, weight_decay=1e-5, grad_clipping=1.0), process_group=pg,
                          use_graphs=not args.no_graphs, compute_dtype=hdt, loss_scale=loss_scale)
    g = torch.Generator(device="cuda").manual_seed(rank)  # each rank owns its shard of the global batch
    images = torch.randn(args.batch, 3, img, img, device="cuda", generator=g)
    labels = torch.randint(0, args.classes, (args.batch,), device="cuda", generator=g)
    eng.set_batch(images, labels)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.train_step()
    sync()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.train_step()
    sync()
    elapsed = time.perf_counter() - t0
    loss = float(eng.loss)
    if world > 1:
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt)
    ms_per_step = elapsed / args.steps * 1e3
    img_s = world * args.batch * args.steps / elapsed
    peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30
    # dominant-kernel duration in its place inside the step: HIP events around every fc1+GELU launch of three further training
    # steps (what a rocprofv3 kernel trace of this run shows for the kernel; profiles/*_kernel_stats.md).  Every rank runs
    # these steps — they contain the gradient collectives.
    k_ms = eng.time_fc1_launches(3)
    sync()

    if rank == 0:
        bb = model.backbone
        M = args.batch * eng.N
        from apla_amd import ops as _ops
        with _ops.use_half(hdt):
            iso_ms, _ = time_dominant_kernel(M, bb.embed_dim, eng.blocks[0].F, hdt)    # back-to-back launches of the same GEMM
        Fdim = eng.blocks[0].F * (2 if eng.swiglu else 1)
        k_tf = 2.0 * M * bb.embed_dim * Fdim / (k_ms * 1e-3) / 1e12
        is_cfg2 = (args.backbone, img, patch, args.batch, args.partial_size) == ("vit_base", 224, 16, 128, 192)
        gf = STEP_GF_PER_IMG.get((args.backbone, img, patch))
        step_tf = img_s / world * gf / 1e3 if gf else None
        out = {
            "metric": "images/sec, ViT-B/16 APLA training step bs=128/GPU (whole job)" if is_cfg2 else
                      f"images/sec, {args.backbone}/{patch} APLA training step bs={args.batch}/GPU (whole job; side measurement)", "value": round(img_s, 1),
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.backbone}/{patch} dinov2-shaped APLA partial_size={args.partial_size} full training "
                                   f"step (fwd+CE+bwd+allreduce+clip+AdamW), {img}x{img}, C={args.classes}, "
                                   f"bs={args.batch}/GPU, residual {args.res_dtype}, grad stream {args.grad_dtype if args.grad_dtype == 'fp32' else args.dtype}"
                                   + (f", loss scale {loss_scale:g}" if loss_scale != 1.0 else ""),
                       "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "hip_graphs": not args.no_graphs},
            "images_per_sec_per_gpu": round(img_s / world, 1), "peak_mem_gib": round(peak_mem, 2),
            "final_loss": round(loss, 4),
            "roofline": {"bound": "mfma", "kernel": f"gemm_pp2_kernel<GELU> (apla_gemm_nt, fc1+GELU launch) M={M} N={eng.blocks[0].F} K={bb.embed_dim}",
                         "achieved": round(k_tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(k_tf / PEAK_BF16_TFLOPS, 4), "traffic": dominant_kernel_traffic() if is_cfg2 else None,
                         "algorithmic_bytes": 2.0 * (M * bb.embed_dim + eng.blocks[0].F * bb.embed_dim + 2 * M * eng.blocks[0].F),
                         "kernel_ms": round(k_ms, 4), "kernel_ms_back_to_back": round(iso_ms, 4),
                         "step_achieved": round(step_tf, 1) if step_tf else None,
                         "step_frac": round(step_tf / PEAK_BF16_TFLOPS, 4) if step_tf else None,
                         "step_gflop_per_image": gf,
                         # dead rows of the last block (everything but CLS after its attention, forward and backward) are
                         # not computed: FLOPs actually executed per image, for the reader who wants the honest MFMA rate
                         "step_gflop_per_image_executed": executed_gflop_per_image(bb, eng, args.classes)},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(64, os.cpu_count() or 1)
            v, n, dt_s = cpu_baseline(args.backbone, args.partial_size, args.classes, img, patch, args.cpu_sample_bs, threads)
            out["cpu_baseline"] = {"value": round(v, 2), "unit": "images/s", "cores": threads, "kind": "port",
                                   "sample": f"{n} full steps of the same model at bs={args.cpu_sample_bs} (fp32 oracle, "
                                             f"{dt_s:.2f} s/step, os.cpu_count()={os.cpu_count()})"}
        print(json.dumps(out), flush=True)
    if world > 1 or force_pg:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
