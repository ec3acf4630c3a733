#!/usr/bin/env python3
"""What regularisation costs the training step at BASELINE config 2 (ViT-B/16, bs 128, r 192, bf16): the fused engine without any, with
stochastic depth (--dpr; inside the LayerNorm kernels, captured graphs), with the reference's nn.Dropout sites (--dr / --adr: mask passes
around the launches, eager, dense last block), and the drop-in module path with the same settings.  One process, one box.  GPU only.

    python3 tools/dropout_bench.py [steps=30]  > profiles/rNN_dropout_bench.md
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apla_amd.engine import AplaTrainEngine, OptimConfig
from apla_amd.models import Classifier
from apla_amd.module_trainer import ModulePathTrainer

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = 128


def build(drop=0.0, attn_drop=0.0, drop_path=0.0):
    torch.manual_seed(0)
    tp = dict(img_size=[224], patch_size=16, pretrained_type="dinov2", is_memory_efficient=True, drop_rate=drop, attn_drop_rate=attn_drop,
              drop_path_rate=drop_path, block_conf=dict(has_layerscale=True, layerscale_init_values=1.0))
    mp = dict(backbone_type="vit_base", n_classes=1000, pretrained=False, transformers_params=tp, adaptation=dict(mode="apla", params=dict(partial_size=192)))
    return Classifier(mp, dict(which_GPUs="0"))


def time_steps(step):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    images = torch.randn(B, 3, 224, 224, device="cuda", generator=g)
    labels = torch.randint(0, 1000, (B,), device="cuda", generator=g)
    print(f"# Step time with regularisation, BASELINE config 2 (ViT-B/16, bs {B}, r 192, bf16), {STEPS} steps after 5 warm-up steps, one process\n")
    print("| path | setting | ms/step | vs the plain fused step |")
    print("|---|---|---:|---:|")
    base = None
    for name, kw in (("none", {}), ("--dpr 0.1 (stochastic depth)", dict(drop_path=0.1)), ("--dr 0.1 (pos_drop, proj_drop, Mlp.drop x 2)", dict(drop=0.1)),
                     ("--adr 0.1 (attention probabilities)", dict(attn_drop=0.1)), ("--dr 0.1 --adr 0.1 --dpr 0.1", dict(drop=0.1, attn_drop=0.1, drop_path=0.1))):
        eng = AplaTrainEngine(build(**kw), B, 224, optim=OptimConfig(lr=1e-4, weight_decay=1e-5, grad_clipping=1.0))
        eng.set_batch(images, labels)
        ms = time_steps(eng.train_step)
        base = base or ms
        mode = "captured graphs" if eng.use_graphs else "eager launches, dense last block"
        print(f"| fused engine ({mode}) | {name} | {ms:.2f} | {ms / base:.3f} |", flush=True)
        del eng
        torch.cuda.empty_cache()
    for name, kw in (("none", {}), ("--dpr 0.1", dict(drop_path=0.1)), ("--dr 0.1", dict(drop=0.1)), ("--dr 0.1 --adr 0.1 --dpr 0.1", dict(drop=0.1, attn_drop=0.1, drop_path=0.1))):
        tr = ModulePathTrainer(build(**kw), lr=1e-4, weight_decay=1e-5, grad_clipping=1.0)
        ms = time_steps(lambda: tr.train_step(images, labels))
        print(f"| module path (autograd over the kernels + FlatAdamW) | {name} | {ms:.2f} | {ms / base:.3f} |", flush=True)
        del tr
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
