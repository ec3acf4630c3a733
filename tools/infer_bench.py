#!/usr/bin/env python3
"""Inference throughput of the engine (AplaTrainEngine.forward_only: the step's forward launches with the forward-only GELU
epilogue) on the bench workload.  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from apla_amd.engine import AplaTrainEngine
B = int(os.environ.get("IB_BATCH", 128))
eng = AplaTrainEngine(bench.build_model("vit_base", 192, 1000, 224, 16), B, 224, use_graphs=False)
g = torch.Generator(device="cuda").manual_seed(0)
images = torch.randn(B, 3, 224, 224, device="cuda", generator=g)
labels = torch.randint(0, 1000, (B,), device="cuda", generator=g)
for _ in range(5):
    eng.forward_only(images, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 30
for _ in range(n):
    eng.forward_only(images, labels)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"forward_only: {dt * 1e3:.2f} ms/batch  {B / dt:.0f} images/s")
