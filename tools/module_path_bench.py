#!/usr/bin/env python3
"""Step time of the DROP-IN module path (apla_amd.vit + build_apla modules, torch autograd, torch.optim.AdamW) on the bench
workload, next to the fused engine: what a user of the plugin API gets without adopting AplaTrainEngine.  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from apla_amd.models import get_params_groups

B = int(os.environ.get("MB_BATCH", 128))
model = bench.build_model("vit_base", 192, 1000, 224, 16).cuda()
opt = torch.optim.AdamW(get_params_groups(model), lr=1e-4, weight_decay=1e-5)
g = torch.Generator(device="cuda").manual_seed(0)
images = torch.randn(B, 3, 224, 224, device="cuda", generator=g)
labels = torch.randint(0, 1000, (B,), device="cuda", generator=g)
crit = torch.nn.CrossEntropyLoss()


def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(model(images).float(), labels)
    loss.backward()
    torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.requires_grad], 1.0)
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"module path: {dt * 1e3:.2f} ms/step  {B / dt:.0f} images/s  loss {float(loss):.4f}  peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
