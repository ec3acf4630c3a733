#!/usr/bin/env python3
"""Microbenchmark of apla_proj_dw (column-masked projection weight gradient) at the ViT-B/16 bs=128 shape.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apla_amd import _lib
if os.environ.get("APLA_LIB"):
    _lib.LIB_PATH = os.environ["APLA_LIB"]
from apla_amd import ops

M, r, D = int(os.environ.get("DW_M", 25216)), int(os.environ.get("DW_R", 192)), int(os.environ.get("DW_D", 768))   # e.g. DW_M=65792 DW_R=256 DW_D=1024 (config 3), DW_M=58496 DW_R=128 (self-supervised student)
dyg = torch.randn(M, r, device="cuda").to(torch.bfloat16)
x = torch.randn(M, D, device="cuda").to(torch.bfloat16)
dW, db = torch.zeros(r, D, device="cuda"), torch.zeros(r, device="cuda")
ws = torch.empty(256 * (r * D + r), device="cuda")
for _ in range(3):
    ops.proj_dw(dyg, x, dW, db, workspace=ws)
ref = dyg.float().T @ x.float()
print("rel err", float((dW - ref).abs().max() / ref.abs().max()))
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.proj_dw(dyg, x, dW, db, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
print(f"proj_dw M={M} r={r} D={D}: {sorted(ts)[2]:.1f} us (partial + reduce)  slabs={os.environ.get('APLA_DW_SLABS')}")
