#!/usr/bin/env python3
"""Run the attention forward and backward of one shape a few times (for rocprofv3 --pmc).  usage: attn_one.py [B N H variant iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apla_amd import ops
B, N, H, variant, iters = (int(x) for x in (sys.argv[1:6] + ["128", "197", "12", "0", "6"][len(sys.argv) - 1:]))
D, scale = 64 * H, 64 ** -0.5
qkv = torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16)
do = torch.randn(B * N, D, device="cuda").to(torch.bfloat16)
ops.set_attn_variant(variant)
for _ in range(iters):
    o, lse = ops.attn_fwd(qkv, B, N, H, scale)
    ops.attn_bwd(qkv, o, do, lse, B, N, H, scale)
torch.cuda.synchronize()
