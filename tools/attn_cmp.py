#!/usr/bin/env python3
"""Fused vs split attention backward on identical inputs (same o / lse): determinism and element-wise agreement."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apla_amd import ops
from apla_amd._lib import lib
B, N, H = int(os.environ.get("ATT_B", 16)), int(os.environ.get("ATT_N", 197)), 12
D = 64 * H; scale = 64 ** -0.5
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(B * N, 3 * D, device="cuda", generator=g) * float(os.environ.get("ATT_STD", 1.0))).to(torch.bfloat16)
do = torch.randn(B * N, D, device="cuda", generator=g).to(torch.bfloat16)
o, lse = ops.attn_fwd(qkv, B, N, H, scale)
ops.set_attn_variant(0)
a1 = ops.attn_bwd(qkv, o, do, lse, B, N, H, scale).clone()
a2 = ops.attn_bwd(qkv, o, do, lse, B, N, H, scale).clone()
ops.set_attn_variant(1)
b1 = ops.attn_bwd(qkv, o, do, lse, B, N, H, scale).clone()
ops.set_attn_variant(0)
print("fused deterministic:", torch.equal(a1, a2))
d = (a1.float() - b1.float()).abs()
nz = d > 0
print("differing elements:", int(nz.sum()), "of", d.numel(), "max abs", float(d.max()), "max rel", float((d / (b1.float().abs() + 1e-6))[nz].max()) if nz.any() else 0.0)
for name, sl in (("dQ", slice(0, D)), ("dK", slice(D, 2 * D)), ("dV", slice(2 * D, 3 * D))):
    print(name, int(nz[:, sl].sum()))
rows = nz.any(1).nonzero().flatten()
print("rows with differences (token index within sequence):", sorted(set((rows % N).tolist()))[:40])
