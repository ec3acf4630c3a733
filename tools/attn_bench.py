#!/usr/bin/env python3
"""Microbenchmark of the attention kernels at the ViT-B/16 bs=128 shape (B=128, N=197, H=12), per variant.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apla_amd import _lib
if os.environ.get("APLA_LIB"):
    _lib.LIB_PATH = os.environ["APLA_LIB"]
from apla_amd import ops
from apla_amd._lib import lib

B, N, H = int(os.environ.get("ATT_B", 128)), int(os.environ.get("ATT_N", 197)), 12
D = 64 * H
scale = 64 ** -0.5
qkv = torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16)
do = torch.randn(B * N, D, device="cuda").to(torch.bfloat16)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(ts)[2]


res = {}
for v in (1, 2, 3, 0):
    ops.set_attn_variant(v)
    o, lse = ops.attn_fwd(qkv, B, N, H, scale)
    dqkv = ops.attn_bwd(qkv, o, do, lse, B, N, H, scale)
    res[v] = (o.clone(), dqkv.clone())
    tf = timeit(lambda: ops.attn_fwd(qkv, B, N, H, scale))
    tb = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, B, N, H, scale))
    print(f"variant {v}: fwd {tf:7.1f} us   bwd {tb:7.1f} us", flush=True)
for v in (2, 3, 0):
    print(f"variant {v} vs 1: fwd max diff", float((res[v][0].float() - res[1][0].float()).abs().max()),
          "bwd max diff", float((res[v][1].float() - res[1][1].float()).abs().max()))
