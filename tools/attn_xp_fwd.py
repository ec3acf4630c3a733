#!/usr/bin/env python3
"""Timing harness of the attention forward in one process (interleaved rounds, four rotating operand sets; ATT_BASE = kernel variant:
0 auto / persistent, 2 one workgroup per head).  Round 5's ablation flags travelled in bits 8.. of `variant` (ATT_XP) and were removed
from the kernels afterwards: profiles/r05_attn_experiments.md.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apla_amd import ops
B, N, H = int(os.environ.get("ATT_B", 128)), int(os.environ.get("ATT_N", 197)), int(os.environ.get("ATT_H", 12))
XPS = [int(x) for x in os.environ.get("ATT_XP", "0").split(",")]
BASE = int(os.environ.get("ATT_BASE", 0))
D = 64 * H
scale = 64 ** -0.5
NBUF = 4
qkvs = [torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16) for _ in range(NBUF)]
ops.set_attn_variant(2)
ref = ops.attn_fwd(qkvs[0], B, N, H, scale)
times = {x: [] for x in XPS}
for rnd in range(7):
    for x in XPS:
        ops.set_attn_variant(BASE | (x << 8))
        for i in range(2):
            ops.attn_fwd(qkvs[i], B, N, H, scale)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(20):
            ops.attn_fwd(qkvs[it % NBUF], B, N, H, scale)
        e1.record()
        torch.cuda.synchronize()
        times[x].append(e0.elapsed_time(e1) / 20 * 1e3)
        r = ops.attn_fwd(qkvs[0], B, N, H, scale)
        if not os.environ.get("ATT_NOEQ"):
            assert torch.equal(r[0], ref[0]) and torch.equal(r[1], ref[1]), f"xp {x}: forward differs from variant 2"
print(f"B={B} N={N} H={H} base variant {BASE}, {NBUF} rotating operand sets")
for x in XPS:
    t = sorted(times[x])
    print(f"fwd xp {x}: median {t[3]:.1f} us  min {t[0]:.1f}")
