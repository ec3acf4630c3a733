#!/usr/bin/env python3
"""A/B timing harness of the attention backward in ONE process: interleaved rounds on four rotating operand sets, median and minimum
per entry of ATT_XP (cdna guide, methodology rule 24).  Round 5 ran its schedule experiments and ablations through it as run-time flags
in bits 8.. of `variant` (results: profiles/r05_attn_experiments.md); the flags were removed from the kernels afterwards, the library
masks those bits off, so all entries now time the same kernel (base variant ATT_BASE, default 3 = persistent).  ATT_B / ATT_N.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apla_amd import ops

B, N, H = int(os.environ.get("ATT_B", 128)), int(os.environ.get("ATT_N", 197)), int(os.environ.get("ATT_H", 12))
XPS = [int(x) for x in os.environ.get("ATT_XP", "0,1,2,3,4,6").split(",")]
BASE = int(os.environ.get("ATT_BASE", 3))
D = 64 * H
scale = 64 ** -0.5
NBUF = 4   # rotating operand sets (the step never re-reads a warm buffer)
qkvs = [torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16) for _ in range(NBUF)]
dos = [torch.randn(B * N, D, device="cuda").to(torch.bfloat16) for _ in range(NBUF)]
ops.set_attn_variant(BASE)
outs = [ops.attn_fwd(q, B, N, H, scale) for q in qkvs]
ref = None
times = {x: [] for x in XPS}
times_f = []
for rnd in range(7):
    for x in XPS:
        ops.set_attn_variant(BASE | (x << 8))
        for i in range(2):
            ops.attn_bwd(qkvs[i], outs[i][0], dos[i], outs[i][1], B, N, H, scale)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(20):
            i = it % NBUF
            r = ops.attn_bwd(qkvs[i], outs[i][0], dos[i], outs[i][1], B, N, H, scale)
        e1.record()
        torch.cuda.synchronize()
        times[x].append(e0.elapsed_time(e1) / 20 * 1e3)
        r0 = ops.attn_bwd(qkvs[0], outs[0][0], dos[0], outs[0][1], B, N, H, scale)
        if ref is None:
            ops.set_attn_variant(2)
            ref = ops.attn_bwd(qkvs[0], outs[0][0], dos[0], outs[0][1], B, N, H, scale).clone()
        assert torch.equal(r0, ref) or os.environ.get("ATT_NOEQ"), f"xp {x}: result differs from the one-workgroup-per-head kernel"
    ops.set_attn_variant(BASE)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(20):
        ops.attn_fwd(qkvs[it % NBUF], B, N, H, scale)
    e1.record()
    torch.cuda.synchronize()
    times_f.append(e0.elapsed_time(e1) / 20 * 1e3)
print(f"B={B} N={N} H={H} base variant {BASE}, {NBUF} rotating operand sets")
print(f"fwd: median {sorted(times_f)[3]:.1f} us  min {min(times_f):.1f}")
for x in XPS:
    t = sorted(times[x])
    print(f"bwd xp {x}: median {t[3]:.1f} us  min {t[0]:.1f}")
