#!/usr/bin/env python3
"""Dump per-phase s_memtime stamps of workgroup 0 of the pp2 GEMM kernel (diagnostic build, variant 205)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apla_amd import ops
from apla_amd._lib import lib
M, N, K = 25216, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 3072
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
dbg = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)  # aux_out: reused as the stamp buffer (8 waves x 512 x 8 B)
p = ops.lib()
lib().apla_gemm_set_variant(9)
for _ in range(3):
    ops.gemm_nt(a, w, None, out=out)
lib().apla_gemm_set_variant(205)
rc = p.apla_gemm_nt(a.data_ptr(), K, w.data_ptr(), K, None, out.data_ptr(), N, M, N, K, 0, 0, None, 0, dbg.data_ptr(), N, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
st = dbg.view(torch.int64).flatten()[:8 * 512].reshape(8, 512).cpu()
base = int(st[0, 0])
for wv in (0, 4):
    t = (st[wv] - base).tolist()
    n = max(i for i, x in enumerate(st[wv].tolist()) if x != 0) + 1
    print(f"wave {wv}: {n} stamps; first 40 deltas:", [t[i + 1] - t[i] for i in range(min(40, n - 1))])
    if n > 120:
        print(f"   steady-state deltas [100:124]:", [t[i + 1] - t[i] for i in range(100, 124)])
