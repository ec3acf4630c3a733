#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 --pmc).  usage: gemm_one.py N K epi [variant] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apla_amd import _lib
if os.environ.get("APLA_LIB"):  # another build of the library (tools/build_ablations.sh)
    _lib.LIB_PATH = os.environ["APLA_LIB"]
from apla_amd import ops
from apla_amd._lib import lib
N, K, epi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 4
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
M = int(os.environ.get("GEMM_M", 25216))
dev = "cuda"
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device=dev)
kw = {}
out_dtype = torch.bfloat16
if epi == ops.EPI_RESIDUAL:
    kw = dict(aux_in=torch.randn(M, N, device=dev)); out_dtype = torch.float32
elif epi == ops.EPI_GELU:
    kw = dict(aux_out=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
elif epi == ops.EPI_MUL:
    kw = dict(aux_in=torch.randn(M, N, device=dev).to(torch.bfloat16))
out = torch.empty(M, N, device=dev, dtype=out_dtype)
ops.set_gemm_variant(variant % 100)
if variant >= 100:   # 100 + schedule: W as its K-panel image
    w = ops.k_panels(w)
for _ in range(iters):
    ops.gemm_nt(a, w, bias, epilogue=epi, out=out, **kw)
torch.cuda.synchronize()
