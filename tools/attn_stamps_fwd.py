#!/usr/bin/env python3
"""Cycle split of the short-sequence attention forward (diagnostic build -DAPLA_ATT_STAMPS; see tools/attn_stamps.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from apla_amd import _lib
_lib.LIB_PATH = os.environ.get("APLA_LIB", os.path.join(os.path.dirname(_lib.__file__), "build", "exp", "libapla_STAMPS.so"))
from apla_amd import ops
B, N, H = int(os.environ.get("ATT_B", 128)), int(os.environ.get("ATT_N", 197)), 12
D = 64 * H
qkv = torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16)
for _ in range(3):
    ops.attn_fwd(qkv, B, N, H, 64 ** -0.5)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
W = min(B * H, 1024)
n = W * 16 * 4
buf = (ctypes.c_ulonglong * n)()
assert raw.apla_attn_debug_dump(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(W, 16, 4).astype(np.float64)
nw = (N + 31) // 32
print(f"forward B={B} N={N}: cycles per wave (mean over {W} workgroups): [0] issue loads  [1] wait K,V,q + barrier  [2] key loop  [3] stores")
for w in range(nw):
    print(f" wave {w}: total {a[:, w].sum(-1).mean():8.0f}   " + "  ".join(f"[{k}] {a[:, w, k].mean():7.0f}" for k in range(4)))
