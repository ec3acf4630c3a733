#!/usr/bin/env python3
"""In-kernel cycle stamps of the persistent attention forward (diagnostic build -DAPLA_ATT_STAMPS, apla_amd/build/exp/libapla_STAMPS.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from apla_amd import _lib
_lib.LIB_PATH = os.environ.get("APLA_LIB", os.path.join(os.path.dirname(_lib.__file__), "build", "exp", "libapla_STAMPS.so"))
from apla_amd import ops
B, N, H = int(os.environ.get("ATT_B", 128)), int(os.environ.get("ATT_N", 197)), 12
D = 64 * H
scale = 64 ** -0.5
qkv = torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16)
ops.set_attn_variant(3)
for _ in range(3):
    ops.attn_fwd(qkv, B, N, H, scale)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
G = min(B * H, 256)
n = G * 8 * 16
buf = (ctypes.c_ulonglong * n)()
assert raw.apla_attn_debug_dump(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(G, 8, 16).astype(np.float64)
heads = -(-B * H // G)
nt = (N + 31) // 32
names = ["barrier", "deferred store", "S products", "softmax", "PV", "q wait + loop"]
print(f"persistent forward: {G} workgroups x {heads} heads, N={N}; cycles per wave per HEAD (mean over workgroups)")
for w in range(nt):
    print(f" wave {w}: total {a[:, w, :6].sum(-1).mean() / heads:8.0f}   " + "  ".join(f"{names[k]} {a[:, w, k].mean() / heads:6.0f}" for k in range(6)))
w = nt
print(f" loader: total {a[:, w, :3].sum(-1).mean() / heads:8.0f}   issue {a[:, w, 0].mean() / heads:6.0f}  wait {a[:, w, 1].mean() / heads:6.0f}  barrier {a[:, w, 2].mean() / heads:6.0f}")
