#!/usr/bin/env python3
"""Where the fused attention backward spends its cycles (diagnostic build -DAPLA_ATT_STAMPS of attention.hip, linked as
apla_amd/build/exp/libapla_STAMPS.so): per-wave s_memtime sums of the kernel's segments, averaged over workgroups.
The stamped build forbids overlaps the product build has: read the SHARES, not the total (cdna guide §7, in-kernel stamps)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from apla_amd import _lib
_lib.LIB_PATH = os.environ.get("APLA_LIB", os.path.join(os.path.dirname(_lib.__file__), "build", "exp", "libapla_STAMPS.so"))
from apla_amd import ops
from apla_amd._lib import lib

B, N, H = int(os.environ.get("ATT_B", 128)), int(os.environ.get("ATT_N", 197)), 12
D = 64 * H
scale = 64 ** -0.5
qkv = torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16)
do = torch.randn(B * N, D, device="cuda").to(torch.bfloat16)
o, lse = ops.attn_fwd(qkv, B, N, H, scale)
for _ in range(3):
    ops.attn_bwd(qkv, o, do, lse, B, N, H, scale)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
n = min(B * H, 4096) * 4 * 16
buf = (ctypes.c_ulonglong * n)()
rc = raw.apla_attn_debug_dump(buf, n)
assert rc == 0, rc
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4, 16).astype(np.float64)
names = ["P1 block start (row loads, delta)", "P1 wait K,V / rows", "P1 reads+stage-1 issue", "P1 softmax VALU", "P1 stage-2 issue",
         "P1 dQ stores", "barrier after P1", "P2 block start", "P2 wait Q,dO / rows", "P2 reads+stage-1 issue", "P2 softmax VALU",
         "P2 stage-2 issue", "P2 dK/dV stores"]
tot = a[:, :, :13].sum(-1)
print(f"B={B} N={N} H={H}: workgroups {a.shape[0]}; cycles per wave per head, mean over workgroups (100 MHz-independent: s_memtime ticks)")
for w in range(4):
    print(f" wave {w}: total {tot[:, w].mean():9.0f}   " + "  ".join(f"[{k}] {a[:, w, k].mean():7.0f}" for k in range(13)))
print("segments:", "; ".join(f"[{k}] {nm}" for k, nm in enumerate(names)))
nt = (N + 31) // 32
for w in (0, 3):
    blocks = len(range(w, nt, 4))
    if blocks:
        print(f" wave {w} ({blocks} blocks): per tile  P1 = {a[:, w, 2].mean() / blocks / nt:6.0f} + {a[:, w, 3].mean() / blocks / nt:6.0f} + {a[:, w, 4].mean() / blocks / nt:6.0f}"
              f"   P2 = {a[:, w, 9].mean() / blocks / nt:6.0f} + {a[:, w, 10].mean() / blocks / nt:6.0f} + {a[:, w, 11].mean() / blocks / nt:6.0f}")

# persistent kernel (variant 3): 8 waves per workgroup, one workgroup per CU, stamps of blockIdx < 512
ops.set_attn_variant(3)
for _ in range(3):
    ops.attn_bwd(qkv, o, do, lse, B, N, H, scale)
torch.cuda.synchronize()
G = min(B * H, 256)
n = G * 8 * 16
buf = (ctypes.c_ulonglong * n)()
assert raw.apla_attn_debug_dump(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(G, 8, 16).astype(np.float64)
heads = -(-B * H // G)
pn = ["first rows+setA", "barrier X", "P1 start (DMA issue, delta)", "P1 loop", "dQ stores", "wait set B", "barrier Y", "P2 start (DMA issue)",
      "P2 loop", "prefetch + dK/dV stores", "wait rows/set A"]
print(f"persistent kernel: {G} workgroups x {heads} heads; cycles per wave per HEAD (mean over workgroups)")
for w in range(8):
    print(f" wave {w}: total {a[:, w, :11].sum(-1).mean() / heads:8.0f}   " + "  ".join(f"[{k}] {a[:, w, k].mean() / heads:6.0f}" for k in range(11)))
print("segments:", "; ".join(f"[{k}] {nm}" for k, nm in enumerate(pn)))
