import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from apla_amd import _lib
if os.environ.get("APLA_LIB"): _lib.LIB_PATH = os.environ["APLA_LIB"]
from apla_amd import ops
B, N, H = 128, 197, 12; D = 64 * H; scale = 64 ** -0.5
qkv = torch.randn(B * N, 3 * D, device="cuda").to(torch.bfloat16); do = torch.randn(B * N, D, device="cuda").to(torch.bfloat16)
o, lse = ops.attn_fwd(qkv, B, N, H, scale)
for _ in range(30): ops.attn_bwd(qkv, o, do, lse, B, N, H, scale)
torch.cuda.synchronize()
