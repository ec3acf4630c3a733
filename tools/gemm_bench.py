#!/usr/bin/env python3
"""Microbenchmark of apla_gemm_nt on the GEMM shapes of the APLA step (ViT-B/16, bs=128 => M=25216), per schedule
variant, interleaved rounds in one process (cdna guide rule 24), random data (rule 25).  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apla_amd import _lib
if os.environ.get("APLA_LIB"):  # A/B experiments against another build of the library (tool only)
    _lib.LIB_PATH = os.environ["APLA_LIB"]
from apla_amd import ops
from apla_amd._lib import lib

M = int(os.environ.get("GEMM_M", 25216))
ONLY = os.environ.get("GEMM_ONLY")
SHAPES = [("qkv", 2304, 768, ops.EPI_STORE), ("proj", 768, 768, ops.EPI_STORE), ("fc2", 768, 3072, ops.EPI_STORE),
          ("fc1+gelu_fwd", 3072, 768, ops.EPI_GELU_FWD), ("proj+res", 768, 768, ops.EPI_RESIDUAL), ("fc1+gelu", 3072, 768, ops.EPI_GELU),
          ("fc2+res", 768, 3072, ops.EPI_RESIDUAL), ("dfc2*g", 3072, 768, ops.EPI_MUL), ("dfc1", 768, 3072, ops.EPI_STORE),
          ("dproj", 768, 768, ops.EPI_STORE), ("dqkv", 768, 2304, ops.EPI_STORE)]
if os.environ.get("GEMM_SHAPES"):  # custom shapes "name:N:K[:epilogue]", e.g. GEMM_M=65792 GEMM_SHAPES=fc2:1024:4096,qkv:3072:1024,fc1:4096:1024:gelu
    EPIS = {"store": ops.EPI_STORE, "gelu": ops.EPI_GELU, "gelu_fwd": ops.EPI_GELU_FWD, "mul": ops.EPI_MUL}
    SHAPES = [(f[0], int(f[1]), int(f[2]), EPIS[f[3]] if len(f) > 3 else ops.EPI_STORE) for f in (t.split(":") for t in os.environ["GEMM_SHAPES"].split(","))]
if os.environ.get("GEMM_SQ"):  # square problems (e.g. GEMM_SQ=4096,8192) to compare with published figures for other kernels
    SHAPES = [(f"sq{v}", int(v), int(v), ops.EPI_STORE, int(v)) for v in os.environ["GEMM_SQ"].split(",")]
VARIANTS = [int(v) for v in os.environ.get("GEMM_VARIANTS", "0,2,3").split(",")]
ROUNDS, ITERS = 5, 10
ROTATE = int(os.environ.get("GEMM_ROTATE", 1))   # distinct output (and second-operand) buffers walked round-robin, as the step does
                                                 # with its per-layer saved activations (cold pages / cache state instead of a warm loop)


def main():
    dev = "cuda"
    res = {}
    for name, N, K, epi, *rest in SHAPES:
        M = rest[0] if rest else globals()["M"]
        if ONLY and name not in ONLY.split(','):
            continue
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        kw = {}
        if epi == ops.EPI_RESIDUAL:
            kw = dict(aux_in=torch.randn(M, N, device=dev), out_dtype=torch.float32)
        elif epi == ops.EPI_GELU:
            kw = dict(aux_out=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
        elif epi == ops.EPI_MUL:
            kw = dict(aux_in=torch.randn(M, N, device=dev).to(torch.bfloat16))
        out = torch.empty(M, N, device=dev, dtype=kw.get("out_dtype", torch.bfloat16))
        if os.environ.get("GEMM_IMAGES") and epi in (ops.EPI_GELU, ops.EPI_GELU_FWD, ops.EPI_MUL):   # outputs / gelu' as K-panel images, as the step keeps them
            out = out.view(N // 32, M, 32)
            if "aux_out" in kw:
                kw["aux_out"] = kw["aux_out"].view(N // 32, M, 32)
            if "aux_in" in kw:
                kw["aux_in"] = ops.k_panels(kw["aux_in"])
        rot = [(out, kw)]
        for _ in range(ROTATE - 1):
            rot.append((torch.empty_like(out), {k: (torch.empty_like(x) if k == "aux_out" else x.clone() if torch.is_tensor(x) else x)
                                                for k, x in kw.items()}))
        ref = None
        w_rm, a_rm = w, a
        for v in VARIANTS:
            ops.set_gemm_variant(v % 100)
            ops._GEMM_EXP = (v // 1000) % 8       # 1000 * e + ...: experiment e of the 4-wave kernel's K loop (GemmParams::exp)
            # v = 100 + schedule: W passed as its K-panel image; 300 + schedule: A too (ping-pong kernel only)
            w = ops.k_panels(w_rm) if (v % 1000) >= 100 else w_rm
            a = ops.k_panels(a_rm) if (v % 1000) >= 300 else a_rm
            o = ops.gemm_nt(a, w, bias, epilogue=epi, out=out, **{k: x for k, x in kw.items() if k != "out_dtype"}).clone()
            if ref is None:
                ref = o
            else:
                assert torch.equal(o, ref), f"variant {v} differs on {name}: {(o.float() - ref.float()).abs().max()}"
        times = {v: [] for v in VARIANTS}
        for _ in range(ROUNDS):
            for v in VARIANTS:
                ops.set_gemm_variant(v % 100)
                ops._GEMM_EXP = (v // 1000) % 8
                w = ops.k_panels(w_rm) if (v % 1000) >= 100 else w_rm
                a = ops.k_panels(a_rm) if (v % 1000) >= 300 else a_rm
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for it in range(ITERS):
                    o_, kw_ = rot[it % ROTATE]
                    ops.gemm_nt(a, w, bias, epilogue=epi, out=o_, **{k: x for k, x in kw_.items() if k != "out_dtype"})
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / ITERS)
        fl = 2.0 * M * N * K
        line = f"{name:10s} N={N:5d} K={K:5d} "
        for v in VARIANTS:
            t = sorted(times[v])[len(times[v]) // 2]
            res.setdefault(v, 0.0)
            res[v] += t
            line += f"| v{v}: {t * 1e3:7.1f} us {fl / t / 1e9:7.1f} TF "
        print(line, flush=True)
    print("sum of medians (one of each GEMM): " + "  ".join(f"v{v}: {t:.3f} ms" for v, t in res.items()))


if __name__ == "__main__":
    main()
