#!/usr/bin/env python3
"""Where do the two wave roles of the tile-alternating GEMM (gemm_tp.hip, schedule 17) spend their cycles?  Uses the TPSTAMPS build
(tools/build_ablations.sh tp): wave 0 of each wave group accumulates s_memtime differences per activity.  GPU only.

    bash tools/build_ablations.sh tp && python3 tools/tp_stamps.py > profiles/r04_tp_stamps.md
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apla_amd import _lib
_lib.LIB_PATH = os.environ.get("APLA_LIB") or os.path.join(os.path.dirname(_lib.__file__), "build", "exp", "libapla_TPSTAMPS.so")
from apla_amd import ops
from apla_amd._lib import lib

M = 25216
NAMES = ["(unused)", "compute periods (asm block: barriers, LDS-DMA, products)", "barrier (service role)", "whole run, 100 MHz ticks", "slices", "vmcnt waits (service role)", "whole run, core cycles", "K-steps computed"]


def main():
    lib().apla_abl_tp_stamps.argtypes = [ctypes.c_void_p]
    cases = [("qkv", 2304, 768, ops.EPI_STORE), ("fc2", 768, 3072, ops.EPI_STORE), ("fc1+GELU_FWD", 3072, 768, ops.EPI_GELU_FWD), ("fc1+GELU", 3072, 768, ops.EPI_GELU)]
    print(f"# gemm_tp.hip role stamps (median workgroup, core cycles; M = {M}); images: W as K-panel image, outputs of the GELU epilogues as images\n")
    for name, N, K, epi in cases:
        for exp in (0,):
            a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            w = ops.k_panels((torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16))
            bias = torch.randn(N, device="cuda")
            kw = {}
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            if epi != ops.EPI_STORE:
                out = out.view(N // 32, M, 32)
            if epi == ops.EPI_GELU:
                kw["aux_out"] = torch.empty(N // 32, M, 32, device="cuda", dtype=torch.bfloat16)
            ops.set_gemm_variant(17)
            ops._GEMM_EXP = exp
            for _ in range(20):
                ops.gemm_nt(a, w, bias, epilogue=epi, out=out, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm_nt(a, w, bias, epilogue=epi, out=out, **kw)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            buf = (ctypes.c_ulonglong * (256 * 2 * 8))()
            assert lib().apla_abl_tp_stamps(buf) == 0
            print(f"## {name} N={N} K={K} ring {5 if exp == 0 else 4}: {us:.1f} us per launch (stamped build)\n")
            print("| activity | group 0 median | group 1 median |")
            print("|---|---:|---:|")
            for i, nm in enumerate(NAMES):
                cols = []
                for g in range(2):
                    v = sorted(buf[(b * 2 + g) * 8 + i] for b in range(256))
                    cols.append(v[128])
                print(f"| {nm} | {cols[0]} | {cols[1]} |")
            clk = sorted(buf[(b * 2) * 8 + 6] / max(1, buf[(b * 2) * 8 + 3]) * 0.1 for b in range(256))
            fl = 2.0 * M * N * K / 256
            cyc = sorted(buf[(b * 2) * 8 + 6] for b in range(256))[128]
            print(f"\ncore clock inside the launch (median workgroup): {clk[128]:.2f} GHz (min {clk[0]:.2f}, max {clk[-1]:.2f}); "
                  f"{fl / cyc:.0f} FLOP per CU and core cycle = {fl / cyc / 4096:.2f} of the matrix pipes' 4096\n")


if __name__ == "__main__":
    main()
