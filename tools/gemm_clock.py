#!/usr/bin/env python3
"""At which core clock does a GEMM launch run, and what does that make of the roofline?  (MI355X lowers its clock under a dense
MFMA stream: MI355X_MICROARCH.md "DVFS give-back", items 1 and 6; tools/mfma_peak.hip measures the register-only loop.)

Uses the CLOCK build of the library (tools/build_ablations.sh): every workgroup of the three tiled GEMM kernels stamps s_memtime
(core clock) and s_memrealtime (100 MHz) around its whole run into a buffer of its own.  Per shape and kernel: SECONDS of back-to-back
launches first (the clock settles over milliseconds), then the launch time and the median workgroup's clock — on random operands
and on all-zero operands (same instructions, no toggling in the data paths).

    bash tools/build_ablations.sh && python3 tools/gemm_clock.py [seconds=1.0] > profiles/r03_gemm_clock.md       # GPU only
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apla_amd import _lib
_lib.LIB_PATH = os.environ.get("APLA_LIB") or os.path.join(os.path.dirname(_lib.__file__), "build", "exp", "libapla_CLOCK.so")
from apla_amd import ops
from apla_amd._lib import lib

M = 25216
LAST_CYCLES = 0
PER_CU_CYCLE = 4096.0   # bf16 FLOP per CU and core cycle: 4 SIMDs x one 16x16x32 MFMA (16 384 FLOP) per 16 cycles


def clock_of(reader, n_wg):
    buf = (ctypes.c_ulonglong * 1024)()
    rc = getattr(lib(), reader)(buf)
    assert rc == 0, rc
    g = sorted(buf[2 * i] / buf[2 * i + 1] * 0.1 for i in range(n_wg) if buf[2 * i + 1])
    global LAST_CYCLES
    cyc = sorted(buf[2 * i] for i in range(n_wg) if buf[2 * i + 1])
    LAST_CYCLES = cyc[len(cyc) // 2]          # median workgroup's whole run in core cycles
    return g[len(g) // 2], g[0], g[-1]


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    for fn in ("apla_abl_clock_nt", "apla_abl_clock_w4", "apla_abl_clock_pp2", "apla_abl_clock_lw"):
        if hasattr(lib(), fn):
            getattr(lib(), fn).argtypes = [ctypes.c_void_p]
    cases = [("qkv", 2304, 768, ops.EPI_STORE, 16, "wide 4-wave", "apla_abl_clock_w4", 512),
             ("qkv", 2304, 768, ops.EPI_STORE, 9, "ping-pong", "apla_abl_clock_pp2", 256),
             ("proj", 768, 768, ops.EPI_STORE, 16, "wide 4-wave", "apla_abl_clock_w4", 474),
             ("fc2", 768, 3072, ops.EPI_STORE, 9, "ping-pong", "apla_abl_clock_pp2", 237),
             ("dqkv", 768, 2304, ops.EPI_STORE, 9, "ping-pong", "apla_abl_clock_pp2", 237),
             ("fc1+GELU", 3072, 768, ops.EPI_GELU, 15, "4-wave 128-wide", "apla_abl_clock_nt", 512),
             ("dfc2*gelu'", 3072, 768, ops.EPI_MUL, 15, "4-wave 128-wide", "apla_abl_clock_nt", 512),
             ("fc1+GELU", 3072, 768, ops.EPI_GELU, 18, "loader-wave", "apla_abl_clock_lw", 512),
             ("dfc2*gelu'", 3072, 768, ops.EPI_MUL, 18, "loader-wave", "apla_abl_clock_lw", 512)]
    print(f"# Core clock inside the GEMM launches of the step (config 2, M = {M}; {seconds:g} s of back-to-back launches before each reading)\n")
    print("| launch | kernel | operands | launch us | TFLOP/s | core clock GHz (median workgroup; min / max) | FLOP per CU and core cycle | of 4096 |")
    print("|---|---|---|---:|---:|---:|---:|---:|")
    if os.environ.get("CLOCK_ONLY"):   # e.g. CLOCK_ONLY=qkv,fc1 with an ablation build of one kernel
        cases = [c for c in cases if any(c[0].startswith(k) for k in os.environ["CLOCK_ONLY"].split(",")) and (os.environ.get("CLOCK_KERNEL", "") in c[5])]
    for name, N, K, epi, variant, kname, reader, n_wg in cases:
        for data in ("random", "zeros"):
            if data == "random":
                a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
                w_rm = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
                aux = torch.randn(M, N, device="cuda").to(torch.bfloat16)
            else:
                a, w_rm, aux = (torch.zeros(M, K, device="cuda", dtype=torch.bfloat16), torch.zeros(N, K, device="cuda", dtype=torch.bfloat16),
                                torch.zeros(M, N, device="cuda", dtype=torch.bfloat16))
            w = ops.k_panels(w_rm) if epi == ops.EPI_STORE else w_rm
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            kw = {"aux_out": aux} if epi == ops.EPI_GELU else ({"aux_in": aux} if epi == ops.EPI_MUL else {})
            ops.set_gemm_variant(variant)
            run = lambda: ops.gemm_nt(a, w, None, epilogue=epi, out=out, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            n_settle = int(seconds * 1e3 / (e0.elapsed_time(e1) / 20)) + 1
            for _ in range(n_settle):
                run()
            e0.record()
            for _ in range(50):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 50 * 1e3
            ghz, lo, hi = clock_of(reader, n_wg)
            flop = 2.0 * M * N * K
            per_cycle = flop / 256 / (us * 1e-6 * ghz * 1e9)
            ksteps = (-(-M // (160 if variant in (15, 18) else 320))) * (N // (128 if variant in (15, 18) else 256)) * (K // 64) / n_wg   # mean 64-wide K-steps per workgroup
            print(f"| {name} N={N} K={K} | {kname} | {data} | {us:.1f} | {flop / us / 1e6:.0f} | {ghz:.3f} ({lo:.3f} / {hi:.3f}) | {per_cycle:.0f} | {per_cycle / PER_CU_CYCLE:.3f} | {LAST_CYCLES} cycles per workgroup = {LAST_CYCLES / ksteps:.0f} per 64-wide K-step (epilogue share included) |", flush=True)
    ops.set_gemm_variant(0)


if __name__ == "__main__":
    main()
