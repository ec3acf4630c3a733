#!/usr/bin/env python3
"""Socket power and core clock of the MI355X while ONE GEMM launch is repeated back to back — the same shape on the different
schedules, on random and on all-zero operands.  The driver exposes both per GPU under /sys/class/drm/card*/device/hwmon/ (power1_input in
microwatts, freq1_input = sclk in Hz, power1_cap); a thread samples them every 20 ms while the main thread keeps the queue full.

What it is for: DESIGN.md section 0d says the GEMM launches of the step sit on a power limit (issue rate x clock is what the chip trades).
This prints the evidence: watts and MHz per case next to the microseconds per launch.

    python3 tools/power_probe.py > gpurun_out/power_probe.md
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apla_amd import ops  # noqa: E402

M = int(os.environ.get("GEMM_M", 25216))
SECONDS = float(os.environ.get("PROBE_SECONDS", 2.5))


from apla_amd.telemetry import Sampler as _Sampler, find_hwmon  # noqa: E402


class Sampler(_Sampler):
    @property
    def stop_flag(self):
        return self._stop_flag


def run_case(hw, fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s = Sampler(hw)
    s.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < SECONDS:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        torch.cuda.synchronize()
        n += 100
        last_us = e0.elapsed_time(e1) * 10.0
    s.stop()
    rows = [r for r in s.rows if r[0] - t0 > 0.5]        # after the clock has settled
    pw = sum(r[1] for r in rows) / max(len(rows), 1)
    fq = sum(r[2] for r in rows) / max(len(rows), 1)
    return last_us, pw, fq, len(rows)


def main():
    hw = find_hwmon()
    cap = int(open(os.path.join(hw, "power1_cap")).read()) / 1e6
    print(f"# Socket power and clock under one repeated GEMM launch (M = {M}, {SECONDS} s per case, sampled every 20 ms from {hw}; power cap {cap:.0f} W)\n")
    print("| launch | schedule | operands | us / launch | TFLOP/s | socket W | sclk MHz (driver) | samples |")
    print("|---|---|---|---:|---:|---:|---:|---:|")
    shapes = [("fc2 (plain store)", 768, 3072, ops.EPI_STORE), ("qkv (plain store)", 2304, 768, ops.EPI_STORE),
              ("fc1 + GELU_FWD", 3072, 768, ops.EPI_GELU_FWD), ("fc1 + GELU (two outputs)", 3072, 768, ops.EPI_GELU)]
    sched = {"ping-pong": 9, "wide 4-wave": 16, "tile-alternating": 17, "4-wave persistent": 15}
    for name, N, K, epi in shapes:
        for zero in (False, True):
            a = (torch.zeros(M, K, device="cuda") if zero else torch.randn(M, K, device="cuda")).to(torch.bfloat16)
            w = (torch.zeros(N, K, device="cuda") if zero else torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
            wi = ops.k_panels(w)
            bias = torch.zeros(N, device="cuda") if zero else torch.randn(N, device="cuda")
            img = epi in (ops.EPI_GELU, ops.EPI_GELU_FWD)
            out = torch.empty((N // 32, M, 32) if img else (M, N), device="cuda", dtype=torch.bfloat16)
            kw = {"aux_out": torch.empty_like(out)} if epi == ops.EPI_GELU else {}
            for sname, v in sched.items():
                if zero and sname not in ("ping-pong", "4-wave persistent", "tile-alternating"):
                    continue
                ops.set_gemm_variant(v)
                image_w = sname != "4-wave persistent"
                try:
                    kn = ops.gemm_kernel_name(M, N, K, epi, out_image=img, aux_image=img and epi == ops.EPI_GELU)
                    want = {"ping-pong": "gemm_pp2", "wide 4-wave": "gemm_w4", "tile-alternating": "gemm_tp", "4-wave persistent": "gemm_persist"}[sname]
                    if not kn.startswith(want):
                        continue                       # this schedule has no instance for the case
                    fn = lambda: ops.gemm_nt(a, wi if image_w else w, bias, epilogue=epi, out=out, **kw)   # noqa: E731
                    us, pw, fq, ns = run_case(hw, fn)
                except Exception as e:   # noqa: BLE001  (a schedule that refuses the case is not a failure of the probe)
                    print(f"| {name} | {sname} | {'zeros' if zero else 'random'} | - | - | - | - | {type(e).__name__} |")
                    continue
                print(f"| {name} | {sname} | {'zeros' if zero else 'random'} | {us:.1f} | {2.0 * M * N * K / us / 1e6:.0f} | {pw:.0f} | {fq:.0f} | {ns} |", flush=True)
    ops.set_gemm_variant(0)
    # the register-only MFMA loop (tools/mfma_peak: no operand movement at all) under the same sampler
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_peak")
    if os.path.exists(exe):
        import json
        import subprocess
        torch.cuda.synchronize()
        smp = Sampler(hw)
        smp.start()
        t0 = time.perf_counter()
        r = subprocess.run([exe, "1.0"], capture_output=True, text=True)
        smp.stop()
        rows = [x for x in smp.rows if x[0] - t0 > 0.3]
        try:
            rec = json.loads(r.stdout.strip().splitlines()[-1])
            top = max(x[1] for x in rows) if rows else float("nan")
            mean = sum(x[1] for x in rows) / max(len(rows), 1)
            print(f"| MFMA only, operands in registers (five back-to-back variants of ~1 s) | tools/mfma_peak | random | - | {rec.get('bf16_16x16x32_1wave_per_simd', 0):.0f} (16x16x32), "
                  f"{rec.get('bf16_32x32x16_1wave_per_simd', 0):.0f} (32x32x16) | mean {mean:.0f}, max {top:.0f} | in-kernel {rec.get('clock_ghz', {}).get('bf16_16x16x32', 0) * 1000:.0f} | {len(rows)} |")
        except (ValueError, IndexError):
            print(f"| MFMA only | tools/mfma_peak | - | - | - | - | - | failed: {r.stderr[-80:]} |")
    # idle reference
    time.sleep(1.0)
    p = int(open(os.path.join(hw, "power1_input")).read()) / 1e6
    print(f"\nidle after the run: {p:.0f} W")


if __name__ == "__main__":
    main()
