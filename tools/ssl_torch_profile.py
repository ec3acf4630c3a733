#!/usr/bin/env python3
"""Which torch operators of the self-supervised iteration cost GPU time?  (torch.profiler over three iterations at the config-4 shape:
operator name, input shapes, calls and device time per iteration; the HIP kernels of this repo show up as ctypes launches without an
operator and are listed by tools/r3_call.sh ssl instead.)  GPU only.

    python3 tools/ssl_torch_profile.py > gpurun_out/ssl_torch_ops.md
"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_bench import build_cfg4


def main():
    tr, batch = build_cfg4()
    for _ in range(3):
        tr.global_step(batch)
    torch.cuda.synchronize()
    n = 3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(n):
            tr.global_step(batch)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dev = getattr(e, "self_device_time_total", None)
        if dev is None:
            dev = getattr(e, "self_cuda_time_total", 0.0)
        if dev > 0:
            rows.append((dev / n / 1e3, e.count / n, e.key, str(e.input_shapes)[:150]))
    rows.sort(reverse=True)
    print("| ms per iteration (self device time) | calls per iteration | operator | input shapes |\n|---:|---:|---|---|")
    for ms, c, k, sh in rows[:45]:
        print(f"| {ms:.3f} | {c:.1f} | {k[:60]} | {sh} |")


if __name__ == "__main__":
    main()
