#!/usr/bin/env python3
"""What a plain device copy / fill / read reaches on this GPU (the practical HBM ceiling the row kernels are judged against)."""
import torch
n = 1 << 28  # 1 GiB fp32
x = torch.randn(n, device="cuda"); y = torch.empty_like(x)
def t(fn, iters=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
c = t(lambda: y.copy_(x)); f = t(lambda: y.fill_(1.0)); r = t(lambda: x.sum())
print(f"copy (1 GiB read + 1 GiB write): {2 * 4 * n / c / 1e12:.2f} TB/s   fill (write only): {4 * n / f / 1e12:.2f} TB/s   sum (read only): {4 * n / r / 1e12:.2f} TB/s")
