#!/usr/bin/env python3
"""Which lines of apla_amd launch torch's own kernels (cat / copy / cast / elementwise) inside one DINOv2-APLA iteration at the
shape of BASELINE config 4, and what they cost on the device: torch.profiler with Python stacks over a few iterations, every
device kernel that is NOT one of libapla_hip's attributed to the innermost apla_amd frame of the op that launched it.

    python tools/ssl_torch_ops.py [--batch 64] [--steps 3] [--dtype bf16|fp16] > gpurun_out/ssl_torch_ops.md
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    args = ap.parse_args()
    from ssl_bench import build_cfg4
    from torch.profiler import ProfilerActivity, profile
    tr, batch = build_cfg4(args.batch, dtype=torch.float16 if args.dtype == "fp16" else torch.bfloat16)
    for _ in range(3):
        tr.global_step(batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for _ in range(args.steps):
            tr.global_step(batch)
        torch.cuda.synchronize()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = collections.defaultdict(lambda: [0, 0.0])
    total_torch = total_all = 0.0
    for ev in prof.events():
        if not ev.kernels:
            continue
        def ours(n):
            return any(t in n for t in ("apla", "gemm_", "attn_", "ln_fwd", "ln_bwd", "distill", "softmax_center", "adamw", "sumsq", "proj_dw",
                                        "pack_", "gather_cols", "colsum", "patchify", "koleo", "assemble", "splitk"))
        total_all += sum(k.duration for k in ev.kernels)
        dev_us = sum(k.duration for k in ev.kernels if not ours(k.name))
        if dev_us == 0:
            continue
        where = "?"
        for fr in ev.stack or []:
            if "apla_amd" in fr or "ssl_bench" in fr:
                where = fr.replace(root + "/", "")
                break
        key = (ev.name + " " + str(getattr(ev, "input_shapes", ""))[:90], where)
        rows[key][0] += 1
        rows[key][1] += dev_us
        total_torch += dev_us
    copies = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.kernels:
            continue
        for k in ev.kernels:
            if "copyBuffer" in k.name or "Memcpy" in k.name:
                key = ev.name + " " + str(getattr(ev, "input_shapes", ""))[:90]
                copies[key][0] += 1
                copies[key][1] += k.duration
    # every runtime copy call (hipMemcpyAsync / hipMemcpyWithStream: the blit kernel of rocprofv3's table), by the operator chain above it
    calls = collections.defaultdict(int)
    for ev in prof.events():
        if not ev.name.startswith("hipMemcpy"):
            continue
        chain, par = [], ev.cpu_parent
        while par is not None:
            chain.append(par.name + " " + str(getattr(par, "input_shapes", ""))[:60])
            par = par.cpu_parent
        calls[(ev.name, " <- ".join(chain[:3]) or "(no operator: called from Python / the library)")] += 1
    print(f"torch-launched kernels of one iteration ({args.dtype}, batch {args.batch}): {total_torch / args.steps / 1e3:.3f} ms/step of "
          f"{total_all / args.steps / 1e3:.3f} ms/step attributed device time\n")
    print("| op | launched from | launches/step | device us/step |")
    print("|---|---|---:|---:|")
    for (name, where), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"| {name} | {where} | {n / args.steps:.1f} | {us / args.steps:.1f} |")
    if copies:
        print("\nruntime device-to-device copies by operator:\n\n| op | copies/step | device us/step |\n|---|---:|---:|")
        for name, (n, us) in sorted(copies.items(), key=lambda kv: -kv[1][0])[:25]:
            print(f"| {name} | {n / args.steps:.1f} | {us / args.steps:.1f} |")
    if calls:
        print("\nruntime copy calls by operator chain:\n\n| call | under | calls/step |\n|---|---|---:|")
        for (name, chain), c in sorted(calls.items(), key=lambda kv: -kv[1])[:30]:
            print(f"| {name} | {chain} | {c / args.steps:.1f} |")


if __name__ == "__main__":
    main()
