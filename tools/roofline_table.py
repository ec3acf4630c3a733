#!/usr/bin/env python3
"""Machine-readable roofline table of a profiled bench.py run at BASELINE config 2 (ViT-B/16, bs 128, r 192: M = 25 216 token rows,
D = 768, F = 3072, N = 197, H = 12): the top kernels of a rocprofv3 `*_kernel_stats.csv`, each with the ALGORITHMIC work of one
launch (2 M N K per product; every operand and result once), its measured average duration, the roofline that
bounds it and the fraction reached — plus, where `--pmc-fetch` / `--pmc-write` directories of separate `rocprofv3 --pmc FETCH_SIZE`
/ `--pmc WRITE_SIZE` passes over the SAME command are given, the HBM traffic the counters saw per launch (FETCH_SIZE doubled on
gfx950, MI355X_MICROARCH.md section HBM).

    tools/roofline_table.py <kernel_stats.csv> <timed steps in the trace> [--pmc-fetch DIR] [--pmc-write DIR] [--top 8] [--tag r06_a]

Writes JSON to stdout: {"config": ..., "kernels": [{kernel, call_site, launches_per_step, us, flop, bytes, bound, achieved, peak,
unit, frac, traffic_bytes}, ...]}.  bench.py embeds the committed file as `roofline.kernels` (a stored record, labelled as such).
"""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_prof import short   # noqa: E402  (kernel-name shortening shared with the markdown tables)

B, N, D, F, H, R, C = 128, 197, 768, 3072, 12, 192, 1000
M = B * N
PEAK_TF, PEAK_TBS = 2500.0, 8.0      # MI355X_MICROARCH.md: dense bf16 MFMA, HBM3E


def gemm(n, k, extra_bytes=0, m=M):
    return 2.0 * m * n * k, 2.0 * (m * k + n * k + m * n) + extra_bytes


# label fragment -> (call site, flop per launch, algorithmic bytes per launch, bound)
def model(label):
    tag = re.search(r"\[(\w+)\]", label)
    tag = tag.group(1) if tag else None
    if label.startswith(("gemm_persist_kernel<GELU", "gemm_lw_kernel<GELU", "gemm_pp2_kernel<GELU")):
        fl, by = gemm(F, D, extra_bytes=2.0 * M * F)          # h and GELU' both written
        return "fc1 + GELU + GELU'", fl, by, "mfma"
    if label.startswith(("gemm_persist_kernel<MUL", "gemm_lw_kernel<MUL", "gemm_pp2_kernel<MUL", "gemm_w4_kernel<MUL")):
        fl, by = gemm(F, D, extra_bytes=2.0 * M * F)          # GELU' read, product written
        return "dfc2 x GELU'", fl, by, "mfma"
    sites = {"qkv": (3 * D, D), "proj": (D, D), "fc2": (D, F), "dfc1": (D, F), "dproj": (D, D), "dqkv": (D, 3 * D), "patch": (D, D)}
    if tag in sites:
        n, k = sites[tag]
        fl, by = gemm(n, k, m=(B * 196 if tag == "patch" else M))
        return tag, fl, by, "mfma"
    if label.startswith("attn_bwd_persist_kernel") or label.startswith("attn_bwd_small_kernel"):
        # 2 x (4 N^2 d) per head of products the reference's autograd executes (bwd counts 2 * att); q, k, v, o, dO read, dq, dk, dv written
        return "attention backward", 2.0 * 4 * N * N * 64 * B * H, 2.0 * M * D * 8, "hbm"
    if label.startswith("attn_fwd_persist_kernel") or label.startswith("attn_fwd_small_kernel"):
        return "attention forward", 4.0 * N * N * 64 * B * H, 2.0 * M * D * 4 + 4.0 * B * H * N, "hbm"
    if label.startswith("ln_fwd_kernel<float, bf16"):
        return "residual add + LayerNorm", 0.0, M * D * (4 + 2 + 4 + 2.0), "hbm"      # res fp32 in / out, branch in, x-hat out
    if label.startswith("ln_bwd2_kernel"):
        return "LayerNorm backward" + (" + trainable columns" if "<true" in label else ""), 0.0, M * D * (2 + 2 + 2 + 2.0) + (2.0 * M * R if "<true" in label else 0), "hbm"
    if label.startswith("proj_dw_partial_kernel"):
        return "column-masked dW1 (4 blocks per launch)", 4 * 2.0 * M * R * D, 4 * 2.0 * (M * R + M * D), "hbm"
    return None


# launches per training step by call site (L = 12 blocks; the last block runs its CLS-only forms): the trace also
# holds warm-up, graph-capture and the bench's own timing launches, so Calls / steps is only approximate
LAUNCHES = {"fc1 + GELU + GELU'": 11, "dfc2 x GELU'": 11, "fc2": 11, "dfc1": 11, "dqkv": 11, "qkv": 12, "proj": 11, "dproj": 10, "patch": 1,
            "attention backward": 10, "attention forward": 11, "residual add + LayerNorm": 24, "LayerNorm backward": 11,
            "LayerNorm backward + trainable columns": 12, "column-masked dW1 (4 blocks per launch)": 3}


def pmc_per_kernel(d, counter):
    """{short kernel label: mean counter value per dispatch} from a rocprofv3 --pmc output directory."""
    if not d:
        return {}
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            acc.setdefault(r.get("Kernel_Name", ""), []).append(float(r["Counter_Value"]))
    return {short(k): sum(v) / len(v) for k, v in acc.items() if v}


def main():
    args = sys.argv[1:]
    def opt(name, default=None):
        if name in args:
            i = args.index(name)
            v = args[i + 1]
            del args[i:i + 2]
            return v
        return default
    fetch_dir, write_dir, top, tag = opt("--pmc-fetch"), opt("--pmc-write"), int(opt("--top", "8")), opt("--tag", "")
    path, steps = args[0], float(args[1])
    fetch, write = pmc_per_kernel(fetch_dir, "FETCH_SIZE"), pmc_per_kernel(write_dir, "WRITE_SIZE")
    rows = sorted(csv.DictReader(open(path)), key=lambda r: -float(r["TotalDurationNs"]))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    out = []
    for r in rows:
        label = short(r["Name"])
        m = model(label)
        if m is None:
            continue
        site, fl, by, bound = m
        us = float(r["AverageNs"]) / 1e3
        per_step = LAUNCHES.get(site, round(float(r["Calls"]) / steps, 2))
        rec = {"kernel": label, "call_site": site, "launches_per_step": per_step, "launches_in_trace": int(r["Calls"]), "us": round(us, 1),
               "ms_per_step": round(us * per_step / 1e3, 3), "share_of_kernel_time": round(float(r["TotalDurationNs"]) / total, 4),
               "flop": fl, "bytes": by, "bound": bound}
        if bound == "mfma":
            rec.update(achieved=round(fl / us / 1e6, 1), peak=PEAK_TF, unit="TFLOP/s", frac=round(fl / us / 1e6 / PEAK_TF, 4))
        else:
            rec.update(achieved=round(by / us / 1e6, 3), peak=PEAK_TBS, unit="TB/s", frac=round(by / us / 1e6 / PEAK_TBS, 4))
        if label in fetch and label in write:
            rec["traffic_bytes"] = round((2.0 * fetch[label] + write[label]) * 1024.0)     # KiB counters; FETCH doubled (gfx950)
            rec["traffic_over_algorithmic"] = round(rec["traffic_bytes"] / by, 3)
        else:
            rec["traffic_bytes"] = None
        out.append(rec)
        if len(out) >= top:
            break
    json.dump({"config": "BASELINE config 2 (ViT-B/16, bs 128, r 192, bf16): M = 25216, D = 768, F = 3072, N = 197, H = 12", "taken_at": tag,
               "source": os.path.basename(path), "steps_in_trace": steps, "kernel_time_ms_per_step": round(total / 1e6 / steps, 3),
               "peak": {"mfma_tflops": PEAK_TF, "hbm_tb_s": PEAK_TBS}, "kernels": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
