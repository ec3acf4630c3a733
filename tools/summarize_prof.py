#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` into a short markdown table (kernel names shortened).
usage: tools/summarize_prof.py <kernel_stats.csv> [steps] [--cfg2]  > profiles/<name>.md
--cfg2: the run was bench.py's default workload (ViT-B/16, bs 128: M = 25216 tokens, D = 768, F = 3072); the GEMM rows then
carry their call site (the kernel's profiling tag, include/apla_hip.h:apla_gemm_nt_ex) and the TFLOP/s of that shape."""
import csv
import re
import sys


EPI = {"0": "STORE", "1": "GELU", "2": "RESIDUAL", "3": "MUL", "4": "SWIGLU", "5": "SWIGLU_BWD", "6": "GELU_FWD"}
TAGS = {"2": "qkv", "3": "proj", "4": "fc2", "5": "dfc1", "6": "dproj", "7": "dqkv", "8": "patch"}
# (N, K) of the call sites at BASELINE config 2; M = 25216 (patch embedding: 25088 rows).  The CLS-only last block runs its
# K/V-only qkv (N = 1536) under the qkv tag: 1 of 12 launches, the average TFLOP/s of that row is read with that in mind.
CFG2 = {"qkv": (2304, 768), "proj": (768, 768), "fc2": (768, 3072), "dfc1": (768, 3072), "dproj": (768, 768), "dqkv": (768, 2304),
        "patch": (768, 768), "GELU": (3072, 768), "MUL": (3072, 768)}


def demangle(name):
    if name.startswith("_Z"):
        import subprocess
        try:
            return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            return name
    return name


def manual_demangle(name):
    """llvm-cxxfilt of this ROCm does not know DF16b (__bf16): decode `_ZN12_GLOBAL__N_1<len><ident>I<targs>E...` by hand."""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    ident = name[m.end():m.end() + n]
    rest = name[m.end() + n:]
    targs = []
    if rest.startswith("I"):
        body = rest[1:]
        while body and not body.startswith("E"):
            if body.startswith("Li"):
                v = re.match(r"Li(\d+)E", body); targs.append(v.group(1)); body = body[v.end():]
            elif body.startswith("Lb"):
                targs.append("true" if body[2] == "1" else "false"); body = body[4:]
            elif body.startswith("DF16b"):
                targs.append("bf16"); body = body[5:]
            elif body[0] == "f":
                targs.append("float"); body = body[1:]
            else:
                break
    return f"{ident}<{', '.join(targs)}>" if targs else ident


def short(name):
    if name.startswith("_ZN12_GLOBAL__N_1") and "DF16b" in name:
        name = manual_demangle(name)
    else:
        name = demangle(name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = name.replace("__bf16", "bf16")
    # the profiler's own demangler garbles <EPI, __bf16, MI, EXP> (DF16b Li5E Li2E -> "bool _Accum, int, ELi2E"): only MI = 5 carries
    # the fourth argument (the K-loop schedule of gemm_persist_kernel, GemmParams::exp)
    name = re.sub(r"bool _Accum, int, ELi(\d+)E", r"bf16, 5, sched \1", name)
    name = re.sub(r"bool _Accum, bool, E, (\d+), (\d+)", r"bf16, true, \1, \2", name)      # gemm_w4_kernel<EPI, __bf16, true, VNS, TAG>
    m = re.match(r"(?:void )?(gemm_\w+_kernel|gemm_nt_kernel)<(\d), (\w+)(.*)>", name)
    if m:
        rest = m.group(4)
        if m.group(1) == "gemm_pp2_kernel":      # <EPI, T, TAG, MI>; the profiler's demangler garbles TAG = 5 into "5, sched MI" (above)
            t = re.match(r",\s*(\d+)(?:,\s*(?:sched )?(\d+))?$", rest)
            if t:
                rest = (f" [{TAGS[t.group(1)]}]" if t.group(1) in TAGS else "") + (", 256 rows" if t.group(2) == "4" else "")
        elif m.group(1) == "gemm_w4_kernel":   # <EPI, T, PRIO, VNS, TAG[, KSPLIT, MI]>
            t = re.match(r",\s*(\w+),\s*(\d+),\s*(\d+)(?:,\s*(\w+),\s*(\d+))?$", rest)
            if t:
                rest = (f" [{TAGS[t.group(3)]}]" if t.group(3) in TAGS else "") + ("" if (t.group(1), t.group(2)) in (("true", "2"), ("1", "2")) else f" prio={t.group(1)} ring={t.group(2)}") \
                       + (", split-K" if t.group(4) in ("true", "1") else "") + (", 128 rows" if t.group(5) == "4" else "")
        return f"{m.group(1)}<{EPI.get(m.group(2), m.group(2))},{m.group(3)}{rest}>"
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    if name.startswith("at::native"):
        name = "torch:" + re.sub(r"<.*", "", name.split("::")[2] if name.count("::") >= 2 else name)
    return name[:70]


def shape_tf(label, avg_us):
    """TFLOP/s of a GEMM row at config 2, from its call-site tag or epilogue; None for other kernels."""
    m = re.search(r"\[(\w+)\]", label)
    big = label.startswith(("gemm_pp2_kernel", "gemm_persist_kernel", "gemm_lw_kernel", "gemm_w4_kernel", "gemm_tp_kernel"))
    key = m.group(1) if m else ("GELU" if big and "<GELU," in label else "MUL" if big and "<MUL," in label else None)
    if key not in CFG2:
        return None
    n, k = CFG2[key]
    return 2.0 * (25088 if key == "patch" else 25216) * n * k / (avg_us * 1e-6) / 1e12


def main():
    cfg2 = "--cfg2" in sys.argv
    if cfg2:
        sys.argv.remove("--cfg2")
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"source: {path}\n")
    print("| kernel | calls | avg us | total ms | % |" + (" ms/step |" if steps else "") + (" TFLOP/s (cfg 2 shape) |" if cfg2 else ""))
    print("|---|---:|---:|---:|---:|" + ("---:|" if steps else "") + ("---:|" if cfg2 else ""))
    for r in rows:
        t = float(r["TotalDurationNs"])
        if t / total < 0.0005:
            continue
        line = f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {t / 1e6:.2f} | {100 * t / total:.1f} |"
        if steps:
            line += f" {t / 1e6 / steps:.3f} |"
        if cfg2:
            tf = shape_tf(short(r["Name"]), float(r["AverageNs"]) / 1e3)
            line += f" {tf:.0f} |" if tf else " |"
        print(line)
    print(f"\ntotal kernel time {total / 1e6:.2f} ms" + (f" = {total / 1e6 / steps:.2f} ms/step over {steps:g} steps" if steps else ""))


if __name__ == "__main__":
    main()
