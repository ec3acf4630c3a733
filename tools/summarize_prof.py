#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` into a short markdown table (kernel names shortened).
usage: tools/summarize_prof.py <kernel_stats.csv> [steps]  > profiles/<name>.md"""
import csv
import re
import sys


EPI = {"0": "STORE", "1": "GELU", "2": "RESIDUAL", "3": "MUL", "4": "SWIGLU", "5": "SWIGLU_BWD"}


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
    m = re.match(r"(?:void )?(gemm_\w+_kernel|gemm_nt_kernel)<(\d), (\w+)(.*)>", name)
    if m:
        return f"{m.group(1)}<{EPI.get(m.group(2), m.group(2))},{m.group(3)}{m.group(4)}>"
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    if name.startswith("at::native"):
        name = "torch:" + re.sub(r"<.*", "", name.split("::")[2] if name.count("::") >= 2 else name)
    return name[:70]


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"source: {path}\n")
    print("| kernel | calls | avg us | total ms | % |" + (" ms/step |" if steps else ""))
    print("|---|---:|---:|---:|---:|" + ("---:|" if steps else ""))
    for r in rows:
        t = float(r["TotalDurationNs"])
        if t / total < 0.0005:
            continue
        line = f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {t / 1e6:.2f} | {100 * t / total:.1f} |"
        if steps:
            line += f" {t / 1e6 / steps:.3f} |"
        print(line)
    print(f"\ntotal kernel time {total / 1e6:.2f} ms" + (f" = {total / 1e6 / steps:.2f} ms/step over {steps:g} steps" if steps else ""))


if __name__ == "__main__":
    main()
