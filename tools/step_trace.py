#!/usr/bin/env python3
"""Which kernels make up ONE fused training step?  (torch launches / buffer copies inside the step.)

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/steptrace -- python3 tools/step_trace.py run
    python3 tools/step_trace.py parse $(find gpurun_out/steptrace -name "*kernel_trace.csv") > profiles/r03_step_launches.md

`run` builds BASELINE config 2, warms up, and brackets ONE eager step (same launch sequence as the captured graphs) and ONE
graph-replayed step with marker launches — a torch.cumsum on a 3 x 5 tensor, a kernel nothing else in the process runs.
`parse` lists, in launch order, every kernel between the markers with its duration, and counts the torch / runtime-copy rows.
"""
import csv
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from bench import build_model
    from apla_amd.engine import AplaTrainEngine, OptimConfig
    model = build_model("vit_base", 192, 1000, 224, 16)
    eng = AplaTrainEngine(model, 128, 224, optim=OptimConfig(lr=1e-4, weight_decay=1e-5, grad_clipping=1.0))
    g = torch.Generator(device="cuda").manual_seed(0)
    eng.set_batch(torch.randn(128, 3, 224, 224, device="cuda", generator=g), torch.randint(0, 1000, (128,), device="cuda", generator=g))
    for _ in range(3):
        eng.train_step()
    torch.cuda.synchronize()
    mark = torch.ones(3, 5, device="cuda")

    def marker():
        torch.cumsum(mark, 1)

    marker()                      # eager step
    for k in range(len(eng.seg_cuts)):
        eng._segment(k)
        eng.exchanger.launch_chunk(k)
    eng.exchanger.wait()
    eng.optimizer_step()
    marker()                      # graph-replayed step
    eng.train_step()
    marker()
    torch.cuda.synchronize()


def parse(path):
    from summarize_prof import short
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "scan" in r["Kernel_Name"].lower() or "cumsum" in r["Kernel_Name"].lower()]
    if len(marks) < 3:
        sys.exit(f"expected 3 marker launches, found {len(marks)}")
    a, b, c = marks[-3:]
    for title, lo, hi in (("eager step", a, b), ("graph-replayed step", b, c)):
        seg = rows[lo + 1:hi]
        t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
        names = [short(r["Kernel_Name"]) for r in seg]
        torch_rows = [n for n in names if n.startswith("torch:") or "at::native" in n]
        copies = [n for n in names if "copyBuffer" in n or "fillBuffer" in n]
        print(f"## {title}: {len(seg)} launches, {(t1 - t0) / 1e6:.3f} ms first start to last end, {busy / 1e6:.3f} ms of kernel time, "
              f"{(t1 - t0 - busy) / 1e6:.3f} ms between kernels; torch launches: {len(torch_rows)}, runtime copies / fills: {len(copies)}\n")
        if title == "eager step":
            print("| # | kernel | us |\n|---:|---|---:|")
            for i, (r, n) in enumerate(zip(seg, names)):
                print(f"| {i} | {re.sub(r'[|]', '/', n)[:110]} | {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f} |")
            print()
        else:
            agg = {}
            for r, n in zip(seg, names):
                d = agg.setdefault(n, [0, 0])
                d[0] += 1
                d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            print("| kernel | launches | us total |\n|---|---:|---:|")
            for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print(f"| {n[:110]} | {k} | {t / 1e3:.1f} |")
            print()


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) >= 3 and sys.argv[1] == "parse":
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        parse(sys.argv[2])
    else:
        sys.exit(__doc__)
