#!/usr/bin/env python3
"""Idle time between kernels in a rocprofv3 --kernel-trace csv: busy vs span per window, and the largest gaps with the kernels on
either side.  Usage: python3 tools/trace_gaps.py <kernel_trace.csv> [n_largest] [skip_first_fraction]"""
import csv
import sys


def main():
    path = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
    rows.sort()
    rows = rows[int(len(rows) * skip):]          # the steady part of the run
    span = rows[-1][1] - rows[0][0]
    busy, gaps, end = 0, [], rows[0][0]
    for a, b, name in rows:
        if a > end:
            gaps.append((a - end, name))
        busy += max(0, b - max(a, end))
        end = max(end, b)
    print(f"kernels {len(rows)}, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), idle {(span - busy) / 1e6:.2f} ms in {len(gaps)} gaps "
          f"(mean {sum(g for g, _ in gaps) / max(len(gaps), 1) / 1e3:.2f} us)")
    hist = {}
    for g, _ in gaps:
        k = "<2us" if g < 2000 else "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"
        hist[k] = hist.get(k, [0, 0])
        hist[k][0] += 1
        hist[k][1] += g
    for k in ("<2us", "<5us", "<20us", "<100us", ">=100us"):
        if k in hist:
            print(f"  gaps {k:>7}: {hist[k][0]:6d}  total {hist[k][1] / 1e6:.3f} ms")
    prev = None
    named = []
    end = rows[0][0]
    for a, b, name in rows:
        if a > end and prev is not None:
            named.append((a - end, prev, name))
        if b >= end:
            prev = name
        end = max(end, b)
    print("\nlargest gaps (us): after kernel -> before kernel")
    for g, p, q in sorted(named, reverse=True)[:n]:
        print(f"  {g / 1e3:8.1f}  {p}  ->  {q}")


if __name__ == "__main__":
    main()


def neighbours(path, needle, skip=0.5, top=20):
    """Histogram of (kernel before, kernel after) around every kernel whose name contains `needle`."""
    import collections
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"][:60]))
    rows.sort()
    rows = rows[int(len(rows) * skip):]
    h = collections.Counter()
    for i, (_, n) in enumerate(rows):
        if needle in n and 0 < i < len(rows) - 1:
            h[(rows[i - 1][1], rows[i + 1][1])] += 1
    print(f"\nneighbours of '{needle}' ({sum(h.values())} occurrences):")
    for (a, b), c in h.most_common(top):
        print(f"  {c:5d}  {a}  ->  [{needle}]  ->  {b}")


if __name__ == "__main__" and len(sys.argv) > 4:
    neighbours(sys.argv[1], sys.argv[4], float(sys.argv[3]))
