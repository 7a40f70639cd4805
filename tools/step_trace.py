#!/usr/bin/env python3
"""tools/step_trace.py <rocprofv3 output dir> [n_steps]: the GPU timeline of the headline step as rocprofv3 saw it —
every kernel dispatch and memory copy of a few steps from the middle of the run, in start order, with its duration and the
idle gap in front of it; then launches per step and the per-step sums.  Input: `rocprofv3 --kernel-trace --memory-copy-trace
--output-format csv -d DIR -- python bench.py --steps 50 --no-extra --no-cpu-baseline --no-ceiling`."""
import csv
import glob
import os
import sys


def rows(d):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", r.get("Name", "?"))))
    return sorted(out)


def short(name):
    name = name.replace("cnsn::", "")
    return name if len(name) < 70 else name[:67] + "..."


def main():
    d = sys.argv[1]
    n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    ev = rows(d)
    bwd = [i for i, e in enumerate(ev) if "bwd" in e[2] and "cnsn::" in e[2]]
    if len(bwd) < n_steps + 12:
        print("too few backward launches in the trace:", len(bwd))
        return
    first, last = bwd[len(bwd) // 2], bwd[len(bwd) // 2 + n_steps]
    print(f"{'start us':>10} {'dur us':>8} {'gap us':>8}  what")
    t0 = ev[first][0]
    for i in range(first, last + 1):
        s, e, name = ev[i]
        gap = (s - ev[i - 1][1]) / 1e3
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {gap:8.1f}  {short(name)}")
    # per-step accounting over the middle half of the run
    a, b = bwd[len(bwd) // 4], bwd[3 * len(bwd) // 4]
    steps = sum(1 for i in bwd if a <= i < b)
    span = (ev[b][0] - ev[a][0]) / 1e3
    busy = sum((e - s) for s, e, _ in ev[a:b]) / 1e3
    small = sum((e - s) for s, e, n in ev[a:b] if "cnsn::" not in n) / 1e3
    print(f"\n{steps} steps: {(b - a) / steps:.2f} GPU operations per step, {span / steps:.1f} us per step, "
          f"{busy / steps:.1f} us busy ({small / steps:.1f} us of it in operations that are not the op's kernels), "
          f"{(span - busy) / steps:.1f} us idle")


if __name__ == "__main__":
    main()
