#!/usr/bin/env python3
"""gpurun_out/parity_margins.jsonl (written by tests/test_gpu_full_size.py::check_case) -> markdown: per dtype and output
the worst observed error relative to its scale, the bound it was held to, and how much of the bound was used."""
import collections
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
print(f"{len(rows)} (case, output) comparisons\n")
groups = collections.defaultdict(list)
for r in rows:
    groups[(r["dtype"], r["out"])].append(r)
print("| dtype | output | cases | worst err / scale | median err / scale | worst err / bound | where the worst is |")
print("|---|---|---|---|---|---|---|")
for (dt, out), rs in sorted(groups.items()):
    rel = sorted(r["err"] / r["scale"] for r in rs)
    worst = max(rs, key=lambda r: r["err"] / r["bound"] if r["bound"] else 0)
    print(f"| {dt} | {out} | {len(rs)} | {rel[-1]:.2e} | {rel[len(rel) // 2]:.2e} | {worst['err'] / worst['bound']:.2f} | "
          f"{tuple(worst['shape'])} {worst['kind']}/{worst['crop']} |")
if any(r["dtype"] == "fp32" for r in rows):
    fp = [r for r in rows if r["dtype"] == "fp32"]
    by_noise = [r for r in fp if 2 * r["oracle32_noise"] > r["rel_tol"] * r["scale"]]
    print(f"\nfp32: {len(by_noise)} of {len(fp)} comparisons were held to 2 x |oracle32 - truth64| rather than to rel_tol x scale; "
          f"in those the device's error was at most {max((r['err'] / r['oracle32_noise'] for r in by_noise), default=0):.2f} x the oracle's own fp32 error "
          f"(median {sorted(r['err'] / r['oracle32_noise'] for r in by_noise)[len(by_noise) // 2] if by_noise else 0:.2f} x).")
