#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel.

    python profiles/pmc_summary.py <dir with *_counter_collection.csv> [name filter]

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
coalesced streams, i.e. HALF the bytes actually read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is
uncalibrated — both are calibrated here against kernels whose traffic is known exactly (the two-pass
apply kernel reads E*b and writes E*b).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, flt=""):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if flt and flt not in name:
                continue
            acc[name[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, ctrs in sorted(acc.items()):
        for c, v in ctrs.items():
            print(f"{name:92s} {c:12s} n={len(v):4d} mean={sum(v) / len(v):14.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
