#!/usr/bin/env python3
"""Trim a rocprofv3 `*_kernel_stats.csv` (kernel names can be kilobytes long) to a readable table.

    python profiles/summarize.py gpurun_out/prof/runc/NNN_kernel_stats.csv profiles/r01_xxx.csv
"""
import csv
import sys


def main(src, dst, width=110):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            name = r["Name"]
            if len(name) > width:
                name = name[:width] + "..."
            w.writerow([name, r["Calls"], r["TotalDurationNs"], f'{float(r["AverageNs"]):.0f}', r["Percentage"],
                        r["MinNs"], r["MaxNs"]])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
