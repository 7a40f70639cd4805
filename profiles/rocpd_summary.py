#!/usr/bin/env python3
"""Summarise rocprofv3's rocpd SQLite output (ROCm 7.2 writes `<name>_results.db` instead of CSV files).

    python profiles/rocpd_summary.py stats  <results.db> [out.csv]      # per-kernel calls / total / average / min / max (ns)
    python profiles/rocpd_summary.py pmc    <results.db> [name filter]  # per-kernel average of every collected counter

Kernel names are the demangled names truncated to 110 characters (they can be kilobytes long)."""
import csv
import sqlite3
import sys


def stats(db, out=None, width=110):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    w = csv.writer(open(out, "w", newline="") if out else sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows:
        name = name if len(name) <= width else name[:width] + "..."
        w.writerow([name, calls, int(tot), f"{avg:.0f}", f"{100.0 * tot / total:.2f}", int(mn), int(mx)])


def pmc(db, flt=""):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events "
                       "group by name, counter_name order by name").fetchall()
    for name, ctr, n, mean in rows:
        if flt and flt not in name:
            continue
        print(f"{name[:92]:92s} {ctr:12s} n={n:4d} mean={mean:14.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
