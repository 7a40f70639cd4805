"""rocprofv3 --kernel-trace CSV of a bench.py run -> the kernels of three mid-run steps with the idle time before each, and
the busy / idle totals per step (profiles/r03_sn_cluster.md section 9).  usage: gaps.py <rocprofv3 output dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the timed loop: consecutive fwd/bwd pipe kernels
idx = [i for i, r in enumerate(rows) if "resident_bwd_pipe" in r["Kernel_Name"]]
lo, hi = idx[len(idx) // 2], idx[-2]
prev_end = None
print("window of kernels between two backward launches (mid-run):")
for r in rows[lo:idx[len(idx) // 2 + 2] + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"  gap {gap:8.1f} us | {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:90]}")
    prev_end = e
# totals over the second half
span = int(rows[hi]["End_Timestamp"]) - int(rows[lo]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[lo + 1:hi + 1])
steps = len([i for i in idx if lo < i <= hi])
print(f"steps {steps}: span {span / steps / 1e3:.1f} us per step, kernels busy {busy / steps / 1e3:.1f} us per step, idle {(span - busy) / steps / 1e3:.1f} us per step")
