cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_sites -- python $GRAFT_REPO_ROOT/tools/nhwc_sites.py bf16 > /tmp/sites.log 2>&1
grep "^|" /tmp/sites.log
g=$(find /tmp/rp_sites -name "*kernel_trace.csv" | head -1); python3 - $g <<'PY'
import csv,sys,collections
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r['Start_Timestamp']))
by=collections.OrderedDict()
for r in rows:
    n=r['Kernel_Name']
    if 'nhwc' not in n: continue
    by.setdefault(n[:70],[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
sites=["56x56 C256 (411 MB)","28x28 C512 (206 MB)","14x14 C1024 (103 MB)","7x7 C2048 (51 MB)"]
for n,d in by.items():
    k=len(d)//4
    print(n, len(d))
    for i,s in enumerate(sites):
        ch=d[i*k:(i+1)*k][4*(k//14):]
        print(f"    {s:24s} avg us {sum(ch)/len(ch):8.1f}  (n {len(ch)})")
PY
