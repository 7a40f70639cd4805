cd /tmp && export TMPDIR=/tmp
for lib in main; do
echo "=== $lib"
if [ $lib = main ]; then unset CNSN_LIB_PATH; else export CNSN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libcnsn_$lib.so; fi
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_sites_$lib -- python $GRAFT_REPO_ROOT/tools/nhwc_sites.py bf16 > /tmp/sites.log 2>&1
grep "^| (" /tmp/sites.log | cut -d'|' -f2,4
g=$(find /tmp/rp_sites_$lib -name "*kernel_trace.csv" | head -1); python3 - $g <<'PY'
import csv,sys,collections
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r['Start_Timestamp']))
by=collections.OrderedDict()
for r in rows:
    n=r['Kernel_Name']
    if 'nhwc' not in n or 'finish' in n or 'saved_rows' in n: continue
    by.setdefault(n[11:40],[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for n,d in by.items():
    k=len(d)//4
    out=[]
    for i in range(4):
        ch=d[i*k:(i+1)*k][4:]
        out.append(f"{sum(ch)/len(ch):7.1f}")
    print(f"  {n:30s} 56x56/28x28/14x14/7x7 us:", " ".join(out))
PY
done
cd $GRAFT_REPO_ROOT; python -m pytest tests/test_gpu_nhwc.py -q -m gpu -x 2>&1 | tail -2
