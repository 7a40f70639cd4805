cd /tmp && export TMPDIR=/tmp
for blk in 1024; do
echo "== CNSN_MID_BLOCK=$blk"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_sites_$blk -- python $GRAFT_REPO_ROOT/tools/nhwc_sites.py bf16 > /tmp/sites.log 2>&1
grep "^|" /tmp/sites.log
g=$(find /tmp/rp_sites_$blk -name "*kernel_trace.csv" | head -1); python3 - $g <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.OrderedDict()
for r in rows:
    k=(r['Kernel_Name'][:60], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Workgroup_Size_X') or r.get('Workgroup_Size'))
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=d
for k,(n,t) in agg.items():
    if 'mid_' in k[0]:
        print(f"{k[0]:60s} grid {k[1]:>9} wg {k[2]:>5} n {n:4d} avg us {t/n:8.1f}")
PY
done
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_nhwc.py -q -m gpu -x 2>&1 | tail -2
python -m pytest tests -q -m gpu -x -k "two_pass or twopass or stream or strategy or parity" 2>&1 | tail -2
