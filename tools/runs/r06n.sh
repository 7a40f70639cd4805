# round 6: whole GPU suite with the fused bottleneck tail; per-site times / kernel stats / PMC of the bn-block launches
mkdir -p gpurun_out; R=$(pwd)
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r06n_pytest_gpu.txt; tail -3 gpurun_out/r06n_pytest_gpu.txt
python tools/bn_block_sites.py 2>/dev/null | tee gpurun_out/r06n_bn_block_sites.txt
python tools/bn_block_sites.py down 2>/dev/null | tee -a gpurun_out/r06n_bn_block_sites.txt
cd /tmp; export TMPDIR=/tmp
rm -f $R/gpurun_out/r06n_bn_block_pmc.txt
for site in 0 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b$site -- python $R/tools/bn_block_sites.py site$site fused > /dev/null 2>&1
  find /tmp/prof_b$site -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/r06n_bn_block_site${site}_kernel_stats.csv \;
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_b${site}_$ctr
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_b${site}_$ctr -- python $R/tools/bn_block_sites.py site$site fused > /dev/null 2>&1
    echo "site$site $ctr" >> $R/gpurun_out/r06n_bn_block_pmc.txt
    python $R/profiles/pmc_summary.py /tmp/pmc_b${site}_$ctr nhwc_bnhead >> $R/gpurun_out/r06n_bn_block_pmc.txt
  done
done
cd $R; python - <<'PY'
import csv,re
for s in range(4):
    for r in csv.DictReader(open(f"gpurun_out/r06n_bn_block_site{s}_kernel_stats.csv")):
        if "cnsn::nhwc" in r["Name"]:
            print("site",s,r["Name"][11:60],r["Calls"],"avg us",round(float(r["AverageNs"])/1e3,1),"min",round(float(r["MinNs"])/1e3,1))
unit={0:256*256*56*56*2/1e6,1:256*512*28*28*2/1e6,2:256*1024*14*14*2/1e6,3:256*2048*7*7*2/1e6}
cur=None; d={}
for line in open('gpurun_out/r06n_bn_block_pmc.txt'):
    m=re.match(r'site(\d) (\w+)',line)
    if m: cur=(int(m.group(1)),m.group(2)); continue
    m=re.search(r'nhwc_bnhead_(fwd|bwd)_kernel.*mean=\s*([\d.]+)',line)
    if m: d[(cur[0],cur[1],m.group(1))]=float(m.group(2))
for s in range(4):
    u=unit[s]
    try:
        fr=2*d[(s,'FETCH_SIZE','fwd')]*1e3/1e6; fw=d[(s,'WRITE_SIZE','fwd')]*1e3/1e6
        br=2*d[(s,'FETCH_SIZE','bwd')]*1e3/1e6; bw=d[(s,'WRITE_SIZE','bwd')]*1e3/1e6
        print(s, f"{u:.0f} MB | fwd read {fr:.0f} / written {fw:.0f} = {fr/(4*u):.2f} / {fw/(1*u):.2f} of 4 + 1 | bwd {br:.0f} / {bw:.0f} = {br/(6*u):.2f} / {bw/(2*u):.2f} of 6 + 2")
    except KeyError as e: print("missing", e)
PY
CNSN_NO_GLUE=1 python bench.py --workload resnet50 --steps 30 --warmup 8 2>/dev/null | tail -1 | cut -c100-330
