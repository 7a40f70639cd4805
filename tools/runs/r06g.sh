# round 6: 16-byte write-through side arrays + XCD-aware phase B: parity, per-site kernel times and PMC
mkdir -p gpurun_out; R=$(pwd)
python -m pytest tests/test_gpu_nhwc.py tests/test_gpu_saved_contract.py tests/test_gpu_context.py tests/test_gpu_step_guard.py -q -m gpu -x 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
rm -f $R/gpurun_out/r06g_sites_pmc.txt
for site in 0 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s$site -- python $R/tools/nhwc_sites.py bf16 cl site$site > /tmp/s$site.txt 2>&1
  find /tmp/prof_s$site -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/r06g_site${site}_kernel_stats.csv \;
  grep "^| (" /tmp/s$site.txt
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_s${site}_$ctr
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_s${site}_$ctr -- python $R/tools/nhwc_sites.py bf16 cl site$site > /dev/null 2>&1
    echo "site$site $ctr" >> $R/gpurun_out/r06g_sites_pmc.txt
    python $R/profiles/pmc_summary.py /tmp/pmc_s${site}_$ctr nhwc_fused >> $R/gpurun_out/r06g_sites_pmc.txt
  done
done
cd $R; python - <<'PY'
import csv
for s in range(4):
    for r in csv.DictReader(open(f"gpurun_out/r06g_site{s}_kernel_stats.csv")):
        if "cnsn::nhwc" in r["Name"]:
            print("site",s,r["Name"][11:60],r["Calls"],"avg us",round(float(r["AverageNs"])/1e3,1),"min",round(float(r["MinNs"])/1e3,1))
PY
grep -v "^site" gpurun_out/r06g_sites_pmc.txt | awk '{print $NF, $(NF-3)}' | paste - - - - | head -8
python bench.py --workload resnet50 --steps 30 --warmup 8 2>/dev/null | tail -1 | cut -c1-330 | tee gpurun_out/r06g_model.txt
