# round 6: single-launch channels-last kernels with the sharded barrier and write-through side arrays
mkdir -p gpurun_out; R=$(pwd)
python -m pytest tests/test_gpu_nhwc.py tests/test_gpu_saved_contract.py tests/test_gpu_context.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r06c_tests.txt
tail -5 gpurun_out/r06c_tests.txt
cd /tmp; export TMPDIR=/tmp
for f in 1 0; do
  CNSN_NHWC_FUSED=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f$f -- python $R/tools/nhwc_sites.py bf16 cl > /tmp/sites_f$f.txt 2>&1
  find /tmp/prof_f$f -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/r06c_sites_fused${f}_kernel_stats.csv \;
  grep -v "^W2026\|^E2026" /tmp/sites_f$f.txt | tail -6
done
cd $R; python - <<'PY'
import csv
for f in (1,0):
    print("fused",f)
    for r in csv.DictReader(open(f"gpurun_out/r06c_sites_fused{f}_kernel_stats.csv")):
        if "cnsn" in r["Name"] and "arena" not in r["Name"]:
            print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.1f} min {float(r["MinNs"])/1e3:8.1f} max {float(r["MaxNs"])/1e3:8.1f}')
PY
