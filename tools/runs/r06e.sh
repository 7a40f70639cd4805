# round 6: per-kernel profile of the channels-last ResNet-50 step (single-launch sites, kept sum) + per-site kernel sums
mkdir -p gpurun_out; R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_model -- python $R/bench.py --workload resnet50 --steps 10 --warmup 4 > /tmp/model.txt 2>&1
find /tmp/prof_model -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/r06e_resnet50_channels_last_step_kernel_stats.csv \;
tail -1 /tmp/model.txt | cut -c1-300
for site in 0 1 2 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s$site -- python $R/tools/nhwc_sites.py bf16 cl site$site > /tmp/s$site.txt 2>&1
  find /tmp/prof_s$site -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/r06e_site${site}_kernel_stats.csv \;
  grep "^| (" /tmp/s$site.txt
done
cd $R; python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r06e_resnet50_channels_last_step_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (14 steps incl. warm-up):", tot/1e6/14)
for r in rows[:25]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>5s} total ms/step {float(r["TotalDurationNs"])/1e6/14:7.3f} avg us {float(r["AverageNs"])/1e3:8.1f}')
cn=sum(float(r["TotalDurationNs"]) for r in rows if "cnsn" in r["Name"] and "arena" not in r["Name"])
print("cnsn kernels ms/step:", cn/1e6/14)
for s in range(4):
    rr=list(csv.DictReader(open(f"gpurun_out/r06e_site{s}_kernel_stats.csv")))
    for r in rr:
        if "cnsn::nhwc" in r["Name"] or "cnsn::mid" in r["Name"]:
            print("site",s,r["Name"][:70],r["Calls"],"avg us",round(float(r["AverageNs"])/1e3,1),"min",round(float(r["MinNs"])/1e3,1))
PY
