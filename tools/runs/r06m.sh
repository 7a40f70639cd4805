# round 6: per-kernel profile of the ResNet-50 step with the fused bottleneck tail; JSD workload; site-level kernel times
mkdir -p gpurun_out; R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_model -- python $R/bench.py --workload resnet50 --steps 20 --warmup 6 > /tmp/model.txt 2>&1
find /tmp/prof_model -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/r06m_resnet50_bn_block_step_kernel_stats.csv \;
tail -1 /tmp/model.txt | cut -c1-300
cd $R; python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r06m_resnet50_bn_block_step_kernel_stats.csv")))
steps=26
skip=("naive_conv","Im2d2Col","Col2Im")
tot=sum(float(r["TotalDurationNs"]) for r in rows if not r["Name"].startswith(skip))
print("kernel ms per step (find-phase kernels left out, /26):", round(tot/1e6/steps,2))
for r in rows[:40]:
    if r["Name"].startswith(skip): continue
    print(f'{r["Name"][:95]:95s} calls/step {int(r["Calls"])/steps:6.1f} ms/step {float(r["TotalDurationNs"])/1e6/steps:6.3f} avg us {float(r["AverageNs"])/1e3:7.1f}')
for key in ("cnsn::","MIOpenBatchNorm"):
    print(key, "ms/step:", round(sum(float(r["TotalDurationNs"]) for r in rows if key in r["Name"] and "arena" not in r["Name"])/1e6/steps,3))
PY
python bench.py --workload resnet50_jsd --steps 20 --warmup 6 2>/dev/null | tail -1 | cut -c1-300 | tee gpurun_out/r06m_jsd.txt
CNSN_BN_BLOCK=0 python bench.py --workload resnet50_jsd --steps 20 --warmup 6 2>/dev/null | tail -1 | cut -c1-300 | tee -a gpurun_out/r06m_jsd.txt
