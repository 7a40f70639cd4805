# round 6: per-kernel profile of the WideResNet-40-2 step (config 2) and a second headline line (another box)
mkdir -p gpurun_out; R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wrn -- python $R/bench.py --workload wrn40 --no-graph --steps 30 --warmup 10 > /tmp/wrn.txt 2>&1
find /tmp/prof_wrn -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/r06j_wrn40_step_kernel_stats.csv \;
tail -1 /tmp/wrn.txt | cut -c1-260
cd $R; python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r06j_wrn40_step_kernel_stats.csv")))
steps=40
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step:", round(tot/1e6/steps,3))
for r in rows[:28]:
    print(f'{r["Name"][:110]:110s} calls/step {int(r["Calls"])/steps:6.1f} ms/step {float(r["TotalDurationNs"])/1e6/steps:6.3f} avg us {float(r["AverageNs"])/1e3:7.1f}')
cn=sum(float(r["TotalDurationNs"]) for r in rows if "cnsn" in r["Name"] and "arena" not in r["Name"])
print("cnsn kernels ms/step:", round(cn/1e6/steps,3))
PY
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','fwd_ms','bwd_ms','ms_per_step_plain_allocator')}, d['roofline']['frac'], d['roofline'].get('ceiling',{}).get('resident_order_triad_GBps'))"
