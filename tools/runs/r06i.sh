# round 6: (1) column-slice passes over a channels-last tensor (tools/slice_probe), (2) the default bench line on this commit
mkdir -p gpurun_out
timeout 120 tools/slice_probe | tee gpurun_out/r06i_slice_probe.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06i_bench.json 2> gpurun_out/r06i_bench.err; tail -c 600 gpurun_out/r06i_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06i_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","fwd_ms","bwd_ms") if k in d}, d["roofline"]["frac"], d.get("ms_per_step_plain_allocator"))
for k,v in d["roofline_bf16"].items():
    if isinstance(v,dict): print(k, v["forward"], v["backward"], v.get("kernels"))
print(d["extra"]["resnet50_bs256_bf16"], d["extra"]["seg_bs16_512"]["bf16"])
PY
