cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05r; mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
(time python bench.py --steps 20 --warmup 5) > $O/bench_driver_style.json 2> $O/bench_driver_style.err; tail -3 $O/bench_driver_style.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05r/bench_driver_style.json').read().strip().splitlines()[-1])
print({k:v for k,v in d.items() if k.startswith(('ms_per','frac','value')) and k!='value_note'}, d['fwd_ms'], d['bwd_ms'])
print(d['arena'].get('tries'), d['arena'].get('probed'))
e=d['extra']
for k in ('f32_crop_both','bf16_crop_neither','bf16_crop_both','resnet50_bs256_bf16','resnet50_bs256_bf16_nchw','seg_bs16_512'):
    print(k, json.dumps(e.get(k))[:300])
PY
for w in "resnet50_jsd" "resnet50_jsd --nchw" "wrn40" "wrn40 --channels-last"; do timeout 400 python bench.py --workload $w --steps 20 --warmup 6 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'])" | tee -a $O/model_lines.txt; done
