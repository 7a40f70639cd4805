cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
(time python bench.py --steps 20 --warmup 5) > $O/r05l_driver_style.json 2> $O/r05l_driver_style.err
tail -4 $O/r05l_driver_style.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05l_driver_style.json').read().strip().splitlines()[-1])
print({k:v for k,v in d.items() if k.startswith(('ms_per','frac','value'))})
print(d['roofline']['frac'], d['roofline']['kernel_us'], d['roofline']['forward']['kernel_us'], d['roofline'].get('traffic_over_bytes'))
print(d['arena'])
print(d['cpu_baseline'])
e=d['extra']
for k,v in e.items():
    print(k, json.dumps(v)[:300])
PY
python -m pytest tests -q -m gpu -x 2>&1 | tail -3
