cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --prospect 12 > $O/r05f_bench1.json 2>$O/r05f_bench1.err
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --prospect 12 > $O/r05f_bench2.json 2>$O/r05f_bench2.err
for f in $O/r05f_bench*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:v for k,v in d.items() if k.startswith("ms_per") or k.startswith("frac")}, d["fwd_ms"], d["bwd_ms"], d["arena"].get("prospect"))
PY
done
tail -3 $O/r05f_bench1.err
