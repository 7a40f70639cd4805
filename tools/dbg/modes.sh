R=$PWD
run() { # label, env..., args
  label=$1; shift
  env "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['ms_per_step'], d['fwd_ms'], d['bwd_ms'], d['config']['paths'])"
}
for dt in f32 bf16; do for crop in neither both; do for st in auto resident two_pass; do
  run "$dt $crop $st" CNSN_X=1 python $R/bench.py --steps 100 --warmup 20 --no-extra --no-cpu-baseline --no-ceiling --dtype $dt --crop $crop --strategy $st
done; 
  run "$dt $crop resident PIPE=2" CNSN_PIPE=2 python $R/bench.py --steps 100 --warmup 20 --no-extra --no-cpu-baseline --no-ceiling --dtype $dt --crop $crop --strategy resident
done; done
