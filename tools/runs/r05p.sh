cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_arena.py tests/test_gpu_placement.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
for t in 1 4 8; do CNSN_ARENA_TRIES=$t python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-ceiling --prospect 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tries $t', {k:v for k,v in d.items() if k.startswith(('ms_per','frac_of_hbm_peak_bytes'))}, d['fwd_ms'], d['bwd_ms'], d['arena'].get('probed'))"; done; done
for i in 1 2; do
for t in 1 4; do CNSN_ARENA_TRIES=$t python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-ceiling --prospect 0 --dtype bf16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 tries $t', {k:v for k,v in d.items() if k.startswith(('ms_per',))}, d['fwd_ms'], d['bwd_ms'])"; done; done
