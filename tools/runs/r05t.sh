cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for sp in 0 16 48; do (time CNSN_ARENA_SPREAD_GB=$sp python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-ceiling 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spread $sp', d['ms_per_step'], d['fwd_ms'], d['bwd_ms'], 'plain', d.get('ms_per_step_plain_allocator'))") 2>&1 | grep -E "spread|real"; done; done
