python -m pytest tests/test_gpu_arena.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2 3; do
for t in 8 4; do
  CNSN_ARENA_TRIES=$t python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); a=d.get('arena',{}); print('tries $t', 'arena', d['ms_per_step'], 'plain', d.get('ms_per_step_plain_allocator'), 'fwd', d['fwd_ms'], 'bwd', d['bwd_ms'], {k:a.get(k) for k in ('blocks','probed','mapped_bytes','block_gbps') if k in a})
"
done; done
