# round 6: what a slow-write box looks like from outside (clocks), and whether more candidates help the headline there
mkdir -p gpurun_out
(rocm-smi --showclocks --showperflevel --showpower --showmemuse 2>&1 | grep -v "^=\|^$" | head -30) > gpurun_out/r06p_smi.txt; cat gpurun_out/r06p_smi.txt | head -20
for t in 8 16; do
  CNSN_ARENA_TRIES=$t python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --prospect 12 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); a=d.get('arena',{}); print('tries $t', 'arena', d['ms_per_step'], 'plain', d.get('ms_per_step_plain_allocator'), 'prospected', d.get('ms_per_step_prospected'), 'fwd', d['fwd_ms'], 'bwd', d['bwd_ms'], {k:a.get(k) for k in ('blocks','probed','block_gbps') if k in a}, a.get('prospect'))
" | tee -a gpurun_out/r06p_lines.txt
done
