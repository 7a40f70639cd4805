cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05y; mkdir -p $O
R=$GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
stats() { name=$1; shift; rm -rf /tmp/rp_$name
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$name -- "$@" > /tmp/rp_$name.log 2>&1
  f=$(for g in $(find /tmp/rp_$name -name "*kernel_stats.csv"); do echo "$(grep -c cnsn:: $g) $g"; done | sort -rn | head -1 | cut -d" " -f2)
  [ -n "$f" ] && python $R/profiles/summarize.py "$f" $O/${name}_kernel_stats.csv; grep cnsn $O/${name}_kernel_stats.csv | cut -c1-160; }
pmc() { name=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pmc_${name}_$ctr
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${name}_$ctr -- "$@" > /dev/null 2>&1
    python $R/profiles/pmc_summary.py /tmp/pmc_${name}_$ctr cnsn >> $O/${name}_pmc.txt; done; cat $O/${name}_pmc.txt | cut -c1-150; }
stats bench_f32_neither python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-alt
stats bench_f32_neither_plain python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-alt --no-arena
pmc bench_f32_neither python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-alt
cd $R
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-ceiling 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v for k,v in d.items() if k.startswith(('ms_per','frac_of_hbm_peak_bytes'))}, d['fwd_ms'], d['bwd_ms'])" | tee -a $O/driver_style_x3.txt; done
