#!/bin/bash
# round 5, first call: does the arena work, where are its blocks fast, A/B of the headline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_arena.py tests/test_gpu_launch_args.py -x -q -m gpu > $O/r05a_tests.log 2>&1; echo "tests rc=$?" >> $O/r05a_tests.log
tail -5 $O/r05a_tests.log
python tools/arena_probe.py 4 > $O/r05a_probe_f32.md 2>$O/r05a_probe_f32.err
PROBE_DTYPE=bf16 python tools/arena_probe.py 3 2,8,56,196 > $O/r05a_probe_bf16.md 2>>$O/r05a_probe_f32.err
for i in 1 2; do
CNSN_ARENA=0 python bench.py --steps 20 --warmup 5 --no-placement --no-extra --no-cpu-baseline --no-ceiling > $O/r05a_bench_plain_$i.json 2>$O/r05a_bench_plain_$i.err
python bench.py --steps 20 --warmup 5 --no-placement --no-extra --no-cpu-baseline --no-ceiling > $O/r05a_bench_arena_$i.json 2>$O/r05a_bench_arena_$i.err
done
CNSN_ARENA=0 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-ceiling > $O/r05a_bench_placed.json 2>$O/r05a_bench_placed.err
for f in $O/r05a_bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], d["fwd_ms"], d["bwd_ms"], d.get("frac_of_hbm_peak_bytes_needed"))
except Exception as e: print("ERR", e)
PY
done
rocm-smi --showmemuse --showmeminfo vram 2>/dev/null | head -20 > $O/r05a_smi.txt
rocm-smi --showcomputepartition --showmemorypartition >> $O/r05a_smi.txt 2>&1
