#!/bin/bash
# tools/collect_profiles.sh <outdir>: the rocprofv3 passes behind profiles/r03_* (run on the GPU box through gpurun).
# Every rocprofv3 call is bounded by `timeout`; counters are collected in their own passes (--pmc without --stats).
OUT=$(realpath -m $1); mkdir -p $OUT   # (absolute: the rocprofv3 passes run from /tmp)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
stats() {  # name, command...
  name=$1; shift
  rm -rf /tmp/rp_$name
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$name -- "$@" > /tmp/rp_$name.log 2>&1
  # (bench.py spawns tools/pattern_bench: one stats file per process — take the one with the library's kernels in it)
  f=$(for g in $(find /tmp/rp_$name -name "*kernel_stats.csv"); do echo "$(grep -c cnsn:: $g) $g"; done | sort -rn | head -1 | cut -d" " -f2)
  [ -n "$f" ] && python $R/profiles/summarize.py "$f" $OUT/${name}_kernel_stats.csv
  echo "stats $name: $(grep -c cnsn $OUT/${name}_kernel_stats.csv) cnsn rows"
}
pmc() {  # name, command...
  name=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${name}_$ctr
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${name}_$ctr -- "$@" > /dev/null 2>&1
    python $R/profiles/pmc_summary.py /tmp/pmc_${name}_$ctr cnsn >> $OUT/${name}_pmc.txt
  done
  echo "pmc $name: $(wc -l < $OUT/${name}_pmc.txt) rows"
}
stats bench_f32_neither python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-alt
for c in boxed_f32 boxed_bf16 neither_bf16 block_bf16 block_f32 sn_bf16; do stats $c python $R/tools/run_cases.py $c 8; done
for c in boxed_f32 boxed_bf16 neither_bf16 block_bf16; do pmc $c python $R/tools/run_cases.py $c 4; done
pmc bench_f32_neither python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-alt
stats resnet50_step python $R/bench.py --workload resnet50 --steps 12 --warmup 4
cd $R
timeout 200 tools/pattern_bench 256 256 > $OUT/pattern_bench.txt 2>&1
timeout 600 python bench.py --sweep > $OUT/shape_sweep.md 2> /dev/null
timeout 600 python bench.py --steps 50 --warmup 10 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err
for w in resnet50 resnet50_jsd wrn40 seg; do timeout 300 python bench.py --workload $w --steps 30 --warmup 8 2>/dev/null | tail -1 >> $OUT/model_workloads.jsonl; done
timeout 300 python bench.py --workload wrn40 --steps 60 --warmup 15 --no-graph 2>/dev/null | tail -1 >> $OUT/model_workloads.jsonl
CNSN_FUSE_TAIL=0 timeout 300 python bench.py --workload wrn40 --steps 60 --warmup 15 --no-graph 2>/dev/null | tail -1 > $OUT/wrn40_eager_no_tail.json
ls -la $OUT
