cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pipe.py tests/test_gpu_parity.py tests/test_gpu_launch_args.py tests/test_gpu_context.py tests/test_gpu_robustness.py -x -q -m gpu 2>&1 | tail -3
export CNSN_LIB_PATH=$GRAFT_REPO_ROOT/tools/ab/libcnsn_prof.so
python tools/prof_pipe.py f32 256 256 56 56 2>&1 | tail -6; python tools/prof_pipe.py bf16 256 256 56 56 2>&1 | tail -6; python tools/prof_pipe.py bf16 256 256 56 56 both 2>&1 | tail -6; python tools/prof_pipe.py f32 256 256 56 56 both 2>&1 | tail -6
unset CNSN_LIB_PATH
run() { tag=$1; lib=$2; shift 2
  env CNSN_LIB_PATH=$lib python bench.py --steps 40 --warmup 10 --no-extra --no-cpu-baseline --no-ceiling --prospect 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', '$(basename $lib)', d['ms_per_step'], 'fwd', d['fwd_ms'], 'bwd', d['bwd_ms'], 'plain', d.get('ms_per_step_plain_allocator'))
"
}
OLD=$GRAFT_REPO_ROOT/tools/ab/libcnsn_before.so; NEW=$GRAFT_REPO_ROOT/crossnorm-selfnorm_amd/libcnsn_hip.so
for i in 1 2; do
run f32 $OLD; run f32 $NEW
run bf16 $OLD --dtype bf16; run bf16 $NEW --dtype bf16
run f32both $OLD --crop both; run f32both $NEW --crop both
run bf16both $OLD --crop both --dtype bf16; run bf16both $NEW --crop both --dtype bf16
done
