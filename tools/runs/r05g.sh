cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
run() { # tag env... -- args
  tag=$1; shift
  env "$1" python bench.py --steps 40 --warmup 10 --no-extra --no-cpu-baseline --no-ceiling --prospect 0 "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', '$1', d['ms_per_step'], 'fwd', d['fwd_ms'], 'bwd', d['bwd_ms'], 'plain', d.get('ms_per_step_plain_allocator'), 'timeouts', d['resident_timeouts'])
"
}
for i in 1 2; do
run f32 CNSN_XCD=0
run f32 CNSN_XCD=1
done
for i in 1 2; do
run bf16 CNSN_XCD=0 --dtype bf16
run bf16 CNSN_XCD=1 --dtype bf16
done
run f32both CNSN_XCD=0 --crop both
run f32both CNSN_XCD=1 --crop both
run bf16both CNSN_XCD=0 --crop both --dtype bf16
run bf16both CNSN_XCD=1 --crop both --dtype bf16
run bf16sn CNSN_XCD=0 --kind sn --dtype bf16
run bf16sn CNSN_XCD=1 --kind sn --dtype bf16
run f32sn CNSN_XCD=0 --kind sn
run f32sn CNSN_XCD=1 --kind sn
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipe.py tests/test_gpu_sn_cluster.py tests/test_gpu_cn_partial.py -x -q -m gpu 2>&1 | tail -3
