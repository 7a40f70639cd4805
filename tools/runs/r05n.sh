cd $GRAFT_REPO_ROOT
run() { tag=$1; lib=$2; shift 2
  env CNSN_LIB_PATH=$lib python bench.py --steps 40 --warmup 10 --no-extra --no-cpu-baseline --no-ceiling --prospect 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', '$(basename $lib)', d['ms_per_step'], 'fwd', d['fwd_ms'], 'bwd', d['bwd_ms'])
"
}
BASE=$GRAFT_REPO_ROOT/crossnorm-selfnorm_amd/libcnsn_hip.so; P1=$GRAFT_REPO_ROOT/tools/ab/libcnsn_prio1.so; P2=$GRAFT_REPO_ROOT/tools/ab/libcnsn_prio2.so
for i in 1 2; do
for L in $BASE $P1 $P2; do run f32 $L; done
for L in $BASE $P1 $P2; do run bf16 $L --dtype bf16; done
for L in $BASE $P1 $P2; do run bf16both $L --dtype bf16 --crop both; done
done
