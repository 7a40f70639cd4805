cd $GRAFT_REPO_ROOT
run() { tag=$1; lib=$2; shift 2
  env CNSN_LIB_PATH=$lib python bench.py --steps 40 --warmup 10 --no-extra --no-cpu-baseline --no-ceiling --prospect 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', '$(basename $lib)', d['ms_per_step'], 'fwd', d['fwd_ms'], 'bwd', d['bwd_ms'])
"
}
A=$GRAFT_REPO_ROOT/tools/ab/libcnsn_base.so; B=$GRAFT_REPO_ROOT/tools/ab/libcnsn_snprio.so
for i in 1 2; do
for L in $A $B; do run f32 $L; done
for L in $A $B; do run bf16 $L --dtype bf16; done
for L in $A $B; do run f32sn $L --kind sn; done
for L in $A $B; do run bf16sn $L --kind sn --dtype bf16; done
for L in $A $B; do run bf16sn28 $L --kind sn --dtype bf16 --shape 256,512,28,28; done
for L in $A $B; do run bf16both $L --dtype bf16 --crop both; done
done
