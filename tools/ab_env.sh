# usage: ab_env.sh "<ENV=VAL>" "<ENV=VAL>" reps -- bench args...   (alternating processes on one box)
A=$1; B=$2; REPS=$3; shift 4
R=$PWD
for i in $(seq $REPS); do for E in "$A" "$B"; do
  env $E python $R/bench.py --no-extra --no-cpu-baseline --no-ceiling "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$E', d['ms_per_step'], d['fwd_ms'], d['bwd_ms'])"
done; done
