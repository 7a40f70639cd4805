# A/B: real masks vs free masks (CNSN_FAKE_MASKS build, wrong results) on the boxed kernels
for rep in 1 2; do
for lib in "" tools/ab/libcnsn_fake.so; do
for dt in bf16 f32; do
  echo "== lib=${lib:-default} dtype=$dt"
  CNSN_LIB_PATH=${lib:+$PWD/$lib} python bench.py --crop both --dtype $dt --no-extra --no-alt --steps 40 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('ms_per_step',d['ms_per_step'], {k:v for k,v in d.items() if 'fwd' in k or 'bwd' in k or k=='phases'})
"
done; done; done
