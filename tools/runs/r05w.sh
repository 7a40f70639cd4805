python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for w in resnet50 resnet50_jsd; do timeout 300 python bench.py --workload $w --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:70], d['value'], d['unit'], d['ms_per_step'])"; done
python tools/nhwc_sites.py bf16 2>/dev/null | grep "^|"
