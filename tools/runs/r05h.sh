cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
tools/coop_probe | tee $O/r05h_coop.txt
python -m pytest tests/test_gpu_foreign_kernel.py tests/test_gpu_step_guard.py tests/test_gpu_arena.py -x -q -m gpu 2>&1 | tail -4
cat $O/foreign_kernel_headroom.json | python -c "
import json,sys
for r in json.load(sys.stdin): print(r)
"
python bench.py --workload seg --steps 6 --warmup 3 2>&1 | tail -1 > $O/r05h_seg_f32.json; cat $O/r05h_seg_f32.json | cut -c1-600
python bench.py --workload seg --steps 6 --warmup 3 --dtype bf16 2>&1 | tail -1 > $O/r05h_seg_bf16.json; cat $O/r05h_seg_bf16.json | cut -c1-600
