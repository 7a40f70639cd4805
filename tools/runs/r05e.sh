cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_arena.py tests/test_gpu_launch_args.py -q -m gpu 2>&1 | tail -15
python tools/arena_probe.py 2 8,56,784 2>&1 | grep -v amdgpu.ids | grep -v '^{' | tee $O/r05e_probe.md
