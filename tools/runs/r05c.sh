cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do python -m pytest tests/test_gpu_arena.py -q -m gpu -k results_do_not 2>&1 | grep -E "^E +AssertionError|^E +assert|passed|failed" | head -8; done
