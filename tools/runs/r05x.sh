cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_nhwc.py -q -m gpu -x 2>&1 | tail -2
python -m pytest tests -q -m gpu -x -k "two_pass or twopass or stream or strategy or parity or golden" 2>&1 | tail -2
python tools/nhwc_sites.py bf16 2>/dev/null | grep "^|"
