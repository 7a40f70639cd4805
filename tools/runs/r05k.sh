cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_sn_cluster.py tests/test_gpu_cn_partial.py tests/test_gpu_pipe.py tests/test_gpu_fused_block.py tests/test_gpu_saved_contract.py tests/test_gpu_resident_instantiations.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -3
