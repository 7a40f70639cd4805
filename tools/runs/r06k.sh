# round 6: the block's last BatchNorm2d folded into the op's launch: parity, then the ResNet-50 step with / without it
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bn_block.py -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r06k_tests.txt; tail -25 gpurun_out/r06k_tests.txt
for b in 1 0; do
  echo "== resnet50 CNSN_BN_BLOCK=$b" | tee -a gpurun_out/r06k_model.txt
  CNSN_BN_BLOCK=$b timeout 400 python bench.py --workload resnet50 --steps 30 --warmup 8 2>/dev/null | tail -1 | cut -c1-330 | tee -a gpurun_out/r06k_model.txt
done
