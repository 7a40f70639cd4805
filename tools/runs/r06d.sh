# round 6: the kept sum (ABI 8) on the channels-last sites: parity, site times, ResNet-50 step
mkdir -p gpurun_out
python -m pytest tests/test_gpu_nhwc.py tests/test_gpu_saved_contract.py tests/test_gpu_fused_block.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r06d_tests.txt
tail -6 gpurun_out/r06d_tests.txt
for k in 1 0; do for f in 1 0; do
  echo "== CNSN_KEEP_SUM=$k CNSN_NHWC_FUSED=$f" | tee -a gpurun_out/r06d_sites.txt
  CNSN_KEEP_SUM=$k CNSN_NHWC_FUSED=$f python tools/nhwc_sites.py bf16 cl 2>/dev/null | tail -4 | tee -a gpurun_out/r06d_sites.txt
done; done
for k in 1 0; do for f in 1 0; do
  echo "== resnet50 CNSN_KEEP_SUM=$k CNSN_NHWC_FUSED=$f" | tee -a gpurun_out/r06d_model.txt
  CNSN_KEEP_SUM=$k CNSN_NHWC_FUSED=$f python bench.py --workload resnet50 --steps 20 --warmup 8 2>/dev/null | tail -1 | cut -c1-400 | tee -a gpurun_out/r06d_model.txt
done; done
