# round 6, first GPU call of the re-entered session: the single-launch channels-last kernels (parity, sites), headline line
mkdir -p gpurun_out
python -m pytest tests/test_gpu_nhwc.py tests/test_gpu_saved_contract.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r06a_nhwc_tests.txt
for f in 0 1; do
  echo "== CNSN_NHWC_FUSED=$f" >> gpurun_out/r06a_sites.txt
  CNSN_NHWC_FUSED=$f python tools/nhwc_sites.py bf16 >> gpurun_out/r06a_sites.txt 2>&1
done
python bench.py --steps 20 --warmup 5 > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err
tail -c 3000 gpurun_out/r06a_nhwc_tests.txt; cat gpurun_out/r06a_sites.txt; tail -c 1500 gpurun_out/r06a_bench.json
