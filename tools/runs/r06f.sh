# round 6: (1) the whole GPU suite on the build with ABI 8 / retired pipe classes / float64 SelfNorm, (2) PMC traffic of the
# channels-last sites, (3) ResNet-50 with the single-launch kernels limited by tensor size
mkdir -p gpurun_out; R=$(pwd)
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r06f_pytest_gpu.txt; tail -3 gpurun_out/r06f_pytest_gpu.txt
cd /tmp; export TMPDIR=/tmp
rm -f $R/gpurun_out/r06f_sites_pmc.txt
for site in 0 1 2 3; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_s${site}_$ctr
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_s${site}_$ctr -- python $R/tools/nhwc_sites.py bf16 cl site$site > /dev/null 2>&1
    echo "site$site $ctr" >> $R/gpurun_out/r06f_sites_pmc.txt
    python $R/profiles/pmc_summary.py /tmp/pmc_s${site}_$ctr nhwc_fused >> $R/gpurun_out/r06f_sites_pmc.txt
  done
done
cat $R/gpurun_out/r06f_sites_pmc.txt
cd $R
for f in 1 160 320; do
  echo "== resnet50 CNSN_NHWC_FUSED=$f" | tee -a gpurun_out/r06f_model.txt
  CNSN_NHWC_FUSED=$f python bench.py --workload resnet50 --steps 30 --warmup 8 2>/dev/null | tail -1 | cut -c1-330 | tee -a gpurun_out/r06f_model.txt
done
