# round 6: per-kernel durations of the channels-last sites, single-launch vs two-pass
mkdir -p gpurun_out; R=$(pwd); cd /tmp; export TMPDIR=/tmp
for f in 1 0; do
  CNSN_NHWC_FUSED=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f$f -- python $R/tools/nhwc_sites.py bf16 cl > /tmp/sites_f$f.txt 2>&1
  find /tmp/prof_f$f -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/r06b_sites_fused${f}_kernel_stats.csv \;
  grep -v "^W2026\|^E2026" /tmp/sites_f$f.txt | tail -12
done
cd $R; head -12 gpurun_out/r06b_sites_fused1_kernel_stats.csv | cut -c1-220; head -14 gpurun_out/r06b_sites_fused0_kernel_stats.csv | cut -c1-220
