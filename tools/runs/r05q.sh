cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05q; mkdir -p $O
rm -rf /tmp/rp_rn
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_rn -- python $R/bench.py --workload resnet50 --steps 12 --warmup 4 > /tmp/rp_rn.log 2>&1
f=$(for g in $(find /tmp/rp_rn -name "*kernel_stats.csv"); do echo "$(grep -c cnsn:: $g) $g"; done | sort -rn | head -1 | cut -d" " -f2)
python $R/profiles/summarize.py "$f" $O/resnet50_cl_step_kernel_stats.csv
head -45 $O/resnet50_cl_step_kernel_stats.csv | cut -c1-190
