cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh $GRAFT_REPO_ROOT/gpurun_out/r05z > gpurun_out/r05z_collect.log 2>&1
tail -5 gpurun_out/r05z_collect.log
