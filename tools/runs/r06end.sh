# round 6, end: the profile set behind profiles/r06end_* (tools/collect_profiles.sh) + the channels-last site profiles
bash tools/collect_profiles.sh gpurun_out/r06end 2>&1 | tail -30
