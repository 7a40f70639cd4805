cd $GRAFT_REPO_ROOT
python tools/runs/r05d.py 2>&1 | grep -v amdgpu.ids
echo "=== NO_TRIM"
NO_TRIM=1 python tools/runs/r05d.py 2>&1 | grep -v amdgpu.ids | grep -v "mismatches 0"
