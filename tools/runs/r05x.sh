cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x -k "mono or wrn or wide or callers or parity or golden or pipe" 2>&1 | tail -2
python tools/auto_audit.py > gpurun_out/r05_auto_audit_box3_after.md 2> gpurun_out/r05_auto_audit_err.txt; tail -3 gpurun_out/r05_auto_audit_err.txt; grep -c revisit gpurun_out/r05_auto_audit_box3_after.md
