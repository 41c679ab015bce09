cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/suite.txt 2>&1
grep -n "passed\|failed\|error" gpurun_out/suite.txt | tail -5
