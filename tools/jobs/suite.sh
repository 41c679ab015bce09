cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -m gpu -x > gpurun_out/suite.txt 2>&1
grep -n "passed\|failed\|error" gpurun_out/suite.txt | tail -5
