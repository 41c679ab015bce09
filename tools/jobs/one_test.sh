cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -v -x > gpurun_out/one_test_full.txt 2>&1
grep -n "PASSED\|FAILED" gpurun_out/one_test_full.txt | tail -3
grep -n "fault\|Fatal\|Abort\|error" gpurun_out/one_test_full.txt | head -5
grep -n "test_gpu_parity.py\", line" gpurun_out/one_test_full.txt | head -5
