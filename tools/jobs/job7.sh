cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_windows.py -q -x 2>&1 | grep -v "^$" | tail -40
timeout 900 python -m pytest tests/test_gpu_multirank.py -q -x -k rccl 2>&1 | grep -v "NCCL WARN\|^$" | tail -15
