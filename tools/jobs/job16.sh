cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "configs1_against" 2>&1 | grep -v "NCCL WARN\|^$" | tail -8
