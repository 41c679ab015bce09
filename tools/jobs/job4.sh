cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_multirank.py -q -x 2>&1 | tail -40
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "nary or another_context" 2>&1 | tail -5
