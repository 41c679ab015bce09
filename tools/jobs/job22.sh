cd $GRAFT_REPO_ROOT
echo "persistent forced on"; DSI_PERSISTENT=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_process.py -q -x 2>&1 | grep "passed\|failed" | tail -2
echo "persistent forced off"; DSI_PERSISTENT=0 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_windows.py -q -x 2>&1 | grep "passed\|failed" | tail -2
echo "two-set vector fill"; DSI_EXPERIMENT=3 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lane_mappings or duplicate or hand_scheduled or baseline_large" 2>&1 | grep "passed\|failed" | tail -2
