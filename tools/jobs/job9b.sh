cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_windows.py tests/test_cpp_adapter.py tests/test_gpu_pipelined_fusion.py -q -m gpu 2>&1 | grep -v "NCCL WARN\|^$" | tail -8
bash tools/jobs/job9.sh
