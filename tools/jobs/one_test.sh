cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "configs4_camera" --durations=3 2>&1 | tail -12
