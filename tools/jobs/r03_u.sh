#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_fused_vote.py tests/test_cpp_adapter.py tests/test_gpu_process.py -m gpu -x -q 2>&1 | tail -12
timeout 300 python tools/fused_trace.py 2>&1 | tail -32
bash tools/jobs/r03_w.sh
