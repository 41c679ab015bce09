#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fused_vote.py -x -q 2>&1 | tail -25
timeout 300 python tools/fused_trace.py 2>&1 | tail -20
timeout 300 python tools/fused_trace.py 500000 5 2>&1 | tail -20
