#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_fused_vote.py -m gpu -x -q 2>&1 | tail -4
bash tools/jobs/r03_w.sh
