#!/bin/bash
cd "$(dirname "$0")/../.."
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_fused_vote.py -m gpu -x -q -k "threads" 2>&1 | tail -3; done
