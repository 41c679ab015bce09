#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r03_pytest_gpu.txt | tail -3
