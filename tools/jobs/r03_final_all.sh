#!/bin/bash
# everything the round's numbers come from, on one box: the GPU test suite, the profiles, the bench lines
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r03_pytest_gpu.txt | tail -3
bash tools/profile_r03.sh > gpurun_out/r03_profile.log 2>&1; echo "profile rc=$?"; grep " rc=" gpurun_out/r03_profile.log
bash tools/jobs/r03_final_bench.sh
