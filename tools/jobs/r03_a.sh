#!/bin/bash
# round 3, first GPU job: new fused-vote tests, windows bench with / without the fused kernel
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fused_vote.py -x -q 2>&1 | tail -25
for mode in "--fused-vote" "--no-fused-vote"; do
  timeout 600 python bench.py --workload windows --no-cpu --no-host-fed $mode > gpurun_out/r03a_windows$mode.json 2> gpurun_out/r03a_windows$mode.err
  echo "windows $mode rc=$?"; tail -c 1500 gpurun_out/r03a_windows$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'])"
done
