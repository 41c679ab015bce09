#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fused_vote.py tests/test_gpu_first_principles.py -x -q 2>&1 | tail -15
timeout 300 python tools/fused_trace.py 2>&1 | tail -12
timeout 600 python bench.py --workload windows --no-cpu --no-host-fed > gpurun_out/r03d_windows.json 2> gpurun_out/r03d_windows.err
tail -c 4000 gpurun_out/r03d_windows.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline'].get('frac_issued'), d['step_ms'])"
