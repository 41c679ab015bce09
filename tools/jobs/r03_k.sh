#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_fused_vote.py tests/test_gpu_parity.py tests/test_golden.py -x -q 2>&1 | tail -6
timeout 600 python bench.py --no-cpu --no-host-fed --no-extra > gpurun_out/r03k_stereo.json 2> gpurun_out/r03k_stereo.err
tail -c 6000 gpurun_out/r03k_stereo.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunks', d['config']['chunks'], d['ms_per_step'], d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline'].get('frac_issued'), d['step_ms'])"
