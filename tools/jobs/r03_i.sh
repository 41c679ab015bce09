#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_parity.py -x -q -k "every_device or gm_tree or configs4_end_to_end or another_context or nary" 2>&1 | tail -15
timeout 600 python bench.py --workload cameras4 --no-cpu > gpurun_out/r03i_cam4.json 2> gpurun_out/r03i_cam4.err
tail -c 6000 gpurun_out/r03i_cam4.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline'].get('frac_issued')); print(json.dumps(d['stream_kernels'], indent=0))"
