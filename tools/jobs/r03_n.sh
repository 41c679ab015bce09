#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_vote.py tests/test_golden.py -x -q 2>&1 | tail -4
for a in "--dims 1024 1024 256" "--dims 640 480 100" "--workload cameras4"; do
timeout 600 python bench.py $a --no-cpu --no-host-fed --no-extra 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$a', round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['frac'],4))"
done
