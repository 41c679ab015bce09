#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 1500 python -m pytest tests/test_gpu_fused_vote.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
for lg in 0 1 2 3; do
  for mode in "" "--serial-windows"; do
    timeout 600 python bench.py --workload windows --no-cpu --no-host-fed --pass-lg $lg $mode --steps 200 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('windows lg $lg $mode', round(d['ms_per_step'],4), 'ms/window kernel', round(d['roofline']['kernel_avg_ms'],4))
"
  done
done
for lg in 0 3 4; do
timeout 600 python bench.py --no-cpu --no-host-fed --no-extra --pass-lg $lg --steps 100 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('stereo lg $lg', round(d['ms_per_step'],4), 'ms/step kernel', round(d['roofline']['kernel_avg_ms'],4))
"
done
for lg in 1 2 3; do
timeout 600 python bench.py --events 1000000 --no-cpu --no-host-fed --no-extra --pass-lg $lg --steps 100 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('stereo 1M lg $lg', round(d['ms_per_step'],4), 'ms/step kernel', round(d['roofline']['kernel_avg_ms'],4))
"
done
