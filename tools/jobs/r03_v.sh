#!/bin/bash
# windows stream: pass size of the dealt packed stream
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for rep in 1 2; do
for lg in 0 1 2 3; do
  for mode in "" "--serial-windows"; do
    timeout 600 python bench.py --workload windows --no-cpu --no-host-fed --pass-lg $lg $mode --steps 200 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('lg $lg $mode', round(d['ms_per_step'],4), 'ms/window kernel', round(d['roofline']['kernel_avg_ms'],4))
"
  done
done
done
