#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for rep in 1 2; do
  for dpt in 2 3 4; do
    timeout 600 python bench.py --workload windows --no-cpu --no-host-fed --window-depth $dpt --steps 240 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('windows depth $dpt', round(d['ms_per_step'],4), 'ms/window kernel', round(d['roofline']['kernel_avg_ms'],4))
"
  done
done
