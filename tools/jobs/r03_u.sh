#!/bin/bash
cd "$(dirname "$0")/../.."
for q in 4 8 16; do
  for mode in "" "--prepare-ahead" "--window-depth 3"; do
    GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --workload windows --no-cpu --no-host-fed $mode --steps 240 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('queues $q windows $mode', round(d['ms_per_step'],4), 'ms/window kernel', round(d['roofline']['kernel_avg_ms'],4))
"
  done
done
