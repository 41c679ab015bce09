#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 1500 python -m pytest tests/test_gpu_fused_vote.py tests/test_gpu_process.py -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do
  for mode in "" "--serial-windows"; do
    timeout 600 python bench.py --workload windows --no-cpu --no-host-fed $mode --steps 200 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('windows $mode', round(d['ms_per_step'],4), 'ms/window kernel', round(d['roofline']['kernel_avg_ms'],4))
"
  done
done
timeout 300 python tools/fused_trace.py 2>&1 | head -13
echo "== dealt (balance off)"; FUSED_NO_CUT_TABLE=1 timeout 300 python tools/fused_trace.py 2>&1 | head -13 | grep -E "span|wave done|phase total|consume"
