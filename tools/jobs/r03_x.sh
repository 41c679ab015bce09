#!/bin/bash
# pass size of the dealt packed stream at mid-range packet counts
cd "$(dirname "$0")/../.."
for ev in 1000000 2000000 4000000; do
  for lg in 2 3 4 5; do
    timeout 600 python bench.py --events $ev --no-cpu --no-host-fed --no-extra --pass-lg $lg --steps 100 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('stereo ev $ev lg $lg', round(d['ms_per_step'],4), 'ms/step kernel', round(d['roofline']['kernel_avg_ms'],4))
"
  done
done
for lg in 3 4; do
timeout 600 python bench.py --workload windows --no-cpu --no-host-fed --pass-lg $lg --steps 200 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('windows lg $lg', round(d['ms_per_step'],4), 'ms/window kernel', round(d['roofline']['kernel_avg_ms'],4))
"
done
for ev in 1000000 4000000; do
  for lg in 2 3 4; do
    timeout 600 python bench.py --dims 512 512 200 --events $ev --no-cpu --no-host-fed --no-extra --pass-lg $lg --steps 50 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('512 ev $ev lg $lg', round(d['ms_per_step'],4), 'ms/step kernel', round(d['roofline']['kernel_avg_ms'],4))
"
  done
done
