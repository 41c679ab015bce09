#!/bin/bash
# band_rows chunks block sweeps of the LDS voting kernel (one summary line each)
cd "$(dirname "$0")/.."
EV=${EV:-4000000}
CFGS=${CFGS:-"0,0,0 52,16,1024 52,8,512 26,8,512 26,8,256 26,16,512 20,8,256 13,8,256"}
for cfg in $CFGS; do
  c=$(echo $cfg | tr ',' ' ')
  echo -n "band $c : "
  timeout 300 python bench.py --events $EV --steps 5 --warmup 2 --no-cpu --band $c 2>&1 | tail -1 | python -c "
import sys, json
try:
    j = json.loads(sys.stdin.read())
    print('value %.1f Mev/s  kernel %.3f ms  kernel %.1f Mev/s  frac %.3f  bands %d rows %d chunks %d block %d' % (j['value'], j['roofline']['kernel_avg_ms'], j['roofline']['kernel_Mevents_per_s'], j['roofline']['frac'], j['config']['bands'], j['config']['band_rows'], j['config']['chunks'], j['config']['block_threads']))
except Exception as e:
    print('FAILED', e)
"
done
