#!/bin/bash
cd "$(dirname "$0")/../.."
python dvs_mcemvs_amd/build.py --force --experiments > /dev/null 2>&1
for a in "--workload cameras4" "--dims 1024 1024 256" "--dims 640 480 100"; do
for lg in 0 4 5; do for ex in 100 101 102 104; do
DSI_PASS_LG=$lg DSI_EXPERIMENT=$ex timeout 600 python bench.py $a --no-cpu --no-host-fed --no-extra --steps 30 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$a lg $lg guided $ex', round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['frac'],4))"
done; done; done
