#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -k "hand_scheduled or lane_mappings or decomposition or golden or full_size_properties" 2>&1 | tail -3
python dvs_mcemvs_amd/build.py --force --experiments > /dev/null 2>&1
for rep in 1 2 3; do for ex in 0 200; do
DSI_EXPERIMENT=$ex timeout 600 python bench.py --no-cpu --no-host-fed --no-extra --steps 50 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('leftover rule $ex', round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['frac'],4))"
done; done
for ex in 0 200; do
DSI_EXPERIMENT=$ex timeout 600 python bench.py --dims 240 180 100 --no-cpu --no-host-fed --no-extra --steps 50 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('240x180 leftover rule $ex', round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4), d['config']['bands'], d['config']['chunks'])"
done
