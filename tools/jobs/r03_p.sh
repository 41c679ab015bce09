#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_vote.py -x -q -k "hand_scheduled or lane_mappings or fused" 2>&1 | tail -3
for dims in "346 260 100" "512 512 200" "640 480 100" "1024 1024 256" "240 180 100" "480 360 100" "800 600 128"; do
for cfg in "--packed -1" "--packed 7"; do
timeout 600 python bench.py --dims $dims $cfg --no-cpu --no-host-fed --no-extra --steps 30 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$dims $cfg', round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['frac'],4), 'mapping', d['config']['packed_lanes'], 'bands', d['config']['bands'], 'chunks', d['config']['chunks'])"
done; done
timeout 600 python bench.py --workload windows --no-cpu --no-host-fed 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('windows', d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
