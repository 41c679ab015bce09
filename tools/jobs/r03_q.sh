#!/bin/bash
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for ch in 1 2 3 4; do
timeout 600 python bench.py --band 0 $ch 0 --no-cpu --no-host-fed --no-extra --steps 50 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('chunks $ch', round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['frac'],4), 'mapping', d['config']['packed_lanes'])"
done; done
for lg in 3 4 5; do
timeout 600 python bench.py --pass-lg $lg --no-cpu --no-host-fed --no-extra --steps 50 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('lg $lg', round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['frac'],4), 'mapping', d['config']['packed_lanes'])"
done
for lg in 4 5 6; do
timeout 600 python bench.py --dims 512 512 200 --pass-lg $lg --no-cpu --no-host-fed --no-extra --steps 30 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('512 lg $lg', round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['frac'],4), 'mapping', d['config']['packed_lanes'])"
done
