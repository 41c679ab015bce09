#!/bin/bash
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for br in 0 13 15 17 20 22; do
timeout 600 python bench.py --band $br 1 0 --no-cpu --no-host-fed --no-extra --steps 50 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('band_rows $br', round(d['ms_per_step'],4), round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['frac'],4), 'bands', d['config']['bands'], 'chunks', d['config']['chunks'])"
done; done
