#!/bin/bash
# packed (dealt) vs vector fill at the shapes around the planner's threshold
cd "$(dirname "$0")/../.."
run() { timeout 600 python bench.py "$@" --no-cpu --no-host-fed --no-extra --steps 30 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
c=d['config']
print('$*', '->', round(d['ms_per_step'],4), 'ms/step kernel', d['roofline']['kernel'], round(d['roofline']['kernel_avg_ms'],4), 'frac', round(d['roofline']['frac'],4), 'bands', c.get('bands'), 'rows', c.get('band_rows'), 'block', c.get('block_threads'), 'packed', c.get('packed_lanes'))
"; }
for dims in "640 480 100" "800 600 128" "720 540 100" "512 512 200"; do
  for pk in -1 7 5; do
    run --dims $dims --packed $pk
  done
done
