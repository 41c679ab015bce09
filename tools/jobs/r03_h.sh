#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
echo skip tests
for mode in "" "--serial-windows"; do
timeout 600 python bench.py --workload windows --no-cpu --no-host-fed $mode > gpurun_out/r03h_windows$mode.json 2> gpurun_out/r03h_windows$mode.err
tail -c 5000 gpurun_out/r03h_windows$mode.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_avg_ms'], d['roofline']['kernel_launches'], d['roofline']['frac'], d['roofline'].get('frac_issued'), d['step_ms'])"
done
timeout 600 python bench.py --workload windows --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_fed'])"
