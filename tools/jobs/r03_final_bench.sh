#!/bin/bash
# the bench lines quoted in DESIGN.md section 5 / README (round 3)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
run() { local tag=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/r03_bench_line$tag.json 2> gpurun_out/r03_bench_line$tag.err; echo "$tag rc=$?"; tail -1 gpurun_out/r03_bench_line$tag.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value'],1), d['unit'], 'kernel', d['roofline']['kernel'], round(d['roofline']['kernel_avg_ms'],4), 'frac', round(d['roofline']['frac'],4), 'issued', d['roofline'].get('frac_issued'))
"; }
run "" 
run _windows --workload windows
run _windows_serial --workload windows --serial-windows --no-cpu
run _windows_unfused --workload windows --no-fused-vote --no-cpu --no-host-fed
run _cameras4 --workload cameras4 --no-cpu
run _cameras4_log --workload cameras4 --gm log --no-cpu --no-host-fed
run _1024 --dims 1024 1024 256 --no-cpu --no-host-fed --no-extra
run _640 --dims 640 480 100 --no-cpu --no-host-fed --no-extra
