#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r03_windows_a
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py --workload windows --no-cpu --no-host-fed --steps 50 --warmup 5 > $OUT/trace.log 2>&1
python tools/rocpd_summary.py $OUT/trace/*.db > $OUT/kernel_trace_stats.txt 2>&1
cat $OUT/kernel_trace_stats.txt
tail -1 $OUT/trace.log | cut -c1-300
rm -rf $OUT/trace
