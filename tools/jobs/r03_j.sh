#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for ch in 0 1 2; do
timeout 600 python bench.py --no-cpu --no-host-fed --no-extra --band 0 $ch 0 > gpurun_out/r03j_stereo_ch$ch.json 2> gpurun_out/r03j_stereo_ch$ch.err
tail -c 6000 gpurun_out/r03j_stereo_ch$ch.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunks', d['config']['chunks'], d['ms_per_step'], d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline'].get('frac_issued'), d['step_ms'])"
done
OUT=gpurun_out/profiles_r03_stereo_a
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py --no-cpu --no-host-fed --no-extra --steps 50 --warmup 5 > $OUT/trace.log 2>&1
python tools/rocpd_summary.py $OUT/trace/*.db > $OUT/kernel_trace_stats.txt 2>&1
cat $OUT/kernel_trace_stats.txt
rm -rf $OUT/trace
