#!/bin/bash
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r03_fetch_calib
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib 2>/dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc1 -o pmc1 -- tools/fetch_calib > $OUT/run.log 2>&1
python tools/rocpd_summary.py --all $OUT/pmc1/*.db > $OUT/pmc_counters.txt 2>&1
cat $OUT/run.log | tail -3
cat $OUT/pmc_counters.txt
rm -rf $OUT/pmc1
