#!/bin/bash
# rocprofv3 passes for one round: kernel trace + stats of the default bench, then PMC counters
# in separate passes (TCC: FETCH_SIZE and WRITE_SIZE cannot share a pass; SQ in two groups).
# Usage (on the GPU box): tools/profile_round.sh r01      -> gpurun_out/profiles_r01/*.txt
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r01}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --no-cpu --no-sensitivity --no-extra"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH --steps 10 --warmup 2 > $OUT/trace.log 2>&1
echo "trace rc=$?"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc$i -o pmc$i -- $BENCH --steps 3 --warmup 1 > $OUT/pmc$i.log 2>&1
  echo "pmc$i ($grp) rc=$?"
done
python tools/rocpd_summary.py $OUT/trace/*.db > $OUT/kernel_trace_stats.txt 2>&1
python tools/rocpd_summary.py $OUT/pmc*/*.db > $OUT/pmc_counters.txt 2>&1
grep "^{" $OUT/trace.log | tail -1 > $OUT/bench_line_under_rocprof.json
python tools/make_traffic_json.py $OUT/pmc_counters.txt 512 512 200 > $OUT/traffic.json
rm -rf $OUT/trace $OUT/pmc[0-9]*  # keep the summaries (the .db files are large)
ls -la $OUT
