#!/bin/bash
# Per-round profiles of every workload (on the GPU box): rocprofv3 kernel trace + stats of the timed steps, PMC counters in their own
# passes (never combined with other trace domains), for the default bench (configs[1]), the configs[2] window stream
# (one stream, so that per-kernel durations are not inflated by the overlap of consecutive windows), the configs[4]
# shape and the 1024x1024x256 stereo shape.   tools/profile_workloads.sh r04 [stereo windows windows_two_streams cameras4 cameras4_unfused cameras4_full 1024]  ->  gpurun_out/profiles_r04_*/
set -u
cd "$(dirname "$0")/.."
R=${1:-r04}
shift || true
WHICH=${*:-stereo windows windows_two_streams cameras4 1024}
want() { case " $WHICH " in *" $1 "*) return 0;; esac; return 1; }
export TMPDIR=/tmp
SQ1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES"
SQ3="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
one() {  # TAG "bench args for the trace" "bench args for the PMC passes" groups...
  local TAG=$1 TARGS=$2 PARGS=$3; shift 3
  local OUT=gpurun_out/profiles_$TAG
  mkdir -p $OUT
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py --no-cpu --no-extra $TARGS > $OUT/trace.log 2>&1
  echo "$TAG trace rc=$?"
  python tools/rocpd_summary.py $OUT/trace/*.db > $OUT/kernel_trace_stats.txt 2>&1
  grep "^{" $OUT/trace.log | tail -1 > $OUT/bench_line_under_rocprof.json
  local i=0
  for grp in "$@"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc$i -o pmc$i -- python bench.py --no-cpu --no-extra $PARGS > $OUT/pmc$i.log 2>&1
    echo "$TAG pmc$i ($grp) rc=$?"
  done
  python tools/rocpd_summary.py $OUT/pmc*/*.db > $OUT/pmc_counters.txt 2>&1
  rm -rf $OUT/trace $OUT/pmc[0-9]*
}
want stereo && one ${R}_stereo "--no-host-fed --steps 50 --warmup 5" "--no-sensitivity --steps 3 --warmup 1" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "$SQ1" "$SQ2" "$SQ3" "GRBM_GUI_ACTIVE GRBM_COUNT"
want stereo && python tools/make_traffic_json.py gpurun_out/profiles_${R}_stereo/pmc_counters.txt 512 512 200 > gpurun_out/profiles_${R}_stereo/traffic.json
want windows && one ${R}_windows "--workload windows --serial-windows --no-host-fed --steps 50 --warmup 5" "--workload windows --serial-windows --no-host-fed --steps 3 --warmup 1" "$SQ1" "$SQ2" "$SQ3" "FETCH_SIZE" "WRITE_SIZE"
want windows_two_streams && one ${R}_windows_two_streams "--workload windows --no-host-fed --steps 50 --warmup 5" "" 
want cameras4 && one ${R}_cameras4 "--workload cameras4 --camera-streams 1 --no-host-fed --steps 10 --warmup 2" "--workload cameras4 --camera-streams 1 --no-host-fed --steps 3 --warmup 1 --clock-ramp 0" "$SQ1" "$SQ2" "$SQ3" "FETCH_SIZE" "WRITE_SIZE"
# the same shape through the banded kernels with the DSIs written (the round-5 path of this workload), for comparison
want cameras4_unfused && one ${R}_cameras4_unfused "--workload cameras4 --no-fused-vote --camera-streams 1 --no-host-fed --steps 10 --warmup 2" "--workload cameras4 --no-fused-vote --camera-streams 1 --no-host-fed --steps 3 --warmup 1 --clock-ramp 0" "$SQ1" "$SQ2"
# BASELINE configs[4] at its own size (4 x 100 M events): the kernel trace of the bench's cameras4_full sub-run + its counters
want cameras4_full && one ${R}_cameras4_full "--workload cameras4 --events 100000000 --tile 10 --camera-streams 1 --no-host-fed --steps 3 --warmup 1 --clock-ramp 1" "--workload cameras4 --events 100000000 --tile 10 --camera-streams 1 --no-host-fed --steps 1 --warmup 1 --clock-ramp 0" "$SQ1" "$SQ2" "FETCH_SIZE" "WRITE_SIZE"
want 1024 && one ${R}_1024 "--dims 1024 1024 256 --events 10000000 --no-host-fed --steps 10 --warmup 2" "--dims 1024 1024 256 --events 10000000 --no-host-fed --steps 3 --warmup 1" "$SQ1" "$SQ2" "$SQ3"
ls -la gpurun_out/profiles_${R}_*
