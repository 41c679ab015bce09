# the default workload's trace + PMC passes + traffic.json with the final kernel source, and the same SQ counters on the
# dense-scene input of the sensitivity block (200 k scene points)
cd $GRAFT_REPO_ROOT
R=${1:-r04}
bash tools/profile_round.sh $R > gpurun_out/profile_round_$R.log 2>&1
tail -3 gpurun_out/profile_round_$R.log
export TMPDIR=/tmp
SQ1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
SQ3="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
OUT=gpurun_out/profiles_${R}_dense
mkdir -p $OUT
i=0
for grp in "$SQ1" "$SQ3"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc$i -o pmc$i -- python bench.py --no-cpu --no-extra --no-sensitivity --no-host-fed --points 200000 --steps 3 --warmup 1 > $OUT/pmc$i.log 2>&1
  echo "dense pmc$i rc=$?"
done
python tools/rocpd_summary.py $OUT/pmc*/*.db > $OUT/pmc_counters.txt 2>&1
rm -rf $OUT/pmc[0-9]*
export TMPDIR=/tmp; mkdir -p gpurun_out/tr
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-host-fed --no-extra --no-sensitivity --steps 50 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/${R}_trace_bench.log 2>&1)
python tools/rocpd_summary.py gpurun_out/tr/*.db > gpurun_out/${R}_kernel_trace_stats_timed_only.txt 2>&1; rm -rf gpurun_out/tr
head -8 gpurun_out/${R}_kernel_trace_stats_timed_only.txt
