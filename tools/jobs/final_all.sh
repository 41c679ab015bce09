# end-of-round job: re-stamp traffic.json on the final kernel source (tools/profile_round.sh), the workloads' traces, then
# the -m gpu suite, smoke() and the driver's bench command
cd $GRAFT_REPO_ROOT
R=${1:-r05}
bash tools/profile_round.sh $R > gpurun_out/profile_round_$R.log 2>&1
tail -2 gpurun_out/profile_round_$R.log
cp gpurun_out/profiles_$R/traffic.json profiles/traffic.json
bash tools/profile_workloads.sh $R windows cameras4 cameras4_unfused cameras4_full 1024 > gpurun_out/profile_workloads_$R.log 2>&1
# the "why" counters and the HBM-side traffic of every workload's voting kernel, stamped with the kernel source's hash
# (bench.py quotes them only while that hash is the running source's): copy to profiles/counters.json
python tools/make_counters_json.py stereo=gpurun_out/profiles_$R/pmc_counters.txt:9999360 \
    windows=gpurun_out/profiles_${R}_windows/pmc_counters.txt cameras4=gpurun_out/profiles_${R}_cameras4/pmc_counters.txt \
    cameras4_full=gpurun_out/profiles_${R}_cameras4_full/pmc_counters.txt 1024x1024x256=gpurun_out/profiles_${R}_1024/pmc_counters.txt \
    > gpurun_out/counters_$R.json 2> gpurun_out/counters_$R.err
cp gpurun_out/counters_$R.json profiles/counters.json
export TMPDIR=/tmp; mkdir -p gpurun_out/tr
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-host-fed --no-extra --no-sensitivity --steps 50 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/${R}_trace_bench.log 2>&1)
python tools/rocpd_summary.py gpurun_out/tr/*.db > gpurun_out/${R}_kernel_trace_stats_timed_only.txt 2>&1; rm -rf gpurun_out/tr
bash tools/jobs/suite_and_bench.sh
