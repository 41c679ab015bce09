# local (not on the GPU box): copy what final_all.sh rNN merged into gpurun_out/ to profiles/ under the round's names
R=${1:-r06}; G=gpurun_out
cp $G/profiles_$R/kernel_trace_stats.txt profiles/${R}_kernel_trace_stats.txt
cp $G/profiles_$R/pmc_counters.txt profiles/${R}_pmc_counters.txt
cp $G/profiles_$R/bench_line_under_rocprof.json profiles/${R}_bench_line_under_rocprof.json
cp $G/profiles_$R/traffic.json profiles/traffic.json
cp $G/${R}_kernel_trace_stats_timed_only.txt profiles/${R}_kernel_trace_stats_timed_only.txt
cp $G/bench_line.json profiles/${R}_bench_line.json
cp $G/suite.txt profiles/${R}_pytest_gpu.txt
cp $G/counters_$R.json profiles/counters.json
for w in windows cameras4 cameras4_full cameras4_unfused 1024; do
  for f in kernel_trace_stats.txt pmc_counters.txt bench_line_under_rocprof.json; do cp $G/profiles_${R}_$w/$f profiles/${R}_${w}_$f; done
done
