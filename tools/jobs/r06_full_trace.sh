cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_workloads.sh r06 cameras4_full > gpurun_out/pw_full.log 2>&1
cat gpurun_out/profiles_r06_cameras4_full/kernel_trace_stats.txt | head -30
cat gpurun_out/profiles_r06_cameras4_full/bench_line_under_rocprof.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
