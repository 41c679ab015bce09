cd $GRAFT_REPO_ROOT
bash tools/profile_r02.sh > gpurun_out/profile_r02.log 2>&1
tail -5 gpurun_out/profile_r02.log
export TMPDIR=/tmp; mkdir -p gpurun_out/tr; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-host-fed --steps 50 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/r02_trace_bench.log 2>&1)
python tools/rocpd_summary.py gpurun_out/tr/*.db > gpurun_out/r02_kernel_trace_stats_timed_only.txt 2>&1; rm -rf gpurun_out/tr; head -14 gpurun_out/r02_kernel_trace_stats_timed_only.txt; tail -1 gpurun_out/r02_trace_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench under rocprof: kernel_avg_ms', d['roofline']['kernel_avg_ms'], 'launches', d['roofline']['kernel_launches'])"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu > gpurun_out/r02_bench_torchrun1.json 2>> gpurun_out/r02_bench.err
tail -1 gpurun_out/r02_bench_torchrun1.json | cut -c1-300
