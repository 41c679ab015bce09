cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "NCCL WARN\|^$" | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench.err; tail -c 400 gpurun_out/r02_bench.err
timeout 900 python bench.py --workload windows > gpurun_out/r02_bench_line_windows.json 2>> gpurun_out/r02_bench.err
timeout 900 python bench.py --workload cameras4 > gpurun_out/r02_bench_line_cameras4.json 2>> gpurun_out/r02_bench.err
timeout 900 python bench.py --dims 1024 1024 256 --steps 5 --warmup 1 --no-cpu > gpurun_out/r02_bench_line_1024.json 2>> gpurun_out/r02_bench.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu > gpurun_out/r02_bench_torchrun1.json 2>> gpurun_out/r02_bench.err
for f in gpurun_out/r02_bench_line*.json gpurun_out/r02_bench_torchrun1.json; do tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', '%.0f Mev/s step %.3f ms kern %.4f ms frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['frac']))"; done
export TMPDIR=/tmp; mkdir -p gpurun_out/tr; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-host-fed --steps 50 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/r02_trace_bench.log 2>&1)
python tools/rocpd_summary.py gpurun_out/tr/*.db > gpurun_out/r02_kernel_trace_stats_timed_only.txt 2>&1; rm -rf gpurun_out/tr; head -14 gpurun_out/r02_kernel_trace_stats_timed_only.txt; tail -1 gpurun_out/r02_trace_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench under rocprof: kernel_avg_ms', d['roofline']['kernel_avg_ms'], 'launches', d['roofline']['kernel_launches'])"
