cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/suite2.txt 2>&1
grep -n "passed\|failed\|error" gpurun_out/suite2.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "NCCL WARN" | tail -1
timeout 900 python bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench.err
timeout 900 python bench.py --workload windows > gpurun_out/r02_bench_line_windows.json 2>> gpurun_out/r02_bench.err
timeout 900 python bench.py --workload cameras4 > gpurun_out/r02_bench_line_cameras4.json 2>> gpurun_out/r02_bench.err
timeout 900 python bench.py --dims 1024 1024 256 --steps 5 --warmup 1 --no-cpu > gpurun_out/r02_bench_line_1024.json 2>> gpurun_out/r02_bench.err
for f in gpurun_out/r02_bench_line*.json; do tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', '%.0f Mev/s step %.3f ms kern %.4f ms frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['frac']))"; done
