cd $GRAFT_REPO_ROOT
SECONDS=0
timeout 1500 python bench.py --workload cameras4 --events 100000000 --steps 3 --warmup 1 --no-cpu --no-host-fed > gpurun_out/r02_bench_line_cameras4_full.json 2> gpurun_out/cam4full.err
echo "wall $SECONDS s"
tail -1 gpurun_out/r02_bench_line_cameras4_full.json | cut -c1-700
tail -3 gpurun_out/cam4full.err
