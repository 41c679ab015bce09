cd $GRAFT_REPO_ROOT
DSI_EXPERIMENT=4 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -q -x -k "lane_mappings or duplicate or hand_scheduled or baseline_large or plane_sharded or four_cameras" 2>&1 | grep "passed\|failed" | tail -3
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu --no-host-fed $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms frac %.3f bands %d rows %d pk %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['frac'], d['config']['bands'], d['config']['band_rows'], d['config']['packed_lanes']))"
}
run "1024 three sets" "A=1" "--dims 1024 1024 256 --steps 5 --warmup 1"
run "1024 bperm-ahead" "DSI_EXPERIMENT=4" "--dims 1024 1024 256 --steps 5 --warmup 1"
run "1024 two sets" "DSI_EXPERIMENT=3" "--dims 1024 1024 256 --steps 5 --warmup 1"
run "cameras4 bperm-ahead" "DSI_EXPERIMENT=4" "--workload cameras4"
run "windows pk5 bperm-ahead" "DSI_EXPERIMENT=4" "--workload windows --packed 5"
run "640x480 pk5 bperm-ahead" "DSI_EXPERIMENT=4" "--dims 640 480 100 --steps 10 --warmup 2 --packed 5"
