cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -q -x -k "lane_mappings or duplicate or hand_scheduled or baseline_large or plane_sharded or four_cameras or fuzz or vector" 2>&1 | grep "passed\|failed" | tail -3
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu --no-host-fed $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms frac %.3f bands %d rows %d pk %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['frac'], d['config']['bands'], d['config']['band_rows'], d['config']['packed_lanes']))"
}
run "1024" "A=1" "--dims 1024 1024 256 --steps 5 --warmup 1"
run "1024 2set" "DSI_EXPERIMENT=3" "--dims 1024 1024 256 --steps 5 --warmup 1"
run "cameras4" "A=1" "--workload cameras4"
run "windows pk5" "A=1" "--workload windows --packed 5"
run "stereo pk5" "A=1" "--packed 5 --steps 20"
