cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -q -x 2>&1 | grep "passed\|failed" | tail -2
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu --no-host-fed $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms frac %.3f chunks %d pk %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['frac'], d['config']['chunks'], d['config']['packed_lanes']))"
}
run "stereo" "A=1" ""
run "stereo novote" "DSI_EXPERIMENT=1" ""
run "windows" "A=1" "--workload windows"
run "windows novote" "DSI_EXPERIMENT=1" "--workload windows"
run "cameras4" "A=1" "--workload cameras4"
run "1024" "A=1" "--dims 1024 1024 256 --steps 5 --warmup 1"
run "640x480" "A=1" "--dims 640 480 100 --steps 10 --warmup 2"
run "stereo 1M" "A=1" "--events 1000000"
