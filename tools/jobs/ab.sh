cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_windows.py tests/test_gpu_process.py -q -x 2>&1 | grep "passed\|failed" | tail -3
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu --no-host-fed $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms frac %.3f bands %d rows %d chunks %d pk %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['frac'], d['config']['bands'], d['config']['band_rows'], d['config']['chunks'], d['config']['packed_lanes']))"
}
run "stereo" "A=1" "--steps 30 --warmup 3"
run "windows" "A=1" "--workload windows"
run "640x480" "A=1" "--dims 640 480 100 --steps 10 --warmup 2"
run "480x360x100" "A=1" "--dims 480 360 100 --steps 10 --warmup 2"
run "400x300x64" "A=1" "--dims 400 300 64 --steps 10 --warmup 2"
run "240x180x100" "A=1" "--dims 240 180 100 --steps 10 --warmup 2"
run "1280x720x64" "A=1" "--dims 1280 720 64 --steps 10 --warmup 2"
