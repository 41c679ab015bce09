cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -q -x 2>&1 | grep -v "NCCL WARN\|^$" | tail -6
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms kernMev/s %.0f frac %.3f bands %d rows %d chunks %d pk %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['kernel_Mevents_per_s'], r['frac'], d['config']['bands'], d['config']['band_rows'], d['config']['chunks'], d['config']['packed_lanes']))"
}
for P in 0 1; do
  run "stereo persistent=$P" "DSI_PERSISTENT=$P" "--steps 30 --warmup 3"
  run "windows persistent=$P" "DSI_PERSISTENT=$P" "--workload windows"
  run "windows pk5 rows18 b512 persistent=$P" "DSI_PERSISTENT=$P" "--workload windows --packed 5 --band 18 0 512"
  run "windows rows18 persistent=$P" "DSI_PERSISTENT=$P" "--workload windows --band 18 0 0"
  run "cameras4 persistent=$P" "DSI_PERSISTENT=$P" "--workload cameras4"
  run "1024 10M persistent=$P" "DSI_PERSISTENT=$P" "--dims 1024 1024 256 --steps 5 --warmup 1"
  run "640x480 persistent=$P" "DSI_PERSISTENT=$P" "--dims 640 480 100 --steps 10 --warmup 2"
done
run "stereo 1chunk persistent" "DSI_PERSISTENT=1" "--steps 30 --warmup 3 --band 0 1 0"
run "stereo 2chunk persistent" "DSI_PERSISTENT=1" "--steps 30 --warmup 3 --band 0 2 0"
