cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "NCCL WARN\|^$" | tail -4
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms kernMev/s %.0f frac %.3f bands %d rows %d chunks %d pk %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['kernel_Mevents_per_s'], r['frac'], d['config']['bands'], d['config']['band_rows'], d['config']['chunks'], d['config']['packed_lanes']))"
}
run "stereo" "A=1" ""
run "stereo nonpersistent" "DSI_PERSISTENT=0" ""
run "stereo chunks1" "A=1" "--band 0 1 0"
run "stereo chunks3" "A=1" "--band 0 3 0"
run "stereo chunks4" "A=1" "--band 0 4 0"
run "windows" "A=1" "--workload windows"
run "windows rows18" "A=1" "--workload windows --band 18 0 0"
run "windows pk5 rows18" "A=1" "--workload windows --band 18 0 0 --packed 5"
run "cameras4" "A=1" "--workload cameras4"
run "1024 10M" "A=1" "--dims 1024 1024 256 --steps 5 --warmup 1"
run "1024 10M rows8" "A=1" "--dims 1024 1024 256 --steps 5 --warmup 1 --band 8 0 0"
run "640x480" "A=1" "--dims 640 480 100 --steps 10 --warmup 2"
