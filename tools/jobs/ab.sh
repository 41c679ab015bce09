cd $GRAFT_REPO_ROOT
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu --no-host-fed $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms frac %.3f bands %d rows %d chunks %d pk %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['frac'], d['config']['bands'], d['config']['band_rows'], d['config']['chunks'], d['config']['packed_lanes']))"
}
run "640x480 auto" "A=1" "--dims 640 480 100 --steps 10 --warmup 2"
run "640x480 pk5 2perCU" "DSI_VFILL_CAPPED=1" "--dims 640 480 100 --steps 10 --warmup 2 --packed 5 --band 14 0 1024"
run "640x480 pk5 rows14 1perCU" "A=1" "--dims 640 480 100 --steps 10 --warmup 2 --packed 5 --band 14 0 1024"
run "512x512x200 auto" "A=1" "--dims 512 512 200 --steps 10 --warmup 2"
run "512x512x200 pk5 2perCU" "DSI_VFILL_CAPPED=1" "--dims 512 512 200 --steps 10 --warmup 2 --packed 5 --band 18 0 1024"
run "800x600x128 auto" "A=1" "--dims 800 600 128 --steps 10 --warmup 2"
run "800x600x128 pk5 2perCU" "DSI_VFILL_CAPPED=1" "--dims 800 600 128 --steps 10 --warmup 2 --packed 5 --band 11 0 1024"
run "1024 auto" "A=1" "--dims 1024 1024 256 --steps 5 --warmup 1"
run "1024 pk5 2perCU" "DSI_VFILL_CAPPED=1" "--dims 1024 1024 256 --steps 5 --warmup 1 --packed 5 --band 8 0 1024"
run "windows pk5 2perCU" "DSI_VFILL_CAPPED=1" "--workload windows --packed 5 --band 18 0 1024"
