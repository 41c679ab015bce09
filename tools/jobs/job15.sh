cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lane_mappings or duplicate or hand_scheduled or baseline_large or plane_sharded" 2>&1 | grep -v "NCCL WARN\|^$" | tail -3
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms kernMev/s %.0f frac %.3f bands %d rows %d chunks %d pk %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['kernel_Mevents_per_s'], r['frac'], d['config']['bands'], d['config']['band_rows'], d['config']['chunks'], d['config']['packed_lanes']))"
}
run "stereo" "A=1" ""
run "1024 10M" "A=1" "--dims 1024 1024 256 --steps 5 --warmup 1"
run "1024 10M b512 (8 waves/CU)" "A=1" "--dims 1024 1024 256 --steps 5 --warmup 1 --band 17 0 512"
run "1024 10M b256 (4 waves/CU)" "A=1" "--dims 1024 1024 256 --steps 3 --warmup 1 --band 17 0 256"
run "1024 10M lg5" "DSI_PASS_LG=5" "--dims 1024 1024 256 --steps 5 --warmup 1"
run "1024 10M lg4" "DSI_PASS_LG=4" "--dims 1024 1024 256 --steps 5 --warmup 1"
run "1024 10M pk1" "A=1" "--dims 1024 1024 256 --steps 5 --warmup 1 --packed 1"
run "1024 10M nonpersistent" "DSI_PERSISTENT=0" "--dims 1024 1024 256 --steps 5 --warmup 1"
run "windows" "A=1" "--workload windows"
run "windows b512" "A=1" "--workload windows --band 37 0 512"
