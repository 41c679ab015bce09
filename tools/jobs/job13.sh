cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "NCCL WARN\|^$" | tail -6
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms kernMev/s %.0f frac %.3f bands %d rows %d chunks %d pk %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['kernel_Mevents_per_s'], r['frac'], d['config']['bands'], d['config']['band_rows'], d['config']['chunks'], d['config']['packed_lanes']))"
}
run "stereo" "A=1" ""
run "windows" "A=1" "--workload windows"
run "cameras4" "A=1" "--workload cameras4"
run "640x480" "A=1" "--dims 640 480 100 --steps 10 --warmup 2"
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/trace13 -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 20 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/trace13.log 2>&1; cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py gpurun_out/trace13/*.db 2>/dev/null | head -20; find gpurun_out/trace13 -name "*.db" | head; rm -rf gpurun_out/trace13
