cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lane_mappings or duplicate_events or hand_scheduled or nary or another_context" 2>&1 | tail -15
run() { # tag, args
  timeout 600 python bench.py --no-cpu $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms frac %.3f bands %d rows %d chunks %d lds %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['lds_atomics']['frac_of_conflict_free_peak'], d['config']['bands'], d['config']['band_rows'], d['config']['chunks'], d['config']['lds_bytes']))"
}
for pk in 1 5; do
  run "512 pk$pk" "--dims 512 512 200 --events 500000 --packed $pk --steps 20 --warmup 3"
  run "1024 pk$pk" "--dims 1024 1024 256 --events 10000000 --packed $pk --steps 5 --warmup 1"
  run "346 pk$pk" "--packed $pk --steps 10 --warmup 2"
done
run "512 pk5 rows18" "--dims 512 512 200 --events 500000 --packed 5 --band 18 0 0 --steps 20 --warmup 3"
run "512 pk5 rows18 b512" "--dims 512 512 200 --events 500000 --packed 5 --band 18 0 512 --steps 20 --warmup 3"
run "1024 pk5 rows8" "--dims 1024 1024 256 --events 10000000 --packed 5 --band 8 0 0 --steps 5 --warmup 1"
run "1024 pk5 rows8 b512" "--dims 1024 1024 256 --events 10000000 --packed 5 --band 8 0 512 --steps 5 --warmup 1"
run "346 pk5 rows24" "--packed 5 --band 24 0 0 --steps 10 --warmup 2"
run "640x480 pk1" "--dims 640 480 100 --packed 1 --steps 5 --warmup 1"
run "640x480 pk5" "--dims 640 480 100 --packed 5 --steps 5 --warmup 1"
