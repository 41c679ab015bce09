cd $GRAFT_REPO_ROOT
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms kernMev/s %.0f frac %.3f bands %d rows %d chunks %d pk %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['kernel_Mevents_per_s'], r['frac'], d['config']['bands'], d['config']['band_rows'], d['config']['chunks'], d['config']['packed_lanes']))"
}
for ev in 250000 500000 1000000 2000000 4000000 8000000; do
  run "512 ev$ev pk1" "A=1" "--dims 512 512 200 --events $ev --steps 20 --warmup 3 --packed 1"
done
for ev in 500000 2000000 8000000; do
  run "512 ev$ev pk5 rows18 b512" "A=1" "--dims 512 512 200 --events $ev --steps 20 --warmup 3 --packed 5 --band 18 0 512"
done
for ch in 1 2 4 8; do
  run "512 ev8M pk1 chunks$ch" "A=1" "--dims 512 512 200 --events 8000000 --steps 10 --warmup 2 --packed 1 --band 0 $ch 0"
done
