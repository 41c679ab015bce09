cd $GRAFT_REPO_ROOT
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu --no-host-fed $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms frac %.3f bands %d rows %d chunks %d pk %d' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['frac'], d['config']['bands'], d['config']['band_rows'], d['config']['chunks'], d['config']['packed_lanes']))"
}
for ev in 2500000 5000000 10000000 20000000 40000000; do run "events $ev" "A=1" "--events $ev --steps 10 --warmup 2"; done
