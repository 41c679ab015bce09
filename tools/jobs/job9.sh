cd $GRAFT_REPO_ROOT
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu --workload windows $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms frac %.3f bands %d rows %d chunks %d pk %d' % (d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['config']['bands'], d['config']['band_rows'], d['config']['chunks'], d['config']['packed_lanes']))"
}
for lg in 2 3 4 5 6; do run "pk1 lg$lg" "DSI_PASS_LG=$lg" "--packed 1"; done
for lg in 3 4 5 6; do run "pk5 lg$lg" "DSI_PASS_LG=$lg" "--packed 5"; done
run "pk1 lg5 rows18" "DSI_PASS_LG=5" "--packed 1 --band 18 0 0"
run "pk1 lg4 rows18 b512" "DSI_PASS_LG=4" "--packed 1 --band 18 0 512"
run "pk1 lg5 rows18 b512" "DSI_PASS_LG=5" "--packed 1 --band 18 0 512"
run "pk5 lg5 rows18 b512" "DSI_PASS_LG=5" "--packed 5 --band 18 0 512"
run "pk5 lg6 rows18 b512" "DSI_PASS_LG=6" "--packed 5 --band 18 0 512"
