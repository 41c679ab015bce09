cd $GRAFT_REPO_ROOT
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu --no-host-fed $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['frac']))"
}
for lg in 6 5 4 3; do run "stereo lg$lg" "DSI_PASS_LG=$lg" "--steps 30 --warmup 3"; done
for lg in 4 3 2 1; do run "windows lg$lg" "DSI_PASS_LG=$lg" "--workload windows"; done
run "stereo default" "A=1" "--steps 30 --warmup 3"
run "windows default" "A=1" "--workload windows"
