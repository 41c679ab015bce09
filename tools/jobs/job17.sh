cd $GRAFT_REPO_ROOT
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu --no-host-fed $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'step %.3f ms kern %.4f ms' % (d['ms_per_step'], r['kernel_avg_ms']))"
}
for E in 0 1 2; do
  run "windows experiment=$E" "DSI_EXPERIMENT=$E" "--workload windows"
  run "windows nonpersistent experiment=$E" "DSI_EXPERIMENT=$E DSI_PERSISTENT=0" "--workload windows"
done
run "1024 experiment=1" "DSI_EXPERIMENT=1" "--dims 1024 1024 256 --steps 5 --warmup 1"
run "1024 experiment=2" "DSI_EXPERIMENT=2" "--dims 1024 1024 256 --steps 5 --warmup 1"
run "stereo persistent experiment=1" "DSI_EXPERIMENT=1 DSI_PERSISTENT=1" ""
run "stereo experiment=1" "DSI_EXPERIMENT=1" ""
