cd $GRAFT_REPO_ROOT
run() { # tag, env, args
  env $2 timeout 600 python bench.py --no-cpu --no-host-fed $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '%.0f Mev/s step %.3f ms kern %.4f ms frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['frac']))"
}
for i in 1 2; do
run "cameras4 prefetch" "A=1" "--workload cameras4"
run "cameras4 no prefetch" "DSI_EXPERIMENT=7" "--workload cameras4"
run "1024 1M prefetch" "A=1" "--dims 1024 1024 256 --events 1000000 --steps 10 --warmup 2"
run "1024 1M no prefetch" "DSI_EXPERIMENT=7" "--dims 1024 1024 256 --events 1000000 --steps 10 --warmup 2"
run "windows pk5 prefetch" "A=1" "--workload windows --packed 5"
run "windows pk5 no prefetch" "DSI_EXPERIMENT=7" "--workload windows --packed 5"
done
