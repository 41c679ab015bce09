cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | grep "passed\|failed" | tail -2
for s in 1 2 4; do
  echo "== DSI_ROW_SUB=$s"
  DSI_ROW_SUB=$s timeout 300 python tools/run_stats.py 2>&1 | grep "run/accepted"
  DSI_ROW_SUB=$s timeout 300 python bench.py --no-cpu --no-host-fed --steps 30 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('stereo %.0f Mev/s step %.3f ms kern %.4f ms' % (d['value'], d['ms_per_step'], r['kernel_avg_ms']))"
  DSI_ROW_SUB=$s timeout 300 python bench.py --no-cpu --no-host-fed --workload windows --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('windows %.0f Mev/s step %.3f ms kern %.4f ms' % (d['value'], d['ms_per_step'], r['kernel_avg_ms']))"
  DSI_ROW_SUB=$s timeout 300 python bench.py --no-cpu --no-host-fed --dims 1024 1024 256 --steps 5 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('1024 %.0f Mev/s step %.3f ms kern %.4f ms' % (d['value'], d['ms_per_step'], r['kernel_avg_ms']))"
done
