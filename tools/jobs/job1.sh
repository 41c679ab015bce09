set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lane_mappings or duplicate_events or hand_scheduled or fusion_ops" 2>&1 | tail -15
for pk in 1 5; do
  timeout 300 python bench.py --no-cpu --dims 512 512 200 --events 500000 --packed $pk --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('512 packed',$pk, d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['lds_atomics']['frac_of_conflict_free_peak'], d['config']['bands'], d['config']['band_rows'])"
  timeout 600 python bench.py --no-cpu --dims 1024 1024 256 --events 10000000 --packed $pk --steps 5 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1024 packed',$pk, d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['lds_atomics']['frac_of_conflict_free_peak'], d['config']['bands'], d['config']['band_rows'])"
  timeout 600 python bench.py --no-cpu --packed $pk --steps 10 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('346 packed',$pk, d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['lds_atomics']['frac_of_conflict_free_peak'], d['config']['bands'], d['config']['band_rows'])"
done
