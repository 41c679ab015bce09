cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fused_vote.py -x -q 2>&1 | tail -3
for args in "" "--band 17 0 0" "--band 12 0 0" "--serial-windows" "--serial-windows --band 17 0 0"; do
  timeout 300 python bench.py --workload windows --no-cpu --no-host-fed --no-extra $args 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('windows [$args]: %.4f ms/window  kernel %.4f ms  frac %.3f  bands %d rows %d lds %d' % (d['ms_per_step'], r['kernel_avg_ms'], r['frac'], d['config']['bands'], d['config']['band_rows'], d['config']['lds_bytes']))"
done
