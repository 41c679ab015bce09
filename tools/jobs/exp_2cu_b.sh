cd $GRAFT_REPO_ROOT
export DSI_ENGINE_EXPERIMENTS=1
for env in "DSI_FUSED_2CU=1" "DSI_FUSED_2CU=0"; do
for args in "--serial-windows --band 17 0 0" "--serial-windows --band 17 0 0 --packed 3"; do
  env $env timeout 300 python bench.py --workload windows --no-cpu --no-host-fed --no-extra $args 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$env windows [$args]: %.4f ms/window  kernel %.4f ms  frac %.3f  bands %d rows %d lds %d' % (d['ms_per_step'], r['kernel_avg_ms'], r['frac'], d['config']['bands'], d['config']['band_rows'], d['config']['lds_bytes']))"
done; done
