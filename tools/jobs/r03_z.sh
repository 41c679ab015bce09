#!/bin/bash
# what the driver runs at round end: smoke, the default bench line, the torchrun form at N=1
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -2
SECONDS=0; timeout 900 python bench.py > gpurun_out/r03_default_bench.json 2> gpurun_out/r03_default_bench.err; echo "bench rc=$?"
echo "wall ${SECONDS}s"
tail -1 gpurun_out/r03_default_bench.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('metric','value','unit','ms_per_step','n_gpus','steps','warmup','dtype','scaling','vs_baseline')})
print('roofline', d['roofline'])
print('cpu_baseline', d['cpu_baseline'])
print('parity', d.get('parity'))
for o in d.get('other_workloads', []): print('other', {k:o.get(k) for k in ('workload','ms_per_step','value','unit')})
print('step_ms', d.get('step_ms'))
"
