#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r03m_torchrun1.json 2> gpurun_out/r03m_torchrun1.err
echo "torchrun rc=$?"
tail -1 gpurun_out/r03m_torchrun1.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('frac_issued'), d.get('argmax_agree_frac'), d.get('near_tie_frac'))
print('parity', d['parity'])
print('step_ms', d['step_ms'])
print('others', {k:(v.get('ms_per_step'), (v.get('roofline') or {}).get('frac')) for k,v in (d.get('other_workloads') or {}).items()})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
tail -3 gpurun_out/r03m_torchrun1.err
timeout 600 python -m pytest tests/test_cpp_adapter.py -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
