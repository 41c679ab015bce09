cd $GRAFT_REPO_ROOT
cat > /tmp/t1.py <<'PY'
import sys, numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    print("torch imported first", torch.__version__)
import dvs_mcemvs_amd as d
ctx = d.Context(0)
try:
    comms = d.Comm.create_all([ctx]); print("create_all ok", comms[0].size)
    g = d.Grid3D(ctx, 8, 8, 4); v = np.arange(256, dtype=np.float32).reshape(4, 8, 8); g.upload(v)
    d.allreduce_all(comms, [g], 0); print("allreduce_all ok", np.array_equal(g.download(), v))
    comms[0].close()
    c = d.Comm(ctx, d.Comm.unique_id(), 1, 0); g.allReduce(c, 0); print("rank comm ok", np.array_equal(g.download(), v)); c.close()
except Exception as e:
    print("FAILED", e)
import subprocess, os
print(subprocess.run("grep -i 'rccl\|amdhip' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid(), shell=True, capture_output=True, text=True).stdout)
PY
echo "--- no torch"; NCCL_DEBUG=WARN PYTHONPATH=$GRAFT_REPO_ROOT timeout 120 python /tmp/t1.py 2>&1 | tail -15
echo "--- torch first"; NCCL_DEBUG=WARN PYTHONPATH=$GRAFT_REPO_ROOT timeout 120 python /tmp/t1.py torch 2>&1 | tail -15
echo "--- windows bench after copy-stream fetch"
timeout 600 python bench.py --workload windows --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('windows: %.0f Mev/s, %.3f ms/window, %.0f windows/s, x%.0f real time; host-fed %s' % (d['value'], d['ms_per_step'], d['windows_per_s'], d['x_real_time'], d['host_fed']))"
timeout 900 python -m pytest tests/test_gpu_windows.py tests/test_gpu_parity.py -q -x -k "windows or full_sequence or depth_map or another_context or two_contexts" 2>&1 | tail -5
