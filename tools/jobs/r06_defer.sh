cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_fused_vote.py tests/test_gpu_windows.py -q -m gpu -x > gpurun_out/defer_tests.txt 2>&1
tail -3 gpurun_out/defer_tests.txt
B="python bench.py --workload windows --no-cpu --no-host-fed --no-extra"
for i in 1 2; do
DSI_ENGINE_EXPERIMENTS=1 DSI_FUSED_DEFER=0 $B > gpurun_out/win_nodefer_$i.json 2> gpurun_out/win_nodefer_$i.err
DSI_ENGINE_EXPERIMENTS=1 $B > gpurun_out/win_defer_$i.json 2> gpurun_out/win_defer_$i.err
DSI_ENGINE_EXPERIMENTS=1 DSI_FUSED_DEFER=0 $B --serial-windows > gpurun_out/wins_nodefer_$i.json 2> gpurun_out/wins_nodefer_$i.err
DSI_ENGINE_EXPERIMENTS=1 $B --serial-windows > gpurun_out/wins_defer_$i.json 2> gpurun_out/wins_defer_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/win*_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, "ms/step %.4f kernel %.4f frac %.3f" % (d["ms_per_step"], r["kernel_avg_ms"], r["frac"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-400:])
PY
