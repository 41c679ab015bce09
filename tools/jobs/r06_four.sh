cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_fused_vote.py -q -m gpu -x -k "four_cameras or three_cameras or fused_equals" > gpurun_out/four_tests.txt 2>&1
tail -5 gpurun_out/four_tests.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "paired" > gpurun_out/paired_tests2.txt 2>&1
tail -3 gpurun_out/paired_tests2.txt
B="python bench.py --workload cameras4 --no-cpu --no-host-fed --no-extra"
$B > gpurun_out/cam4_unfused.json 2> gpurun_out/cam4_unfused.err
$B --fused-vote > gpurun_out/cam4_fused.json 2> gpurun_out/cam4_fused.err
$B --fused-vote --packed 1 > gpurun_out/cam4_fused_p1.json 2> gpurun_out/cam4_fused_p1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/cam4_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, "ms/step %.4f kernel %.4f frac %.3f bands %s rows %s" % (d["ms_per_step"], r["kernel_avg_ms"], r["frac"], d["config"]["bands"], d["config"]["band_rows"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
