# round 6 item 2a: the vector fill's inline cuts -- tests, then A/B bench lines
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "inline_cuts or hand_scheduled or both_lane or baseline_large or configs4_camera" > gpurun_out/inline_tests.txt 2>&1
tail -3 gpurun_out/inline_tests.txt
B="python bench.py --no-cpu --no-host-fed --no-extra --no-sensitivity"
for ic in 1000000000 0; do
  $B --workload cameras4 --events 100000000 --tile 10 --steps 3 --warmup 1 --clock-ramp 1 --inline-cuts $ic > gpurun_out/full_ic$ic.json 2> gpurun_out/full_ic$ic.err
  $B --dims 1024 1024 256 --events 10000000 --steps 10 --warmup 2 --inline-cuts $ic > gpurun_out/d1024_ic$ic.json 2> gpurun_out/d1024_ic$ic.err
  $B --workload cameras4 --inline-cuts $ic > gpurun_out/cam4_ic$ic.json 2> gpurun_out/cam4_ic$ic.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/*_ic*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, "ms/step %.3f kernel %.4f frac %.3f" % (d["ms_per_step"], r["kernel_avg_ms"], r["frac"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-400:])
PY
