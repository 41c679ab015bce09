cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "paired" -s > gpurun_out/paired_tests.txt 2>&1
tail -15 gpurun_out/paired_tests.txt
python bench.py --no-cpu --no-host-fed --no-sensitivity --steps 20 --warmup 5 > gpurun_out/paired_bench.json 2> gpurun_out/paired_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/paired_bench.json").read().strip().splitlines()[-1])
print("exact: ms/step %.3f kernel %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_avg_ms"], d["roofline"]["frac"]))
print("paired:", json.dumps(d.get("paired_mode")))
PY
tail -3 gpurun_out/paired_bench.err
