# the -m gpu suite, smoke(), then the default bench line exactly as the driver runs it
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/suite.txt 2>&1
grep -n "passed\|failed\|error" gpurun_out/suite.txt | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "NCCL WARN" | tail -1
( time timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_line.json 2> gpurun_out/bench.err ) 2>&1 | grep real
tail -c 600 gpurun_out/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.0f Mev/s  step %.3f ms  kernel %.4f ms  frac %.3f  frac_issued %.3f" % (d["value"], d["ms_per_step"], r["kernel_avg_ms"], r["frac"], r["frac_issued"]))
for c in (d.get("sensitivity") or {}).get("cases", []):
    print("  sensitivity %-16s %.4f ms frac %.3f issued %.3f merge %.3f prep %.1f s" % (c["case"], c["kernel_avg_ms"], c["frac"], c["frac_issued"], c["records_per_accepted_event_plane"], c["prepare_s"]))
for k, v in (d.get("other_workloads") or {}).items():
    print("  other", k, v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"), v.get("error"))
print("  parity", d.get("argmax_agree_frac"), d.get("near_tie_frac"))
PY
