cd $GRAFT_REPO_ROOT
export DSI_ENGINE_EXPERIMENTS=1
B="python bench.py --no-cpu --no-host-fed --no-extra"
for il in 0 1; do
  DSI_FUSED_INTERLEAVE=$il $B --workload cameras4 > gpurun_out/il_cam4_$il.json 2> gpurun_out/il_cam4_$il.err
  DSI_FUSED_INTERLEAVE=$il $B --workload windows > gpurun_out/il_win_$il.json 2> gpurun_out/il_win_$il.err
  DSI_FUSED_INTERLEAVE=$il $B --workload windows --serial-windows > gpurun_out/il_wins_$il.json 2> gpurun_out/il_wins_$il.err
done
DSI_FUSED_INTERLEAVE=1 timeout 900 python -m pytest tests/test_gpu_fused_vote.py -q -m gpu -x -k "four_cameras or fused_equals or three_cameras" 2>&1 | tail -2
unset DSI_ENGINE_EXPERIMENTS
for ch in 1 2 4; do
  $B --workload cameras4 --events 100000000 --tile 10 --steps 3 --warmup 1 --clock-ramp 1 --band 0 $ch 0 > gpurun_out/full_ch$ch.json 2> gpurun_out/full_ch$ch.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/il_*.json")+glob.glob("gpurun_out/full_ch*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, "ms/step %.4f kernel %.4f frac %.3f chunks %s" % (d["ms_per_step"], r["kernel_avg_ms"], r["frac"], d["config"]["chunks"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-400:])
PY
