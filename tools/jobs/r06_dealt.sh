# A/B of the fused kernel's pair assignment (experiments flavour): 0 contiguous pieces, 1 pairs in turn, 2 pairs drawn
cd $GRAFT_REPO_ROOT
export DSI_ENGINE_EXPERIMENTS=1
B="python bench.py --no-cpu --no-host-fed --no-extra"
DSI_FUSED_INTERLEAVE=2 timeout 1200 python -m pytest tests/test_gpu_fused_vote.py tests/test_gpu_windows.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2; do
for il in 0 1 2; do
  DSI_FUSED_INTERLEAVE=$il $B --workload windows > gpurun_out/dl_win_${il}_$rep.json 2> gpurun_out/dl_win_${il}_$rep.err
  DSI_FUSED_INTERLEAVE=$il $B --workload windows --serial-windows > gpurun_out/dl_wins_${il}_$rep.json 2> gpurun_out/dl_wins_${il}_$rep.err
  DSI_FUSED_INTERLEAVE=$il $B --workload cameras4 > gpurun_out/dl_cam4_${il}_$rep.json 2> gpurun_out/dl_cam4_${il}_$rep.err
done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/dl_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, "ms/step %.4f kernel %.4f frac %.3f" % (d["ms_per_step"], r["kernel_avg_ms"], r["frac"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-400:])
PY
