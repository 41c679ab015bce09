"""BASELINE configs[2] (512x512x200, 2 x 500 k events per 50 ms window, camera HM) with the exact tie resolver in the
loop: ms per window of WindowStream(fused_vote=True) (one kernel, no DSI written), of the unfused path the resolver
needs (camera DSIs written, fusion inside the arg-max) and of that path + resolveNearTies (exact_ties=True), windows
resident in HBM as packetised batches.  Run on the GPU box; under rocprofv3 --kernel-trace --stats for the split."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import dvs_mcemvs_amd as d
from dvs_mcemvs_amd import process as proc, synthetic as syn

NX, NY, NZ, EV, DUR, NW = 512, 512, 200, 500_000, 0.05, 8
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rig = syn.stereo_rig(NW * EV, width=640, height=480, t0=10.0, duration=NW * DUR, seed=77, n_points=6000)
ctx = d.Context(0)
shape = d.ShapeDSI(NX, NY, NZ, 4.0, 200.0, 0.0)
bounds = proc.window_bounds(rig["t0"], rig["t1"] + 1e-9, DUR, DUR)[:NW]
wins = []
for a, b in bounds:
    T_rv_w = proc.reference_view_process1(rig["trajectories"][0], b)
    per_cam = []
    for c in range(2):
        ev = proc.window_events(rig["events"][c], a, b)
        first, Rt = d.packetize(ev[2], rig["trajectories"][c], T_rv_w)
        per_cam.append(d.EventBatch(ctx, ev[0], ev[1], Rt, first))
    wins.append((per_cam, b))
out = {}
for name, kw in (("fused_vote", dict(fused_vote=True, materialize_fused=False)),
                 ("unfused", dict(materialize_fused=False)),
                 ("unfused_exact_ties", dict(materialize_fused=False, exact_ties=True))):
    ws = proc.WindowStream(ctx, (rig["cam"],) * 2, shape, d.FUSE_HM, depth=1, **kw)
    infos = []
    for r in range(reps + 8):
        if r == 8:
            ctx.synchronize()
            t = time.perf_counter()
        per_cam, ts = wins[r % NW]
        ws.fetch(ws.submit(None, rig["trajectories"], ts, batches=per_cam))
        if ws.last_resolve is not None and r >= 8:
            infos.append(ws.last_resolve)
    ctx.synchronize()
    out[name] = 1e3 * (time.perf_counter() - t) / reps
    if infos:
        print("resolver per window: elapsed_ms mean %.3f, columns %d..%d, voxels %d..%d, votes %d..%d, changed %d, premise_ok %s, widenings %d"
              % (np.mean([i["elapsed_ms"] for i in infos]), min(i["near_tie_pixels"] for i in infos),
                 max(i["near_tie_pixels"] for i in infos), min(i["candidate_voxels"] for i in infos),
                 max(i["candidate_voxels"] for i in infos), min(i["votes"] for i in infos), max(i["votes"] for i in infos),
                 sum(i["changed_pixels"] for i in infos), all(i["premise_ok"] for i in infos),
                 max(i["gap_widenings"] for i in infos)))
    ws.close()
print("ms per window (one stream, depth 1, fetch included): " + ", ".join("%s %.3f" % kv for kv in out.items()))
print("exact ties cost over the fused kernel: +%.3f ms per window" % (out["unfused_exact_ties"] - out["fused_vote"]))
