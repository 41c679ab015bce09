"""configs[2] shape as a STREAM: 50 ms windows of 2 x 500 k events arrive in host memory; per window
upload + evaluate both cameras + harmonic fusion + arg-max + fetch the depth map.  Prints windows/s,
first window by window (fetch right after compute), then pipelined (window w+1 is uploaded and
queued before window w's depth map is fetched; two fused grids / extraction mappers alternate)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dvs_mcemvs_amd as d
from dvs_mcemvs_amd import synthetic as syn

nx, ny, nz, ev, nwin = 512, 512, 200, 500_000, 40
ctx = d.Context(0)
rig = syn.stereo_rig(ev * 4, width=nx, height=ny, duration=0.2, seed=7)
shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
mappers = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(2)]
fused = d.Grid3D(ctx, nx, ny, nz)
wins = []
for w in range(4):
    per_cam = []
    for c in range(2):
        x, y, ts = (a[w * ev:(w + 1) * ev] for a in rig["events"][c])
        first, Rt = d.packetize(ts, rig["trajectories"][c], rig["T_rv_w"])
        per_cam.append((x, y, Rt, first))
    wins.append(per_cam)


def window(w):
    bs = []
    for c in range(2):
        x, y, Rt, first = wins[w % 4][c]
        b = d.EventBatch(ctx, x, y, Rt, first)
        mappers[c].evaluateDSI_batch(b)
        bs.append(b)
    fused.setToFusionOf(mappers[0].dsi_, mappers[1].dsi_, d.FUSE_HM)
    mappers[0].computeDepthMap(fused)
    out = mappers[0].fetchDepthMap()
    for b in bs:
        b.close()
    return out


for w in range(4):
    window(w)
fused2 = [fused, d.Grid3D(ctx, nx, ny, nz)]
mf = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(2)]   # the reference's "mapper_fused"


def enqueue(w):
    bs = []
    for c in range(2):
        x, y, Rt, first = wins[w % 4][c]
        b = d.EventBatch(ctx, x, y, Rt, first)
        mappers[c].evaluateDSI_batch(b)
        bs.append(b)
    fused2[w % 2].setToFusionOf(mappers[0].dsi_, mappers[1].dsi_, d.FUSE_HM)
    mf[w % 2].computeDepthMap(fused2[w % 2])
    for b in bs:
        b.close()


enqueue(0)
t0 = time.perf_counter()
for w in range(1, nwin + 1):
    enqueue(w)
    mf[(w - 1) % 2].fetchDepthMap()
dtp = (time.perf_counter() - t0) / nwin
mf[nwin % 2].fetchDepthMap()
t0 = time.perf_counter()
for w in range(nwin):
    window(w)
dt = (time.perf_counter() - t0) / nwin
print("window by window: %.3f ms per 50 ms window = %.0f windows/s = %.0fx real time; %.0f Mevents/s incl. upload and depth-map fetch"
      % (dt * 1e3, 1 / dt, 0.05 / dt, 2 * (ev // 1024) * 1024 / dt / 1e6))
print("pipelined:        %.3f ms per 50 ms window = %.0f windows/s = %.0fx real time; %.0f Mevents/s"
      % (dtp * 1e3, 1 / dtp, 0.05 / dtp, 2 * (ev // 1024) * 1024 / dtp / 1e6))
