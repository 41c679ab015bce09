"""Where a phase of the fused vote -> fusion -> arg-max kernel spends its time (development aid).

Runs one 512x512x200 window (2 x 500 k events) through dsi_mapper_depth_map_of_events with the test hook
that stamps every (workgroup, phase, wave) with the 100 MHz clock at: stream begins / stream ends / after
barrier + read-back + clear / after the closing barrier, and prints the averages.
Usage (GPU box): python tools/fused_trace.py [events_per_window] [packed] [fixed cost of a phase in records] [pass_lg]
"""
import ctypes as C
import os
import sys
os.environ["DSI_ENGINE_EXPERIMENTS"] = "1"   # the hooks used below exist only in the experiments flavour (build.py --experiments)

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dvs_mcemvs_amd as d  # noqa: E402
from dvs_mcemvs_amd import process as proc, synthetic as syn  # noqa: E402

ev_win = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
packed = int(sys.argv[2]) if len(sys.argv) > 2 else -1
NX, NY, NZ = 512, 512, 200
ctx = d.Context(0)
rig = syn.stereo_rig(2 * ev_win, width=640, height=480, t0=10.0, duration=2 * ev_win / 10e6, seed=77, n_points=6000)
shape = d.ShapeDSI(NX, NY, NZ, 4.0, 200.0, 0.0)
ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(2)]
for m in ms:
    m.set_packed_lanes(packed)
a, b = proc.window_bounds(rig["t0"], rig["t1"] + 1e-9, ev_win / 10e6, ev_win / 10e6)[0]
T = proc.reference_view_process1(rig["trajectories"][0], b)
batches = []
for c in range(2):
    ev = proc.window_events(rig["events"][c], a, b)
    first, Rt = d.packetize(ev[2], rig["trajectories"][c], T)
    batches.append(d.EventBatch(ctx, ev[0], ev[1], Rt, first))
L = d.load_library()
for _ in range(3):
    ms[0].computeDepthMapOfEvents(ms, batches, d.FUSE_HM)
ctx.synchronize()
n = C.c_size_t()
if len(sys.argv) > 3:
    L.dsi_test_fused_fixed_cost.argtypes = [C.c_void_p, C.c_int]
    assert L.dsi_test_fused_fixed_cost(ms[0]._h, int(sys.argv[3])) == 0
if len(sys.argv) > 4:
    L.dsi_test_pass_lg.argtypes = [C.c_void_p, C.c_int]
    for m in ms:
        assert L.dsi_test_pass_lg(m._h, int(sys.argv[4])) == 0
L.dsi_test_fused_trace_enable.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
L.dsi_test_fused_trace_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert L.dsi_test_fused_trace_enable(ms[0]._h, C.byref(n)) == 0
ms[0].computeDepthMapOfEvents(ms, batches, d.FUSE_HM)
buf = np.zeros(n.value, np.uint64)
assert L.dsi_test_fused_trace_read(ms[0]._h, buf.ctypes.data, n.value) == 0
t = buf.reshape(-1, 64, 16, 4).astype(np.float64) * 0.01          # microseconds
valid = t[..., 3] > 0
wg_valid = valid.any(axis=(1, 2))
print("workgroups with work: %d, phases per workgroup: %.1f" % (wg_valid.sum(), valid[:, :, 0].sum(1)[wg_valid].mean()))
t0 = np.where(valid, t[..., 0], np.inf).min()
t1 = t[..., 3].max()
print("kernel span (first stamp -> last stamp): %.1f us" % (t1 - t0))
ends = t[..., 3].max(axis=(1, 2))[wg_valid] - t0
print("workgroup finish times: min %.1f  median %.1f  max %.1f us" % (ends.min(), np.median(ends), ends.max()))
ph = []
for w in np.nonzero(wg_valid)[0]:
    for p in range(64):
        if not valid[w, p].all():
            continue
        s = t[w, p]
        start = s[:, 0].min()
        ph.append((s[:, 0].max() - start,                 # skew of the stream starts
                   s[:, 1].min() - start,                 # first wave done
                   np.median(s[:, 1]) - start,
                   s[:, 1].mean() - start,                # mean wave done = the work per wave, pass switches included
                   s[:, 1].max() - start,                 # last wave done
                   s[:, 2].max() - s[:, 1].max(),         # barrier + read-back
                   s[:, 3].max() - s[:, 2].max(),         # closing barrier
                   s[:, 3].max() - start))
ph = np.array(ph)
par = []          # (phases alternate camera 0 / camera 1 with two cameras: the same table per camera)
for w in np.nonzero(wg_valid)[0]:
    for p in range(64):
        if valid[w, p].all():
            par.append(p & 1)
par = np.array(par)
names = ["start skew", "first wave done", "median wave done", "mean wave done", "last wave done", "barrier+consume", "closing barrier", "phase total"]
for k, nm in enumerate(names):
    print("%-18s mean %7.2f us   p10 %7.2f   p90 %7.2f" % (nm, ph[:, k].mean(), np.percentile(ph[:, k], 10), np.percentile(ph[:, k], 90)))
print("phases: %d; sum of phase totals / workgroups = %.1f us" % (len(ph), ph[:, 7].sum() / wg_valid.sum()))
for c in (0, 1):
    sel = par == c
    print("camera %d phases: " % c + ", ".join("%s %.2f" % (nm, ph[sel, k].mean()) for k, nm in enumerate(names)))
fin = np.where(wg_valid, t[..., 3].max(axis=(1, 2)) - t0, np.nan)
sta = np.where(wg_valid, np.where(valid, t[..., 0], np.inf).min(axis=(1, 2)) - t0, np.nan)
for x in range(8):
    sel = np.arange(fin.shape[0]) % 8 == x
    print("xcd %d: workgroups start %.1f..%.1f us, finish min %.1f mean %.1f max %.1f us" %
          (x, np.nanmin(sta[sel]), np.nanmax(sta[sel]), np.nanmin(fin[sel]), np.nanmean(fin[sel]), np.nanmax(fin[sel])))
# the slowest and the fastest workgroups: phases, mean stream time, first / last stamp
order = np.argsort(-np.where(wg_valid, t[..., 3].max(axis=(1, 2)), 0))
for w in list(order[:6]) + list(order[wg_valid.sum() - 3:wg_valid.sum()]):
    nph = int(valid[w, :, 0].sum())
    s_ = t[w, :nph]
    dur = s_[:, :, 3].max(axis=1) - s_[:, :, 0].min(axis=1)
    stream = s_[:, :, 1].max(axis=1) - s_[:, :, 0].min(axis=1)
    print("wg %3d (xcd %d, slot %2d): %2d phases, %.1f -> %.1f us, phase mean %.1f (stream %.1f) max %.1f us"
          % (w, w & 7, w >> 3, nph, s_[..., 0].min() - t0, s_[..., 3].max() - t0, dur.mean(), stream.mean(), dur.max()))
for o in ms + batches:
    o.close()
ctx.close()
