"""The exact tie resolver (dsi_mapper_resolve_near_ties) at BASELINE configs[1]'s size, three calls in a row: the first
allocates the scratch it keeps, the others show the cost in a stream of steps.  Run on the GPU box, e.g. under
rocprofv3 --kernel-trace --stats for the per-kernel split (k_tie_columns, k_tie_contenders, k_tie_hits_binned, the sort, k_tie_sums2, k_tie_pick;
tools/trace_timeline.py prints the dispatches of one call)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import dvs_mcemvs_amd as d
from dvs_mcemvs_amd import synthetic as syn
nx, ny, nz, ev = 346, 260, 100, 10_000_000
rig = syn.stereo_rig(ev, width=nx, height=ny, t0=10.0, duration=0.5, seed=1234, n_points=5000)
ctx = d.Context(0)
shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(2)]
bs = []
for c in range(2):
    first, Rt = d.packetize(rig["events"][c][2], rig["trajectories"][c], rig["T_rv_w"])
    bs.append(d.EventBatch(ctx, rig["events"][c][0], rig["events"][c][1], Rt, first))
    ms[c].evaluateDSI_batch(bs[c])
fused = d.Grid3D(ctx, nx, ny, nz)
for rep in range(3):
    fused.setToFusionOf(ms[0].dsi_, ms[1].dsi_, d.FUSE_HM)
    ms[0].computeDepthMap(fused)
    ctx.synchronize()
    t = time.perf_counter()
    res = ms[0].resolveNearTies(ms, bs, d.FUSE_HM)
    print("resolver wall %.1f ms" % (1e3 * (time.perf_counter() - t)), res)
# the premise as a per-column proof (dsi_mapper_prove_near_ties): with the resolver's default gap, then with the gap it asks for
t = time.perf_counter()
proof = ms[0].proveNearTies(ms, bs, d.FUSE_HM)
print("proof wall %.1f ms" % (1e3 * (time.perf_counter() - t)), proof)
if proof["columns_unproven"]:
    gap = proof["gap_needed"] * 1.05
    fused.setToFusionOf(ms[0].dsi_, ms[1].dsi_, d.FUSE_HM)
    ms[0].computeDepthMap(fused)
    for rep in range(2):
        ctx.synchronize()
        t = time.perf_counter()
        res = ms[0].resolveNearTies(ms, bs, d.FUSE_HM, rel_gap=gap)
        print("resolver at the gap the proof needs: wall %.1f ms" % (1e3 * (time.perf_counter() - t)), res)
    print("proof again", ms[0].proveNearTies(ms, bs, d.FUSE_HM, rel_gap=gap))
