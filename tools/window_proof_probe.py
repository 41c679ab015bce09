"""One 50 ms window at BASELINE configs[2] shape (512 x 512 x 200, 2 x 500 k events): the exact tie resolver, then its premise\nas a per-column proof (dsi_mapper_prove_near_ties) and proven mode (process.resolve_near_ties_proven).  Run on the GPU box."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import dvs_mcemvs_amd as d
from dvs_mcemvs_amd import synthetic as syn, process as proc
NX, NY, NZ, EV, DUR = 512, 512, 200, 500_000, 0.05
rig = syn.stereo_rig(EV, width=640, height=480, t0=10.0, duration=DUR, seed=77, n_points=6000)
ctx = d.Context(0)
shape = d.ShapeDSI(NX, NY, NZ, 4.0, 200.0, 0.0)
ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(2)]
bs = []
for c in range(2):
    first, Rt = d.packetize(rig["events"][c][2], rig["trajectories"][c], rig["T_rv_w"])
    bs.append(d.EventBatch(ctx, rig["events"][c][0], rig["events"][c][1], Rt, first))
    ms[c].evaluateDSI_batch(bs[c])
out = d.MapperEMVS(ctx, rig["cam"], shape)
out.computeDepthMapOfFusion(ms[0].dsi_, ms[1].dsi_, d.FUSE_HM)
res = out.resolveNearTies(ms, bs, d.FUSE_HM)
print("resolver", {k: res[k] for k in ("near_tie_pixels", "candidate_voxels", "votes", "elapsed_ms", "rel_gap")})
p = out.proveNearTies(ms, bs, d.FUSE_HM, rel_gap=res["rel_gap"])
print("proof", p)
out.computeDepthMapOfFusion(ms[0].dsi_, ms[1].dsi_, d.FUSE_HM)
info, proof = proc.resolve_near_ties_proven(out, ms, bs, d.FUSE_HM)
print("proven mode", {k: info[k] for k in ("near_tie_pixels", "candidate_voxels", "elapsed_ms", "rel_gap")}, proof)
