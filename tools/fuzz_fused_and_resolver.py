"""Longer fuzz runs than the test suite affords (GPU box): (1) the fused vote -> fusion -> arg-max kernel -- including
its two-workgroups-per-CU variant, which small bands take -- against vote -> fuse -> collapse over random shapes, band
heights, lane mappings, ops and camera counts (the suite's test with more seeds); (2) the exact tie resolver against the
oracle's arg-max on random small problems.  Usage: python tools/fuzz_fused_and_resolver.py [first_seed] [n_seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dvs_mcemvs_amd as d  # noqa: E402
from dvs_mcemvs_amd import synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle_pipeline import OracleMapper  # noqa: E402
import test_gpu_fused_vote as tf  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 16
count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
ctx = d.Context(0)
bad = 0
for seed in range(first, first + count):
    try:
        tf.test_fused_fuzz_over_shapes_bands_and_mappings(ctx, seed)
    except AssertionError as e:
        bad += 1
        print("FUSED FUZZ FAILURE seed", seed, str(e)[:300])
print("fused fuzz: %d seeds, %d failures" % (count, bad))

bad2 = bad3 = 0
for seed in range(first, first + max(1, count // 5)):
    rng = np.random.default_rng(7000 + seed)
    nx, ny, nz = int(rng.integers(16, 160)), int(rng.integers(12, 120)), int(rng.integers(2, 48))
    n_cams = int(rng.integers(1, 3))
    op = int(rng.integers(1, 7)) if n_cams == 2 else 0
    n_ev = int(rng.integers(3_000, 120_000))
    rig = syn.stereo_rig(n_ev, width=nx, height=ny, duration=0.25, seed=seed, n_points=int(rng.integers(50, 1500)))
    shape = d.ShapeDSI(0, 0, nz, 4.0, float(rng.uniform(30, 200)), 0.0)
    batches, ms, dsis = [], [], []
    for c in range(n_cams):
        pk = d.packetize(rig["events"][c][2], rig["trajectories"][c], rig["T_rv_w"])
        first_, Rt = pk
        b = d.EventBatch(ctx, rig["events"][c][0], rig["events"][c][1], Rt, first_)
        m = d.MapperEMVS(ctx, rig["cam"], shape)
        m.evaluateDSI_batch(b)
        r = OracleMapper(rig["cam"], dimX=nx, dimY=ny, dimZ=nz, min_depth=4.0, max_depth=shape.max_depth_)
        assert r.evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
        batches.append(b); ms.append(m); dsis.append(r.dsi)
    out = d.MapperEMVS(ctx, rig["cam"], shape)
    if n_cams == 1:
        out.computeDepthMap(ms[0].dsi_)
        ref = dsis[0]
    else:
        out.computeDepthMapOfFusion(ms[0].dsi_, ms[1].dsi_, op)
        ref = orc.fuse2(dsis[0].copy(), dsis[1], op)
    info = out.resolveNearTies(ms, batches, op)
    _, _, idx = out.fetchDepthMap()
    if not np.array_equal(idx, ref.argmax(axis=0)):
        bad2 += 1
        print("RESOLVER FUZZ FAILURE seed", seed, (nx, ny, nz), n_cams, op, n_ev, int((idx != ref.argmax(axis=0)).sum()), info)
    # proven mode on the same problem (round 6): every column proven by bounds or re-summed on all planes, same map; and the
    # proof must never call a column proven whose resolved plane differs from the oracle's
    from dvs_mcemvs_amd import process as proc
    if n_cams == 1:
        out.computeDepthMap(ms[0].dsi_)
    else:
        out.computeDepthMapOfFusion(ms[0].dsi_, ms[1].dsi_, op)
    info_p, proof = proc.resolve_near_ties_proven(out, ms, batches, op, max_full_columns=10 ** 9)
    _, _, idx_p = out.fetchDepthMap()
    if proof["columns_unproven"] != proof["columns_resolved_fully"] or not np.array_equal(idx_p, ref.argmax(axis=0)):
        bad3 += 1
        print("PROVEN MODE FUZZ FAILURE seed", seed, (nx, ny, nz), n_cams, op, n_ev, int((idx_p != ref.argmax(axis=0)).sum()), proof)
    for o in ms + [out] + batches:
        o.close()
print("resolver fuzz: %d seeds, %d failures; proven mode: %d failures" % (max(1, count // 5), bad2, bad3))
sys.exit(1 if bad or bad2 or bad3 else 0)
