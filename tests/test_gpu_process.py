"""process_1 / process_2 orchestration (process1.cpp, process2.cpp) on the GPU engine vs the
same orchestration assembled from oracle pieces."""
import numpy as np
import pytest

import dvs_mcemvs_amd as d
from dvs_mcemvs_amd import process, synthetic as syn
from oracle_pipeline import OracleMapper, argmax_report, oracle_process_1, oracle_process_2

pytestmark = pytest.mark.gpu
TOL = 2e-4


def close(got, ref, tol=TOL):
    err = np.abs(got.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= tol, "max rel err %g" % err.max()


@pytest.mark.parametrize("fusion_method", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("n_cams", [2, 3])
def test_process_1(ctx, fusion_method, n_cams):
    rig = syn.stereo_rig(12000, width=72, height=54, duration=0.25, n_cams=n_cams, baseline=0.4, seed=31)
    shape = d.ShapeDSI(0, 0, 16, 4.0, 150.0, 0.0)
    mappers = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(n_cams)]
    fused = d.MapperEMVS(ctx, rig["cam"], shape)
    ts = rig["t0"] + 0.2
    process.process_1(mappers, rig["events"], rig["trajectories"], fused, ts, fusion_method, rv_pos=0.1)
    dsis, ref = oracle_process_1(lambda: OracleMapper(rig["cam"], dimZ=16, min_depth=4.0, max_depth=150.0),
                                 rig["events"], rig["trajectories"], ts, fusion_method, rv_pos=0.1)
    for m, r in zip(mappers, dsis):
        close(m.dsi_.download(), r, 1e-4)
    close(fused.dsi_.download(), ref)
    for m in mappers + [fused]:
        m.close()


def test_process_1_bad_method(ctx):
    rig = syn.stereo_rig(3000, width=40, height=30, duration=0.2, seed=2)
    shape = d.ShapeDSI(0, 0, 6, 4.0, 150.0, 0.0)
    ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(3)]
    with pytest.raises(d.DsiError) as e:
        process.process_1(ms[:2], rig["events"], rig["trajectories"], ms[2], rig["t0"] + 0.1, 9)
    assert e.value.code == 5


@pytest.mark.parametrize("stereo_fusion,temporal_fusion", [(2, 2), (2, 4), (3, 2), (4, 4), (1, 2), (5, 3)])
def test_process_2(ctx, stereo_fusion, temporal_fusion):
    rig = syn.stereo_rig(18000, width=64, height=48, duration=0.3, seed=41)
    shape = d.ShapeDSI(0, 0, 12, 4.0, 150.0, 0.0)
    fused = d.MapperEMVS(ctx, rig["cam"], shape)
    cam_time = d.MapperEMVS(ctx, rig["cam"], shape)
    ts = rig["t0"] + 0.15
    out = process.process_2(ctx, [rig["cam"]] * 2, shape, rig["events"], rig["trajectories"], 4, fused,
                            cam_time, ts, stereo_fusion, temporal_fusion)
    ref = oracle_process_2(lambda: OracleMapper(rig["cam"], dimZ=12, min_depth=4.0, max_depth=150.0),
                           rig["events"], rig["trajectories"], 4, ts, stereo_fusion, temporal_fusion)
    close(out["left"].download(), ref["left"])
    close(out["right"].download(), ref["right"])
    close(fused.dsi_.download(), ref["fused"])
    close(cam_time.dsi_.download(), ref["camera_time"], 4e-4)
    fused.close()
    cam_time.close()


def test_process_5_shuffled_right_camera(ctx):
    """process5.cpp: right-camera sub-intervals start at n/2 and wrap around the event vector."""
    rig = syn.stereo_rig(18000 + 37, width=64, height=48, duration=0.3, seed=43)   # remainder != 0
    shape = d.ShapeDSI(0, 0, 12, 4.0, 150.0, 0.0)
    fused = d.MapperEMVS(ctx, rig["cam"], shape)
    cam_time = d.MapperEMVS(ctx, rig["cam"], shape)
    ts = rig["t0"] + 0.15
    out = process.process_5(ctx, [rig["cam"]] * 2, shape, rig["events"], rig["trajectories"], 4, fused,
                            cam_time, ts, 2, 2)
    ref = oracle_process_2(lambda: OracleMapper(rig["cam"], dimZ=12, min_depth=4.0, max_depth=150.0),
                           rig["events"], rig["trajectories"], 4, ts, 2, 2, shuffle_right=True)
    plain = oracle_process_2(lambda: OracleMapper(rig["cam"], dimZ=12, min_depth=4.0, max_depth=150.0),
                             rig["events"], rig["trajectories"], 4, ts, 2, 2)
    close(out["right"].download(), ref["right"])
    close(fused.dsi_.download(), ref["fused"])
    assert np.abs(ref["fused"] - plain["fused"]).max() > 1e-3   # the shuffle really changes the result
    sel = process.shuffled_subintervals(18037, 4)
    assert sel[0][0] == 2 * (18037 // 4) and len(sel[1]) == 18037 // 4
    assert any(s[-1] < s[0] for s in sel)                       # one sub-interval wraps: tail then head
    fused.close()
    cam_time.close()


def test_configs3_eight_time_slices_at_full_size(ctx):
    """BASELINE.json configs[3] at FULL size on one GPU: "Stereo DSEC, 8 independent time-slice DSIs" -- 10 M
    events per camera cut into 8 sub-intervals of 1.25 M by event count (process2.cpp:46-47), per slice the two
    camera DSIs and their harmonic mean, temporal harmonic mean over the slices (Alg. 2, process2.cpp:98-249),
    arg-max.  Against the oracle's process_2: every voxel of the left / right / fused temporal DSIs and of the
    converse (camera fusion of the temporal DSIs) within tolerance, and for EVERY pixel the GPU's plane is the
    oracle's or a provable near-tie.  (The multi-rank tests shard exactly these slices over ranks, with smaller
    slices; the sums they all-reduce are these accumulators.)"""
    rig = syn.stereo_rig(10_000_000, seed=1234)
    shape = d.ShapeDSI(0, 0, 100, 4.0, 200.0, 0.0)
    fused = d.MapperEMVS(ctx, rig["cam"], shape)
    cam_time = d.MapperEMVS(ctx, rig["cam"], shape)
    ts = rig["t1"]
    out = process.process_2(ctx, [rig["cam"]] * 2, shape, rig["events"], rig["trajectories"], 8, fused, cam_time, ts, 2, 2)
    ref = oracle_process_2(lambda: OracleMapper(rig["cam"], dimZ=100, min_depth=4.0, max_depth=200.0),
                           rig["events"], rig["trajectories"], 8, ts, 2, 2)
    close(out["left"].download(), ref["left"])
    close(out["right"].download(), ref["right"])
    got = fused.dsi_.download()
    close(got, ref["fused"])
    close(cam_time.dsi_.download(), ref["camera_time"], 4e-4)
    assert ref["fused"].max() > 1.0
    fused.computeDepthMap()
    depth, conf, idx = fused.fetchDepthMap()
    rep = argmax_report(idx, ref["fused"], TOL)
    print("configs[3] full size: %r" % rep)
    assert rep["violations"] == 0 and rep["argmax_agree_frac"] > 0.99, rep
    assert np.array_equal(depth, fused.raw_depths_vec_[idx])
    # ... and through the resolver's building blocks (process.exact_depth_map_process_2: near-tie columns of the final
    # fused DSI, their voxels of all 16 (slice, camera) DSIs re-summed in the reference's order, process_2's scalar
    # ops on the host) the plane index map IS the oracle's on all 89,960 pixels
    exact = d.MapperEMVS(ctx, rig["cam"], shape)
    info = process.exact_depth_map_process_2(ctx, [rig["cam"]] * 2, shape, rig["events"], rig["trajectories"], 8, exact, ts, 2, 2,
                                             prove=True)
    # (prove=True, ABI 10: the resolution's premise proven column by column through interval grids -- all 16 (slice, camera)
    #  DSIs' votes counted, their bounds carried through the harmonic camera fusion and the harmonic temporal fusion)
    proof = info.pop("proof")
    print("configs[3] proof: %r" % (proof,))
    assert proof["columns"] == 346 * 260 and proof["columns_unproven"] == proof["columns_resolved_fully"], proof
    assert proof["columns_proven"] > 0.99 * 346 * 260 and proof["max_votes"] > 100
    depth2, conf2, idx2 = exact.fetchDepthMap()
    ridx = ref["fused"].argmax(axis=0)
    print("configs[3] exact: %r; %d pixels differed before" % (info, int((idx != ridx).sum())))
    assert np.array_equal(exact.dsi_.download(), got)                   # the same fused DSI as process_2's
    assert np.array_equal(idx2, ridx), "%d pixels differ" % (idx2 != ridx).sum()
    assert np.array_equal(depth2, fused.raw_depths_vec_[ridx])
    assert info["changed_pixels"] == int((idx != ridx).sum()) and 8 * info["max_order_diff"] < 2.5e-4
    exact.close()
    fused.close()
    cam_time.close()


@pytest.mark.parametrize("stereo_fusion,temporal_fusion", [(2, 2), (3, 4), (6, 2), (4, 3)])
def test_exact_depth_map_process_2_small(ctx, stereo_fusion, temporal_fusion):
    """Alg. 2 with few events: most columns tie exactly or nearly; the resolved index map equals the oracle's
    process_2 + first-maximum arg-max on every pixel (temporal_fusion 3 does nothing in the reference: zero DSI)."""
    rig = syn.stereo_rig(30_000, width=64, height=48, duration=0.3, seed=47, n_points=500)
    shape = d.ShapeDSI(0, 0, 14, 4.0, 150.0, 0.0)
    exact = d.MapperEMVS(ctx, rig["cam"], shape)
    ts = rig["t0"] + 0.15
    info = process.exact_depth_map_process_2(ctx, [rig["cam"]] * 2, shape, rig["events"], rig["trajectories"], 4, exact, ts,
                                             stereo_fusion, temporal_fusion)
    ref = oracle_process_2(lambda: OracleMapper(rig["cam"], dimZ=14, min_depth=4.0, max_depth=150.0),
                           rig["events"], rig["trajectories"], 4, ts, stereo_fusion, temporal_fusion)
    depth, conf, idx = exact.fetchDepthMap()
    assert np.array_equal(idx, ref["fused"].argmax(axis=0)), info
    close(exact.dsi_.download(), ref["fused"])
    if temporal_fusion in (2, 4):
        assert info["near_tie_pixels"] > 0
    exact.close()


@pytest.mark.parametrize("stereo_fusion,temporal_fusion,events", [(2, 2, 30_000), (3, 4, 30_000), (6, 2, 30_000), (2, 2, 400_000),
                                                                  (1, 4, 400_000)])
def test_process_2_proven_through_interval_grids(ctx, stereo_fusion, temporal_fusion, events):
    """Alg. 2's resolution with its premise PROVEN (ABI 10, interval grids): every (sub-interval, camera) DSI's votes counted
    and bounded, the bounds carried through the camera fusion and the temporal accumulate / finalize like the values, every
    column either proven or re-summed on all planes; the map is the oracle's process_2 arg-max.  With a gap of rounding size
    the same bounds leave the contested columns unproven: they are not vacuous."""
    rig = syn.stereo_rig(events, width=64, height=48, duration=0.3, seed=47, n_points=500)
    shape = d.ShapeDSI(0, 0, 14, 4.0, 150.0, 0.0)
    exact = d.MapperEMVS(ctx, rig["cam"], shape)
    ts = rig["t0"] + 0.15
    info = process.exact_depth_map_process_2(ctx, [rig["cam"]] * 2, shape, rig["events"], rig["trajectories"], 4, exact, ts,
                                             stereo_fusion, temporal_fusion, prove=True)
    proof = info["proof"]
    assert proof["columns"] == 64 * 48 and proof["columns_proven"] + proof["columns_unproven"] == 64 * 48
    assert proof["columns_unproven"] == proof["columns_resolved_fully"], proof
    assert proof["max_votes"] > 5
    ref = oracle_process_2(lambda: OracleMapper(rig["cam"], dimZ=14, min_depth=4.0, max_depth=150.0),
                           rig["events"], rig["trajectories"], 4, ts, stereo_fusion, temporal_fusion)
    depth, conf, idx = exact.fetchDepthMap()
    assert np.array_equal(idx, ref["fused"].argmax(axis=0)), proof
    tiny = process.exact_depth_map_process_2(ctx, [rig["cam"]] * 2, shape, rig["events"], rig["trajectories"], 4, exact, ts,
                                             stereo_fusion, temporal_fusion, rel_gap=1e-7, prove=True)["proof"]
    assert tiny["columns_unproven"] >= proof["columns_unproven"] - proof["columns_resolved_fully"]
    if events >= 100_000:
        # voxels with hundreds of votes: bounds of 1e-5, which a gap of 1e-7 cannot cover -- and most columns settled by bounds
        assert tiny["columns_unproven"] > 0 and tiny["gap_needed"] > 1e-7, tiny
        assert proof["columns_proven"] > 0.5 * 64 * 48, proof
    exact.close()


def test_resolver_building_blocks(ctx):
    """dsi_mapper_exact_voxels on arbitrary voxels (any order, duplicates) = the oracle's DSI at those voxels, bit for
    bit; the host reference ops = the device's; dsi_mapper_patch_depth_map writes what it is told."""
    from dvs_mcemvs_amd import engine
    from oracle import oracle as orc
    nx, ny, nz = 80, 60, 20
    rig = syn.stereo_rig(40_000, width=nx, height=ny, duration=0.25, seed=19, n_points=600)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    first, Rt = d.packetize(rig["events"][0][2], rig["trajectories"][0], rig["T_rv_w"])
    batch = d.EventBatch(ctx, rig["events"][0][0], rig["events"][0][1], Rt, first)
    m = d.MapperEMVS(ctx, rig["cam"], shape)
    r = OracleMapper(rig["cam"], dimZ=nz, min_depth=4.0, max_depth=200.0)
    assert r.evaluateDSI(rig["events"][0], rig["trajectories"][0], rig["T_rv_w"])
    rng = np.random.default_rng(3)
    vox = rng.integers(0, nx * ny * nz, 5000).astype(np.uint32)
    vox[:50] = vox[50:100]                                   # duplicates
    top = np.argsort(r.dsi.reshape(-1))[-200:].astype(np.uint32)   # the most-voted voxels
    vox = np.concatenate([vox, top])
    values, votes = m.exactVoxels(batch, vox)               # (the mapper's grid was never evaluated: not needed)
    assert np.array_equal(values, r.dsi.reshape(-1)[vox])
    assert votes[-1] > 10 and (votes[values == 0] == 0).all()
    a = rng.uniform(0, 50, 4096).astype(np.float32)
    g = rng.uniform(0, 50, 4096).astype(np.float32)
    a[:8] = 0
    for op in range(1, 7):
        assert np.array_equal(engine.reference_fuse2(op, a, g), orc.fuse2(a.copy(), g, op))
    for mode in (0, 1):
        acc = engine.reference_accumulate(mode, a, g)
        assert np.array_equal(acc, orc.accumulate(a.copy(), g, mode))
        assert np.array_equal(engine.reference_finalize(mode, acc, 5), orc.finalize(acc.copy(), mode, 5))
    m.evaluateDSI_batch(batch)
    m.computeDepthMap()
    depth0, conf0, idx0 = m.fetchDepthMap()
    m.patchDepthMap([5, 7 * nx + 3], [4, 19], [1.5, 2.5])
    depth1, conf1, idx1 = m.fetchDepthMap()
    assert idx1.reshape(-1)[5] == 4 and idx1[7, 3] == 19 and conf1[7, 3] == 2.5
    assert depth1[7, 3] == m.raw_depths_vec_[19]
    idx1.reshape(-1)[[5, 7 * nx + 3]] = idx0.reshape(-1)[[5, 7 * nx + 3]]
    assert np.array_equal(idx1, idx0)
    with pytest.raises(d.DsiError):
        m.patchDepthMap([nx * ny], [0], [1.0])
    with pytest.raises(d.DsiError):
        m.exactVoxels(batch, [nx * ny * nz])
    m.close()
    batch.close()


def test_exact_voxels_on_degenerate_plane_transfers(ctx):
    """The inverted event pass of the resolver (k_tie_hits_binned) asks, per packet and voxel, which z0 locations the plane
    transfer can take into the voxel's neighbourhood -- the pre-image of an affine map.  Where the map degenerates the
    pass must fall back, not guess: camera centre ON a plane (a = 0: every event lands on one location), z0 - Cz = 0
    (d = 0: no finite coordinate), a centre between planes (negative slope on some), far behind everything, NaN.  EVERY
    voxel of the DSI, re-summed in the reference's order, equals the oracle's DSI bit for bit."""
    nx, ny, nz = 40, 30, 6
    cam = (nx, ny, 30.0, 30.0, 20.0, 15.0)
    shape = d.ShapeDSI(0, 0, nz, 1.0, 4.0, 0.0)
    m = d.MapperEMVS(ctx, cam, shape)
    planes = m.raw_depths_vec_
    centers = np.array([(0, 0, 0), (0.2, -0.1, planes[2]), (0.1, 0.1, planes[0]), (0.0, 0.0, planes[3] + 0.01),
                        (5.0, -7.0, 100.0), (np.nan, 0.0, 0.0), (0.3, 0.2, -0.5), (-0.4, 0.1, 0.2)], np.float32)
    npk = centers.shape[0]
    Rt = np.zeros((npk, 12), np.float32)
    Rt[:, 0] = Rt[:, 4] = Rt[:, 8] = 1.0
    Rt[:, 9:12] = -centers                      # C = -R^T t (mapper_emvs_stereo.cpp:108)
    rng = np.random.default_rng(5)
    x = rng.integers(0, nx, npk * 1024).astype(np.uint16)
    y = rng.integers(0, ny, npk * 1024).astype(np.uint16)
    x[:64] = 7                                  # a burst on one pixel: more hits per (voxel, packet) than a wave queues
    y[:64] = 9
    batch = d.EventBatch(ctx, x, y, Rt)
    r = OracleMapper(cam, dimZ=nz, min_depth=1.0, max_depth=4.0)
    first = np.arange(npk, dtype=np.uint32) * 1024
    r.evaluate_packets(x, y, first, Rt)
    ref = r.dsi
    vox = np.arange(nx * ny * nz, dtype=np.uint32)
    values, votes = m.exactVoxels(batch, vox)
    assert np.array_equal(values, ref.reshape(-1)), "%d voxels differ" % int((values != ref.reshape(-1)).sum())
    assert votes.sum() > 4 * 1024   # (zero-weight votes count: integer coordinates give weight-0 corners)
    m.close()
    batch.close()
