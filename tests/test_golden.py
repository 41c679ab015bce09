"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle must reproduce them bit for bit (they are regression vectors of the
oracle itself -- the reference ships none and cannot be run here).
GPU (-m gpu): the HIP path is compared with the fixtures, so the GPU box checks against
committed data and not only against an oracle compiled on the spot.
"""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as orc
from oracle_pipeline import OracleMapper

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "case_*.npz")))
DSI_TOL = 1e-4


def load(path):
    z = np.load(path)
    g = {k: z[k] for k in z.files}
    g["cam_t"] = (int(g["cam"][0]), int(g["cam"][1])) + tuple(float(v) for v in g["cam"][2:])
    g["lut_arr"] = g["lut"] if int(g["has_lut"]) else None
    return g


def oracle_mapper(g):
    return OracleMapper(g["cam_t"], dimX=int(g["dimX"]), dimY=int(g["dimY"]), dimZ=int(g["nz"]),
                        min_depth=float(g["dmin"]), max_depth=float(g["dmax"]), fov=float(g["fov"]),
                        lut=g["lut_arr"], inverse_depth=bool(g["inverse"]))


def test_fixtures_present():
    assert len(CASES) >= 2


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_oracle_reproduces_golden(path):
    g = load(path)
    dsis = []
    for c in range(2):
        m = oracle_mapper(g)
        x, y = g["x%d" % c].astype(np.uint16), g["y%d" % c].astype(np.uint16)
        pk = m.packetize(g["ts%d" % c], (g["traj_t%d" % c], g["traj_p%d" % c]), g["T_rv_w"])
        assert np.array_equal(pk[0], g["first%d" % c])
        assert np.array_equal(pk[1], g["Rt%d" % c])
        xy, centers = m.evaluate_packets(x, y, *pk)
        assert np.array_equal(centers, g["centers%d" % c])
        assert np.array_equal(m.dsi, g["dsi%d" % c])
        dsis.append(m.dsi.copy())
        if c == 0:
            assert np.array_equal(xy, g["xy0"])
            assert np.array_equal(m.planes, g["planes"]) and np.array_equal(m.Kv, g["Kv"])
            depth, conf, idx = m.depth_map()
            assert np.array_equal(depth, g["depth0"]) and np.array_equal(conf, g["conf0"])
            assert np.array_equal(idx, g["idx0"])
            assert orc.mean_square(m.dsi) == float(g["mean_square0"])
    for op in range(1, 7):
        f = orc.fuse2(dsis[0], dsis[1], op)
        if op == 2:
            assert np.array_equal(f, g["fused2"])
        else:
            assert np.array_equal(f.reshape(-1)[::7], g["fused%d_s7" % op])
            assert f.astype(np.float64).sum() == float(g["fused%d_sum" % op])
    hm3 = orc.fuse_hm_n(g["fused2"], dsis[0], 3)
    assert np.array_equal(hm3.reshape(-1)[::7], g["fused_hm3_s7"])
    conf, idx = orc.collapse_max_z(g["fused2"])
    assert np.array_equal(conf, g["conf_fused2"]) and np.array_equal(idx, g["idx_fused2"])
    assert np.array_equal(orc.indices_to_depth(idx, g["planes"]), g["depth_fused2"])


# ------------------------------------------------------------------------- GPU
def _close(got, ref, tol=DSI_TOL):
    err = np.abs(got.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= tol, "max rel err %g" % err.max()


@pytest.mark.gpu
@pytest.mark.parametrize("algo", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_hip_matches_golden(ctx, path, algo):
    # 3..6 = LDS bands with the packed (asm) / grouped / packed (compiled) / grouped (asm) lane mapping
    packed = {3: 1, 4: 2, 5: 3, 6: 4}.get(algo, 0)
    algo = min(algo, 2)
    import dvs_mcemvs_amd as d
    g = load(path)
    shape = d.ShapeDSI(int(g["dimX"]), int(g["dimY"]), int(g["nz"]), float(g["dmin"]), float(g["dmax"]),
                       float(g["fov"]))
    mappers = []
    for c in range(2):
        m = d.MapperEMVS(ctx, g["cam_t"], shape, lut=g["lut_arr"], inverse_depth=bool(g["inverse"]))
        m.set_vote_algo(algo)
        m.set_packed_lanes(packed)
        assert np.array_equal(m.raw_depths_vec_, g["planes"])
        assert np.array_equal(np.array(m.virtual_cam_, np.float32), g["Kv"])
        x, y = g["x%d" % c].astype(np.uint16), g["y%d" % c].astype(np.uint16)
        # (a) full evaluateDSI from timestamps + trajectory (host packetisation in the C ABI)
        assert m.evaluateDSI((x, y, g["ts%d" % c]), (g["traj_t%d" % c], g["traj_p%d" % c]), g["T_rv_w"])
        assert m.n_voted == g["first%d" % c].shape[0] * 1024
        _close(m.dsi_.download(), g["dsi%d" % c])
        # (b) from the stored packetisation (device-resident batch)
        b = d.EventBatch(ctx, x, y, g["Rt%d" % c], g["first%d" % c])
        m.evaluateDSI_batch(b)
        _close(m.dsi_.download(), g["dsi%d" % c])
        b.close()
        mappers.append(m)
    # (c) the exact fillVoxelGrid boundary
    m0 = mappers[0]
    m0.dsi_.resetGrid()
    m0.fillVoxelGrid(g["xy0"], g["centers0"])
    _close(m0.dsi_.download(), g["dsi0"])
    depth, conf, idx = m0.getDepthMapFromDSI()
    vol = m0.dsi_.download()
    rconf, ridx = orc.collapse_max_z(vol)
    assert np.array_equal(conf, rconf) and np.array_equal(idx, ridx)
    srt = np.sort(g["dsi0"], axis=0)
    safe = (srt[-1] - srt[-2]) > 2 * DSI_TOL * np.maximum(1.0, srt[-1])
    assert np.array_equal(idx[safe], g["idx0"][safe])
    assert np.abs(depth - g["depth0"])[safe].max() <= 1e-4
    assert m0.dsi_.computeMeanSquare() == pytest.approx(float(g["mean_square0"]), rel=1e-5)
    # fusion on the golden volumes themselves: bit exact
    A = d.Grid3D(ctx, *m0.dsi_.getDimensions())
    B = d.Grid3D(ctx, *m0.dsi_.getDimensions())
    B.upload(g["dsi1"])
    for op in range(1, 7):
        A.upload(g["dsi0"])
        A.fuseTwoGrids(B, op)
        f = A.download()
        if op == 2:
            assert np.array_equal(f, g["fused2"])
            d2, c2, i2 = m0.getDepthMapFromDSI(A)
            assert np.array_equal(c2, g["conf_fused2"]) and np.array_equal(i2, g["idx_fused2"])
            assert np.array_equal(d2, g["depth_fused2"])
        else:
            assert np.array_equal(f.reshape(-1)[::7], g["fused%d_s7" % op])
    B.upload(g["dsi0"])
    A.upload(g["fused2"])
    A.harmonicMeanTwoGrids(B, 3)
    assert np.array_equal(A.download().reshape(-1)[::7], g["fused_hm3_s7"])
    if "temporal_hm" in g:
        # process_2 temporal fusion of camera 0 over sub-intervals by event count
        n = int(g["slices"])
        x, y, ts = g["x0"].astype(np.uint16), g["y0"].astype(np.uint16), g["ts0"]
        per = x.shape[0] // n
        hm = d.Grid3D(ctx, *m0.dsi_.getDimensions())
        am = d.Grid3D(ctx, *m0.dsi_.getDimensions())
        for k in range(n):
            sl = slice(k * per, (k + 1) * per)
            assert m0.evaluateDSI((x[sl], y[sl], ts[sl]), (g["traj_t0"], g["traj_p0"]), g["T_rv_w"])
            hm.addInverseOfTwoGrids(m0.dsi_)
            am.addTwoGrids(m0.dsi_)
        hm.computeHMfromSumOfInv(n)
        am.computeAMfromSum(n)
        _close(hm.download(), g["temporal_hm"], tol=2e-4)
        _close(am.download().reshape(-1)[::7], g["temporal_am_s7"], tol=2e-4)
    for m in mappers:
        m.close()
