"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Run on a real MI355X with `pytest -m gpu`.

Stated tolerances
  * coordinates are computed with the reference's operations in the reference's order
    (no FMA), so the ONLY difference between GPU and oracle DSIs is the order in which
    votes are summed into a voxel:  |gpu - cpu| <= 1e-4 * max(1, |cpu|)  for EVERY voxel
    (observed ~1e-6).
  * fusion ops, arg-max, index->depth: bit-exact on identical inputs.
  * mean square (double accumulation, different order): rel 1e-12.
"""
import os

import numpy as np
import pytest

import dvs_mcemvs_amd as d
from dvs_mcemvs_amd import synthetic as syn
from oracle import oracle as orc
from oracle_pipeline import OracleMapper, argmax_report

pytestmark = pytest.mark.gpu

DSI_TOL = 1e-4


def assert_dsi_close(got, ref, tol=DSI_TOL):
    err = np.abs(got.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= tol, "max rel err %g at %s (gpu %g cpu %g)" % (
        err.max(), np.unravel_index(err.argmax(), err.shape), got.flat[err.argmax()],
        ref.flat[err.argmax()])


def random_packets(rng, n_packets, nx, ny, spread=0.3, cz_spread=0.5):
    """z0 locations + camera centres as fillVoxelGrid receives them."""
    xy = np.empty((n_packets * 1024, 2), np.float32)
    xy[:, 0] = rng.uniform(-0.1 * nx, 1.1 * nx, xy.shape[0])
    xy[:, 1] = rng.uniform(-0.1 * ny, 1.1 * ny, xy.shape[0])
    centers = rng.normal(0, spread, (n_packets, 3)).astype(np.float32)
    centers[:, 2] = rng.normal(0, cz_spread, n_packets)
    return xy, centers


def make_mapper(ctx, cam, nz, dmin, dmax, algo, dimX=0, dimY=0, fov=0.0, lut=None, inverse=False,
                band=None, packed=None, inline_cuts=None):
    m = d.MapperEMVS(ctx, cam, d.ShapeDSI(dimX, dimY, nz, dmin, dmax, fov), lut=lut,
                     inverse_depth=inverse)
    m.set_vote_algo(algo)
    if band:
        m.set_band_params(*band)
    if packed is not None:
        m.set_packed_lanes(packed)
    if inline_cuts is not None:
        m.set_inline_cuts(inline_cuts)
    return m


ALGOS = [d.VOTE_GLOBAL_ATOMIC, d.VOTE_LDS_BANDS]


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("shape", [(64, 48, 16), (96, 72, 32), (346, 260, 20), (130, 97, 7)])
def test_fill_voxel_grid_matches_oracle(ctx, algo, shape):
    nx, ny, nz = shape
    rng = np.random.default_rng(100 + nx)
    cam = (nx, ny, 0.8 * nx, 0.8 * nx, 0.5 * nx, 0.5 * ny)
    m = make_mapper(ctx, cam, nz, 1.0, 6.5, algo)
    xy, centers = random_packets(rng, 9, nx, ny)
    ref = orc.fill_voxel_grid(xy, centers, m.raw_depths_vec_, np.array(m.virtual_cam_, np.float32),
                              nx, ny)
    m.dsi_.resetGrid()
    m.fillVoxelGrid(xy, centers)
    assert_dsi_close(m.dsi_.download(), ref)
    # fillVoxelGrid accumulates (the reset belongs to evaluateDSI): voting twice doubles
    m.fillVoxelGrid(xy, centers)
    assert_dsi_close(m.dsi_.download(), 2 * ref, tol=2 * DSI_TOL)
    m.close()


@pytest.mark.parametrize("band", [(5, 1, 256), (7, 3, 512), (16, 8, 1024), (48, 2, 256)])
def test_lds_band_decompositions_agree(ctx, band):
    """Every band height / chunk count / block size gives the same DSI (ragged last band,
    bands of different sizes, more chunks than packets)."""
    nx, ny, nz = 70, 48, 9
    rng = np.random.default_rng(5)
    cam = (nx, ny, 60.0, 60.0, 35.0, 24.0)
    xy, centers = random_packets(rng, 5, nx, ny)
    m = make_mapper(ctx, cam, nz, 0.5, 4.0, d.VOTE_LDS_BANDS, band=band)
    ref = orc.fill_voxel_grid(xy, centers, m.raw_depths_vec_, np.array(m.virtual_cam_, np.float32),
                              nx, ny)
    m.fillVoxelGrid(xy, centers)
    info = m.last_vote_info()
    assert info["algo"] == d.VOTE_LDS_BANDS and info["band_rows"] == band[0]
    assert_dsi_close(m.dsi_.download(), ref)
    m.close()


@pytest.mark.parametrize("algo", ALGOS)
def test_vote_edge_cases(ctx, algo):
    """Border drop (x+1 < Nx), negative / NaN / inf coordinates, camera centre at or beyond a
    plane (d = 0, negative alpha), all rejected or accepted exactly like the oracle."""
    nx, ny, nz = 40, 30, 6
    cam = (nx, ny, 30.0, 30.0, 20.0, 15.0)
    m = make_mapper(ctx, cam, nz, 1.0, 4.0, algo, band=(9, 2, 256))
    planes = m.raw_depths_vec_
    rng = np.random.default_rng(11)
    xy, centers = random_packets(rng, 6, nx, ny)
    special = np.array([[0.0, 0.0], [nx - 1.0, 3.0], [nx - 1.0001, 3.0], [3.0, ny - 1.0],
                        [-0.0, 5.0], [-1e-7, 5.0], [np.nan, 1.0], [1.0, np.nan], [np.inf, 2.0],
                        [2.0, -np.inf], [1e30, 1.0], [nx - 2.0, ny - 2.0], [3.4e38, 3.4e38]],
                       np.float32)
    for k in range(6):
        xy[k * 1024:k * 1024 + special.shape[0]] = special
    centers[0] = (0, 0, 0)                       # identity pose: every plane gets the z0 vote
    centers[1] = (0.2, -0.1, planes[2])          # zi - Cz = 0 on plane 2 -> a = 0
    centers[2] = (0.1, 0.1, planes[0])           # z0 - Cz = 0 -> d = 0 on every plane
    centers[3] = (0.0, 0.0, planes[3] + 0.01)    # camera between planes: negative alpha on some
    centers[4] = (5.0, -7.0, 100.0)              # far behind everything
    centers[5] = (np.nan, 0.0, 0.0)
    ref = orc.fill_voxel_grid(xy, centers, planes, np.array(m.virtual_cam_, np.float32), nx, ny)
    m.fillVoxelGrid(xy, centers)
    assert_dsi_close(m.dsi_.download(), ref)
    m.close()


@pytest.mark.parametrize("algo", ALGOS)
def test_identity_pose_known_answer(ctx, algo):
    """Identity pose => X_i = x0, Y_i = y0 on every plane (SURVEY 8c KAT)."""
    nx, ny, nz = 32, 24, 5
    m = make_mapper(ctx, (nx, ny, 25.0, 25.0, 16.0, 12.0), nz, 1.0, 3.0, algo)
    xy = np.zeros((1024, 2), np.float32)
    xy[:] = (2.25, 3.5)
    m.fillVoxelGrid(xy, np.zeros((1, 3), np.float32))
    got = m.dsi_.download()
    for z in range(nz):
        assert got[z, 3, 2] == pytest.approx(1024 * 0.375, rel=1e-6)
        assert got[z, 3, 3] == pytest.approx(1024 * 0.125, rel=1e-6)
        assert got[z, 4, 2] == pytest.approx(1024 * 0.375, rel=1e-6)
        assert got[z, 4, 3] == pytest.approx(1024 * 0.125, rel=1e-6)
        assert got[z].sum() == pytest.approx(1024.0, rel=1e-6)
    m.close()


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("variant", ["plain", "lut", "inverse_depth", "fov_dims"])
def test_evaluate_dsi_matches_oracle(ctx, algo, variant):
    """Full evaluateDSI (host packetisation + pose interpolation, device stage A + B)."""
    rig = syn.stereo_rig(30000, width=96, height=72, duration=0.3, seed=21)
    cam = rig["cam"]
    kw = dict(dimZ=24, min_depth=4.0, max_depth=200.0)
    lut, inverse, dimX, dimY, fov = None, False, 0, 0, 0.0
    if variant == "lut":
        lut = syn.radial_lut(cam)
    elif variant == "inverse_depth":
        inverse = True
    elif variant == "fov_dims":
        dimX, dimY, fov = 80, 64, 60.0
    for c in range(2):
        m = make_mapper(ctx, cam, 24, 4.0, 200.0, algo, dimX=dimX, dimY=dimY, fov=fov, lut=lut,
                        inverse=inverse)
        r = OracleMapper(cam, dimX=dimX, dimY=dimY, fov=fov, lut=lut, inverse_depth=inverse, **kw)
        assert np.array_equal(m.raw_depths_vec_, r.planes)
        assert np.array_equal(np.array(m.virtual_cam_, np.float32), r.Kv)
        assert m.evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
        assert r.evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
        assert m.n_voted == r.n_voted
        assert_dsi_close(m.dsi_.download(), r.dsi)
        # evaluateDSI resets the grid first (:145): a second call gives the same DSI
        assert m.evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
        assert_dsi_close(m.dsi_.download(), r.dsi)
        m.close()


def test_events_outside_the_sensor_are_dropped(ctx):
    """An event whose pixel lies outside the sensor has no LUT entry (the reference would read past its
    matrix, mapper_emvs_stereo.cpp:134): the engine gives it a non-finite z0 location, which no plane
    accepts -- same DSI as if its LUT entry were NaN, and no out-of-bounds access on the device."""
    rig = syn.stereo_rig(20000, width=96, height=72, duration=0.3, seed=23)
    cam = rig["cam"]
    lut = syn.radial_lut(cam).copy()
    lut.reshape(72, 96, 2)[0, 0] = np.nan
    x, y, ts = rig["events"][0]
    bad = np.zeros(x.shape, bool)
    bad[5::37] = True
    x_out, y_out = x.copy(), y.copy()
    x_out[bad & (np.arange(x.size) % 2 == 0)] = 96          # one past the last column
    y_out[bad & (np.arange(x.size) % 2 == 1)] = 65535       # far outside
    x_nan, y_nan = x.copy(), y.copy()
    x_nan[bad], y_nan[bad] = 0, 0                            # the pixel whose LUT entry is NaN
    got = []
    for xs, ys in ((x_out, y_out), (x_nan, y_nan)):
        m = make_mapper(ctx, cam, 16, 4.0, 200.0, d.VOTE_LDS_BANDS, lut=lut)
        assert m.evaluateDSI((xs, ys, ts), rig["trajectories"][0], rig["T_rv_w"])
        got.append(m.dsi_.download())
        m.close()
    assert np.array_equal(got[0], got[1]) and got[0].any()
    r = OracleMapper(cam, dimZ=16, min_depth=4.0, max_depth=200.0, lut=lut)
    assert r.evaluateDSI((x_nan, y_nan, ts), rig["trajectories"][0], rig["T_rv_w"])
    assert_dsi_close(got[0], r.dsi)


def test_evaluate_dsi_packet_counts(ctx):
    """mapper_emvs_stereo.cpp:71-75, :88: 1023 -> false, 1024 -> true with 0 packets,
    1025 -> 1 packet, 2048 -> 1, 2049 -> 2."""
    rig = syn.stereo_rig(4096, width=64, height=48, duration=0.2, seed=3)
    m = make_mapper(ctx, rig["cam"], 8, 4.0, 100.0, d.VOTE_AUTO)
    x, y, ts = rig["events"][0]
    for n, expect in [(1023, None), (1024, 0), (1025, 1), (2048, 1), (2049, 2)]:
        ok = m.evaluateDSI((x[:n], y[:n], ts[:n]), rig["trajectories"][0], rig["T_rv_w"])
        if expect is None:
            assert ok is False
        else:
            assert ok is True and m.n_voted == expect * 1024
            if expect == 0:
                assert not m.dsi_.download().any()
    m.close()


def test_pose_lookup_failure_shifts_packets(ctx):
    """mapper_emvs_stereo.cpp:95-99: a failed pose lookup slides the packet start by one event."""
    rig = syn.stereo_rig(6000, width=64, height=48, duration=0.2, seed=4)
    x, y, ts = rig["events"][0]
    times, poses = rig["trajectories"][0]
    late = ts[700]  # trajectory starts after the mid-timestamp of the first candidates
    keep = times > late
    traj = (times[keep], poses[keep])
    m = make_mapper(ctx, rig["cam"], 8, 4.0, 100.0, d.VOTE_AUTO)
    r = OracleMapper(rig["cam"], dimZ=8, min_depth=4.0, max_depth=100.0)
    assert m.evaluateDSI((x, y, ts), traj, rig["T_rv_w"])
    assert r.evaluateDSI((x, y, ts), traj, rig["T_rv_w"])
    assert m.n_voted == r.n_voted and m.n_voted > 0
    first, _ = r.packetize(ts, traj, rig["T_rv_w"])
    assert first[0] % 1024 != 0  # genuinely misaligned
    assert_dsi_close(m.dsi_.download(), r.dsi)
    m.close()


@pytest.mark.parametrize("n", [1, 5, 4096 + 3, 64 * 48 * 16])
def test_fusion_ops_bit_exact(ctx, n):
    rng = np.random.default_rng(n)
    nx, ny, nz = (n, 1, 1) if n < 4096 + 4 else (64, 48, 16)
    a = rng.gamma(2.0, 8.0, (nz, ny, nx)).astype(np.float32)
    g = rng.gamma(2.0, 8.0, (nz, ny, nx)).astype(np.float32)
    a.flat[::7] = 0.0  # empty voxels are the common case
    g.flat[::5] = 0.0
    A, G = d.Grid3D(ctx, nx, ny, nz), d.Grid3D(ctx, nx, ny, nz)
    G.upload(g)
    for op in range(1, 7):
        A.upload(a)
        A.fuseTwoGrids(G, op)
        assert np.array_equal(A.download(), orc.fuse2(a, g, op)), "op %d" % op
    for nmaps in (3, 4):
        A.upload(a)
        A.harmonicMeanTwoGrids(G, nmaps)
        assert np.array_equal(A.download(), orc.fuse_hm_n(a, g, nmaps))
    for mode, fin in ((d.ACC_SUM, A.computeAMfromSum), (d.ACC_INV_SUM, A.computeHMfromSumOfInv)):
        A.upload(a)
        ref = a
        for _ in range(3):
            (A.addTwoGrids if mode == d.ACC_SUM else A.addInverseOfTwoGrids)(G)
            ref = orc.accumulate(ref, g, mode)
        assert np.array_equal(A.download(), ref)
        fin(4)
        assert np.array_equal(A.download(), orc.finalize(ref, mode, 4))
    A.close()
    G.close()


def _ulp_diff(x, y):
    xi = x.view(np.int32).astype(np.int64)
    yi = y.view(np.int32).astype(np.int64)
    return np.abs(xi - yi)


@pytest.mark.parametrize("n_maps", [2, 3, 4, 8])
def test_nary_fusion_modes_bit_exact(ctx, n_maps):
    """n-ary camera fusion (accumulate / finalize modes LOG_SUM = GM, SQ_SUM = RMS, MIN, MAX, SUM =
    AM): the HIP kernels against the oracle, bit for bit, on vote-count-like values with zeros,
    denormals, huge values and +inf mixed in.  The reference has no n-ary GM / AM / RMS
    (process1.cpp:169-191 drops camera 3); for n = 2 the relation to its 2-ary ops
    (cartesian3dgrid.h:111-190) is checked too: min / max / AM equal, RMS within 2 ulp, GM within
    16 ulp for values in [2^-20, 2^20] (the accumulator is an fp32 sum of logs)."""
    rng = np.random.default_rng(40 + n_maps)
    nx, ny, nz = 67, 33, 5   # n % 4 != 0: the scalar tail of the float4 kernels runs too
    maps = []
    for k in range(n_maps):
        v = rng.gamma(2.0, 8.0, (nz, ny, nx)).astype(np.float32)
        v.flat[k::11] = 0.0                                   # empty voxels
        v.flat[3 + k::97] = np.float32(1e-41)                 # denormal
        v.flat[5 + k::101] = np.float32(3e37)
        v.flat[7 + k::103] = np.float32(2.0 ** -20)
        maps.append(v)
    maps[0].flat[50] = np.inf
    G = [d.Grid3D(ctx, nx, ny, nz) for _ in range(n_maps)]
    for g, v in zip(G, maps):
        g.upload(v)
    A = d.Grid3D(ctx, nx, ny, nz)
    for mode in (d.ACC_SUM, d.ACC_LOG_SUM, d.ACC_SQ_SUM, d.ACC_MIN, d.ACC_MAX):
        A.upload(maps[0])          # whatever was there must not matter
        A.setToFusionOfN(G, mode)
        got = A.download()
        ref = orc.fuse_nary(maps, mode)
        same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
        assert same.all(), "mode %d: %d voxels differ, first at %s: gpu %r cpu %r" % (
            mode, (~same).sum(), np.argwhere(~same)[0], got[~same][0], ref[~same][0])
    # the one-pass kernel (setToFusionOfN -> dsi_grid_fuse_n) and the begin / accumulate / finalize
    # sequence it stands for give the same bits
    for mode in (d.ACC_SUM, d.ACC_INV_SUM, d.ACC_LOG_SUM, d.ACC_SQ_SUM, d.ACC_MIN, d.ACC_MAX):
        A.setToFusionOfN(G, mode)
        one = A.download()
        A.accumulateBegin(mode)
        for g in G:
            A.accumulate(g, mode)
        A.finalize(mode, n_maps)
        seq = A.download()
        assert ((one.view(np.uint32) == seq.view(np.uint32)) | (np.isnan(one) & np.isnan(seq))).all(), mode
    # "0 if any factor is 0" and the all-reduce forms
    gm = orc.fuse_nary(maps, d.ACC_LOG_SUM)
    anyzero = np.zeros(gm.shape, bool)
    anyinf = np.zeros(gm.shape, bool)
    for v in maps:
        anyzero |= v == 0
        anyinf |= np.isinf(v)
    assert np.all(gm[anyzero & ~anyinf] == 0.0)          # (0 * inf has no geometric mean: NaN)
    assert d.acc_reduce_op(d.ACC_LOG_SUM) == d.REDUCE_SUM and d.acc_reduce_op(d.ACC_SQ_SUM) == d.REDUCE_SUM
    assert d.acc_reduce_op(d.ACC_MIN) == d.REDUCE_MIN and d.acc_reduce_op(d.ACC_MAX) == d.REDUCE_MAX
    if n_maps == 2:
        a, g = maps
        for mode, op in ((d.ACC_MIN, 1), (d.ACC_SUM, 4), (d.ACC_MAX, 6)):
            A.setToFusionOfN(G, mode)
            ref2 = orc.fuse2(a, g, op)
            got = A.download()
            ok = np.isfinite(ref2)
            assert np.array_equal(got[ok], ref2[ok]), "mode %d vs 2-ary op %d" % (mode, op)
        A.setToFusionOfN(G, d.ACC_SQ_SUM)
        got, ref2 = A.download(), orc.fuse2(a, g, 5)
        ok = np.isfinite(ref2) & (ref2 > 1e-18) & np.isfinite(a * a + g * g)
        assert _ulp_diff(got[ok], ref2[ok]).max() <= 2
        A.setToFusionOfN(G, d.ACC_LOG_SUM)
        got, ref2 = A.download(), orc.fuse2(a, g, 3)
        lo, hi = 2.0 ** -20, 2.0 ** 20
        ok = (a >= lo) & (a <= hi) & (g >= lo) & (g <= hi)
        assert ok.sum() > 1000 and _ulp_diff(got[ok], ref2[ok]).max() <= 16
    for o in G + [A]:
        o.close()


@pytest.mark.parametrize("n_maps", [2, 4, 8])
def test_gm_tree_is_the_reference_op_applied_pairwise(ctx, n_maps):
    """ACC_GM_TREE: the n-ary geometric mean rooted in the reference -- the balanced tree of its own 2-ary
    sqrt(a*g) (Grid3D::geometricMeanTwoGrids, cartesian3dgrid.h:150-156).  Bit-equal to the oracle's repeated
    2-ary calls; for n = 2 bit-equal to the 2-ary member itself (the LOG_SUM form is only within 16 ulp);
    the arg-max-fused form gives the same bits as fuse-then-collapse; within 1e-5 of exp(mean(log))."""
    rng = np.random.default_rng(60 + n_maps)
    nx, ny, nz = 67, 33, 6
    maps = []
    for k in range(n_maps):
        v = rng.gamma(2.0, 8.0, (nz, ny, nx)).astype(np.float32)
        v.flat[k::13] = 0.0
        v.flat[3 + k::97] = np.float32(1e-41)
        v.flat[5 + k::101] = np.float32(3e37)          # a product of two overflows to inf: sqrt(inf) = inf on both sides
        maps.append(v)
    G = [d.Grid3D(ctx, nx, ny, nz) for _ in range(n_maps)]
    for g, v in zip(G, maps):
        g.upload(v)
    A = d.Grid3D(ctx, nx, ny, nz)
    A.setToFusionOfN(G, d.ACC_GM_TREE)
    got = A.download()
    ref = orc.fuse_gm_tree(maps)
    same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), "%d voxels differ" % (~same).sum()
    if n_maps == 2:
        A.upload(maps[0])
        A.geometricMeanTwoGrids(G[1])                 # the 2-ary member on the GPU
        assert np.array_equal(A.download().view(np.uint32), got.view(np.uint32))
        assert np.array_equal(orc.fuse2(maps[0], maps[1], 3).view(np.uint32), got.view(np.uint32))
    lo, hi = 2.0 ** -20, 2.0 ** 20
    ok = np.ones(got.shape, bool)
    for v in maps:
        ok &= (v >= lo) & (v <= hi)
    logs = orc.fuse_nary(maps, d.ACC_LOG_SUM)
    assert ok.sum() > 1000 and np.allclose(got[ok], logs[ok], rtol=1e-5)
    # inside the arg-max kernel
    m = d.MapperEMVS(ctx, (nx, ny, 50.0, 50.0, 33.0, 16.0), d.ShapeDSI(0, 0, nz, 1.0, 5.0, 0.0))
    finite = [np.where(np.isfinite(v) & (v < 1e30), v, 0).astype(np.float32) for v in maps]
    for g, v in zip(G, finite):
        g.upload(v)
    A.setToFusionOfN(G, d.ACC_GM_TREE)
    m.computeDepthMap(A)
    want = m.fetchDepthMap()
    m.computeDepthMapOfFusionN(G, d.ACC_GM_TREE)
    have = m.fetchDepthMap()
    for a_, b_ in zip(have, want):
        assert np.array_equal(a_, b_)
    # not an accumulation, and only for 2 / 4 / 8 grids
    from dvs_mcemvs_amd import engine
    with pytest.raises(d.DsiError) as e:
        A.accumulate(G[0], d.ACC_GM_TREE)
    assert e.value.code == engine.ERR_BAD_OP
    with pytest.raises(d.DsiError) as e:
        A.setToFusionOfN(G[:1] + G[:1] + G[:1], d.ACC_GM_TREE)
    assert e.value.code == engine.ERR_BAD_OP
    for o in G + [A, m]:
        o.close()


@pytest.mark.parametrize("n_maps", [1, 3, 4, 8])
def test_depth_map_of_nary_fusion_equals_fuse_then_collapse(ctx, n_maps):
    """dsi_mapper_depth_map_of_fusion_n (the n-camera fusion inside the arg-max kernel) = setToFusionOfN
    followed by computeDepthMap, bit for bit, for every accumulate mode -- zeros, ties and an odd
    number of planes included -- and equal to the oracle's fuse_nary + collapse."""
    rng = np.random.default_rng(60 + n_maps)
    nx, ny, nz = 37, 21, 9
    cam = (nx, ny, 30.0, 30.0, 18.0, 10.0)
    m = make_mapper(ctx, cam, nz, 1.0, 6.0, d.VOTE_AUTO)
    maps = []
    for k in range(n_maps):
        v = np.rint(rng.gamma(2.0, 2.0, (nz, ny, nx))).astype(np.float32)   # small integers: many ties
        v.flat[k::7] = 0.0
        maps.append(v)
    G = [d.Grid3D(ctx, nx, ny, nz) for _ in range(n_maps)]
    for g, v in zip(G, maps):
        g.upload(v)
    F = d.Grid3D(ctx, nx, ny, nz)
    for mode in (d.ACC_SUM, d.ACC_INV_SUM, d.ACC_LOG_SUM, d.ACC_SQ_SUM, d.ACC_MIN, d.ACC_MAX):
        m.computeDepthMapOfFusionN(G, mode)
        depth, conf, idx = m.fetchDepthMap()
        F.setToFusionOfN(G, mode)
        m.computeDepthMap(F)
        depth2, conf2, idx2 = m.fetchDepthMap()
        assert np.array_equal(idx, idx2) and np.array_equal(depth, depth2), mode
        assert np.array_equal(conf.view(np.uint32), conf2.view(np.uint32)), mode
        if mode != d.ACC_INV_SUM:
            rconf, ridx = orc.collapse_max_z(orc.fuse_nary(maps, mode))
            assert np.array_equal(idx, ridx) and np.array_equal(conf, rconf), mode
    with pytest.raises(d.DsiError):
        m.computeDepthMapOfFusionN(G, 9)
    for o in G + [F, m]:
        o.close()


def test_fusion_errors(ctx):
    A, B = d.Grid3D(ctx, 8, 8, 4), d.Grid3D(ctx, 8, 8, 5)
    with pytest.raises(d.DsiError) as e:
        A.addTwoGrids(B)
    assert e.value.code == 4  # DSI_ERR_SHAPE (reference: std::out_of_range)
    C_ = d.Grid3D(ctx, 8, 8, 4)
    for bad in (0, 7, -1):
        with pytest.raises(d.DsiError) as e:
            A.fuseTwoGrids(C_, bad)
        assert e.value.code == 5  # "Improper fusion method selected"


@pytest.mark.parametrize("shape", [(33, 17, 1), (64, 48, 16), (50, 40, 100), (31, 9, 256)])
def test_collapse_max_z_exact(ctx, shape):
    nx, ny, nz = shape
    rng = np.random.default_rng(nz)
    v = rng.integers(0, 6, (nz, ny, nx)).astype(np.float32)  # many ties: first max must win
    v[:, 0, 0] = 0.0                                         # empty column -> conf 0, index 0
    if nz > 2:
        v[:, 1, 1] = 3.0                                     # all equal -> index 0
        v[nz - 1, 2, 2] = 99.0                               # max on the last plane
    G = d.Grid3D(ctx, nx, ny, nz)
    G.upload(v)
    conf, idx = G.collapseMaxZSlice()
    rconf, ridx = orc.collapse_max_z(v)
    assert np.array_equal(conf, rconf) and np.array_equal(idx, ridx)
    assert idx[0, 0] == 0 and conf[0, 0] == 0
    ms = G.computeMeanSquare()
    assert ms == pytest.approx(orc.mean_square(v), rel=1e-12)
    G.close()


def test_depth_map_of_fused_grid(ctx):
    rig = syn.stereo_rig(20000, width=80, height=60, duration=0.2, seed=9)
    ms = [make_mapper(ctx, rig["cam"], 20, 4.0, 200.0, d.VOTE_AUTO) for _ in range(2)]
    for c in range(2):
        assert ms[c].evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
    fused = d.Grid3D(ctx, *ms[0].dsi_.getDimensions())
    fused.addTwoGrids(ms[0].dsi_)
    fused.harmonicMeanTwoGrids(ms[1].dsi_)
    vol = fused.download()
    depth, conf, idx = ms[0].getDepthMapFromDSI(fused)
    rconf, ridx = orc.collapse_max_z(vol)
    assert np.array_equal(idx, ridx) and np.array_equal(conf, rconf)
    assert np.array_equal(depth, orc.indices_to_depth(ridx, ms[0].raw_depths_vec_))
    assert np.abs(depth - orc.indices_to_depth(ridx, ms[0].raw_depths_vec_)).max() <= 1e-4


@pytest.mark.parametrize("shape", [(64, 48, 16), (50, 40, 7), (33, 17, 1), (96, 72, 130)])
def test_depth_map_of_fusion_equals_fuse_then_collapse(ctx, shape):
    """dsi_mapper_depth_map_of_fusion (camera fusion inside the arg-max kernel) = the reference's
    sequence resetGrid / addTwoGrids / <op>TwoGrids (process1.cpp:126-158) then collapseMaxZSlice +
    convertDepthIndicesToValues, for every op code, bit for bit -- ties and all-zero columns included."""
    nx, ny, nz = shape
    rng = np.random.default_rng(nx + nz)
    a = np.floor(rng.gamma(1.5, 2.0, (nz, ny, nx))).astype(np.float32)      # integers: many ties
    g = np.floor(rng.gamma(1.5, 2.0, (nz, ny, nx))).astype(np.float32)
    a[:, :2] = 0.0
    g[:, 1:3] = 0.0
    a[:, 5, 5] = 3.25
    g[:, 5, 5] = 3.25                                                         # a whole column tied: index 0
    cam = (nx, ny, 0.8 * nx, 0.8 * nx, 0.5 * nx, 0.5 * ny)
    m = d.MapperEMVS(ctx, cam, d.ShapeDSI(0, 0, nz, 1.0, 9.0, 0.0))
    A, G, F = d.Grid3D(ctx, nx, ny, nz), d.Grid3D(ctx, nx, ny, nz), d.Grid3D(ctx, nx, ny, nz)
    A.upload(a)
    G.upload(g)
    for op in range(1, 7):
        F.setToFusionOf(A, G, op)
        m.computeDepthMap(F)
        want = m.fetchDepthMap()
        m.computeDepthMapOfFusion(A, G, op)
        got = m.fetchDepthMap()
        for x, y in zip(got, want):
            assert np.array_equal(x, y), "op %d" % op
        rconf, ridx = orc.collapse_max_z(orc.fuse2(a, g, op))
        assert np.array_equal(got[2], ridx) and np.array_equal(got[1], rconf)
    with pytest.raises(d.DsiError) as e:
        m.computeDepthMapOfFusion(A, G, 7)
    assert e.value.code == 5
    for o in (m, A, G, F):
        o.close()


_DIV_PROBE = r'''
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %(root)r)
import dvs_mcemvs_amd as d
ctx = d.Context(0)
rng = np.random.default_rng(77)
n = 1 << 22
num = (rng.uniform(-1, 1, n) * np.exp2(rng.uniform(-30, 30, n))).astype(np.float32)
den = (rng.choice([-1.0, 1.0], n) * rng.uniform(1, 2, n) * np.exp2(rng.integers(-40, 40, n))).astype(np.float32)
num[:8] = [0.0, -0.0, 1.0, 3.0, 1e-38, 16777215.0, 0.1, 346.0]
q = np.empty(n, np.float32)
ref = np.empty(n, np.float32)
L = d.load_library()
assert L.dsi_build_flavour() == 1
f32p = C.POINTER(C.c_float)
rc = L.dsi_test_div_probe(ctx._h, num.ctypes.data_as(f32p), den.ctypes.data_as(f32p), n, q.ctypes.data_as(f32p),
                          ref.ctypes.data_as(f32p))
assert rc == 0
assert np.array_equal(ref, num / den)          # the GPU's '/' is IEEE
same = (q == ref) | ((q == 0) & (ref == 0))    # sign of zero may differ; votes identical
assert same.all(), "%%d mismatches" %% (~same).sum()
print("DIV_PROBE_OK")
'''


def test_residual_corrected_division_is_ieee(built):
    """The banded kernel's 5-op division must equal the IEEE divide bit for bit wherever it
    is used (2^-40 <= |d| <= 2^40), otherwise its coordinates would not be the oracle's.  The probe kernel is a
    hook of the EXPERIMENTS flavour of the library (same kernel source; the production library exports no test
    hook), loaded by a child process that opts in with DSI_ENGINE_EXPERIMENTS=1."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _DIV_PROBE % {"root": root}], capture_output=True, text=True,
                       env=dict(os.environ, DSI_ENGINE_EXPERIMENTS="1"), timeout=600)
    assert r.returncode == 0 and "DIV_PROBE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_full_size_properties(ctx):
    """346x260x100 at 1M events/camera: size-independent checks (the oracle is too slow to be
    the checker at BASELINE sizes beyond this)."""
    rig = syn.stereo_rig(1_000_000, seed=1234)
    cam = rig["cam"]
    tot = []
    vols = []
    for algo in ALGOS:
        m = make_mapper(ctx, cam, 100, 4.0, 200.0, algo)
        assert m.evaluateDSI(rig["events"][0], rig["trajectories"][0], rig["T_rv_w"])
        v = m.dsi_.download()
        vols.append(v)
        assert (v >= 0).all()
        # each accepted vote adds exactly 1 to its plane: plane sums are integers <= events voted
        sums = v.reshape(100, -1).sum(axis=1, dtype=np.float64)
        assert (sums <= m.n_voted + 1).all()
        assert np.abs(sums - np.rint(sums)).max() < 0.5
        tot.append(sums)
        assert m.dsi_.computeMeanSquare() == pytest.approx(float((v.astype(np.float64) ** 2).mean()), rel=1e-10)
        m.close()
    # the two kernels agree voxel for voxel
    assert_dsi_close(vols[1], vols[0].astype(np.float64))
    assert np.abs(tot[0] - tot[1]).max() < 0.5
    # linearity: DSI(first half) + DSI(second half) == DSI(all) when the halves are packet aligned
    x, y, ts = rig["events"][0]
    h = 488 * 1024 + 1  # 488 packets + the dropped tail event
    m = make_mapper(ctx, cam, 100, 4.0, 200.0, d.VOTE_LDS_BANDS)
    assert m.evaluateDSI((x[:h], y[:h], ts[:h]), rig["trajectories"][0], rig["T_rv_w"])
    va = m.dsi_.download()
    x2, y2, t2 = x[h - 1:], y[h - 1:], ts[h - 1:]
    assert m.evaluateDSI((x2, y2, t2), rig["trajectories"][0], rig["T_rv_w"])
    vb = m.dsi_.download()
    assert_dsi_close(va + vb, vols[1].astype(np.float64), tol=2e-4)
    m.close()
    # oracle on the full-size case (8 host threads: a few seconds)
    r = OracleMapper(cam, dimZ=100, min_depth=4.0, max_depth=200.0)
    assert r.evaluateDSI(rig["events"][0], rig["trajectories"][0], rig["T_rv_w"])
    assert_dsi_close(vols[1], r.dsi)
    d1, c1, i1 = r.depth_map()
    conf, idx = orc.collapse_max_z(vols[1])
    # arg-max index equal wherever the CPU top-2 gap exceeds the DSI tolerance
    srt = np.sort(r.dsi, axis=0)
    gap = srt[-1] - srt[-2]
    safe = gap > 2 * DSI_TOL * np.maximum(1.0, srt[-1])
    assert np.array_equal(idx[safe], i1[safe])
    assert (np.abs(orc.indices_to_depth(idx, r.planes) - d1)[safe] <= 1e-4).all()
    # ... and on EVERY other pixel the GPU's plane is a provable near-tie of the oracle's column
    rep = argmax_report(idx, r.dsi, DSI_TOL)
    assert rep["violations"] == 0, rep


@pytest.mark.parametrize("dims", [(512, 512, 200), (1024, 1024, 256)])
def test_baseline_large_grids(ctx, dims):
    """BASELINE.json configs[2] (512x512x200) and configs[4] (1024x1024x256) grid shapes at a
    small event count: many narrow bands, rows of 4-8 KB in LDS, 1 GiB volume."""
    nx, ny, nz = dims
    rig = syn.stereo_rig(6 * 1024 + 1, width=nx, height=ny, duration=0.2, seed=77)
    m = make_mapper(ctx, rig["cam"], nz, 4.0, 200.0, d.VOTE_LDS_BANDS)
    r = OracleMapper(rig["cam"], dimZ=nz, min_depth=4.0, max_depth=200.0)
    assert m.evaluateDSI(rig["events"][0], rig["trajectories"][0], rig["T_rv_w"])
    assert r.evaluateDSI(rig["events"][0], rig["trajectories"][0], rig["T_rv_w"])
    info = m.last_vote_info()
    assert info["algo"] == d.VOTE_LDS_BANDS and info["bands"] >= 10
    got = m.dsi_.download()
    assert_dsi_close(got, r.dsi)
    depth, conf, idx = m.getDepthMapFromDSI()
    rconf, ridx = orc.collapse_max_z(got)
    assert np.array_equal(conf, rconf) and np.array_equal(idx, ridx)
    m.close()


def test_configs4_camera_at_full_size(ctx):
    """One camera of BASELINE.json configs[4] at FULL size: 100 M events (97,650 packets) into a
    1024x1024x256 DSI (1 GiB) in one evaluateDSI call.  The oracle cannot follow there (it runs at
    ~2.5 Mevents/s on this shape), so the check is linearity, a size-independent property of the vote:
    the DSI of the whole batch equals the sum of the DSIs of its ten packet-aligned tenths (each
    voxel is an exact fixed-point sum rounded once, so the two differ by at most ten roundings), and
    no plane holds more votes than there are events."""
    nx = ny = 1024
    nz, parts = 256, 10
    rig = syn.stereo_rig(10_000_000, width=nx, height=ny, t0=10.0, duration=0.5, seed=1234)
    x, y, ts = rig["events"][0]
    first, Rt = d.packetize(ts, rig["trajectories"][0], rig["T_rv_w"])
    n_packets = first.shape[0]
    assert np.array_equal(first, np.arange(n_packets, dtype=np.uint32) * 1024)
    n = n_packets * 1024
    x, y = x[:n], y[:n]
    parts_Rt = []
    for i in range(parts):       # the same pixels seen from a rig that has moved on: ten different tenths
        r = Rt.copy()
        r[:, 9] += 0.02 * i
        r[:, 11] += 0.01 * i
        parts_Rt.append(r)
    m = d.MapperEMVS(ctx, rig["cam"], d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0))
    acc = d.Grid3D(ctx, nx, ny, nz)
    acc.resetGrid()
    for i in range(parts):
        b = d.EventBatch(ctx, x, y, parts_Rt[i])
        m.evaluateDSI_batch(b)
        acc.addTwoGrids(m.dsi_)
        ctx.synchronize()
        b.close()
    full = d.EventBatch(ctx, np.tile(x, parts), np.tile(y, parts), np.concatenate(parts_Rt))
    m.evaluateDSI_batch(full)
    info = m.last_vote_info()
    assert info["n_packets"] * 1024 == n * parts == 99_993_600 and info["algo"] == d.VOTE_LDS_BANDS
    whole = m.dsi_.download()
    summed = acc.download()
    sums = whole.reshape(nz, -1).sum(axis=1, dtype=np.float64)
    assert (sums <= n * parts + 64).all() and sums.max() > 0.9 * n * parts
    err = np.abs(whole.astype(np.float64) - summed) / np.maximum(1.0, np.abs(summed))
    assert err.max() < 2e-6, err.max()
    for o in (full, acc, m):
        o.close()


def test_configs4_end_to_end_at_full_size(ctx):
    """BASELINE.json configs[4] END TO END at full size on one GPU: 4 cameras x 100 M events, 1024x1024x256 (1 GiB
    per DSI), n-ary geometric mean, arg-max + depth.
      (a) every plane of every camera DSI holds at most as many votes as there are events, most of them all;
      (b) the camera fusion computed INSIDE the arg-max kernel gives the depth map of fuse-then-collapse bit
          for bit, for both GM forms (tree of the reference's 2-ary op; exp(mean(log)));
      (c) against the CPU oracle on a 64-row strip (all 256 planes, all 4 x 100 M events -- what the oracle
          can afford at this size): every voxel of the four camera DSIs and of the fused DSI within the stated
          tolerance, and for EVERY pixel of the strip the GPU's plane is the oracle's or a provable near-tie."""
    nx = ny = 1024
    nz, parts, n_cams = 256, 10, 4
    r0, rows = 480, 64
    rig = syn.stereo_rig(10_000_000, width=nx, height=ny, t0=10.0, duration=0.5, seed=1234, n_cams=n_cams)
    mappers, batches, host = [], [], []
    for c in range(n_cams):
        x, y, ts = rig["events"][c]
        first, Rt = d.packetize(ts, rig["trajectories"][c], rig["T_rv_w"])
        n = first.shape[0] * 1024
        assert np.array_equal(first, np.arange(first.shape[0], dtype=np.uint32) * 1024)
        rts = []
        for i in range(parts):       # the same pixels seen from a rig that has moved on: ten different tenths
            r = Rt.copy()
            r[:, 9] += 0.02 * i
            r[:, 11] += 0.01 * i
            rts.append(r)
        X, Y, R = np.tile(x[:n], parts), np.tile(y[:n], parts), np.concatenate(rts)
        m = d.MapperEMVS(ctx, rig["cam"], d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0))
        b = d.EventBatch(ctx, X, Y, R)
        m.evaluateDSI_batch(b)
        assert m.last_vote_info()["n_packets"] * 1024 == 99_993_600
        mappers.append(m)
        batches.append(b)
        host.append((X, Y, R))
    dsis = [m.dsi_ for m in mappers]
    fused = d.Grid3D(ctx, nx, ny, nz)
    idx_tree = None
    for mode in (d.ACC_GM_TREE, d.ACC_LOG_SUM):                                         # (b)
        fused.setToFusionOfN(dsis, mode)
        mappers[0].computeDepthMap(fused)
        want = mappers[0].fetchDepthMap()
        mappers[0].computeDepthMapOfFusionN(dsis, mode)
        got = mappers[0].fetchDepthMap()
        for g, w, name in zip(got, want, ("depth", "confidence", "index")):
            assert np.array_equal(g, w), "mode %d: %s differs at %d pixels" % (mode, name, (g != w).sum())
        assert want[1].max() > 10.0
        if mode == d.ACC_GM_TREE:
            idx_tree, conf_tree, depth_tree = want[2], want[1], want[0]
    assert np.array_equal(depth_tree, mappers[0].raw_depths_vec_[idx_tree])
    fused.setToFusionOfN(dsis, d.ACC_GM_TREE)
    fused_strip = fused.download()[:, r0:r0 + rows, :]
    strips = []
    first_all = np.arange(host[0][2].shape[0], dtype=np.int64) * 1024
    for c in range(n_cams):                                                             # (a), (c)
        vol = mappers[c].dsi_.download()
        sums = vol.reshape(nz, -1).sum(axis=1, dtype=np.float64)
        assert (sums <= 99_993_600 + 64).all() and sums.max() > 0.9 * 99_993_600
        o = OracleMapper(rig["cam"], dimZ=nz, min_depth=4.0, max_depth=200.0)
        ref = o.evaluate_packets_rows(host[c][0], host[c][1], first_all, host[c][2], r0, rows)
        assert_dsi_close(vol[:, r0:r0 + rows, :], ref)
        assert ref.max() > 50.0
        strips.append(ref)
        del vol
    ref_fused = orc.fuse_gm_tree(strips)
    assert_dsi_close(fused_strip, ref_fused, tol=4 * DSI_TOL)      # GM of four values each within DSI_TOL
    rep = argmax_report(idx_tree[r0:r0 + rows], ref_fused, 4 * DSI_TOL)
    print("configs[4] full size, rows %d..%d: %r" % (r0, r0 + rows, rep))
    assert rep["violations"] == 0 and rep["argmax_agree_frac"] > 0.9, rep
    # (d) the n-camera exact resolution (process.exact_depth_map_nary: near-tie columns of the fused DSI, their voxels of
    #     all four camera DSIs re-summed in the reference's order over the 4 x 100 M events, the GM tree on the host):
    #     on the strip the oracle covers, the plane index map IS the oracle's
    from dvs_mcemvs_amd import process
    mappers[0].computeDepthMap(fused)
    info = process.exact_depth_map_nary(mappers[0], mappers, batches, d.ACC_GM_TREE, fused_grid=fused)
    _, _, idx_exact = mappers[0].fetchDepthMap()
    ridx = ref_fused.argmax(axis=0)
    before = int((idx_tree[r0:r0 + rows] != ridx).sum())
    print("configs[4] exact n-ary resolution: %r; on the strip %d pixels differed before" % (info, before))
    assert np.array_equal(idx_exact[r0:r0 + rows], ridx), "%d pixels of the strip differ" % (idx_exact[r0:r0 + rows] != ridx).sum()
    # (e) beyond the strip: the resolution's premise as a per-column PROOF over the WHOLE image (dsi_mapper_prove_near_ties_n:
    #     the 4 x 100 M events' votes counted per voxel, the reference's fp32 event-order sums bounded from the counts).  With
    #     tens of thousands of votes in a voxel the bounds are ~1e-3 wide: the default gap proves 99.2 % of the 1,048,576
    #     columns (1,040,633; the rest ask for up to 1.8e-3, and settling them means re-summing their 8 M voxels over
    #     4 x 100 M events -- minutes, in chunks: not in this test).  Asserted: the proof accounts for every column, proves
    #     the bulk, and a column it calls proven on the strip indeed carries the oracle's plane
    proof = mappers[0].proveNearTiesN(mappers, batches, d.ACC_GM_TREE, fused_grid=fused)
    print("configs[4] proof at the default gap: %r" % (proof,))
    assert proof["columns"] == 1024 * 1024 and proof["columns_proven"] + proof["columns_unproven"] == 1024 * 1024
    assert proof["max_votes"] > 1000 and proof["columns_proven"] > 0.95 * 1024 * 1024
    pix, gaps = mappers[0].proofUnproven()
    assert pix.size == proof["columns_unproven"] and (pix.size == 0 or float(gaps.max()) == pytest.approx(proof["gap_needed"], rel=1e-6))
    proven_mask = np.ones(1024 * 1024, bool)
    proven_mask[pix] = False
    strip_proven = proven_mask.reshape(1024, 1024)[r0:r0 + rows]
    assert np.array_equal(idx_exact[r0:r0 + rows][strip_proven], ridx[strip_proven])
    for o_ in mappers + batches + [fused]:
        o_.close()


def test_fuse_into_equals_reference_sequence(ctx):
    """dsi_grid_fuse2_into == resetGrid + addTwoGrids + <op>TwoGrids (process1.cpp:126-158)."""
    rng = np.random.default_rng(12)
    a = rng.gamma(2.0, 8.0, (6, 10, 13)).astype(np.float32)
    g = rng.gamma(2.0, 8.0, (6, 10, 13)).astype(np.float32)
    a.flat[::3] = 0.0
    A, G, F, R = (d.Grid3D(ctx, 13, 10, 6) for _ in range(4))
    A.upload(a)
    G.upload(g)
    for op in range(1, 7):
        R.upload(rng.random((6, 10, 13)).astype(np.float32))  # stale contents must not matter
        R.setToFusionOf(A, G, op)
        F.resetGrid()
        F.addTwoGrids(A)
        F.fuseTwoGrids(G, op)
        assert np.array_equal(R.download(), F.download())
        assert np.array_equal(R.download(), orc.fuse2(a, g, op))
    with pytest.raises(d.DsiError):
        A.setToFusionOf(A, G, 2)


@pytest.mark.parametrize("packed", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("shape,band", [((64, 48, 16), None), ((346, 260, 12), (26, 4, 1024)),
                                        ((130, 97, 7), (5, 3, 256)), ((70, 48, 9), (48, 2, 512)),
                                        ((40, 30, 6), (9, 2, 256))])
def test_both_lane_mappings_match_oracle(ctx, shape, band, packed):
    """The per-packet, packed-lane and packet-group voting kernels are three lane mappings of
    the same work."""
    nx, ny, nz = shape
    rng = np.random.default_rng(300 + nx)
    cam = (nx, ny, 0.8 * nx, 0.8 * nx, 0.5 * nx, 0.5 * ny)
    m = make_mapper(ctx, cam, nz, 1.0, 6.5, d.VOTE_LDS_BANDS, band=band, packed=packed)
    xy, centers = random_packets(rng, 70, nx, ny)     # > 64 packets: more than one wave group
    xy[5 * 1024:6 * 1024] = np.nan                    # an entirely dead packet (run length 0)
    xy[7 * 1024:8 * 1024, 1] = 3.25                   # one row: the whole packet in one band
    centers[9] = (0.1, 0.1, m.raw_depths_vec_[0])     # d = 0 on every plane
    centers[11] = (0.0, 0.0, 1e30)                    # |d| huge -> IEEE-divide path
    ref = orc.fill_voxel_grid(xy, centers, m.raw_depths_vec_, np.array(m.virtual_cam_, np.float32),
                              nx, ny)
    m.fillVoxelGrid(xy, centers)
    assert m.last_vote_info()["packed"] == packed
    assert_dsi_close(m.dsi_.download(), ref)
    m.close()


@pytest.mark.parametrize("shape,band", [((64, 48, 16), None), ((346, 260, 12), (26, 4, 1024)), ((346, 260, 12), None),
                                        ((131, 97, 7), (5, 3, 1024)), ((70, 48, 9), (48, 2, 1024)), ((41, 30, 6), (9, 1, 1024))])
def test_paired_cells_mapping_matches_oracle(ctx, shape, band):
    """Lane mapping 8 (round 6, opt-in): two LDS atomics per vote on paired 32-bit cells, ROUNDED Q.19 weights instead of the
    exact Q33.31 sums -- NOT bit-identical to the other mappings, but inside the tolerance every DSI is held to
    (|gpu - cpu| <= 1e-4 max(1, |cpu|), cartesian3dgrid.h:261-270 summed in fp32 by the reference), odd and even widths,
    several chunks (raw partial volumes), dead packets, one-row packets, d = 0 and IEEE-divide planes; no cell near its
    capacity."""
    nx, ny, nz = shape
    rng = np.random.default_rng(300 + nx)
    cam = (nx, ny, 0.8 * nx, 0.8 * nx, 0.5 * nx, 0.5 * ny)
    m = make_mapper(ctx, cam, nz, 1.0, 6.5, d.VOTE_LDS_BANDS, band=band, packed=8)
    xy, centers = random_packets(rng, 70, nx, ny)
    xy[5 * 1024:6 * 1024] = np.nan
    xy[7 * 1024:8 * 1024, 1] = 3.25
    centers[9] = (0.1, 0.1, m.raw_depths_vec_[0])
    centers[11] = (0.0, 0.0, 1e30)
    ref = orc.fill_voxel_grid(xy, centers, m.raw_depths_vec_, np.array(m.virtual_cam_, np.float32), nx, ny)
    m.fillVoxelGrid(xy, centers)
    assert m.last_vote_info()["packed"] == 8
    got = m.dsi_.download()
    assert_dsi_close(got, ref)
    assert not m.paired_overflow()
    # against the exact mapping: the difference is the weights' rounding, <= 2^-20 per vote (votes <= value / smallest weight...)
    m7 = make_mapper(ctx, cam, nz, 1.0, 6.5, d.VOTE_LDS_BANDS, band=band, packed=7)
    m7.fillVoxelGrid(xy, centers)
    exact = m7.dsi_.download()
    assert np.abs(got.astype(np.float64) - exact).max() <= 2.0 ** -20 * 70 * 1024   # (a loose, rigorous bound)
    assert np.abs(got.astype(np.float64) - exact).max() <= 1e-4
    # the second call accumulates (fillVoxelGrid adds to the grid: raw partial volumes): still inside the tolerance
    m.fillVoxelGrid(xy, centers)
    assert_dsi_close(m.dsi_.download(), 2.0 * ref.astype(np.float64))
    m.close()
    m7.close()


def test_paired_cells_report_a_cell_at_half_capacity(ctx):
    """Identity pose, every event on the integer location (2, 3): all votes are one full weight on ONE paired cell.  3,072
    of them are summed exactly; 5,120 exceed 2^31 in Q.19 (half the cell's capacity): dsi_mapper_paired_overflow says so."""
    nx, ny, nz = 32, 24, 5
    for n_packets, expect in ((3, False), (5, True)):
        # (ONE packet chunk: a cell's capacity is per work item, and small grids are cut into as many chunks as packets)
        m = make_mapper(ctx, (nx, ny, 25.0, 25.0, 16.0, 12.0), nz, 1.0, 3.0, d.VOTE_LDS_BANDS, band=(0, 1, 1024), packed=8)
        xy = np.zeros((n_packets * 1024, 2), np.float32)
        xy[:] = (2.0, 3.0)
        m.fillVoxelGrid(xy, np.zeros((n_packets, 3), np.float32))
        assert m.paired_overflow() is expect
        if not expect:
            got = m.dsi_.download()
            # (X, Y are the integers only up to the transfer's fp32 rounding on some planes: a weight of 1 - 2^-22)
            assert np.allclose(got[:, 3, 2], 1024.0 * n_packets, rtol=1e-6) and got.sum() == pytest.approx(1024.0 * n_packets * nz, rel=1e-6)
        m.close()
    # an exact mapping never reports
    m = make_mapper(ctx, (nx, ny, 25.0, 25.0, 16.0, 12.0), nz, 1.0, 3.0, d.VOTE_LDS_BANDS, packed=7)
    m.fillVoxelGrid(xy, np.zeros((5, 3), np.float32))
    assert not m.paired_overflow() and np.allclose(m.dsi_.download()[:, 3, 2], 5120.0, rtol=1e-6)
    m.close()


def test_paired_cells_at_configs1_full_size(ctx):
    """BASELINE configs[1] with the opt-in paired cells: every voxel of both camera DSIs within 1e-4 of
    the CPU oracle, no cell at half capacity, the HM-fused volume within 3e-4, and -- with the exact tie resolver, which
    re-sums the contending voxels in the reference's order whatever built the DSIs -- the oracle's index map on every pixel."""
    rig = syn.stereo_rig(10_000_000, seed=1234)
    cam = rig["cam"]
    gpu, cpu, batches = [], [], []
    for c in range(2):
        m = make_mapper(ctx, cam, 100, 4.0, 200.0, d.VOTE_LDS_BANDS, packed=8)
        first, Rt = d.packetize(rig["events"][c][2], rig["trajectories"][c], rig["T_rv_w"])
        b = d.EventBatch(ctx, rig["events"][c][0], rig["events"][c][1], Rt, first)
        m.evaluateDSI_batch(b)
        assert m.last_vote_info()["packed"] == 8 and not m.paired_overflow()
        r = OracleMapper(cam, dimZ=100, min_depth=4.0, max_depth=200.0)
        assert r.evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
        got = m.dsi_.download()
        assert_dsi_close(got, r.dsi)
        err = np.abs(got.astype(np.float64) - r.dsi) / np.maximum(1.0, np.abs(r.dsi))
        print("paired cells, camera %d: max rel err %.3g, busiest voxel %.1f" % (c, err.max(), r.dsi.max()))
        gpu.append(m)
        cpu.append(r)
        batches.append(b)
    fused = d.Grid3D(ctx, 346, 260, 100)
    fused.setToFusionOf(gpu[0].dsi_, gpu[1].dsi_, d.FUSE_HM)
    ref = orc.fuse2(cpu[0].dsi.copy(), cpu[1].dsi, 2)
    assert_dsi_close(fused.download(), ref, tol=3 * DSI_TOL)
    gpu[0].computeDepthMap(fused)
    idx = gpu[0].fetchDepthMap()[2]
    rep = argmax_report(idx, ref, 3 * DSI_TOL)
    assert rep["violations"] == 0, rep
    res = gpu[0].resolveNearTies(gpu, batches, d.FUSE_HM)
    idx = gpu[0].fetchDepthMap()[2]
    print("paired cells + resolver: %r; without: %r" % (res, rep))
    assert res["premise_ok"]
    assert np.array_equal(idx, ref.argmax(axis=0))
    for o in gpu + batches + [fused]:
        o.close()


@pytest.mark.parametrize("packed", [5, 6])
@pytest.mark.parametrize("n_packets", [1, 70, 700, 2500])
def test_inline_cuts_equal_the_cut_table_bit_for_bit(ctx, packed, n_packets):
    """Round 6: from 16 k packets on the vector-fill mapping derives every pass's runs in the voting kernel (two entries
    of the packets' transposed row tables per packet and pass) instead of reading k_plane_coef's cut table -- 6.1 GB per
    camera at configs[4]'s size.  Forced on here at small sizes: same DSI bit for bit, and the oracle's
    (mapper_emvs_stereo.cpp:168-203 visits the same events either way).  Dead packets, one-row packets, d = 0 and
    IEEE-divide planes included; 2,500 packets = many dealt stretches per wave."""
    nx, ny, nz = 160, 120, 9
    rng = np.random.default_rng(4100 + n_packets)
    cam = (nx, ny, 0.8 * nx, 0.8 * nx, 0.5 * nx, 0.5 * ny)
    xy, centers = random_packets(rng, n_packets, nx, ny)
    if n_packets > 12:
        xy[5 * 1024:6 * 1024] = np.nan
        xy[7 * 1024:8 * 1024, 1] = 3.25
        centers[11] = (0.0, 0.0, 1e30)
    got = []
    for inline in (None, 0):
        m = make_mapper(ctx, cam, nz, 1.0, 6.5, d.VOTE_LDS_BANDS, band=(9, 1, 1024), packed=packed, inline_cuts=inline)
        if n_packets > 12 and inline is None:
            centers[9] = (0.1, 0.1, m.raw_depths_vec_[0])
        m.fillVoxelGrid(xy, centers)
        assert m.last_vote_info()["packed"] == packed
        got.append(m.dsi_.download())
        if inline is None and n_packets <= 700:
            ref = orc.fill_voxel_grid(xy, centers, m.raw_depths_vec_, np.array(m.virtual_cam_, np.float32), nx, ny)
            assert_dsi_close(got[0], ref)
        m.close()
    assert np.array_equal(got[0], got[1])
    assert got[0].any()


@pytest.mark.parametrize("packed", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("n_pixels", [1, 7, 300, 5000])
def test_duplicate_events_in_a_packet(ctx, packed, n_pixels):
    """Events of a packet drawn from a few pixels (hot pixels, bursts): the packet sort merges
    bit-identical z0 locations into one record with a multiplicity (k_sort_packets); the DSI
    must be what voting every event separately gives.  -0.0 / +0.0 and NaN keys included."""
    nx, ny, nz = 96, 64, 10
    cam = (nx, ny, 70.0, 70.0, 48.0, 32.0)
    rng = np.random.default_rng(700 + n_pixels)
    n_packets = 6
    pool = np.empty((n_pixels, 2), np.float32)
    pool[:, 0] = rng.uniform(-0.05 * nx, 1.05 * nx, n_pixels)
    pool[:, 1] = rng.uniform(-0.05 * ny, 1.05 * ny, n_pixels)
    xy = pool[rng.integers(0, n_pixels, n_packets * 1024)].copy()
    xy[5] = (0.0, 4.0)
    xy[6] = (-0.0, 4.0)
    xy[7] = (np.nan, 4.0)
    xy[8] = (np.nan, 4.0)
    centers = rng.normal(0, 0.2, (n_packets, 3)).astype(np.float32)
    m = make_mapper(ctx, cam, nz, 1.0, 5.0, d.VOTE_LDS_BANDS, packed=packed)
    ref = orc.fill_voxel_grid(xy, centers, m.raw_depths_vec_, np.array(m.virtual_cam_, np.float32),
                              nx, ny)
    m.fillVoxelGrid(xy, centers)
    assert_dsi_close(m.dsi_.download(), ref)
    m.set_vote_algo(d.VOTE_GLOBAL_ATOMIC)
    m.dsi_.resetGrid()
    m.fillVoxelGrid(xy, centers)
    assert_dsi_close(m.dsi_.download(), ref)
    m.close()


@pytest.mark.parametrize("inverse", [False, True])
def test_plane_sharded_mappers_equal_slices_of_the_full_dsi(ctx, inverse):
    """Plane sharding (SURVEY 8e, configs[4]): a mapper that owns planes [b, b+c) of the depth vector
    builds exactly those planes of the unsharded DSI (same z0, same kernels per plane), and the packed
    (confidence, index) keys of the shards combine to the unsharded arg-max and depth."""
    from dvs_mcemvs_amd import distributed as dd
    rig = syn.stereo_rig(30000, width=96, height=72, duration=0.3, seed=5)
    shape = d.ShapeDSI(0, 0, 23, 4.0, 120.0, 0.0)
    full = d.MapperEMVS(ctx, rig["cam"], shape, inverse_depth=inverse)
    assert full.evaluateDSI(rig["events"][0], rig["trajectories"][0], rig["T_rv_w"])
    ref = full.dsi_.download()
    full.computeDepthMap(full.dsi_)
    depth_ref, conf_ref, idx_ref = full.fetchDepthMap()
    keys = None
    for b, c in dd.plane_ranges(23, 4):
        m = d.MapperEMVS(ctx, rig["cam"], shape, inverse_depth=inverse, plane_range=(b, c))
        assert m.plane_begin == b and m.dimZ == c
        assert np.array_equal(m.raw_depths_vec_, full.raw_depths_vec_[b:b + c])
        assert m.evaluateDSI(rig["events"][0], rig["trajectories"][0], rig["T_rv_w"])
        assert np.array_equal(m.dsi_.download(), ref[b:b + c])          # bit for bit
        m.computeDepthMap(m.dsi_)
        _, conf_l, idx_l = m.fetchDepthMap()
        k = dd.pack_argmax_keys(conf_l, idx_l, b)
        keys = k if keys is None else np.maximum(keys, k)
        m.close()
    conf, idx = dd.unpack_argmax_keys(keys)
    assert np.array_equal(conf, conf_ref) and np.array_equal(idx, idx_ref)
    # index -> depth with the full depth vector (mapper_emvs_stereo.cpp:302-313)
    assert np.array_equal(orc.indices_to_depth(idx, full.raw_depths_vec_), depth_ref)
    with pytest.raises(d.DsiError):
        d.MapperEMVS(ctx, rig["cam"], shape, plane_range=(20, 9))
    full.close()


@pytest.mark.parametrize("seed", range(16))
def test_hand_scheduled_loops_equal_the_compiled_loop_bit_for_bit(ctx, seed):
    """The assembly wave loops (lane mappings 1 and 4) do the compiled loop's arithmetic instruction
    for instruction and accumulate in exact fixed point, so on ANY input -- random grid sizes, band
    heights, chunk counts, block sizes, poses that push most events off the grid, planes behind the
    camera -- their DSIs must equal mapping 3's bit for bit (and mapping 0's, the per-packet loop)."""
    rng = np.random.default_rng(9000 + seed)
    nx = int(rng.integers(8, 400))
    ny = int(rng.integers(8, 300))
    nz = int(rng.integers(1, 40))
    n_packets = int(rng.integers(1, 70))
    cam = (nx, ny, float(rng.uniform(0.5, 2.0) * nx), float(rng.uniform(0.5, 2.0) * nx),
           0.5 * nx, 0.5 * ny)
    xy, centers = random_packets(rng, n_packets, nx, ny, spread=float(rng.uniform(0.01, 2.0)),
                                 cz_spread=float(rng.uniform(0.01, 3.0)))
    if seed % 2:
        pool = xy[rng.integers(0, xy.shape[0], 200)]                  # heavy duplicates
        xy[: xy.shape[0] // 2] = pool[rng.integers(0, 200, xy.shape[0] // 2)]
    xy[rng.integers(0, xy.shape[0], 20)] = np.nan
    xy[rng.integers(0, xy.shape[0], 20)] = np.inf
    xy[rng.integers(0, xy.shape[0], 5)] = 3.0e38
    band = (int(rng.integers(1, 12)), int(rng.integers(1, 6)), int(rng.choice([256, 512, 1024])))
    m_max = float(rng.uniform(2.0, 9.0))
    ref = None
    # every voxel is fl(exact 64-bit sum) whatever the lane mapping, band height or packet chunking (round 3: the
    # chunks' partial volumes are raw 64-bit sums), so ALL seven mappings give the same bits -- the grouped
    # mappings 2 / 4 too, although they cut the packets into chunks at other places
    # (5 / 6 twice: with the cut table and with the runs derived in the kernel from the transposed row tables, round 6)
    for packed, inline in ((3, None), (1, None), (7, None), (0, None), (5, None), (6, None), (2, None), (4, None), (5, 0), (6, 0)):
        m = make_mapper(ctx, cam, nz, 0.8, m_max, d.VOTE_LDS_BANDS, band=band, packed=packed, inline_cuts=inline)
        m.fillVoxelGrid(xy, centers)
        got = m.dsi_.download()
        m.close()
        if ref is None:
            ref = got
            orc_ref = orc.fill_voxel_grid(xy, centers, m.raw_depths_vec_, np.array(m.virtual_cam_, np.float32), nx, ny)
            assert_dsi_close(got, orc_ref)
        else:
            assert np.array_equal(got, ref), "lane mapping %d differs from mapping 3: max %g" % (
                packed, np.abs(got.astype(np.float64) - ref).max())


def test_two_contexts_meet_at_the_fusion(ctx):
    """Each camera's mapper on its own context (HIP stream); dsi_context_wait_for orders the streams on
    the device; grid ops accept operands of another context of the same device."""
    rig = syn.stereo_rig(20000, width=80, height=60, duration=0.3, seed=3)
    shape = d.ShapeDSI(0, 0, 12, 4.0, 100.0, 0.0)
    ctx2 = d.Context(0)
    m0 = d.MapperEMVS(ctx, rig["cam"], shape)
    m1 = d.MapperEMVS(ctx2, rig["cam"], shape)
    fused = d.Grid3D(ctx, 80, 60, 12)
    for rep in range(3):
        ctx2.wait_for(ctx)                      # the previous fusion has read m1.dsi_
        assert m0.evaluateDSI(rig["events"][0], rig["trajectories"][0], rig["T_rv_w"])
        assert m1.evaluateDSI(rig["events"][1], rig["trajectories"][1], rig["T_rv_w"])
        ctx.wait_for(ctx2)
        fused.setToFusionOf(m0.dsi_, m1.dsi_, d.FUSE_HM)
    got = fused.download()
    ref = orc.fuse2(m0.dsi_.download(), m1.dsi_.download(), 2)
    assert np.array_equal(got, ref)
    for o in (m0, m1, fused, ctx2):
        o.close()


def test_depth_map_of_a_grid_in_another_context(ctx):
    """A mapper in context A extracts the depth map of a grid that context B is still producing
    (the `mapper_fused` pattern of the pipelined temporal fusion): the arg-max must be ordered
    after B's queued work on the device, without the host waiting."""
    nx, ny, nz = 512, 512, 200
    ctx_b = d.Context(0)
    cam = (nx, ny, 400.0, 400.0, 256.0, 256.0)
    m = d.MapperEMVS(ctx, cam, d.ShapeDSI(0, 0, nz, 1.0, 10.0, 0.0))
    rng = np.random.default_rng(8)
    vol = rng.gamma(2.0, 3.0, (nz, ny, nx)).astype(np.float32)
    src = d.Grid3D(ctx_b, nx, ny, nz)
    src.upload(vol)
    g = d.Grid3D(ctx_b, nx, ny, nz)
    for rep in range(3):
        g.resetGrid()
        for _ in range(40):                    # ~10 ms of queued work on B's stream
            g.addTwoGrids(src)
        g.computeAMfromSum(1)                   # g = 40 * vol, finished only when all of it ran
        m.computeDepthMap(g)                    # queued on A's stream right away
        depth, conf, idx = m.fetchDepthMap()
        got = g.download()
        rconf, ridx = orc.collapse_max_z(got)
        assert np.array_equal(idx, ridx) and np.array_equal(conf, rconf)
        assert conf.max() > 39.0 * vol.max() - 1.0   # not a stale / partial volume
        ctx_b.wait_for(ctx)                     # B overwrites g only after A has read it
    for o in (m, src, g, ctx_b):
        o.close()


def test_configs1_against_the_oracle_at_full_size(ctx):
    """BASELINE.json configs[1] exactly (stereo, 10 M events per camera, 346x260x100, harmonic camera
    fusion, arg-max + depth) against the CPU oracle: every voxel of both DSIs and of the fused DSI
    within the stated tolerance; depth equal (<= 1e-4) wherever the CPU's arg-max is not a near-tie."""
    rig = syn.stereo_rig(10_000_000, seed=1234)
    cam = rig["cam"]
    gpu, cpu = [], []
    for c in range(2):
        m = make_mapper(ctx, cam, 100, 4.0, 200.0, d.VOTE_LDS_BANDS)
        assert m.evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
        assert m.n_voted == 9765 * 1024
        gpu.append(m)
        r = OracleMapper(cam, dimZ=100, min_depth=4.0, max_depth=200.0)
        assert r.evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
        cpu.append(r)
        assert_dsi_close(m.dsi_.download(), r.dsi)
    fused = d.Grid3D(ctx, 346, 260, 100)
    fused.setToFusionOf(gpu[0].dsi_, gpu[1].dsi_, d.FUSE_HM)
    ref = orc.fuse2(cpu[0].dsi.copy(), cpu[1].dsi, 2)
    got = fused.download()
    assert_dsi_close(got, ref, tol=3 * DSI_TOL)      # HM of two values each within DSI_TOL
    gpu[0].computeDepthMap(fused)
    depth, conf, idx = gpu[0].fetchDepthMap()
    rconf, ridx = orc.collapse_max_z(ref)
    srt = np.sort(ref, axis=0)
    safe = (srt[-1] - srt[-2]) > 6 * DSI_TOL * np.maximum(1.0, srt[-1])
    # what the statement "depth map equal to the CPU reference" covers: the pixels whose CPU top-2 gap
    # exceeds the DSI tolerance (elsewhere either plane is a correct arg-max of an equally valid DSI)
    agree = float((idx == ridx).mean())
    print("configs[1] arg-max: %.4f of the pixels have a top-2 gap above 6x tolerance; indices agree on %.5f of "
          "ALL pixels" % (safe.mean(), agree))
    assert safe.mean() > 0.95 and agree > 0.99      # measured: 0.961 safe
    assert np.array_equal(idx[safe], ridx[safe])
    assert (np.abs(depth - orc.indices_to_depth(ridx, cpu[0].planes))[safe] <= 1e-4).all()
    # EVERY pixel: equal index, or the GPU's plane is a provable near-tie of the oracle's fused column
    # (fused volume within 3 x DSI_TOL, asserted above => ref[idx_gpu] >= ref_max - 6 x DSI_TOL x max(1, ref_max))
    rep = argmax_report(idx, ref, 3 * DSI_TOL)
    print("configs[1] depth map, all %d pixels: %r" % (rep["pixels"], rep))
    assert rep["violations"] == 0, rep
    assert np.array_equal(depth, orc.indices_to_depth(idx, cpu[0].planes))   # depth = plane of the index, exactly
    # ... and with the exact tie resolver the index map IS the oracle's, on all 89,960 pixels:
    # the contending voxels of the near-tie columns re-summed in the reference's order (fp32, event order)
    batches = []
    for c in range(2):
        first, Rt = d.packetize(rig["events"][c][2], rig["trajectories"][c], rig["T_rv_w"])
        batches.append(d.EventBatch(ctx, rig["events"][c][0], rig["events"][c][1], Rt, first))
    info = gpu[0].resolveNearTies(gpu, batches, d.FUSE_HM)
    depth2, conf2, idx2 = gpu[0].fetchDepthMap()
    print("configs[1] exact tie resolver: %r; %d pixels differed before" % (info, int((idx != ridx).sum())))
    assert np.array_equal(idx2, ridx)
    assert np.array_equal(depth2, orc.indices_to_depth(ridx, cpu[0].planes))
    assert np.array_equal(conf2[idx != ridx], rconf[idx != ridx])
    assert info["changed_pixels"] == int((idx != ridx).sum())
    assert 8 * info["max_order_diff"] < info["rel_gap"], info   # the premise of the gap, checked on the re-summed voxels
    for o in gpu + [fused] + batches:
        o.close()
