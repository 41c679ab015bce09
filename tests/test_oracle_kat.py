"""Known-answer tests of the CPU oracle (CPU only).

The reference has no tests of its own, so these are analytic /
geometric known answers that WE authored.  They check the oracle's restatement against
closed forms and against first-principles geometry (a 3-D point must be re-found by the
space sweep), independently of the algebra in oracle/dsi_oracle.c.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle_pipeline import OracleMapper
from dvs_mcemvs_amd import synthetic as syn


# ---------------------------------------------------------------- depth planes
def test_linear_depth_planes():
    z = orc.depth_planes(4.0, 200.0, 100, inverse=False)
    assert z[0] == 4.0
    step = (200.0 - 4.0) / 100
    assert np.allclose(z, 4.0 + step * np.arange(100), rtol=1e-6)
    assert z[-1] == pytest.approx(200.0 - step, rel=1e-6)  # last plane is max - step, never max
    # depth_vector.hpp:33-36: swapped bounds are put in order
    assert np.array_equal(orc.depth_planes(200.0, 4.0, 100), z)


def test_inverse_depth_planes():
    z = orc.depth_planes(1.0, 10.0, 9, inverse=True)
    rho = 1.0 / 10 + np.arange(9) * (1.0 - 0.1) / 9
    assert np.allclose(z, 1.0 / rho, rtol=1e-6)
    assert z[0] == pytest.approx(10.0, rel=1e-6)  # index 0 is the FAR plane
    assert np.all(np.diff(z) < 0)


def test_virtual_focal():
    assert orc.virtual_focal(123.5, 0.0, 346) == pytest.approx(123.5)
    assert orc.virtual_focal(123.5, 9.99, 346) == pytest.approx(123.5)
    assert orc.virtual_focal(123.5, 90.0, 346) == pytest.approx(173.0, rel=1e-6)  # 0.5*W/tan(45deg)


# ------------------------------------------------------------------------ vote
def test_bilinear_vote_weights_and_border():
    p = np.zeros((5, 6), np.float32)
    orc.vote(2.25, 3.5, p)
    assert p[3, 2] == 0.375 and p[3, 3] == 0.125 and p[4, 2] == 0.375 and p[4, 3] == 0.125
    assert p.sum() == 1.0
    q = np.zeros((5, 6), np.float32)
    for xy in [(5.0, 1.0), (5.0001, 1.0), (1.0, 4.0), (-1e-6, 1.0), (1.0, -0.5), (np.nan, 1.0),
               (1.0, np.nan), (np.inf, 1.0), (1.0, 1e30), (3e9, 3e9)]:
        orc.vote(xy[0], xy[1], q)       # x+1 < Nx / y+1 < Ny / >= 0 violated (cartesian3dgrid.h:255-259)
    assert not q.any()
    orc.vote(5.0 - 1e-3, 4.0 - 1e-3, q)  # last accepted cell
    assert q.sum() == pytest.approx(1.0, rel=1e-6) and q[3, 4] > 0
    orc.vote(-0.0, 0.0, q)               # -0.0 >= 0 is true
    assert q[0, 0] == 1.0


def test_survey_known_answer_hm():
    """a.vote(2.25,3.5); b.vote(2.75,3.5); a.harmonicMeanTwoGrids(b) -> 0.15625
    at (2,3): 2*0.375*0.125/(0.375+0.125+0.1)."""
    a = np.zeros((1, 6, 6), np.float32)
    b = np.zeros((1, 6, 6), np.float32)
    orc.vote(2.25, 3.5, a[0])
    orc.vote(2.75, 3.5, b[0])
    f = orc.fuse2(a, b, 2)
    assert f[0, 3, 2] == pytest.approx(0.15625, rel=1e-6)


# -------------------------------------------------------------- fillVoxelGrid
def test_identity_pose_votes_every_plane_at_z0_location():
    nx, ny, nz = 20, 16, 7
    planes = orc.depth_planes(1.0, 3.0, nz)
    Kv = np.array([15, 15, 10, 8], np.float32)
    xy = np.tile(np.array([[4.25, 6.5]], np.float32), (1024, 1))
    dsi = orc.fill_voxel_grid(xy, np.zeros((1, 3), np.float32), planes, Kv, nx, ny)
    for z in range(nz):
        assert dsi[z, 6, 4] == pytest.approx(1024 * 0.375)
        assert dsi[z].sum() == pytest.approx(1024.0)


def test_pure_x_translation_closed_form():
    """C = (c,0,0): a = z0*zi, bx = (z0-zi)*c*fx, d = zi*z0 => X_i = x0 + (z0-zi)*c*fx/(zi*z0),
    Y_i = y0 (from mapper_emvs_stereo.cpp:177-195)."""
    nx, ny, nz = 64, 20, 9
    planes = orc.depth_planes(2.0, 11.0, nz)
    fx = 40.0
    Kv = np.array([fx, fx, 32, 10], np.float32)
    c = 0.5
    x0, y0 = 30.0, 7.0
    xy = np.tile(np.array([[x0, y0]], np.float32), (1024, 1))
    dsi = orc.fill_voxel_grid(xy, np.array([[c, 0, 0]], np.float32), planes, Kv, nx, ny)
    z0 = float(planes[0])
    for i, zi in enumerate(planes.astype(np.float64)):
        X = x0 + (z0 - zi) * c * fx / (zi * z0)
        xi = int(np.floor(X))
        w = 1.0 - (X - xi)
        assert dsi[i, 7, xi] == pytest.approx(1024 * w, rel=2e-4, abs=0.05)
        assert dsi[i, 8].sum() == 0.0  # y0 integer: nothing leaks into the next row
        assert dsi[i].sum() == pytest.approx(1024.0, rel=1e-4)  # fp32 += rounds at every vote


def test_omp_and_serial_paths_agree():
    """mapper_emvs_stereo.cpp:168: `omp parallel for if (n >= 20000)`; planes are independent,
    so the threaded and the serial path give the same bits."""
    rng = np.random.default_rng(1)
    nx, ny = 50, 40
    planes = orc.depth_planes(1.0, 5.0, 6)
    Kv = np.array([40, 40, 25, 20], np.float32)
    xy = rng.uniform(-5, 55, (20 * 1024, 2)).astype(np.float32)
    centers = rng.normal(0, 0.2, (20, 3)).astype(np.float32)
    big = orc.fill_voxel_grid(xy, centers, planes, Kv, nx, ny)  # 20480 events -> threaded
    acc = np.zeros_like(big)
    for k in range(20):                                         # 1024 events -> serial
        orc.fill_voxel_grid(xy[k * 1024:(k + 1) * 1024], centers[k:k + 1], planes, Kv, nx, ny, acc)
    assert np.array_equal(big, acc)


# ------------------------------------------------- geometry from first principles
def _project(K, R, t, P):
    pc = R @ P + t
    return np.array([K[0] * pc[0] / pc[2] + K[2], K[1] * pc[1] / pc[2] + K[3]])


def test_homography_maps_plane_z0_points():
    """A 3-D point on plane Z = z0 of the reference view, seen by the event camera at pixel u,
    must be warped by H_z0_px to its virtual-camera pixel (mapper_emvs_stereo.cpp:113-141)."""
    rng = np.random.default_rng(3)
    K = np.array([200.0, 210.0, 160.0, 120.0], np.float32)
    Kv = np.array([180.0, 180.0, 160.0, 120.0], np.float32)
    z0 = 2.5
    for _ in range(20):
        ang = rng.normal(0, 0.05, 3)
        th = np.linalg.norm(ang)
        k = ang / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx   # T_ev_rv rotation
        t = rng.normal(0, 0.2, 3)
        Rt = np.concatenate([R.reshape(-1), t]).astype(np.float32)
        centers, H = orc.packet_geometry(Rt[None], K, Kv, z0)
        assert np.allclose(centers[0], -R.T @ t, atol=1e-5)
        H = H[0].reshape(3, 3).astype(np.float64)
        for _ in range(5):
            uv_virtual = rng.uniform([20, 20], [300, 220])
            P = np.array([(uv_virtual[0] - Kv[2]) / Kv[0] * z0, (uv_virtual[1] - Kv[3]) / Kv[1] * z0, z0])
            u = _project(K, R, t, P)
            w = H @ np.array([u[0], u[1], 1.0])
            assert np.allclose(w[:2] / w[2], uv_virtual, atol=2e-2)


def test_space_sweep_refinds_a_scene_point():
    """Events generated by ONE 3-D point seen from a moving camera: all back-projected rays
    intersect at the point, so the DSI maximum is at its pixel and on its depth plane."""
    cam = (64, 48, 60.0, 60.0, 32.0, 24.0)
    m = OracleMapper(cam, dimZ=20, min_depth=1.0, max_depth=5.0)
    k_true = 7
    depth = float(m.planes[k_true])
    P_rv = np.array([0.3, -0.2, depth])                     # in the reference view
    n_packets = 12
    Rt = np.zeros((n_packets, 12), np.float32)
    ex = np.zeros(n_packets * 1024, np.uint16)
    ey = np.zeros(n_packets * 1024, np.uint16)
    lut = np.zeros((48 * 64, 2), np.float32)                # sub-pixel "undistorted" positions
    for p in range(n_packets):
        c = np.array([-0.5 + p / (n_packets - 1.0), 0.05 * np.sin(p), 0.0])  # camera centre in rv
        R = np.eye(3)                                       # T_ev_rv: x_ev = R x_rv + t, t = -R c
        t = -R @ c
        Rt[p, :9] = R.reshape(-1)
        Rt[p, 9:] = t
        u = _project(cam[2:], R, t, P_rv)
        px, py = 3 + p, 5                                   # a distinct raw pixel per packet ...
        lut[py * 64 + px] = u                               # ... whose LUT entry is the exact projection
        ex[p * 1024:(p + 1) * 1024] = px
        ey[p * 1024:(p + 1) * 1024] = py
    m.lut = lut
    first = np.arange(n_packets, dtype=np.int64) * 1024
    m.evaluate_packets(ex, ey, first, Rt)
    depth_map, conf, idx = m.depth_map()
    v, u = np.unravel_index(conf.argmax(), conf.shape)
    u_true = cam[2] * P_rv[0] / depth + cam[4]
    v_true = cam[3] * P_rv[1] / depth + cam[5]
    assert abs(u - u_true) <= 1 and abs(v - v_true) <= 1
    assert idx[v, u] == k_true
    assert depth_map[v, u] == m.planes[k_true]
    assert conf[v, u] > 0.2 * n_packets * 1024              # most votes meet in one voxel


# ---------------------------------------------------------------- packetisation
def test_packet_counts():
    assert orc.packetize(1023) is None
    for n, expect in [(1024, 0), (1025, 1), (2048, 1), (2049, 2), (10 * 1024 + 1, 10)]:
        first, mid = orc.packetize(n)
        assert first.shape[0] == expect
        assert np.array_equal(first, np.arange(expect) * 1024)
        assert np.array_equal(mid, first + 512)


def test_pose_failure_slides_by_one_event():
    n = 5000
    ok = np.ones(n, np.uint8)
    ok[:700] = 0                      # mid timestamps 512..699 fail -> cur advances 188 times
    first, mid = orc.packetize(n, ok)
    assert first[0] == 700 - 512 and mid[0] == 700
    assert np.array_equal(np.diff(first), np.full(first.shape[0] - 1, 1024))
    assert first[-1] + 1024 < n


# ------------------------------------------------------------- fusion, arg-max
def test_fusion_formulas():
    rng = np.random.default_rng(8)
    a = rng.gamma(2, 5, 5000).astype(np.float32)
    g = rng.gamma(2, 5, 5000).astype(np.float32)
    a[::9] = 0
    g[::4] = 0
    A, G = a.astype(np.float64), g.astype(np.float64)
    assert np.array_equal(orc.fuse2(a, g, 1), np.minimum(a, g))
    assert np.array_equal(orc.fuse2(a, g, 6), np.maximum(a, g))
    assert np.allclose(orc.fuse2(a, g, 2), 2 * A * G / (A + G + np.float32(0.1)), rtol=3e-7)
    assert np.allclose(orc.fuse2(a, g, 3), np.sqrt(A * G), rtol=2e-7)
    assert np.array_equal(orc.fuse2(a, g, 4), ((a + g) * np.float32(0.5)))
    assert np.allclose(orc.fuse2(a, g, 5), np.sqrt(0.5 * (A * A + G * G)), rtol=2e-7)
    # n-ary harmonic mean (cartesian3dgrid.h:130-139)
    av = A / 2
    assert np.allclose(orc.fuse_hm_n(a, g, 3), 3 * av * G / (av + G + np.float32(0.1)), rtol=4e-7)
    with pytest.raises(ValueError):
        orc.fuse2(a, g, 7)
    # eps sits in the denominator only: both zero -> 0, one zero -> 0
    z = np.zeros(3, np.float32)
    assert not orc.fuse2(z, z, 2).any()
    assert not orc.fuse2(z, np.ones(3, np.float32), 2).any()


def test_temporal_accumulators():
    g = np.array([0.0, 1.0, 3.0], np.float32)
    acc = np.zeros(3, np.float32)
    for _ in range(4):
        acc = orc.accumulate(acc, g, 1)
    hm = orc.finalize(acc, 1, 4)
    assert np.allclose(hm, 0.01 + g, rtol=1e-6)  # HM of n equal maps v is v + eps (eps = 1e-2)
    acc = np.zeros(3, np.float32)
    for k in range(4):
        acc = orc.accumulate(acc, g * (k + 1), 0)
    assert np.allclose(orc.finalize(acc, 0, 4), g * 2.5)


def test_argmax_first_maximum_wins():
    v = np.zeros((5, 2, 3), np.float32)
    v[1, 0, 0] = 2
    v[3, 0, 0] = 2                     # tie: first wins
    v[4, 1, 2] = 9                     # last plane
    conf, idx = orc.collapse_max_z(v)
    assert idx[0, 0] == 1 and conf[0, 0] == 2
    assert idx[1, 2] == 4 and conf[1, 2] == 9
    assert idx[0, 1] == 0 and conf[0, 1] == 0   # empty column -> conf 0, index 0
    planes = orc.depth_planes(1.0, 6.0, 5)
    assert np.array_equal(orc.indices_to_depth(idx, planes), planes[idx])
    assert orc.mean_square(v) == pytest.approx((4 + 4 + 81) / 30.0)


# ---------------------------------------------------------------- pose pipeline
def test_pose_interpolation():
    times = np.array([0.0, 1.0, 2.0])
    th = 0.4
    poses = np.array([[0, 0, 0, 1, 0, 0, 0],
                      [2, 0, 0, 1, 0, 0, 0],
                      [2, 0, 4, np.cos(th / 2), 0, np.sin(th / 2), 0]], float)
    assert np.allclose(orc.pose_at(times, poses, 0.25), [0.5, 0, 0, 1, 0, 0, 0])
    mid = orc.pose_at(times, poses, 1.5)   # translation linear, rotation half angle (minkindr)
    assert np.allclose(mid, [2, 0, 2, np.cos(th / 4), 0, np.sin(th / 4), 0], atol=1e-12)
    assert orc.pose_at(times, poses, 0.0) is not None          # t == first time: upper_bound ok
    assert orc.pose_at(times, poses, 2.0) is None              # t == last time: end() -> false
    assert orc.pose_at(times, poses, -0.1) is None and orc.pose_at(times, poses, 2.1) is None


def test_event_pose_Rt_is_inverse_of_relative_pose():
    rig = syn.stereo_rig(2048, width=32, height=24, duration=0.2, seed=1)
    times, poses = rig["trajectories"][1]
    T_w_ev = orc.pose_at(times, poses, rig["t0"] + 0.1)
    Rt = orc.event_pose_Rt(rig["T_rv_w"], T_w_ev)
    R = Rt[:9].reshape(3, 3).astype(np.float64)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6)
    # camera centre in rv coordinates = -R^T t must equal T_rv_w * (position of the camera in w)
    c_rv = -R.T @ Rt[9:].astype(np.float64)
    q = rig["T_rv_w"][3:]
    expect = rig["T_rv_w"][:3] + syn.quat_rotate(q, T_w_ev[:3])
    assert np.allclose(c_rv, expect, atol=1e-5)


def test_pose_pipeline_against_scipy_rotations():
    """Stage A's SE(3) arithmetic (quaternion products, inverse, the SO(3) log / exp of LinearTrajectory::getPoseAt,
    trajectory.hpp:92-126, and the rotation matrix + translation of mapper_emvs_stereo.cpp:101-105) against a
    third-party implementation nobody here wrote: scipy.spatial.transform.Rotation.  Random, LARGE rotations (up to
    170 degrees between control poses, both quaternion signs) -- conventions (Hamilton product, w-first storage,
    active rotation, T1 * T2 = (q1 q2, t1 + q1 t2)) that a small-angle test cannot tell apart.  The oracle and the
    engine's host code (dsi_pose_at, dsi_packetize) are both checked."""
    from scipy.spatial.transform import Rotation as R
    import dvs_mcemvs_amd as d

    rng = np.random.default_rng(42)

    def rand_pose(max_angle=np.pi):
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        ang = rng.uniform(0.05, max_angle)
        q_xyzw = R.from_rotvec(axis * ang).as_quat()
        q = np.array([q_xyzw[3], q_xyzw[0], q_xyzw[1], q_xyzw[2]])     # our storage: w, x, y, z
        if rng.uniform() < 0.5:
            q = -q                                                     # the other sign of the same rotation
        return np.concatenate([rng.uniform(-3, 3, 3), q])

    def rot(p):
        return R.from_quat([p[4], p[5], p[6], p[3]])                   # scipy storage: x, y, z, w

    for _ in range(200):
        T0 = rand_pose()
        # second control pose: T0 composed with a relative rotation of up to 170 degrees
        rel = rand_pose(np.deg2rad(170))
        r1 = rot(T0) * rot(rel)
        q1 = r1.as_quat()
        T1 = np.concatenate([T0[:3] + rot(T0).apply(rel[:3]), [q1[3], q1[0], q1[1], q1[2]]])
        times = np.array([1.0, 3.0])
        poses = np.stack([T0, T1])
        t = rng.uniform(1.0, 3.0)
        delta = (t - 1.0) / 2.0
        # getPoseAt: T0 * exp(delta * log(T0^-1 T1)); minkindr's log / exp treat the translation linearly
        r_rel = rot(T0).inv() * rot(T1)
        t_rel = rot(T0).inv().apply(T1[:3] - T0[:3])
        r_t = rot(T0) * R.from_rotvec(delta * r_rel.as_rotvec())
        p_t = T0[:3] + rot(T0).apply(delta * t_rel)
        for got in (orc.pose_at(times, poses, t), d.pose_at((times, poses), t)):
            assert got is not None
            assert np.allclose(got[:3], p_t, atol=1e-12)
            assert abs(np.linalg.norm(got[3:]) - 1.0) < 1e-12
            assert (rot(got) * r_t.inv()).magnitude() < 1e-12           # the same rotation, whatever the sign
        # T_ev_rv = (T_rv_w * T_w_ev)^-1 as (R row-major, t) in float
        T_rv_w = rand_pose()
        T_w_ev = np.concatenate([p_t, [r_t.as_quat()[3], *r_t.as_quat()[:3]]])
        r_c = rot(T_rv_w) * r_t
        p_c = T_rv_w[:3] + rot(T_rv_w).apply(p_t)
        R_inv, t_inv = r_c.inv().as_matrix(), -r_c.inv().apply(p_c)
        Rt = orc.event_pose_Rt(T_rv_w, T_w_ev)
        assert np.allclose(Rt[:9].reshape(3, 3), R_inv, atol=2e-7) and np.allclose(Rt[9:], t_inv, atol=2e-6)
    # the callers' pose algebra (process.pose_mul, pose_inverse, TrajectoryBase::applyTransformationRight / Left)
    from dvs_mcemvs_amd import process as proc
    for _ in range(50):
        A, B = rand_pose(), rand_pose()
        AB = proc.pose_mul(A, B)
        assert np.allclose(AB[:3], A[:3] + rot(A).apply(B[:3]), atol=1e-12)
        assert (rot(AB) * (rot(A) * rot(B)).inv()).magnitude() < 1e-12
        Ai = syn.pose_inverse(A)
        assert np.allclose(proc.pose_mul(A, Ai)[:3], 0, atol=1e-12) and rot(proc.pose_mul(A, Ai)).magnitude() < 1e-12
        tr = (np.array([0.0, 1.0]), np.stack([A, B]))
        right, left = proc.apply_transformation_right(tr, B), proc.apply_transformation_left(tr, B)
        assert np.allclose(right[1][0], proc.pose_mul(A, B)) and np.allclose(left[1][0], proc.pose_mul(B, A))
    # the packets of a whole camera: every Rt of dsi_packetize against scipy's interpolation at the packet's middle event
    rig = syn.stereo_rig(6000, width=40, height=30, duration=0.3, seed=8)
    x, y, ts = rig["events"][0]
    times, poses = rig["trajectories"][0]
    first, Rts = d.packetize(ts, (times, poses), rig["T_rv_w"])
    assert first.shape[0] >= 4
    for k in range(first.shape[0]):
        t = ts[first[k] + 512]
        i = int(np.searchsorted(times, t, side="right"))              # std::map::upper_bound (trajectory.hpp:98)
        T0, T1 = poses[i - 1], poses[i]
        delta = (t - times[i - 1]) / (times[i] - times[i - 1])
        r_rel = rot(T0).inv() * rot(T1)
        r_t = rot(T0) * R.from_rotvec(delta * r_rel.as_rotvec())
        p_t = T0[:3] + rot(T0).apply(delta * rot(T0).inv().apply(T1[:3] - T0[:3]))
        r_c = rot(rig["T_rv_w"]) * r_t
        p_c = rig["T_rv_w"][:3] + rot(rig["T_rv_w"]).apply(p_t)
        assert np.allclose(Rts[k][:9].reshape(3, 3), r_c.inv().as_matrix(), atol=2e-7)
        assert np.allclose(Rts[k][9:], -r_c.inv().apply(p_c), atol=2e-6)


def _ulp(x, y):
    return np.abs(x.view(np.int32).astype(np.int64) - y.view(np.int32).astype(np.int64))


def test_nary_fusion_modes_against_float64_and_the_two_ary_ops():
    """The n-ary accumulate / finalize modes (an extension: the reference is 2-ary,
    cartesian3dgrid.h:111-190, and drops camera 3 for GM/AM/RMS, process1.cpp:169-191) against
    float64 numpy, and for n = 2 against the restated 2-ary ops."""
    rng = np.random.default_rng(5)
    maps = [rng.gamma(2.0, 8.0, 20000).astype(np.float32) for _ in range(4)]
    for k, v in enumerate(maps):
        v[k::13] = 0.0
    st = np.stack(maps).astype(np.float64)
    gm = orc.fuse_nary(maps, 2)
    with np.errstate(divide="ignore"):
        ref = np.exp(np.mean(np.log(st), axis=0))
    assert np.all(gm[(st == 0).any(axis=0)] == 0.0)
    nz = ref > 0
    assert np.max(np.abs(gm[nz] - ref[nz]) / ref[nz]) < 4e-6      # fp32 sum of four fp32 logs
    rms = orc.fuse_nary(maps, 3)
    assert np.max(np.abs(rms - np.sqrt(np.mean(st * st, axis=0))) / np.maximum(1.0, rms)) < 1e-6
    assert np.array_equal(orc.fuse_nary(maps, 4), np.min(np.stack(maps), axis=0))
    assert np.array_equal(orc.fuse_nary(maps, 5), np.max(np.stack(maps), axis=0))
    am = orc.fuse_nary(maps, 0)
    assert np.max(np.abs(am - st.mean(axis=0))) < 1e-4
    a, g = maps[0], maps[1]
    assert np.array_equal(orc.fuse_nary([a, g], 0), orc.fuse2(a, g, 4))
    assert np.array_equal(orc.fuse_nary([a, g], 4), orc.fuse2(a, g, 1))
    assert np.array_equal(orc.fuse_nary([a, g], 5), orc.fuse2(a, g, 6))
    r5 = orc.fuse2(a, g, 5)
    assert _ulp(orc.fuse_nary([a, g], 3), r5).max() <= 2
    r3 = orc.fuse2(a, g, 3)
    ok = (a > 2.0 ** -20) & (g > 2.0 ** -20)
    assert _ulp(orc.fuse_nary([a, g], 2)[ok], r3[ok]).max() <= 16
    assert np.all(orc.fuse_nary([a, g], 2)[~ok & ((a == 0) | (g == 0))] == 0.0)
    # identities: a rank that owns no map contributes the identity to the all-reduce
    for mode, ident in ((0, 0.0), (2, 0.0), (3, 0.0), (4, np.inf), (5, -np.inf)):
        assert np.all(orc.accumulate_begin((3, 2), mode) == ident)


def test_c_oracle_equals_the_independent_numpy_restatement_bit_for_bit():
    """oracle/dsi_oracle.c against tests/independent_numpy.py -- a second restatement of the same
    reference lines, vectorised and differently structured: depth planes, per-packet geometry, z0
    warp (identity and LUT), fillVoxelGrid (sequential vote order), every fusion op, the temporal
    accumulators and the arg-max.  Bit equality of two independently written programs is what
    stands in for the golden vectors the reference does not have."""
    import independent_numpy as ind
    from dvs_mcemvs_amd import synthetic as syn
    rng = np.random.default_rng(11)
    for inverse in (False, True):
        for nz in (1, 7, 100, 256):
            assert np.array_equal(orc.depth_planes(0.7, 31.0, nz, inverse), ind.depth_planes(0.7, 31.0, nz, inverse))
    nx, ny, nz, npk = 96, 72, 12, 5
    cam = syn.camera(nx, ny)
    K = np.array(cam[2:], np.float32)
    Kv = np.array([K[0], K[0], K[2], K[3]], np.float32)
    planes = orc.depth_planes(1.0, 7.0, nz)
    # per-packet poses: small rotations + translations
    Rt = np.empty((npk, 12), np.float32)
    for k in range(npk):
        ang = rng.normal(0, 0.02, 3)
        Rx = np.array([[1, 0, 0], [0, np.cos(ang[0]), -np.sin(ang[0])], [0, np.sin(ang[0]), np.cos(ang[0])]])
        Ry = np.array([[np.cos(ang[1]), 0, np.sin(ang[1])], [0, 1, 0], [-np.sin(ang[1]), 0, np.cos(ang[1])]])
        Rt[k, :9] = (Rx @ Ry).astype(np.float32).reshape(-1)
        Rt[k, 9:] = rng.normal(0, 0.15, 3)
    centers_c, H_c = orc.packet_geometry(Rt, K, Kv, planes[0])
    ex = rng.integers(0, nx, npk * 1024).astype(np.uint16)
    ey = rng.integers(0, ny, npk * 1024).astype(np.uint16)
    lut = syn.radial_lut(cam)
    for use_lut in (None, lut):
        xy_c = orc.warp_z0(ex, ey, H_c, use_lut, nx)
        for k in range(npk):
            C, H = ind.packet_geometry(Rt[k], K, Kv, planes[0])
            assert np.array_equal(C, centers_c[k]) and np.array_equal(H.reshape(-1), H_c[k])
            sl = slice(k * 1024, (k + 1) * 1024)
            assert np.array_equal(ind.warp_z0(ex[sl], ey[sl], H, use_lut, nx), xy_c[sl])
    xy = orc.warp_z0(ex, ey, H_c, None, nx)
    xy[7] = (np.nan, 3.0)                       # rejected by the >= tests
    xy[9] = (np.inf, 3.0)
    xy[11] = (3.0e38, 3.0)
    xy[13] = (-0.0, 2.5)                        # accepted (>= 0)
    xy[15] = (nx - 1.0, 2.5)                    # x + 1 == nx: rejected
    centers = centers_c.copy()
    centers[3] = (0.1, 0.1, planes[0])          # d == 0 on every plane: inf / nan coordinates
    dsi_c = orc.fill_voxel_grid(xy, centers, planes, Kv, nx, ny)
    dsi_n = ind.fill_voxel_grid(xy, centers, planes, Kv, nx, ny)
    assert dsi_c.sum() > 1000
    assert np.array_equal(dsi_c, dsi_n)
    g = np.roll(dsi_c, 5, axis=2) * np.float32(1.7)
    for op in range(1, 7):
        assert np.array_equal(orc.fuse2(dsi_c, g, op), ind.fuse2(dsi_c, g, op)), op
    for n in (3, 4):
        assert np.array_equal(orc.fuse_hm_n(dsi_c, g, n), ind.fuse_hm_n(dsi_c, g, n))
    for mode in (0, 1):
        acc_c, acc_n = np.zeros_like(g), np.zeros_like(g)
        for v in (dsi_c, g, dsi_c):
            acc_c, acc_n = orc.accumulate(acc_c, v, mode), ind.accumulate(acc_n, v, mode)
        assert np.array_equal(acc_c, acc_n)
        assert np.array_equal(orc.finalize(acc_c, mode, 3), ind.finalize(acc_n, mode, 3))
    conf_c, idx_c = orc.collapse_max_z(dsi_c)
    conf_n, idx_n = ind.collapse_max_z(dsi_c)
    assert np.array_equal(conf_c, conf_n) and np.array_equal(idx_c, idx_n)


def test_gm_tree_is_repeated_two_ary_geometric_mean():
    """oracle.fuse_gm_tree = Grid3D::geometricMeanTwoGrids (cartesian3dgrid.h:150-156) on pairs, then on the
    results: equal to float32 numpy sqrt of float32 products level by level, and to (a b c d)^(1/4) in float64
    within a few ulp."""
    rng = np.random.default_rng(2)
    maps = [rng.gamma(2.0, 5.0, (3, 8, 9)).astype(np.float32) for _ in range(4)]
    maps[1][0, 0, 0] = 0.0
    got = orc.fuse_gm_tree(maps)
    l0 = np.sqrt(maps[0] * maps[1])
    l1 = np.sqrt(maps[2] * maps[3])
    assert np.array_equal(got, np.sqrt(l0 * l1))
    assert got[0, 0, 0] == 0.0
    exact = (maps[0].astype(np.float64) * maps[1] * maps[2] * maps[3]) ** 0.25
    assert np.allclose(got, exact, rtol=4e-7)
    assert np.array_equal(orc.fuse_gm_tree(maps[:2]), orc.fuse2(maps[0], maps[1], 3))
    with pytest.raises(ValueError):
        orc.fuse_gm_tree(maps[:3])


def test_row_strip_oracle_equals_the_rows_of_the_full_dsi():
    """oracle.fill_voxel_grid_rows (test infrastructure for 1024 x 1024 x 256 at 100 M events) is bit-equal to the
    corresponding rows of fill_voxel_grid, including the strip's first and last row (half votes from outside)."""
    rng = np.random.default_rng(9)
    nx, ny = 70, 60
    planes = orc.depth_planes(1.0, 5.0, 7)
    Kv = np.array([50, 50, 35, 30], np.float32)
    xy = rng.uniform(-8, 78, (6 * 1024, 2)).astype(np.float32)
    xy[:40] = [[3.0, 17.0], [3.5, 16.999], [60.25, 28.0], [1.0, 29.5]] * 10     # on the strip's edges at plane 0
    centers = rng.normal(0, 0.2, (6, 3)).astype(np.float32)
    centers[0] = 0
    full = orc.fill_voxel_grid(xy, centers, planes, Kv, nx, ny)
    for r0, rows in ((0, 60), (17, 13), (59, 1), (0, 1)):
        strip = orc.fill_voxel_grid_rows(xy, centers, planes, Kv, nx, ny, r0, rows)
        assert np.array_equal(strip, full[:, r0:r0 + rows, :]), (r0, rows)
    assert full[:, 17:30].sum() > 100
