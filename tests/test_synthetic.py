"""CPU tests of the input generators bench.py and the GPU tests rely on (dvs_mcemvs_amd/synthetic.py): they compute
nothing on the DSI path, but the bench's sensitivity block and the first-principles tests are only as good as they."""
import os

import numpy as np

from dvs_mcemvs_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stereo_rig_shapes_and_order():
    rig = syn.stereo_rig(5000, width=96, height=72, duration=0.2, seed=3, n_cams=3)
    assert len(rig["events"]) == 3 and len(rig["trajectories"]) == 3
    for x, y, ts in rig["events"]:
        assert x.dtype == np.uint16 and y.dtype == np.uint16 and x.shape == (5000,)
        assert x.max() < 96 and y.max() < 72
        assert np.all(np.diff(ts) >= 0) and rig["t0"] <= ts[0] and ts[-1] <= rig["t1"]   # data_loading.cpp:211-216
    times, poses = rig["trajectories"][0]
    assert times[0] < rig["t0"] and times[-1] > rig["t1"]            # control poses bracket the interval
    assert np.allclose(np.linalg.norm(poses[:, 3:], axis=1), 1.0)
    # the cameras sit along the baseline: camera 2 is 0.6 m from camera 0 at every control pose
    d = np.linalg.norm(rig["trajectories"][2][1][:, :3] - poses[:, :3], axis=1)
    assert np.allclose(d, 0.6, atol=1e-9)
    # same seed, same rig
    again = syn.stereo_rig(5000, width=96, height=72, duration=0.2, seed=3, n_cams=3)
    assert all(np.array_equal(a, b) for ea, eb in zip(rig["events"], again["events"]) for a, b in zip(ea, eb))


def test_recorded_rig_follows_the_zurich_city_04_poses():
    """bench.py's sensitivity case "zurich_city_04": events generated along the recorded vehicle trajectory
    (tests/golden/zurich_city_04_poses_9_16s.npz, extracted from the reference's pose.bag)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "zurich_city_04_poses_9_16s.npz"))
    fn = syn.recorded_rig(z["times"], z["poses"])
    # at the control times the interpolant returns the control poses (up to the quaternion's sign)
    pos, q = fn(z["times"][5:20])
    assert np.allclose(pos, z["poses"][5:20, :3], atol=1e-12)
    dots = np.abs(np.sum(q * z["poses"][5:20, 3:], axis=1))
    assert np.allclose(dots, 1.0, atol=1e-12)
    # in between: on the chord, unit quaternion
    tm = 0.5 * (z["times"][7] + z["times"][8])
    pm, qm = fn(np.array(tm))
    assert np.allclose(pm, 0.5 * (z["poses"][7, :3] + z["poses"][8, :3])) and abs(np.linalg.norm(qm) - 1.0) < 1e-12
    rig = syn.stereo_rig(20_000, width=346, height=260, t0=10.0, duration=5.0, seed=9, n_cams=2, pose_fn=fn)
    times, poses = rig["trajectories"][0]
    assert np.array_equal(times, z["times"]) and np.allclose(poses[:, :3], z["poses"][:, :3])
    # the right camera: the recorded poses shifted 0.6 m along each pose's own x axis
    d = np.linalg.norm(rig["trajectories"][1][1][:, :3] - poses[:, :3], axis=1)
    assert np.allclose(d, 0.6, atol=1e-9)
    for x, y, ts in rig["events"]:
        assert x.shape == (20_000,) and x.max() < 346 and y.max() < 260 and np.all(np.diff(ts) >= 0)
        assert 10.0 <= ts[0] and ts[-1] <= 15.0
    # the engine's host pose pipeline interpolates these control poses everywhere inside the window
    import dvs_mcemvs_amd as d_
    assert d_.pose_at(rig["trajectories"][0], 12.3456) is not None


def test_radial_lut_is_identity_at_the_principal_point():
    cam = syn.camera(64, 48)
    lut = syn.radial_lut(cam).reshape(48, 64, 2)
    cx, cy = cam[4], cam[5]
    ix, iy = int(round(cx)), int(round(cy))
    assert abs(lut[iy, ix, 0] - ix) < 0.05 and abs(lut[iy, ix, 1] - iy) < 0.05
    assert lut.dtype == np.float32 and np.isfinite(lut).all()
