"""On-disk formats: .npy DSI dump, depth-points text, ROSBAG v2 pose reader."""
import os

import numpy as np
import pytest

from dvs_mcemvs_amd import io

REF_BAG = "/root/reference/data/DSEC/zurich_city_04-odometry/pose.bag"


def test_grid_npy_layout(tmp_path):
    vol = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)   # [Z][Y][X]
    shape = io.write_grid_npy(tmp_path / "dsi.npy", vol)
    assert shape == (2, 3, 4)                       # cartesian3dgrid_IO.cpp:33: {size_[2], size_[1], size_[0]}
    back = np.load(tmp_path / "dsi.npy")
    assert back.dtype == np.float32 and back.flags["C_CONTIGUOUS"] and np.array_equal(back, vol)
    # the reference's viewers index dsi[z, y, x] (scripts/visualize_dsi_*.py)
    assert back[1, 2, 3] == vol.reshape(-1)[3 + 4 * (2 + 3 * 1)]   # volume[x + dimX*(y + dimY*z)]


def test_depth_points_text(tmp_path):
    depth = np.array([[1.5, 2.25, 0.0], [123.456789, 0.0, 7.0]], np.float32)
    mask = np.array([[1, 0, 0], [1, 0, 1]], np.uint8)
    n = io.save_depth_points(tmp_path / "pts.txt", depth, mask)
    assert n == 3
    lines = open(tmp_path / "pts.txt").read().splitlines()
    assert lines == ["0 0 1.5", "0 1 123.457", "2 1 7"]   # "col row depth", 6 significant digits


def test_pose_bag_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    times = 100.0 + np.sort(rng.uniform(0, 5, 40))
    q = rng.normal(size=(40, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses = np.concatenate([rng.normal(size=(40, 3)), q], axis=1)
    io.write_pose_bag(tmp_path / "pose.bag", times, poses, topic="/pose")
    t, p = io.read_pose_bag(tmp_path / "pose.bag", topic="/pose")
    assert np.allclose(t, times, atol=1e-9) and np.allclose(p, poses, atol=0)
    t2, _ = io.read_pose_bag(tmp_path / "pose.bag", topic="/other")
    assert t2.shape[0] == 0
    with pytest.raises(ValueError):
        open(tmp_path / "junk.bag", "wb").write(b"not a bag")
        io.read_pose_bag(tmp_path / "junk.bag")


def test_parse_rosbag_gt_relative_stamps_and_window(tmp_path):
    """data_loading::parse_rosbag_gt (data_loading.cpp:303-420): stamps relative to the first pose
    message, poses before tmin skipped, the first one beyond tmax still taken, then the scan stops."""
    times = 1000.0 + 0.1 * np.arange(60)
    poses = np.zeros((60, 7))
    poses[:, 0] = np.arange(60)
    poses[:, 3] = 1.0
    io.write_pose_bag(tmp_path / "pose.bag", times, poses, topic="/pose")
    t, p = io.parse_rosbag_gt(tmp_path / "pose.bag", topic="/pose")
    assert t.shape[0] == 60 and t[0] == 0.0 and np.allclose(t, 0.1 * np.arange(60), atol=1e-9)
    t, p = io.parse_rosbag_gt(tmp_path / "pose.bag", topic="/pose", tmin=1.0, tmax=2.0)
    # rel 1.0 .. 2.0 inclusive, plus the first stamp beyond tmax (2.1)
    assert np.allclose(t, 0.1 * np.arange(10, 22), atol=1e-9)
    assert np.array_equal(p[:, 0], np.arange(10, 22))


def test_parse_rosbag_gt_keeps_the_first_of_equal_stamps(tmp_path):
    """The reference stores poses with std::map::insert (data_loading.cpp:303-420): a repeated stamp keeps the
    FIRST pose and adds no control point."""
    times = np.array([5.0, 5.1, 5.1, 5.2, 5.2, 5.2, 5.3])
    poses = np.zeros((7, 7))
    poses[:, 0] = np.arange(7)
    poses[:, 3] = 1.0
    io.write_pose_bag(tmp_path / "pose.bag", times, poses, topic="/pose")
    t, p = io.parse_rosbag_gt(tmp_path / "pose.bag", topic="/pose")
    assert np.allclose(t, [0.0, 0.1, 0.2, 0.3], atol=1e-9)
    assert np.array_equal(p[:, 0], [0, 1, 3, 6])
    assert np.all(np.diff(t) > 0)


@pytest.mark.skipif(not os.path.exists(REF_BAG), reason="reference checkout not present on this box")
def test_reads_the_reference_dsec_odometry_bag():
    """The only real data in the reference repo: LiDAR-IMU odometry of
    DSEC zurich_city_04, 6205 PoseStamped at ~9.92 Hz over 625.7 s, median speed 3.6 m/s."""
    t, p = io.read_pose_bag(REF_BAG, topic="/pose")
    assert t.shape[0] == 6205
    assert t[-1] - t[0] == pytest.approx(625.7, abs=0.1)
    assert np.median(np.diff(t)) == pytest.approx(1 / 9.92, rel=0.02)
    assert np.abs(np.linalg.norm(p[:, 3:], axis=1) - 1).max() < 1e-5
    speed = np.linalg.norm(np.diff(p[:, :3], axis=0), axis=1) / np.diff(t)
    assert np.median(speed) == pytest.approx(3.6, abs=0.1)
    # the engine's trajectory interpolation accepts it (cfg/DSEC/zurich_04_a_full/dsec.conf:13-14 window)
    import dvs_mcemvs_amd as d
    T = d.pose_at((t - t[0], p), 12.5)
    assert T is not None and abs(np.linalg.norm(T[3:]) - 1) < 1e-5


@pytest.mark.parametrize("compression", ["none", "bz2"])
def test_event_bag_reader_follows_parse_rosbag(tmp_path, compression):
    """data_loading.cpp:66-104, 211-216: the first event defines the initial stamp; events with
    relative stamp < tmin are skipped; the message in which tmax is exceeded is still taken whole
    and is the last one read; stamps become ts - initial - events_offset; sorted by stamp."""
    rng = np.random.default_rng(3)
    n = 1000
    t_abs = 1000.25 + np.sort(rng.uniform(0.0, 1.0, n))
    t_abs[0] = 1000.25
    # unsorted inside a message (the final sort fixes it)
    t_abs[[10, 11]] = t_abs[[11, 10]]
    x = rng.integers(0, 346, n).astype(np.uint16)
    y = rng.integers(0, 260, n).astype(np.uint16)
    pol = rng.integers(0, 2, n).astype(np.uint8)
    path = tmp_path / "ev.bag"
    io.write_event_bag(path, x, y, t_abs, pol, topic="/dvs/left/events", events_per_message=100,
                       compression=compression)
    ev = io.read_event_bag(path, "/dvs/left/events")
    assert ev["x"].shape[0] == n and ev["height"] == 260 and ev["width"] == 346
    assert ev["initial_stamp"] == pytest.approx(1000.25, abs=1e-9)
    assert np.all(np.diff(ev["ts"]) >= 0) and ev["ts"][0] == pytest.approx(0.0, abs=1e-9)
    order = np.argsort(t_abs, kind="stable")
    assert np.array_equal(ev["x"], x[order]) and np.array_equal(ev["polarity"], pol[order])
    assert np.allclose(ev["ts"], t_abs[order] - 1000.25, atol=2e-9)

    tmin, tmax, off = 0.2, 0.5, 0.05
    ev = io.read_event_bag(path, "/dvs/left/events", tmin, tmax, off)
    rel = np.rint((t_abs - 1000.25) * 1e9) * 1e-9
    msg = np.arange(n) // 100
    last_msg = msg[np.nonzero(rel > tmax)[0][0]]          # read up to and including this message
    keep = (rel >= tmin) & (msg <= last_msg)
    assert ev["x"].shape[0] == keep.sum()
    assert ev["ts"].max() > tmax - off                     # events beyond tmax of that message stay
    ko = np.nonzero(keep)[0][np.argsort(t_abs[keep], kind="stable")]
    assert np.array_equal(ev["x"], x[ko]) and np.array_equal(ev["y"], y[ko])
    assert np.allclose(ev["ts"], t_abs[ko] - 1000.25 - off, atol=2e-9)
    assert io.read_event_bag(path, "/nope")["x"].shape[0] == 0
