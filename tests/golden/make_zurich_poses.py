"""Extracts the control poses of the reference's only real recording that fall into the window its own
configuration processes, as a small fixture (DATA, not code):

    /root/reference/data/DSEC/zurich_city_04-odometry/pose.bag   (LiDAR-IMU odometry, PoseStamped @ ~10 Hz)
    window: --start_time_s=10 --stop_time_s=15  (mapper_emvs_stereo/cfg/DSEC/zurich_04_a_full/dsec.conf:13-14),
    read like data_loading::parse_rosbag_gt (stamps relative to the first pose message), one second of
    margin on either side so that every event time of the window can be interpolated.

Run in the build container (the reference checkout does not exist on the GPU box):
    python tests/golden/make_zurich_poses.py   ->   tests/golden/zurich_city_04_poses_9_16s.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dvs_mcemvs_amd import io  # noqa: E402

BAG = "/root/reference/data/DSEC/zurich_city_04-odometry/pose.bag"
OUT = os.path.join(ROOT, "tests", "golden", "zurich_city_04_poses_9_16s.npz")

if __name__ == "__main__":
    times, poses = io.parse_rosbag_gt(BAG, topic="/pose", tmin=9.0, tmax=16.0)
    assert 60 <= times.shape[0] <= 80, times.shape
    np.savez_compressed(OUT, times=times, poses=poses,
                        source="zurich_city_04-odometry/pose.bag, /pose, relative stamps 9..16 s "
                               "(tx,ty,tz,qw,qx,qy,qz)")
    d = np.linalg.norm(np.diff(poses[:, :3], axis=0), axis=1)
    print("%d poses, %.3f .. %.3f s, path length %.2f m, median speed %.2f m/s -> %s"
          % (times.shape[0], times[0], times[-1], d.sum(), np.median(d / np.diff(times)), OUT))
