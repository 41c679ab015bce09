"""Synthetic event-camera rigs for tests and bench.py (numpy only; no DSEC data exists
on the GPU box).  Follows SURVEY.md section 8(d): a cloud of 3-D points seen by a rig
that drives forward at ~3.6 m/s with a small lateral sinusoid, control poses sampled
at 10 Hz like the DSEC LiDAR odometry, events = projections of scene points at
uniformly increasing timestamps (+10 % uniform noise events), time-sorted.

This module only GENERATES INPUTS (events, trajectories, calibration).  It computes
nothing on the DSI path.
"""
import numpy as np

# DSEC zurich_city_04_a left event camera (calib.cpp:463-466 of the reference), used as
# realistic intrinsics for the 640x480 case; scaled for the 346x260 case.
DSEC_K = (553.4686750102932, 553.3994078799127, 346.65339162053317, 216.52092103243012)


def camera(width, height):
    """(width, height, fx, fy, cx, cy) of a pinhole sensor with DSEC-like field of view."""
    s = width / 640.0
    fx, fy, cx, cy = DSEC_K
    return (int(width), int(height), fx * s, fy * s, cx * s, cy * height / 480.0)


def quat_from_yaw_pitch(yaw, pitch):
    """Unit quaternions (w,x,y,z) for rotation about y (yaw) then x (pitch); arrays ok."""
    cy, sy = np.cos(yaw / 2), np.sin(yaw / 2)
    cp, sp = np.cos(pitch / 2), np.sin(pitch / 2)
    # q = q_y(yaw) * q_x(pitch)
    w = cy * cp
    x = cy * sp
    y = sy * cp
    z = -sy * sp
    return np.stack([w, x, y, z], axis=-1)


def quat_rotate(q, v):
    """Rotate vectors v [...,3] by unit quaternions q [...,4]."""
    w = q[..., :1]
    u = q[..., 1:]
    uv = 2.0 * np.cross(u, v)
    return v + w * uv + np.cross(u, uv)


def rig_pose(t, speed=3.6, lateral=0.1, yaw_amp=0.01, pitch_amp=0.004):
    """Analytic rig pose T_w_rig(t): position [...,3] and quaternion [...,4]."""
    t = np.asarray(t, np.float64)
    pos = np.stack([lateral * np.sin(2 * np.pi * 0.5 * t), 0.02 * np.sin(2 * np.pi * 0.8 * t),
                    speed * t], axis=-1)
    q = quat_from_yaw_pitch(yaw_amp * np.sin(2 * np.pi * 0.4 * t),
                            pitch_amp * np.sin(2 * np.pi * 0.7 * t))
    return pos, q


def recorded_rig(times, poses):
    """pose_fn(t) -> (position [...,3], quaternion [...,4]) of a RECORDED trajectory (control poses T_w_rig as
    {tx,ty,tz,qw,qx,qy,qz} at ascending times, e.g. tests/golden/zurich_city_04_poses_9_16s.npz): translation
    interpolated linearly, rotation by normalised linear interpolation of the quaternions -- an input generator,
    not LinearTrajectory::getPoseAt (the engine's pose pipeline interpolates the control poses its own way; the
    events only have to be plausible).  `pose_fn.control` = (times, poses) for trajectory()."""
    times = np.asarray(times, np.float64)
    poses = np.asarray(poses, np.float64).copy()
    for i in range(1, poses.shape[0]):                   # keep neighbouring quaternions in one hemisphere
        if np.dot(poses[i, 3:], poses[i - 1, 3:]) < 0:
            poses[i, 3:] = -poses[i, 3:]

    def pose_fn(t):
        t = np.asarray(t, np.float64)
        k = np.clip(np.searchsorted(times, t, side="right") - 1, 0, times.shape[0] - 2)
        w = ((t - times[k]) / (times[k + 1] - times[k]))[..., None]
        pos = (1 - w) * poses[k, :3] + w * poses[k + 1, :3]
        q = (1 - w) * poses[k, 3:] + w * poses[k + 1, 3:]
        return pos, q / np.linalg.norm(q, axis=-1, keepdims=True)

    pose_fn.control = (times, poses)
    return pose_fn


def trajectory(t0, t1, cam_offset_x=0.0, rate_hz=10.0, pose_fn=None, **kw):
    """Control poses T_w_cam at rate_hz covering [t0, t1] with margin (a recorded rig: its own control poses).
    Returns (times float64[m], poses float64[m][7] = tx,ty,tz,qw,qx,qy,qz)."""
    if pose_fn is not None and hasattr(pose_fn, "control"):
        times = pose_fn.control[0]
        pos, q = pose_fn.control[1][:, :3], pose_fn.control[1][:, 3:]
        n = times.shape[0]
        off = np.zeros((n, 3))
        off[:, 0] = cam_offset_x
        return times.copy(), np.concatenate([pos + quat_rotate(q, off), q], axis=1)
    n = int(np.ceil((t1 - t0) * rate_hz)) + 3
    times = t0 - 1.0 / rate_hz + np.arange(n) / rate_hz
    pos, q = (pose_fn or rig_pose)(times, **kw)
    off = np.zeros((n, 3))
    off[:, 0] = cam_offset_x
    pos = pos + quat_rotate(q, off)
    return times, np.concatenate([pos, q], axis=1)


def pose_inverse(p):
    """Inverse of a 7-vector pose (tx,ty,tz,qw,qx,qy,qz)."""
    q = np.array([p[3], -p[4], -p[5], -p[6]])
    t = -quat_rotate(q, np.asarray(p[:3], np.float64))
    return np.concatenate([t, q])


def make_events(n_events, cam, t0, t1, seed, cam_offset_x=0.0, n_points=5000,
                depth_range=(4.8, 160.0), noise_frac=0.10, pose_fn=None, **kw):
    """Events of one camera: (x uint16[n], y uint16[n], ts float64[n]), time-sorted.
    pose_fn: the rig's pose as a function of time (default: the analytic rig_pose; recorded_rig() for a real one)."""
    width, height, fx, fy, cx, cy = cam
    rig_pose = pose_fn or globals()["rig_pose"]
    rng = np.random.default_rng(seed)
    # scene points in the frustum of the view at the middle of the interval
    tm = 0.5 * (t0 + t1)
    pm, qm = rig_pose(np.array(tm), **kw)
    z = rng.uniform(depth_range[0], depth_range[1], n_points)
    u = rng.uniform(-0.15 * width, 1.15 * width, n_points)
    v = rng.uniform(-0.15 * height, 1.15 * height, n_points)
    pts_c = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], axis=-1)
    pts_w = pm + quat_rotate(qm, pts_c)

    out_x = np.empty(0, np.uint16)
    out_y = np.empty(0, np.uint16)
    out_t = np.empty(0, np.float64)
    need = int(n_events)
    n_noise = int(round(noise_frac * need))
    n_sig = need - n_noise
    while out_x.shape[0] < n_sig:
        m = int((n_sig - out_x.shape[0]) * 1.6) + 1024
        ts = rng.uniform(t0, t1, m)
        pid = rng.integers(0, n_points, m)
        pos, q = rig_pose(ts, **kw)
        off = np.zeros((m, 3))
        off[:, 0] = cam_offset_x
        c = pos + quat_rotate(q, off)
        qi = q * np.array([1.0, -1.0, -1.0, -1.0])
        pc = quat_rotate(qi, pts_w[pid] - c)
        ok = pc[:, 2] > 0.5
        px = fx * pc[:, 0] / np.where(ok, pc[:, 2], 1.0) + cx
        py = fy * pc[:, 1] / np.where(ok, pc[:, 2], 1.0) + cy
        ix = np.rint(px)
        iy = np.rint(py)
        ok &= (ix >= 0) & (ix < width) & (iy >= 0) & (iy < height)
        out_x = np.concatenate([out_x, ix[ok].astype(np.uint16)])
        out_y = np.concatenate([out_y, iy[ok].astype(np.uint16)])
        out_t = np.concatenate([out_t, ts[ok]])
    out_x, out_y, out_t = out_x[:n_sig], out_y[:n_sig], out_t[:n_sig]
    nx = rng.integers(0, width, n_noise).astype(np.uint16)
    ny = rng.integers(0, height, n_noise).astype(np.uint16)
    nt = rng.uniform(t0, t1, n_noise)
    x = np.concatenate([out_x, nx])
    y = np.concatenate([out_y, ny])
    ts = np.concatenate([out_t, nt])
    order = np.argsort(ts, kind="stable")  # data_loading.cpp:211-216 sorts by timestamp
    return x[order], y[order], ts[order]


def radial_lut(cam, k1=-0.09, k2=0.19):
    """A closed-form stand-in for precomputeRectifiedPoints (mapper_emvs_stereo.cpp:256-299):
    raw pixel -> undistorted pixel through one fixed-point inversion step of a radial model.
    Returns float32 [H*W][2], column index y*W+x as in the reference."""
    width, height, fx, fy, cx, cy = cam
    ys, xs = np.mgrid[0:height, 0:width]
    xn = (xs - cx) / fx
    yn = (ys - cy) / fy
    r2 = xn * xn + yn * yn
    s = 1.0 + k1 * r2 + k2 * r2 * r2
    xu = xn / s
    yu = yn / s
    lut = np.stack([xu * fx + cx, yu * fy + cy], axis=-1).reshape(-1, 2)
    return np.ascontiguousarray(lut, np.float32)


def stereo_rig(n_events_per_cam, width=346, height=260, t0=10.0, duration=0.5, baseline=0.6,
               n_cams=2, seed=1234, n_points=5000, noise_frac=0.10, pose_fn=None, **kw):
    """A synthetic multi-camera recording: dict with cam, per-camera events and
    trajectories, and the reference-view pose T_rv_w (left camera at the END of the
    interval, i.e. --forward_looking=true as in cfg/DSEC/zurich_04_a_full/dsec.conf)."""
    cam = camera(width, height)
    t1 = t0 + duration
    offsets = [baseline * i / max(1, n_cams - 1) for i in range(n_cams)] if n_cams > 1 else [0.0]
    events, trajs = [], []
    for i, off in enumerate(offsets):
        events.append(make_events(n_events_per_cam, cam, t0, t1, seed + i, cam_offset_x=off,
                                  n_points=n_points, noise_frac=noise_frac, pose_fn=pose_fn, **kw))
        trajs.append(trajectory(t0, t1, cam_offset_x=off, pose_fn=pose_fn, **kw))
    pos, q = (pose_fn or rig_pose)(np.array(t1), **kw)
    T_w_rv = np.concatenate([pos, q])
    return {"cam": cam, "events": events, "trajectories": trajs, "T_rv_w": pose_inverse(T_w_rv),
            "t0": t0, "t1": t1}
