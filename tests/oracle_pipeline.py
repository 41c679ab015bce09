"""evaluateDSI / process_1 / process_2 assembled from the CPU oracle's pieces, for tests
and for bench.py's cpu_baseline leg.  TEST INFRASTRUCTURE (imports oracle/)."""
import numpy as np

from oracle import oracle as orc

PACKET = 1024


class OracleMapper:
    """MapperEMVS restated with oracle calls (mapper_emvs_stereo.cpp:29-64, 208-241)."""

    def __init__(self, cam, dimX=0, dimY=0, dimZ=100, min_depth=0.3, max_depth=5.0, fov=0.0,
                 lut=None, inverse_depth=False):
        w, h, fx, fy, cx, cy = cam
        self.W, self.H = int(w), int(h)
        self.K = np.array([fx, fy, cx, cy], np.float32)
        self.nx = dimX if dimX > 0 else self.W
        self.ny = dimY if dimY > 0 else self.H
        self.nz = dimZ
        self.planes = orc.depth_planes(min_depth, max_depth, dimZ, inverse_depth)
        f = orc.virtual_focal(np.float32(fx), fov, self.nx)
        self.Kv = np.array([f, f, np.float32(cx), np.float32(cy)], np.float32)
        self.lut = None if lut is None else np.ascontiguousarray(lut, np.float32)
        self.dsi = np.zeros((self.nz, self.ny, self.nx), np.float32)

    def packetize(self, ts, trajectory, T_rv_w):
        """mapper_emvs_stereo.cpp:67-105 -> (first[np], Rt[np][12]) or None."""
        times, poses = trajectory
        n = ts.shape[0]
        if n < PACKET:
            return None
        first, Rt = [], []
        cur = 0
        while cur + PACKET < n:
            T = orc.pose_at(times, poses, ts[cur + PACKET // 2])
            if T is None:
                cur += 1
                continue
            first.append(cur)
            Rt.append(orc.event_pose_Rt(T_rv_w, T))
            cur += PACKET
        return (np.array(first, np.int64), np.array(Rt, np.float32).reshape(-1, 12))

    def stage_a(self, x, y, first, Rt):
        centers, H = orc.packet_geometry(Rt, self.K, self.Kv, self.planes[0])
        idx = (first[:, None] + np.arange(PACKET)[None, :]).reshape(-1)
        xy = orc.warp_z0(x[idx], y[idx], H, self.lut, self.W)
        return xy, centers

    def evaluate_packets(self, x, y, first, Rt):
        xy, centers = self.stage_a(x, y, first, Rt)
        self.dsi[:] = 0  # resetGrid, :145
        orc.fill_voxel_grid(xy, centers, self.planes, self.Kv, self.nx, self.ny, self.dsi)
        return xy, centers

    def evaluateDSI(self, events, trajectory, T_rv_w):
        x, y, ts = events
        pk = self.packetize(ts, trajectory, T_rv_w)
        if pk is None:
            return False
        self.evaluate_packets(x, y, *pk)
        self.n_voted = pk[0].shape[0] * PACKET
        return True

    def depth_map(self, dsi=None):
        conf, idx = orc.collapse_max_z(self.dsi if dsi is None else dsi)
        return orc.indices_to_depth(idx, self.planes), conf, idx
