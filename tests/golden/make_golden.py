#!/usr/bin/env python
"""Generates the committed golden fixtures (tests/golden/*.npz).

The reference cannot be built or run in this image and ships no vectors of its own
(SURVEY.md 8c), so these are REGRESSION vectors produced by our CPU oracle
(oracle/dsi_oracle.c) on seeded synthetic inputs -- they pin the oracle and the HIP path
to each other and to this commit, not to an execution of the reference.

    python tests/golden/make_golden.py          # rewrites tests/golden/case_*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from dvs_mcemvs_amd import synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle_pipeline import OracleMapper  # noqa: E402


def case(name, width, height, nz, n_events, seed, dmin, dmax, inverse=False, lut=False, dimX=0,
         dimY=0, fov=0.0, slices=0):
    rig = syn.stereo_rig(n_events, width=width, height=height, duration=0.25, seed=seed)
    cam = rig["cam"]
    lut_arr = syn.radial_lut(cam) if lut else None
    out = {"cam": np.array(cam, np.float64), "nz": nz, "dmin": dmin, "dmax": dmax,
           "inverse": int(inverse), "dimX": dimX, "dimY": dimY, "fov": fov,
           "T_rv_w": rig["T_rv_w"], "has_lut": int(lut)}
    if lut:
        out["lut"] = lut_arr
    dsis = []
    for c in range(2):
        m = OracleMapper(cam, dimX=dimX, dimY=dimY, dimZ=nz, min_depth=dmin, max_depth=dmax, fov=fov,
                         lut=lut_arr, inverse_depth=inverse)
        x, y, ts = rig["events"][c]
        times, poses = rig["trajectories"][c]
        first, Rt = m.packetize(ts, (times, poses), rig["T_rv_w"])
        xy, centers = m.evaluate_packets(x, y, first, Rt)
        out.update({"x%d" % c: x.astype(np.uint8), "y%d" % c: y.astype(np.uint8), "ts%d" % c: ts, "traj_t%d" % c: times,
                    "traj_p%d" % c: poses, "first%d" % c: first.astype(np.uint32), "Rt%d" % c: Rt,
                    "centers%d" % c: centers, "dsi%d" % c: m.dsi.copy()})
        if c == 0:
            out["xy0"] = xy
        dsis.append(m.dsi.copy())
        if c == 0:
            out["planes"] = m.planes
            out["Kv"] = m.Kv
            depth, conf, idx = m.depth_map()
            out.update({"depth0": depth, "conf0": conf, "idx0": idx})
            out["mean_square0"] = np.float64(orc.mean_square(m.dsi))
            if slices:
                # process_2 temporal fusion of camera 0 over `slices` sub-intervals by event count
                # (process2.cpp:46-47, :105-107, :211-242)
                per = x.shape[0] // slices
                acc_hm = np.zeros_like(m.dsi)
                acc_am = np.zeros_like(m.dsi)
                for k in range(slices):
                    sl = slice(k * per, (k + 1) * per)
                    ms = OracleMapper(cam, dimX=dimX, dimY=dimY, dimZ=nz, min_depth=dmin,
                                      max_depth=dmax, fov=fov, lut=lut_arr, inverse_depth=inverse)
                    assert ms.evaluateDSI((x[sl], y[sl], ts[sl]), (times, poses), rig["T_rv_w"])
                    acc_hm = orc.accumulate(acc_hm, ms.dsi, 1)
                    acc_am = orc.accumulate(acc_am, ms.dsi, 0)
                out["temporal_hm"] = orc.finalize(acc_hm, 1, slices)
                am = orc.finalize(acc_am, 0, slices)
                out["temporal_am_s7"] = am.reshape(-1)[::7].copy()
                out["temporal_am_sum"] = np.float64(am.astype(np.float64).sum())
                out["slices"] = slices
    # fused volumes: op 2 (HM, the default --stereo_fusion) in full, the others as a strided
    # sample (every 7th voxel) + a float64 checksum, to keep the fixtures small
    for op in range(1, 7):
        f = orc.fuse2(dsis[0], dsis[1], op)
        if op == 2:
            out["fused2"] = f
        else:
            out["fused%d_s7" % op] = f.reshape(-1)[::7].copy()
            out["fused%d_sum" % op] = np.float64(f.astype(np.float64).sum())
    hm3 = orc.fuse_hm_n(out["fused2"], dsis[0], 3)
    out["fused_hm3_s7"] = hm3.reshape(-1)[::7].copy()
    out["fused_hm3_sum"] = np.float64(hm3.astype(np.float64).sum())
    depth, conf, idx = OracleMapper(cam, dimX=dimX, dimY=dimY, dimZ=nz, min_depth=dmin, max_depth=dmax,
                                    fov=fov, inverse_depth=inverse).depth_map(out["fused2"])
    out.update({"depth_fused2": depth, "conf_fused2": conf, "idx_fused2": idx})
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(path, "%.0f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    case("case_a_linear", 56, 40, 12, 5200, 101, 4.0, 200.0, slices=4)
    case("case_b_inverse_lut_fov", 60, 44, 12, 4200, 202, 2.0, 80.0, inverse=True, lut=True, dimX=48,
         dimY=40, fov=70.0)
