"""ctypes wrapper around oracle/libdsi_oracle.so (plain-C restatement of the
reference's DSI path).  TEST INFRASTRUCTURE ONLY -- see oracle/dsi_oracle.h.

numpy in, numpy out; shapes follow the reference layout volume[x + Nx*(y + Ny*z)]
(cartesian3dgrid.h:34-35), i.e. arrays are [Nz][Ny][Nx] C-contiguous.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PACKET = 1024


_LIB_NATIVE = None
_USE_NATIVE = False


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def use_native(flag):
    """bench.py's "tuned CPU" row: route the calls below through a second build of the same file
    with -O3 -march=native, compiled ON THIS MACHINE (the CPU it runs on decides the ISA).
    Timing only -- every parity test uses the reference-flag build."""
    global _USE_NATIVE, _LIB_NATIVE
    if flag and _LIB_NATIVE is None:
        subprocess.check_call(["make", "-s", "-C", _HERE, "native"])
        _LIB_NATIVE = _load(os.path.join(_HERE, "libdsi_oracle_native.so"))
    _USE_NATIVE = bool(flag)


def lib():
    global _LIB
    if _USE_NATIVE and _LIB_NATIVE is not None:
        return _LIB_NATIVE
    if _LIB is None:
        path = os.path.join(_HERE, "libdsi_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = _load(path)
    return _LIB


def _load(path):
    L = C.CDLL(path)
    f32p, u8p, u16p, f64p = (C.POINTER(C.c_float), C.POINTER(C.c_uint8),
                             C.POINTER(C.c_uint16), C.POINTER(C.c_double))
    szp = C.POINTER(C.c_size_t)
    L.orc_depth_planes.argtypes = [C.c_float, C.c_float, C.c_int, C.c_int, f32p]
    L.orc_virtual_focal.argtypes = [C.c_float, C.c_float, C.c_int]
    L.orc_virtual_focal.restype = C.c_float
    L.orc_inverse3x3.argtypes = [f32p, f32p]
    L.orc_packet_geometry.argtypes = [f32p, f32p, f32p, C.c_float, f32p, f32p]
    L.orc_warp_z0.argtypes = [u16p, u16p, C.c_size_t, f32p, f32p, C.c_int, f32p]
    L.orc_fill_voxel_grid.argtypes = [f32p, f32p, C.c_size_t, f32p, C.c_int, f32p,
                                      C.c_int, C.c_int, f32p]
    L.orc_fill_voxel_grid_rows.argtypes = [f32p, f32p, C.c_size_t, f32p, C.c_int, f32p, C.c_int, C.c_int, C.c_int,
                                           C.c_int, f32p]
    L.orc_vote.argtypes = [C.c_float, C.c_float, f32p, C.c_int, C.c_int]
    L.orc_packetize.argtypes = [C.c_size_t, u8p, szp, szp]
    L.orc_packetize.restype = C.c_long
    L.orc_fuse2.argtypes = [f32p, f32p, C.c_size_t, C.c_int]
    L.orc_fuse2.restype = C.c_int
    L.orc_fuse_hm_n.argtypes = [f32p, f32p, C.c_size_t, C.c_int]
    L.orc_accumulate.argtypes = [f32p, f32p, C.c_size_t, C.c_int]
    L.orc_finalize.argtypes = [f32p, C.c_size_t, C.c_int, C.c_int]
    L.orc_accumulate_begin.argtypes = [f32p, C.c_size_t, C.c_int]
    L.orc_collapse_max_z.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, u8p]
    L.orc_indices_to_depth.argtypes = [u8p, C.c_size_t, f32p, f32p]
    L.orc_mean_square.argtypes = [f32p, C.c_size_t]
    L.orc_mean_square.restype = C.c_double
    L.orc_pose_at.argtypes = [f64p, f64p, C.c_size_t, C.c_double, f64p]
    L.orc_pose_at.restype = C.c_int
    L.orc_event_pose_Rt.argtypes = [f64p, f64p, f32p]
    L.orc_depth_map_filters.argtypes = [f32p, u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                        C.c_double, f32p, u8p, u8p, u8p, f32p]
    L.orc_num_threads.restype = C.c_int
    L.orc_set_num_threads.argtypes = [C.c_int]
    L.orc_set_num_threads.restype = None
    return L


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def depth_planes(min_depth, max_depth, nz, inverse=False):
    out = np.empty(nz, np.float32)
    lib().orc_depth_planes(min_depth, max_depth, nz, int(inverse), _p(out, C.c_float))
    return out


def virtual_focal(cam_fx, fov_deg, dim_x):
    return float(lib().orc_virtual_focal(cam_fx, fov_deg, dim_x))


def inverse3x3(m):
    m = _f32(m).reshape(9)
    out = np.empty(9, np.float32)
    lib().orc_inverse3x3(_p(m, C.c_float), _p(out, C.c_float))
    return out.reshape(3, 3)


def packet_geometry(Rt, K, Kv, z0):
    """Rt: [Np][12] float32 -> (centers [Np][3], H [Np][9])."""
    Rt = _f32(Rt).reshape(-1, 12)
    K, Kv = _f32(K), _f32(Kv)
    n = Rt.shape[0]
    centers = np.empty((n, 3), np.float32)
    H = np.empty((n, 9), np.float32)
    for k in range(n):
        lib().orc_packet_geometry(_p(Rt[k], C.c_float), _p(K, C.c_float), _p(Kv, C.c_float),
                                  np.float32(z0), _p(centers[k], C.c_float), _p(H[k], C.c_float))
    return centers, H


def warp_z0(ex, ey, H, lut, W):
    ex = np.ascontiguousarray(ex, np.uint16)
    ey = np.ascontiguousarray(ey, np.uint16)
    H = _f32(H).reshape(-1, 9)
    n = ex.shape[0]
    assert n <= H.shape[0] * PACKET
    out = np.empty((n, 2), np.float32)
    lp = _p(_f32(lut), C.c_float) if lut is not None else None
    if lut is not None:
        lut = _f32(lut)
        lp = _p(lut, C.c_float)
    lib().orc_warp_z0(_p(ex, C.c_uint16), _p(ey, C.c_uint16), n, _p(H, C.c_float), lp, W,
                      _p(out, C.c_float))
    return out


def fill_voxel_grid(xy_z0, centers, raw_depths, Kv, nx, ny, dsi=None):
    xy_z0 = _f32(xy_z0).reshape(-1, 2)
    centers = _f32(centers).reshape(-1, 3)
    raw_depths = _f32(raw_depths)
    Kv = _f32(Kv)
    npk = centers.shape[0]
    assert xy_z0.shape[0] == npk * PACKET
    nz = raw_depths.shape[0]
    if dsi is None:
        dsi = np.zeros((nz, ny, nx), np.float32)
    lib().orc_fill_voxel_grid(_p(xy_z0, C.c_float), _p(centers, C.c_float), npk,
                              _p(raw_depths, C.c_float), nz, _p(Kv, C.c_float), nx, ny,
                              _p(dsi, C.c_float))
    return dsi


def fill_voxel_grid_rows(xy_z0, centers, raw_depths, Kv, nx, ny, row_begin, row_count, strip=None):
    """Rows [row_begin, row_begin + row_count) of every plane of fill_voxel_grid's DSI (bit-equal to them)."""
    xy_z0 = _f32(xy_z0).reshape(-1, 2)
    centers = _f32(centers).reshape(-1, 3)
    raw_depths = _f32(raw_depths)
    Kv = _f32(Kv)
    npk = centers.shape[0]
    assert xy_z0.shape[0] == npk * PACKET and 0 <= row_begin and row_begin + row_count <= ny
    nz = raw_depths.shape[0]
    if strip is None:
        strip = np.zeros((nz, row_count, nx), np.float32)
    lib().orc_fill_voxel_grid_rows(_p(xy_z0, C.c_float), _p(centers, C.c_float), npk, _p(raw_depths, C.c_float), nz,
                                   _p(Kv, C.c_float), nx, ny, int(row_begin), int(row_count), _p(strip, C.c_float))
    return strip


def vote(x_f, y_f, plane):
    ny, nx = plane.shape
    lib().orc_vote(np.float32(x_f), np.float32(y_f), _p(plane, C.c_float), nx, ny)


def packetize(n_events, pose_ok=None):
    cap = n_events // PACKET + 1
    first = np.zeros(cap, np.uintp)
    mid = np.zeros(cap, np.uintp)
    pk = None
    if pose_ok is not None:
        pose_ok = np.ascontiguousarray(pose_ok, np.uint8)
        pk = _p(pose_ok, C.c_uint8)
    n = lib().orc_packetize(n_events, pk, _p(first, C.c_size_t), _p(mid, C.c_size_t))
    if n < 0:
        return None
    return first[:n].astype(np.int64), mid[:n].astype(np.int64)


def fuse2(a, g, op):
    a = _f32(a).copy()
    g = _f32(g)
    rc = lib().orc_fuse2(_p(a, C.c_float), _p(g, C.c_float), a.size, op)
    if rc != 0:
        raise ValueError("improper fusion method %d" % op)
    return a


def fuse_hm_n(a, g, n_maps):
    a = _f32(a).copy()
    g = _f32(g)
    lib().orc_fuse_hm_n(_p(a, C.c_float), _p(g, C.c_float), a.size, n_maps)
    return a


def accumulate(acc, g, mode):
    acc = _f32(acc).copy()
    g = _f32(g)
    lib().orc_accumulate(_p(acc, C.c_float), _p(g, C.c_float), acc.size, mode)
    return acc


def accumulate_begin(shape, mode):
    acc = np.empty(shape, np.float32)
    lib().orc_accumulate_begin(_p(acc, C.c_float), acc.size, mode)
    return acc


def fuse_nary(grids, mode):
    """n-ary camera fusion in accumulate / finalize form (modes 0 AM, 2 GM, 3 RMS, 4 min, 5 max)."""
    acc = accumulate_begin(np.asarray(grids[0]).shape, mode)
    for g in grids:
        acc = accumulate(acc, g, mode)
    return finalize(acc, mode, len(grids))


def fuse_gm_tree(grids):
    """Geometric mean of 2, 4 or 8 grids as the balanced tree of the reference's 2-ary op
    (Grid3D::geometricMeanTwoGrids, cartesian3dgrid.h:150-156): pairs (0,1), (2,3), ... then pairs of the
    results -- nothing but repeated calls of the restated reference member (fuse2 op 3)."""
    level = [_f32(g) for g in grids]
    if len(level) not in (2, 4, 8):
        raise ValueError("the tree form needs 2, 4 or 8 grids")
    while len(level) > 1:
        level = [fuse2(level[i], level[i + 1], 3) for i in range(0, len(level), 2)]
    return level[0]


def finalize(acc, mode, n_maps):
    acc = _f32(acc).copy()
    lib().orc_finalize(_p(acc, C.c_float), acc.size, mode, n_maps)
    return acc


def collapse_max_z(dsi):
    dsi = _f32(dsi)
    nz, ny, nx = dsi.shape
    conf = np.empty((ny, nx), np.float32)
    idx = np.empty((ny, nx), np.uint8)
    lib().orc_collapse_max_z(_p(dsi, C.c_float), nx, ny, nz, _p(conf, C.c_float),
                             _p(idx, C.c_uint8))
    return conf, idx


def indices_to_depth(idx, raw_depths):
    idx = np.ascontiguousarray(idx, np.uint8)
    raw_depths = _f32(raw_depths)
    out = np.empty(idx.shape, np.float32)
    lib().orc_indices_to_depth(_p(idx, C.c_uint8), idx.size, _p(raw_depths, C.c_float),
                               _p(out, C.c_float))
    return out


def mean_square(dsi):
    dsi = _f32(dsi)
    return float(lib().orc_mean_square(_p(dsi, C.c_float), dsi.size))


def pose_at(times, poses, t):
    times = np.ascontiguousarray(times, np.float64)
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
    out = np.empty(7, np.float64)
    ok = lib().orc_pose_at(_p(times, C.c_double), _p(poses, C.c_double), times.shape[0],
                           float(t), _p(out, C.c_double))
    return out if ok else None


def event_pose_Rt(T_rv_w, T_w_ev):
    a = np.ascontiguousarray(T_rv_w, np.float64)
    b = np.ascontiguousarray(T_w_ev, np.float64)
    out = np.empty(12, np.float32)
    lib().orc_event_pose_Rt(_p(a, C.c_double), _p(b, C.c_double), _p(out, C.c_float))
    return out


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def depth_map_filters(conf, idx, raw_depths, ksize=5, C_=5.0, median_size=5, max_confidence=0.0):
    """Post-arg-max filters of getDepthMapFromDSI (mapper_emvs_stereo.cpp:390-436).
    Returns dict(depth, confidence (with (0,0) overwritten), mask, conf8, idx_filtered)."""
    conf = _f32(conf).copy()
    idx = np.ascontiguousarray(idx, np.uint8)
    raw_depths = _f32(raw_depths)
    ny, nx = conf.shape
    conf8 = np.empty((ny, nx), np.uint8)
    mask = np.empty((ny, nx), np.uint8)
    filt = np.empty((ny, nx), np.uint8)
    depth = np.empty((ny, nx), np.float32)
    lib().orc_depth_map_filters(_p(conf, C.c_float), _p(idx, C.c_uint8), nx, ny, ksize, float(C_),
                                median_size, float(max_confidence), _p(raw_depths, C.c_float),
                                _p(conf8, C.c_uint8), _p(mask, C.c_uint8), _p(filt, C.c_uint8),
                                _p(depth, C.c_float))
    return {"depth": depth, "confidence": conf, "mask": mask, "conf8": conf8, "idx_filtered": filt}
