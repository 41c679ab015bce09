"""A SECOND, independent restatement of the reference's DSI arithmetic, in numpy float32, written
from the reference sources (file:line cited) without looking at oracle/dsi_oracle.c's structure:
vectorised where the reference loops, and with the sequential `+=` order of the votes reproduced
through one unbuffered np.add.at per plane.  TEST INFRASTRUCTURE: it exists so that a misreading of
the reference would have to be made twice, in two differently shaped programs, to go unnoticed
(tests/test_oracle_kat.py compares the C oracle with this bit for bit).

numpy float32 arithmetic is IEEE single with one rounding per operation (no FMA), which is what the
reference's SSE2 build does.
"""
import numpy as np

F = np.float32
PACKET = 1024


def depth_planes(min_depth, max_depth, nz, inverse=False):
    """depth_vector.hpp:76-163.  Linear: vec_[i] = min + i / ((float)nz / (max - min)) (:88-98).
    Inverse: stored inverse depth 1/max + i / mult, raw depth = 1 / that (:131-148)."""
    lo, hi = F(min(min_depth, max_depth)), F(max(min_depth, max_depth))
    i = np.arange(nz).astype(F)
    if not inverse:
        mult = F(nz) / (hi - lo)
        return (lo + i / mult).astype(F)
    inv_min, inv_max = F(1.0) / lo, F(1.0) / hi
    mult = F(nz) / (inv_min - inv_max)
    return (F(1.0) / (inv_max + i / mult)).astype(F)


def _dot3(a0, b0, a1, b1, a2, b2):
    # Eigen 3.3 fixed-size inner product of length 3: x0 + (x1 + x2)
    return a0 * b0 + (a1 * b1 + a2 * b2)


def _inv3(m):
    """Eigen compute_inverse for 3x3 (cofactors; the first column's cofactors give the determinant)."""
    m = m.astype(F)

    def cof(i, j):
        i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
        return m[i1, j1] * m[i2, j2] - m[i1, j2] * m[i2, j1]

    c0, c1, c2 = cof(0, 0), cof(1, 0), cof(2, 0)
    det = _dot3(c0, m[0, 0], c1, m[1, 0], c2, m[2, 0])
    invdet = F(1.0) / det
    out = np.empty((3, 3), F)
    for i in range(3):
        for j in range(3):
            out[i, j] = cof(j, i) * invdet
    return out


def _mul3(a, b):
    out = np.empty((3, 3), F)
    for i in range(3):
        for j in range(3):
            out[i, j] = _dot3(a[i, 0], b[0, j], a[i, 1], b[1, j], a[i, 2], b[2, j])
    return out


def packet_geometry(Rt, K, Kv, z0):
    """mapper_emvs_stereo.cpp:108-120 for one packet: camera centre C = -R^T t (:108),
    H^-1 = z0 * R, col(2) += t (:114-116), H = (K * H^-1 * Kv^-1)^-1 (:119-120)."""
    Rt = np.asarray(Rt, F)
    R, t = Rt[:9].reshape(3, 3), Rt[9:]
    C = np.array([_dot3(-R[0, i], t[0], -R[1, i], t[1], -R[2, i], t[2]) for i in range(3)], F)
    Hinv = (R * F(z0)).astype(F)
    Hinv[:, 2] = Hinv[:, 2] + t
    Km = np.array([[K[0], 0, K[2]], [0, K[1], K[3]], [0, 0, 1]], F)
    Kvm = np.array([[Kv[0], 0, Kv[2]], [0, Kv[1], Kv[3]], [0, 0, 1]], F)
    H = _inv3(_mul3(_mul3(Km, Hinv), _inv3(Kvm)))
    return C, H


def warp_z0(x, y, H, lut=None, W=0):
    """mapper_emvs_stereo.cpp:129-142 for the events of one packet: p = H4 * (u, v, 1, 0), p /= p[2]."""
    if lut is None:
        u, v = x.astype(F), y.astype(F)
    else:
        p = np.asarray(lut, F).reshape(-1, 2)[y.astype(np.int64) * W + x.astype(np.int64)]
        u, v = p[:, 0], p[:, 1]
    one, zero = F(1.0), F(0.0)
    px = ((H[0, 0] * u + H[0, 1] * v) + H[0, 2] * one) + zero
    py = ((H[1, 0] * u + H[1, 1] * v) + H[1, 2] * one) + zero
    pz = ((H[2, 0] * u + H[2, 1] * v) + H[2, 2] * one) + zero
    return np.stack([px / pz, py / pz], axis=1).astype(F)


def fill_voxel_grid(xy_z0, centers, planes, Kv, nx, ny, dsi=None):
    """mapper_emvs_stereo.cpp:151-205 + Grid3D::accumulateGridValueAt (cartesian3dgrid.h:253-273).
    Per plane the votes are applied in packet order, event order, corner order g[0], g[1], g[nx],
    g[nx+1] -- one unbuffered np.add.at over that exact sequence."""
    xy = np.asarray(xy_z0, F).reshape(-1, 2)
    centers = np.asarray(centers, F).reshape(-1, 3)
    planes = np.asarray(planes, F)
    fx, fy, cx, cy = (F(v) for v in Kv)
    nz = planes.shape[0]
    if dsi is None:
        dsi = np.zeros((nz, ny, nx), F)
    z0 = planes[0]                                                    # :163
    npk = centers.shape[0]
    x0 = xy[:, 0].reshape(npk, PACKET)
    y0 = xy[:, 1].reshape(npk, PACKET)
    Cx, Cy, Cz = centers[:, 0:1], centers[:, 1:2], centers[:, 2:3]
    with np.errstate(all="ignore"):
        for k in range(nz):
            zi = planes[k]
            a = z0 * (zi - Cz)                                        # :177
            bx = (z0 - zi) * (Cx * fx + Cz * cx)                      # :178
            by = (z0 - zi) * (Cy * fy + Cz * cy)                      # :179
            d = zi * (z0 - Cz)                                        # :182
            X = ((x0 * a + bx) / d).reshape(-1)                       # :194
            Y = ((y0 * a + by) / d).reshape(-1)                       # :195
            ok = (X >= 0) & (Y >= 0)                                  # cartesian3dgrid.h:255 (NaN fails)
            big = F(2.0 ** 30)
            Xc = np.where(ok & (X < big), X, F(-1.0))
            Yc = np.where(ok & (Y < big), Y, F(-1.0))
            xi = Xc.astype(np.int64)                                  # (int) truncation, :257
            yi = Yc.astype(np.int64)
            ok &= (X < big) & (Y < big) & (xi + 1 < nx) & (yi + 1 < ny)   # :258-259 (huge -> INT_MIN on x86: rejected)
            X, Y, xi, yi = X[ok], Y[ok], xi[ok], yi[ok]
            fxx = X - xi.astype(F)
            fyy = Y - yi.astype(F)
            one = F(1.0)
            w = np.stack([(one - fxx) * (one - fyy), fxx * (one - fyy), (one - fxx) * fyy, fxx * fyy], axis=1)
            base = yi * nx + xi
            idx = np.stack([base, base + 1, base + nx, base + nx + 1], axis=1)
            np.add.at(dsi[k].reshape(-1), idx.reshape(-1), w.astype(F).reshape(-1))   # :261-270, in order
    return dsi


def fuse2(a, g, op):
    """cartesian3dgrid.h:111-192 (op codes of process1.cpp:136-158)."""
    a, g = np.asarray(a, F), np.asarray(g, F)
    with np.errstate(all="ignore"):
        if op == 1:
            return np.where(g < a, g, a)                                      # std::min(a, g), :115
        if op == 2:
            return (F(2.0) * (a * g) / ((a + g) + F(0.1))).astype(F)          # :119-127, eps 1e-1 (float)
        if op == 3:
            return np.sqrt(a * g).astype(F)                                   # :154 (sqrt of the float product)
        if op == 4:
            return (0.5 * (a + g).astype(np.float64)).astype(F)               # :162, 0.5 is a double
        if op == 5:
            ms = (0.5 * (a.astype(np.float64) ** 2 + g.astype(np.float64) ** 2)).astype(F)   # :145, pow in double
            return np.sqrt(ms.astype(np.float64)).astype(F)                   # :146
        if op == 6:
            return np.where(a < g, g, a)                                      # std::max(a, g), :188
    raise ValueError(op)


def fuse_hm_n(a, g, n):
    """cartesian3dgrid.h:130-139."""
    a, g = np.asarray(a, F), np.asarray(g, F)
    av = a / F(n - 1)
    return (F(n) * (av * g) / ((av + g) + F(0.1))).astype(F)


def accumulate(acc, g, mode):
    """cartesian3dgrid.h:64-78."""
    acc, g = np.asarray(acc, F), np.asarray(g, F)
    return (acc + g).astype(F) if mode == 0 else (acc + F(1.0) / (F(0.01) + g)).astype(F)


def finalize(acc, mode, n):
    """cartesian3dgrid.h:80-93."""
    acc = np.asarray(acc, F)
    with np.errstate(all="ignore"):
        return (acc / F(n)).astype(F) if mode == 0 else (F(n) / acc).astype(F)


def collapse_max_z(dsi):
    """cartesian3dgrid.cpp:115-137: std::max_element keeps the FIRST maximum; np.argmax does too."""
    dsi = np.asarray(dsi, F)
    idx = np.argmax(dsi, axis=0)
    return np.take_along_axis(dsi, idx[None], axis=0)[0], idx.astype(np.uint8)
