"""Orchestration of the reference's process_1 / process_2 (Alg. 1 / Alg. 2) over the GPU
engine: which mapper gets which events, where the reference view sits, the fusion order and
op codes, and the temporal accumulators.  Pure call sequencing -- every voxel operation is an
engine kernel.

    process_1   mapper_emvs_stereo/src/process1.cpp:28-224
    process_2   mapper_emvs_stereo/src/process2.cpp:28-302
    process_5   mapper_emvs_stereo/src/process5.cpp:28-260  (process_2 with the right camera's
                sub-intervals circularly shifted by num_subintervals/2)
    full_sequence / WindowStream   mapper_emvs_stereo/src/main.cpp:174-302  (--full_seq: a window of
                `duration` seconds every `out_skip` seconds, process_1 per window)
"""
import numpy as np

from . import engine as E
from . import synthetic as _q  # quaternion helpers only (numpy)


def pose_mul(a, b):
    """T_a * T_b for 7-vector poses (tx,ty,tz,qw,qx,qy,qz)."""
    qa, qb = np.asarray(a[3:], np.float64), np.asarray(b[3:], np.float64)
    w = qa[0] * qb[0] - qa[1] * qb[1] - qa[2] * qb[2] - qa[3] * qb[3]
    x = qa[0] * qb[1] + qa[1] * qb[0] + qa[2] * qb[3] - qa[3] * qb[2]
    y = qa[0] * qb[2] + qa[2] * qb[0] + qa[3] * qb[1] - qa[1] * qb[3]
    z = qa[0] * qb[3] + qa[3] * qb[0] + qa[1] * qb[2] - qa[2] * qb[1]
    t = np.asarray(a[:3], np.float64) + _q.quat_rotate(qa, np.asarray(b[:3], np.float64))
    return np.concatenate([t, [w, x, y, z]])


def apply_transformation_right(trajectory, T):
    """TrajectoryBase::applyTransformationRight (trajectory.hpp:57-63): every control pose becomes pose * T -- how
    main.cpp:199-216 turns the recorded poses into a camera's (hand-eye calibration, then the inverse extrinsics of the
    other cameras).  trajectory = (times, poses[n][7]); returns a new one."""
    times, poses = trajectory
    poses = np.asarray(poses, np.float64).reshape(-1, 7)
    return np.asarray(times, np.float64), np.stack([pose_mul(p, T) for p in poses])


def apply_transformation_left(trajectory, T):
    """TrajectoryBase::applyTransformationLeft (trajectory.hpp:65-71): every control pose becomes T * pose."""
    times, poses = trajectory
    poses = np.asarray(poses, np.float64).reshape(-1, 7)
    return np.asarray(times, np.float64), np.stack([pose_mul(T, p) for p in poses])


def reference_view_process1(trajectory_left, ts, rv_pos=0.0):
    """process1.cpp:56-68: T_w_rv = T_w_l(ts) * baseline(rv_pos along x); returns T_rv_w."""
    T_w_l = E.pose_at(trajectory_left, ts)
    if T_w_l is None:
        raise E.DsiError(E.ERR_INVALID, "no pose at the reference timestamp %r" % ts)
    baseline = np.array([rv_pos, 0, 0, 1, 0, 0, 0], np.float64)
    return _q.pose_inverse(pose_mul(T_w_l, baseline))


def reference_view_process2(trajectory_left, ts):
    """process2.cpp:79-81: T_rv_w = T_w_l(ts)^-1."""
    T = E.pose_at(trajectory_left, ts)
    if T is None:
        raise E.DsiError(E.ERR_INVALID, "no pose at the reference timestamp %r" % ts)
    return _q.pose_inverse(T)


def _fuse_cameras(fused, other, method):
    if method not in (1, 2, 3, 4, 5, 6):
        raise E.DsiError(E.ERR_BAD_OP, "Improper fusion method selected")
    fused.fuseTwoGrids(other, method)


def process_1(mappers, events, trajectories, mapper_fused, ts, fusion_method, rv_pos=0.0):
    """Alg. 1: one DSI per camera, fused across cameras (process1.cpp:54-191).
    mappers/events/trajectories: lists of 2 or 3 (a third camera is the EVIMO2 case).
    Returns T_rv_w; the fused DSI is mapper_fused.dsi_."""
    T_rv_w = reference_view_process1(trajectories[0], ts, rv_pos)
    for m, ev, tr in zip(mappers, events, trajectories):
        m.evaluateDSI(ev, tr, T_rv_w)                      # :76, :94, :110
    mapper_fused.dsi_.resetGrid()                          # :126
    mapper_fused.dsi_.addTwoGrids(mappers[0].dsi_)         # :127
    _fuse_cameras(mapper_fused.dsi_, mappers[1].dsi_, fusion_method)   # :136-158
    if len(mappers) > 2 and events[2][0].shape[0] > 0:     # :169-191
        g = mappers[2].dsi_
        if fusion_method == 1:
            mapper_fused.dsi_.minTwoGrids(g)
        elif fusion_method == 2:
            mapper_fused.dsi_.harmonicMeanTwoGrids(g, 3)
        elif fusion_method == 6:
            mapper_fused.dsi_.maxTwoGrids(g)
        # 3, 4, 5: the reference silently ignores the third camera
    return T_rv_w


def shuffled_subintervals(n_events, num_subintervals):
    """Index arrays of the right camera's sub-intervals in process_5 (process5.cpp:89-93,
    :136-150): start at shift*per with shift = n/2, wrap around the END OF THE EVENT VECTOR
    (not per*n), so a wrapped sub-interval is the tail followed by the head."""
    per = int(n_events) // int(num_subintervals)
    idx = (int(num_subintervals) // 2) * per
    out = []
    for _ in range(int(num_subintervals)):
        if idx + per >= n_events:
            sel = np.concatenate([np.arange(idx, n_events), np.arange(0, idx + per - n_events)])
            idx = idx + per - n_events
        else:
            sel = np.arange(idx, idx + per)
            idx += per
        out.append(sel)
    return out


def process_2(ctx, cams, dsi_shape, events, trajectories, num_subintervals, mapper_fused,
              mapper_fused_camera_time, ts, stereo_fusion, temporal_fusion, luts=(None, None),
              inverse_depth=False, shuffle_right=False):
    """Alg. 2: per sub-interval camera fusion, then temporal fusion; and the converse order
    (process2.cpp:46-289).  shuffle_right=True is process_5.  Returns dict with the left / right
    temporal DSIs (Grid3D)."""
    mapper0 = E.MapperEMVS(ctx, cams[0], dsi_shape, lut=luts[0], inverse_depth=inverse_depth)
    mapper1 = E.MapperEMVS(ctx, cams[1], dsi_shape, lut=luts[1], inverse_depth=inverse_depth)
    dims = mapper0.dsi_.getDimensions()
    sub = E.Grid3D(ctx, *dims)     # mapper_fused_subinterval.dsi_
    left = E.Grid3D(ctx, *dims)    # mapper_fused_left.dsi_
    right = E.Grid3D(ctx, *dims)   # mapper_fused_right.dsi_
    T_rv_w = reference_view_process2(trajectories[0], ts)
    per = [int(events[c][0].shape[0]) // int(num_subintervals) for c in range(2)]   # :46-47
    mapper_fused.dsi_.resetGrid()                                                    # :90
    right_sel = shuffled_subintervals(events[1][0].shape[0], num_subintervals) if shuffle_right else None
    for k in range(num_subintervals):
        for c, m in ((0, mapper0), (1, mapper1)):
            sl = slice(k * per[c], (k + 1) * per[c])                                 # :105-107, :132-134
            if c == 1 and right_sel is not None:
                sl = right_sel[k]                                                    # process5.cpp:136-150
            m.dsi_.resetGrid()
            m.evaluateDSI(tuple(a[sl] for a in events[c]), trajectories[c], T_rv_w)  # :119, :146
        sub.resetGrid()                                                              # :159
        sub.addTwoGrids(mapper0.dsi_)                                                # :160
        _fuse_cameras(sub, mapper1.dsi_, stereo_fusion)                              # :168-189
        if temporal_fusion == 2:                                                     # :216-226
            left.addInverseOfTwoGrids(mapper0.dsi_)
            right.addInverseOfTwoGrids(mapper1.dsi_)
            mapper_fused.dsi_.addInverseOfTwoGrids(sub)
            if k == num_subintervals - 1:
                for g in (left, right, mapper_fused.dsi_):
                    g.computeHMfromSumOfInv(num_subintervals)
        elif temporal_fusion == 4:                                                   # :229-239
            left.addTwoGrids(mapper0.dsi_)
            right.addTwoGrids(mapper1.dsi_)
            mapper_fused.dsi_.addTwoGrids(sub)
            if k == num_subintervals - 1:
                for g in (left, right, mapper_fused.dsi_):
                    g.computeAMfromSum(num_subintervals)
        # temporal_fusion 1, 3, 5, 6: nothing happens (:213, :227, :240-243)
    # converse order: time first, cameras second (:266-289).  NOTE the reference swaps cases 3
    # and 4 here (3 -> arithmetic, 4 -> geometric, process2.cpp:274-279); kept as is.
    mapper_fused_camera_time.dsi_.addTwoGrids(left)                                  # :266
    converse = {1: 1, 2: 2, 3: 4, 4: 3, 5: 5, 6: 6}
    if stereo_fusion not in converse:
        raise E.DsiError(E.ERR_BAD_OP, "Improper stereo fusion method selected")
    mapper_fused_camera_time.dsi_.fuseTwoGrids(right, converse[stereo_fusion])
    mapper0.close()
    mapper1.close()
    sub.close()
    return {"left": left, "right": right, "T_rv_w": T_rv_w}


def exact_depth_map_process_2(ctx, cams, dsi_shape, events, trajectories, num_subintervals, mapper_fused, ts,
                              stereo_fusion, temporal_fusion, luts=(None, None), inverse_depth=False, rel_gap=0.0, prove=False):
    """Alg. 2's fused DSI (camera fusion per sub-interval, then temporal fusion; process2.cpp:98-249) and its arg-max
    with the plane index map EQUAL TO THE CPU REFERENCE'S ON EVERY PIXEL (BASELINE configs[3]).  The engine's exact
    vote sums and the reference's fp32, event-ordered sums agree to ~1e-5, so the first-maximum plane can differ in
    near-tie columns; dsi_mapper_resolve_near_ties covers Alg. 1's topology (one camera fusion), this is the same idea
    built from its public pieces for the camera-then-time topology:
      1. Alg. 2 on the device as process_2 does it, every sub-interval's event batches kept;
      2. near-tie columns of the FINAL fused DSI (Grid3D near-tie voxels);
      3. those voxels of every (sub-interval, camera) DSI re-summed in the reference's order (MapperEMVS.exactVoxels);
      4. the reference's scalar ops on the host (engine.reference_fuse2 / _accumulate / _finalize) in process_2's
         order, the first maximum per column (cartesian3dgrid.cpp:132-134), and the few pixels patched.
    mapper_fused.dsi_ holds the fused DSI afterwards, mapper_fused's depth map (fetchDepthMap) the resolved arg-max.
    temporal_fusion 2 (harmonic) or 4 (arithmetic), like the reference (anything else leaves the fused DSI zero).
    Returns the statistics of the resolution.
    prove=True (ABI 10): the premise of that resolution as a per-column PROOF, for this topology built from INTERVAL GRIDS --
    every (sub-interval, camera) DSI's votes counted and its reference value bounded (MapperEMVS.referenceInterval), the bounds
    carried through the same camera fusion / temporal accumulate / finalize as the values (each widened by its roundings;
    the harmonic accumulation's inverse sums with the directions swapped), MapperEMVS.proveColumns at the end; columns the
    bounds do not settle are re-summed on all their planes (at most 4,096).  info["proof"] holds the statistics:
    columns_unproven == columns_resolved_fully means the index map is the reference's on every pixel by proof."""
    mappers = [E.MapperEMVS(ctx, cams[c], dsi_shape, lut=luts[c], inverse_depth=inverse_depth) for c in range(2)]
    dims = mappers[0].dsi_.getDimensions()
    sub = E.Grid3D(ctx, *dims)
    iv = None
    if prove and {2: E.ACC_INV_SUM, 4: E.ACC_SUM}.get(int(temporal_fusion)) is not None:
        iv = {k: E.Grid3D(ctx, *dims) for k in ("lo0", "hi0", "lo1", "hi1", "acc_lo", "acc_hi")}
        iv["acc_lo"].resetGrid()
        iv["acc_hi"].resetGrid()
    T_rv_w = reference_view_process2(trajectories[0], ts)
    per = [int(events[c][0].shape[0]) // int(num_subintervals) for c in range(2)]
    mode = {2: E.ACC_INV_SUM, 4: E.ACC_SUM}.get(int(temporal_fusion))
    mapper_fused.dsi_.resetGrid()
    batches = []
    for k in range(num_subintervals):
        pair = []
        for c in range(2):
            ev = tuple(a[k * per[c]:(k + 1) * per[c]] for a in events[c])
            pk = E.packetize(ev[2], trajectories[c], T_rv_w)
            first, Rt = pk if pk is not None else (np.zeros(0, np.uint32), np.zeros((0, 12), np.float32))
            b = E.EventBatch(ctx, ev[0], ev[1], Rt, first)
            if pk is None:
                mappers[c].dsi_.resetGrid()          # evaluateDSI returns false: the DSI keeps its reset state
            else:
                mappers[c].evaluateDSI_batch(b)
            pair.append(b)
        batches.append(pair)
        sub.resetGrid()
        sub.addTwoGrids(mappers[0].dsi_)
        _fuse_cameras(sub, mappers[1].dsi_, stereo_fusion)
        if mode is not None:
            mapper_fused.dsi_.accumulate(sub, mode)
            if k == num_subintervals - 1:
                mapper_fused.dsi_.finalize(mode, num_subintervals)
        if iv is not None:
            # the sub-interval's bounds: per camera from the counted votes, through the camera fusion (<= 5 roundings), into
            # the temporal accumulators (3 roundings per step).  The harmonic accumulator holds a SUM OF INVERSES, which falls
            # where the values rise: the lower values' accumulator bounds that sum from ABOVE, so it is widened upwards
            for c in range(2):
                if pair[c].n_packets:
                    mapper_fused.referenceInterval(mappers[c], pair[c], iv["lo%d" % c], iv["hi%d" % c])
                else:
                    iv["lo%d" % c].resetGrid()
                    iv["hi%d" % c].resetGrid()
            _fuse_cameras(iv["lo0"], iv["lo1"], stereo_fusion)       # (in place: lo0 <- op(lo0, lo1), like sub above)
            _fuse_cameras(iv["hi0"], iv["hi1"], stereo_fusion)
            E.widen_interval(iv["lo0"], iv["hi0"], 8)
            iv["acc_lo"].accumulate(iv["lo0"], mode)
            iv["acc_hi"].accumulate(iv["hi0"], mode)
            if mode == E.ACC_INV_SUM:
                E.widen_interval(iv["acc_hi"], iv["acc_lo"], 4)      # (swapped: see above)
            else:
                E.widen_interval(iv["acc_lo"], iv["acc_hi"], 4)
            if k == num_subintervals - 1:
                iv["acc_lo"].finalize(mode, num_subintervals)
                iv["acc_hi"].finalize(mode, num_subintervals)
                E.widen_interval(iv["acc_lo"], iv["acc_hi"], 4)
    mapper_fused.computeDepthMap()
    info = {"near_tie_pixels": 0, "candidate_voxels": 0, "votes": 0, "changed_pixels": 0, "max_order_diff": 0.0}
    if mode is not None:
        vox, cols = mapper_fused.nearTieVoxels(rel_gap=rel_gap)
        info["near_tie_pixels"], info["candidate_voxels"] = int(cols), int(vox.size)
        if vox.size:
            acc = np.zeros(vox.shape, np.float32)
            for k in range(num_subintervals):
                vals = []
                for c in range(2):
                    v, n = mappers[c].exactVoxels(batches[k][c], vox)
                    info["votes"] += int(n.sum())
                    vals.append(v)
                acc = E.reference_accumulate(mode, acc, E.reference_fuse2(stereo_fusion, vals[0], vals[1]))
            final = E.reference_finalize(mode, acc, num_subintervals)
            nx, ny, nz = dims
            pix, new_idx, new_conf = _first_maxima(vox, final, nx * ny)
            _, conf0, idx0 = mapper_fused.fetchDepthMap()
            info["changed_pixels"] = int((idx0.reshape(-1)[pix] != new_idx).sum())
            fused_now = mapper_fused.dsi_.download().reshape(-1)[vox]
            info["max_order_diff"] = float(np.max(np.abs(fused_now.astype(np.float64) - final) /
                                                  np.maximum(1.0, np.abs(final))))
            mapper_fused.patchDepthMap(pix, new_idx, new_conf)
    if iv is not None:
        proof = mapper_fused.proveColumns(mapper_fused.dsi_, iv["acc_lo"], iv["acc_hi"], rel_gap=rel_gap)
        if 0 < proof["columns_unproven"] <= 4096:
            # the columns the bounds do not settle: ALL their planes re-summed, in process_2's order, on the host
            pixels = np.unique(mapper_fused.proofUnproven()[0])
            nx, ny, nz = dims
            vox = (pixels[:, None].astype(np.uint64) + np.arange(nz, dtype=np.uint64)[None, :] * (nx * ny)).reshape(-1).astype(np.uint32)
            acc = np.zeros(vox.shape, np.float32)
            for k in range(num_subintervals):
                vals = [mappers[c].exactVoxels(batches[k][c], vox)[0] for c in range(2)]
                acc = E.reference_accumulate(mode, acc, E.reference_fuse2(stereo_fusion, vals[0], vals[1]))
            final = E.reference_finalize(mode, acc, num_subintervals)
            pix, new_idx, new_conf = _first_maxima(vox, final, nx * ny)
            mapper_fused.patchDepthMap(pix, new_idx, new_conf)
            proof["columns_resolved_fully"] = int(pixels.size)
        info["proof"] = proof
        for g in iv.values():
            g.close()
    for pair in batches:
        for b in pair:
            b.close()
    for o in mappers + [sub]:
        o.close()
    return info


def process_5(*args, **kw):
    """process5.cpp:28-260: process_2 with the right camera's sub-intervals circularly shifted."""
    kw["shuffle_right"] = True
    return process_2(*args, **kw)


def process_1_nary(mappers, events, trajectories, mapper_fused, ts, mode, rv_pos=0.0):
    """Alg. 1 for n >= 2 cameras with an n-ary mean across ALL cameras (BASELINE configs[4]: 4-camera
    rig, geometric mean).  The reference's process_1 fuses two cameras and, for GM / AM / RMS,
    silently drops a third one (process1.cpp:169-191); this is the n-ary form of the same ops
    (engine.ACC_SUM arithmetic, ACC_LOG_SUM geometric, ACC_SQ_SUM rms, ACC_MIN, ACC_MAX;
    include/dsi_engine.h dsi_acc_mode_t).  Returns T_rv_w; the fused DSI is mapper_fused.dsi_."""
    T_rv_w = reference_view_process1(trajectories[0], ts, rv_pos)
    for m, ev, tr in zip(mappers, events, trajectories):
        m.evaluateDSI(ev, tr, T_rv_w)
    mapper_fused.dsi_.setToFusionOfN([m.dsi_ for m in mappers], mode)
    return T_rv_w


def _first_maxima(vox, values, npix):
    """Per column of the (pixel-major, planes ascending) voxel list: pixel, first-maximum plane, its value."""
    pix_of = vox % npix
    z_of = (vox // npix).astype(np.int64)
    starts = np.flatnonzero(np.r_[True, pix_of[1:] != pix_of[:-1]])
    ends = np.r_[starts[1:], vox.size]
    j = np.array([a + int(np.argmax(values[a:b])) for a, b in zip(starts, ends)], np.int64)   # argmax: the first maximum
    return pix_of[starts].astype(np.uint32), z_of[j].astype(np.uint8), values[j].astype(np.float32)


def resolve_columns_fully(mapper_fused, mappers, batches, fusion_method, pixels):
    """ALL planes of the listed columns (pixels y * dimX + x) re-summed per camera in the reference's order
    (MapperEMVS.exactVoxels), fused with the reference's scalar op (engine.reference_fuse2; one camera: as they are), the first
    maximum taken (cartesian3dgrid.cpp:132-134) and the depth map of mapper_fused patched: these columns are then the
    reference's by construction, whatever their values.  For the FEW columns a proof cannot settle -- it costs
    planes x columns voxels.  Returns the number of pixels whose plane changed."""
    pixels = np.unique(np.asarray(pixels, np.uint32))
    if not pixels.size:
        return 0
    nx, ny, nz = mapper_fused.dsi_.getDimensions()
    npix = nx * ny
    vox = (pixels[:, None].astype(np.uint64) + np.arange(nz, dtype=np.uint64)[None, :] * npix).reshape(-1).astype(np.uint32)
    vals = [m.exactVoxels(b, vox)[0] for m, b in zip(mappers, batches)]   # (pixel-major, planes ascending)
    final = vals[0] if len(mappers) == 1 else E.reference_fuse2(fusion_method, vals[0], vals[1])
    pix, new_idx, new_conf = _first_maxima(vox, final, npix)
    idx0 = mapper_fused.fetchDepthMap()[2].reshape(-1)[pix]
    mapper_fused.patchDepthMap(pix, new_idx, new_conf)
    return int((idx0 != new_idx).sum())


def resolve_near_ties_proven(mapper_fused, mappers, batches, fusion_method=E.FUSE_HM, rel_gap=0.0, max_gap=4e-3,
                             max_full_columns=4096):
    """MapperEMVS.resolveNearTies in PROVEN mode (ABI 10): resolve, then MapperEMVS.proveNearTies -- every voxel's votes counted,
    the reference's fp32 event-order sums bounded from the counts.  Columns whose bounds reach beyond the gap: if a gap of at
    most `max_gap` covers some of them the resolver runs again with it (and the proof again); the columns that remain -- maxima
    made of a handful of tiny weights, where the engine's 2^-31 weight grid is as coarse as the values -- are re-summed on ALL
    their planes (resolve_columns_fully; at most `max_full_columns`, else they stay unproven).  Returns (resolver info, proof
    info); proof["columns_unproven"] - proof["columns_resolved_fully"] == 0 means the index map is the reference's on every
    pixel by proof.  The proof passes are verification passes (global-atomic vote counts): not for a timed loop."""
    info = mapper_fused.resolveNearTies(mappers, batches, fusion_method, rel_gap=rel_gap)
    proof = mapper_fused.proveNearTies(mappers, batches, fusion_method, rel_gap=info["rel_gap"])
    if proof["columns_unproven"]:
        pix, gaps = mapper_fused.proofUnproven()
        moderate = gaps[gaps.astype(np.float64) * 1.05 <= max_gap]
        if moderate.size:
            info = mapper_fused.resolveNearTies(mappers, batches, fusion_method, rel_gap=float(moderate.max()) * 1.05)
            proof = mapper_fused.proveNearTies(mappers, batches, fusion_method, rel_gap=info["rel_gap"])
            pix = mapper_fused.proofUnproven()[0] if proof["columns_unproven"] else pix[:0]
        if pix.size and pix.size <= max_full_columns:
            resolve_columns_fully(mapper_fused, mappers, batches, fusion_method, pix)
            proof["columns_resolved_fully"] = int(pix.size)
    return info, proof


def _fuse_nary_host(vals, mode):
    """The n-ary camera fusion of reference-order values on the host, with the reference's scalar ops (bit for bit what
    setToFusionOfN computes on the device)."""
    n = len(vals)
    if mode == E.ACC_GM_TREE:
        if n not in (2, 4, 8):
            raise E.DsiError(E.ERR_BAD_OP, "ACC_GM_TREE needs 2, 4 or 8 cameras")
        level = vals
        while len(level) > 1:       # geometricMeanTwoGrids on pairs, then on the results (cartesian3dgrid.h:150-156)
            level = [E.reference_fuse2(E.FUSE_GM, level[i], level[i + 1]) for i in range(0, len(level), 2)]
        return level[0]
    if mode in (E.ACC_MIN, E.ACC_MAX):
        final = vals[0]
        for v in vals[1:]:
            final = E.reference_fuse2(E.FUSE_MIN if mode == E.ACC_MIN else E.FUSE_MAX, final, v)
        return final
    if mode == E.ACC_SUM:
        acc = np.zeros(vals[0].shape, np.float32)
        for v in vals:
            acc = E.reference_accumulate(E.ACC_SUM, acc, v)
        return E.reference_finalize(E.ACC_SUM, acc, n)
    raise E.DsiError(E.ERR_BAD_OP, "mode %r has no host restatement" % (mode,))


def exact_depth_map_nary_proven(mapper_fused, mappers, batches, mode, rel_gap=0.0, fused_grid=None, max_gap=4e-3,
                                max_full_columns=4096):
    """exact_depth_map_nary in PROVEN mode (ABI 10, dsi_mapper_prove_near_ties_n): resolve, prove from counted votes, once
    more with a moderately wider gap where the bounds ask for it, and the columns no such gap settles re-summed on all their
    planes.  At BASELINE configs[4]'s size the oracle covers a strip of the image; this covers every column.  Returns
    (resolver info, proof info): proof["columns_unproven"] == proof["columns_resolved_fully"] means every column is settled."""
    gap = rel_gap if rel_gap > 0 else 2.5e-4
    info = exact_depth_map_nary(mapper_fused, mappers, batches, mode, rel_gap=gap, fused_grid=fused_grid)
    proof = mapper_fused.proveNearTiesN(mappers, batches, mode, fused_grid=fused_grid, rel_gap=gap)
    if proof["columns_unproven"]:
        pix, gaps = mapper_fused.proofUnproven()
        moderate = gaps[gaps.astype(np.float64) * 1.05 <= max_gap]
        if moderate.size:
            gap = float(moderate.max()) * 1.05
            info = exact_depth_map_nary(mapper_fused, mappers, batches, mode, rel_gap=gap, fused_grid=fused_grid)
            proof = mapper_fused.proveNearTiesN(mappers, batches, mode, fused_grid=fused_grid, rel_gap=gap)
            pix = mapper_fused.proofUnproven()[0] if proof["columns_unproven"] else pix[:0]
        if pix.size and pix.size <= max_full_columns:
            pixels = np.unique(pix)
            nx, ny, nz = mapper_fused.dsi_.getDimensions()
            npix = nx * ny
            vox = (pixels[:, None].astype(np.uint64) + np.arange(nz, dtype=np.uint64)[None, :] * npix).reshape(-1).astype(np.uint32)
            final = _fuse_nary_host([m.exactVoxels(b, vox)[0] for m, b in zip(mappers, batches)], mode)
            p2, new_idx, new_conf = _first_maxima(vox, final, npix)
            mapper_fused.patchDepthMap(p2, new_idx, new_conf)
            proof["columns_resolved_fully"] = int(pixels.size)
    info["rel_gap"] = gap
    return info, proof


def exact_depth_map_nary(mapper_fused, mappers, batches, mode, rel_gap=0.0, fused_grid=None):
    """The n-camera counterpart of MapperEMVS.resolveNearTies (BASELINE configs[4]): mapper_fused.dsi_ holds
    setToFusionOfN([m.dsi_ for m in mappers], mode) of the DSIs the mappers built from `batches`, and mapper_fused
    the depth map of it; the near-tie columns' voxels are re-summed per camera in the reference's order, fused on the
    host with the reference's scalar ops and the first maximum re-picked.  mode: ACC_GM_TREE (the balanced tree of the
    reference's 2-ary sqrt(a*b); n = 2, 4, 8), ACC_MIN, ACC_MAX, ACC_SUM (arithmetic mean).  fused_grid: the grid that
    holds the fusion when it is not mapper_fused.dsi_.  Returns statistics."""
    n = len(mappers)
    nx, ny, nz = mapper_fused.dsi_.getDimensions()
    vox, cols = mapper_fused.nearTieVoxels(grid=fused_grid, rel_gap=rel_gap)
    info = {"near_tie_pixels": int(cols), "candidate_voxels": int(vox.size), "votes": 0, "changed_pixels": 0}
    if not vox.size:
        return info
    vals = []
    for m, b in zip(mappers, batches):
        v, cnt = m.exactVoxels(b, vox)
        info["votes"] += int(cnt.sum())
        vals.append(v)
    final = _fuse_nary_host(vals, mode)
    pix, new_idx, new_conf = _first_maxima(vox, final, nx * ny)
    _, _, idx0 = mapper_fused.fetchDepthMap()
    info["changed_pixels"] = int((idx0.reshape(-1)[pix] != new_idx).sum())
    mapper_fused.patchDepthMap(pix, new_idx, new_conf)
    return info


def window_bounds(start_time_s, stop_time_s, duration, out_skip):
    """The intervals of main.cpp:177: for (t = start; t + duration <= stop; t += out_skip)."""
    out = []
    t = float(start_time_s)
    while t + duration <= stop_time_s:
        out.append((t, t + duration))
        t += out_skip
    return out


def window_events(events, t_start, t_stop):
    """Events of one camera with t_start <= ts <= t_stop, on an already time-sorted array.
    NOT identical to parse_rosbag (data_loading.cpp:272-285) on a real bag: the reference tests its stop flag only
    at the NEXT EventArray message, so it also keeps the first event past t_stop and the rest of that message
    (a few hundred events; the 1024-event packetisation can shift by one packet).  Message boundaries do not
    exist in an event array; io.read_event_bag reproduces the message-granular cut when reading a bag."""
    x, y, ts = events
    a = int(np.searchsorted(ts, t_start, side="left"))
    b = int(np.searchsorted(ts, t_stop, side="right"))
    return x[a:b], y[a:b], ts[a:b]


class WindowStream:
    """The --full_seq loop of main.cpp:174-302 with process_method 1 as a STREAM: per window
    packetise on the host, upload both cameras' events (pooled device blocks on the copy stream, so
    window w+1's upload overlaps window w's voting), evaluateDSI x2 (reset + vote), camera fusion,
    arg-max + depth; the depth map of window w is fetched while window w+1 is already queued
    (`depth` fused grids / extraction mappers alternate).  The reference constructs fresh mappers per
    window (main.cpp:262-275); here they are reused -- evaluateDSI resets the DSI anyway (:145)."""

    def __init__(self, ctx, cams, dsi_shape, fusion_method=E.FUSE_HM, luts=(None, None),
                 inverse_depth=False, depth=2, materialize_fused=True, fused_vote=False, concurrent=False,
                 exact_ties=False):
        """materialize_fused=False: the fused DSI (the reference's mapper_fused.dsi_) is not written;
        the camera fusion happens inside the arg-max kernel (same bits, one pass less over the
        volume) -- for streams that only keep the depth maps.
        fused_vote=True (implies materialize_fused=False): not even the camera DSIs are written -- one
        kernel votes, fuses and keeps the running arg-max on the CU (MapperEMVS.computeDepthMapOfEvents;
        the same depth maps bit for bit).
        exact_ties=True: every window's arg-max goes through the exact tie resolver (MapperEMVS.resolveNearTies), so
        that its plane index map equals the CPU reference's on every pixel; it needs the camera DSIs, so it excludes
        fused_vote; the call's statistics (and cost) are in `last_resolve`."""
        self.ctx = ctx
        self.exact_ties = bool(exact_ties)
        self.last_resolve = None
        fused_vote = bool(fused_vote) and not self.exact_ties
        self.fused_vote = bool(fused_vote)
        self.materialize_fused = bool(materialize_fused) and not self.fused_vote
        self.fusion_method = int(fusion_method)
        # concurrent=True: every slot owns a context (HIP stream) and its own pair of camera mappers, so that
        # consecutive windows are INDEPENDENT streams of work: the next window's preparation kernels and the
        # first workgroups of its voting kernel run on the CUs the current window's tail has already left
        # (the fused kernel is one persistent workgroup per CU; they finish between 0.5x and 1x its duration).
        # Windows are independent in the reference too (main.cpp:177: fresh mappers per window).
        self.concurrent = bool(concurrent) and depth > 1
        self.contexts = [ctx] + ([E.Context(ctx.device) for _ in range(depth - 1)] if self.concurrent else [])
        self._own_contexts = self.contexts[1:]
        sets = len(self.contexts)
        self.mapper_sets = [[E.MapperEMVS(self.contexts[k], cams[c], dsi_shape, lut=luts[c], inverse_depth=inverse_depth)
                             for c in range(2)] for k in range(sets)]
        self.mappers = self.mapper_sets[0]
        dims = self.mappers[0].dsi_.getDimensions()
        self.fused = [E.Grid3D(self.contexts[k % sets], *dims) if self.materialize_fused else None for k in range(depth)]
        self.extract = [E.MapperEMVS(self.contexts[k % sets], cams[0], dsi_shape, inverse_depth=inverse_depth)
                        for k in range(depth)]
        self.k = 0
        self.voted = 0
        self._pins = {}     # (slot, camera) -> page-locked (Rt, packet_first) staging of asynchronous uploads

    def _staging(self, slot, c, n_packets):
        cur = self._pins.get((slot, c))
        if cur is None or cur[0].a.shape[0] < n_packets:
            if cur is not None:
                for a in cur:
                    a.close()
            cap = max(1024, 2 * n_packets)
            cur = (E.PinnedArray((cap, 12), np.float32), E.PinnedArray((cap,), np.uint32))
            self._pins[(slot, c)] = cur
        return cur

    def submit(self, events, trajectories, ts, rv_pos=0.0, batches=None, asynchronous=False):
        """Queue one window (process1.cpp:54-166 + :222).  events: per camera (x, y, ts) of the
        window; batches: optional pre-uploaded EventBatch per camera (device-resident inputs).
        asynchronous: the x / y arrays live in page-locked memory (engine.PinnedArray) and stay
        unchanged until this window's result has been fetched; the uploads then run as plain DMAs
        and the host does not wait for them.  Returns the slot to pass to fetch()."""
        slot = self.k % len(self.fused)
        ctx = self.contexts[slot % len(self.contexts)]
        mappers = self.mapper_sets[slot % len(self.mapper_sets)]
        T_rv_w = reference_view_process1(trajectories[0], ts, rv_pos)
        own, fused_batches, all_batches = [], [], []
        for c in range(2):
            if batches is not None:
                b = batches[c]
            else:
                pk = E.packetize(events[c][2], trajectories[c], T_rv_w)
                if pk is None:                      # evaluateDSI returns false: < 1024 events (:71-75)
                    if not self.fused_vote and not self.exact_ties:
                        mappers[c].dsi_.resetGrid()
                        continue
                    pk = (np.zeros(0, np.uint32), np.zeros((0, 12), np.float32))   # a batch without packets
                first, Rt = pk
                if asynchronous:
                    pr, pf = self._staging(slot, c, first.shape[0])
                    pr.a[:Rt.shape[0]] = Rt
                    pf.a[:first.shape[0]] = first
                    Rt, first = pr.a[:Rt.shape[0]], pf.a[:first.shape[0]]
                b = E.EventBatch(ctx, events[c][0], events[c][1], Rt, first, asynchronous=asynchronous)
                own.append(b)
            all_batches.append(b)
            if self.fused_vote:
                fused_batches.append(b)
            elif b.n_packets:
                mappers[c].evaluateDSI_batch(b)
            else:                                   # a batch without packets: evaluateDSI resets the DSI and returns false;
                mappers[c].dsi_.resetGrid()         # the resolver takes the batch as it is -- no votes from this camera
            self.voted += b.n_packets * E.PACKET_SIZE
        if self.fused_vote:
            self.extract[slot].computeDepthMapOfEvents(mappers, fused_batches, self.fusion_method)
        elif self.materialize_fused:
            self.fused[slot].setToFusionOf(mappers[0].dsi_, mappers[1].dsi_, self.fusion_method)
            self.extract[slot].computeDepthMap(self.fused[slot])
        else:
            self.extract[slot].computeDepthMapOfFusion(mappers[0].dsi_, mappers[1].dsi_,
                                                       self.fusion_method)
        if self.exact_ties:
            self.last_resolve = self.extract[slot].resolveNearTies(mappers, all_batches, self.fusion_method)
        for b in own:
            b.close()                               # the block returns to the pool once its readers are done
        self.k += 1
        return slot

    def fetch(self, slot, options_depth_map=None):
        """(depth, confidence, indices) of the window submitted into `slot` (synchronises).  With
        options_depth_map (OptionsDepthMap): the reference's per-window outputs instead --
        (depth_map, confidence_map, mask) after the adaptive threshold, masked median and border removal
        of getDepthMapFromDSI (main.cpp:281 -> mapper_emvs_stereo.cpp:390-437), whichever way the
        window's arg-max was produced."""
        if options_depth_map is not None:
            return self.extract[slot].filterDepthMap(options_depth_map)
        # (every slot has its own context when the stream is concurrent: the copies follow the window's kernels on its
        #  compute stream -- one stream per window in flight, plus the device's shared upload stream)
        return self.extract[slot].fetchDepthMap(in_order=self.concurrent)

    def fused_grid(self, slot):
        return self.fused[slot]

    def context_of_slot(self, slot):
        """The context window `slot` runs in (pre-uploaded batches must live there).  Lifetime: with concurrent=True
        the stream OWNS the contexts of slots >= 1 and close() destroys them -- after closing every object still alive
        in them (Context.close() does that: a caller's EventBatch created here is closed with the stream; closing it
        again later is a no-op).  At the C level dsi_context_destroy refuses while children are alive."""
        return self.contexts[slot % len(self.contexts)]

    def close(self):
        for o in [m for ms in self.mapper_sets for m in ms] + [f for f in self.fused if f is not None] + self.extract:
            o.close()
        for c in self._own_contexts:
            c.close()
        self._own_contexts = []
        for pair in self._pins.values():
            for a in pair:
                a.close()
        self._pins = {}


def full_sequence(ctx, cams, dsi_shape, events, trajectories, start_time_s, stop_time_s, duration,
                  out_skip, fusion_method=E.FUSE_HM, forward_looking=True, rv_pos=0.0, options_depth_map=None, **kw):
    """Generator over the windows of main.cpp:177-302: yields (ts, depth, confidence, indices) per
    window, pipelined one window deep; with options_depth_map, (ts, depth_map, confidence_map, mask)
    -- the filtered outputs the reference saves per window."""
    ws = WindowStream(ctx, cams, dsi_shape, fusion_method, **kw)
    pending = None
    try:
        for t0, t1 in window_bounds(start_time_s, stop_time_s, duration, out_skip):
            ts = t1 if forward_looking else 0.5 * (t0 + t1)          # main.cpp:185-189
            ev = [window_events(events[c], t0, t1) for c in range(2)]
            slot = ws.submit(ev, trajectories, ts, rv_pos)
            if pending is not None:
                yield (pending[0],) + ws.fetch(pending[1], options_depth_map)
            pending = (ts, slot)
        if pending is not None:
            yield (pending[0],) + ws.fetch(pending[1], options_depth_map)
    finally:
        ws.close()
