"""The fused vote -> camera fusion -> arg-max kernel (dsi_mapper_depth_map_of_events) against the
unfused sequence it replaces, and the property that makes the comparison exact: the DSI's bits do not
depend on the band decomposition (seam rows are rounded once from the exact 64-bit sums).

Reference sequence: process1.cpp:76-166 (evaluateDSI x 2, camera fusion) + :222 ->
mapper_emvs_stereo.cpp:368 (collapseMaxZSlice) + :302-313 (index -> depth).

Every comparison here is `array_equal`: same confidence bits, same indices, same depths.
"""
import numpy as np
import pytest

import dvs_mcemvs_amd as d
from dvs_mcemvs_amd import engine, process as proc, synthetic as syn
from oracle import oracle as orc
from oracle_pipeline import OracleMapper

pytestmark = pytest.mark.gpu


def rig_batches(ctx, rig, n_cams=2):
    out = []
    for c in range(n_cams):
        first, Rt = d.packetize(rig["events"][c][2], rig["trajectories"][c], rig["T_rv_w"])
        out.append(d.EventBatch(ctx, rig["events"][c][0], rig["events"][c][1], Rt, first))
    return out


def unfused(ctx, mappers, batches, op):
    for m, b in zip(mappers, batches):
        m.evaluateDSI_batch(b)
    if len(mappers) == 2:
        mappers[0].computeDepthMapOfFusion(mappers[0].dsi_, mappers[1].dsi_, op)
    else:
        mappers[0].computeDepthMap()
    return mappers[0].fetchDepthMap()


@pytest.mark.parametrize("packed", [1, 3, 5, 6])
@pytest.mark.parametrize("shape,band_rows", [((96, 72, 32), 0), ((96, 72, 32), 7), ((130, 97, 9), 5),
                                             ((346, 260, 20), 0), ((64, 48, 256), 11)])
def test_fused_equals_vote_fuse_collapse(ctx, packed, shape, band_rows):
    nx, ny, nz = shape
    rig = syn.stereo_rig(30_000, width=nx, height=ny, duration=0.25, seed=31 + nx, n_points=900)
    dsi_shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    batches = rig_batches(ctx, rig)
    ref_m = [d.MapperEMVS(ctx, rig["cam"], dsi_shape) for _ in range(2)]
    fus_m = [d.MapperEMVS(ctx, rig["cam"], dsi_shape) for _ in range(2)]
    out = d.MapperEMVS(ctx, rig["cam"], dsi_shape)
    for m in fus_m + [out]:          # (the fused path reads its knobs from the OUTPUT mapper of the call)
        m.set_packed_lanes(packed)
        if band_rows:
            m.set_band_params(band_rows, 0, 0)
    for op in (d.FUSE_MIN, d.FUSE_HM, d.FUSE_GM, d.FUSE_AM, d.FUSE_RMS, d.FUSE_MAX):
        want = unfused(ctx, ref_m, batches, op)
        out.computeDepthMapOfEvents(fus_m, batches, op)
        got = out.fetchDepthMap()
        info = fus_m[0].last_vote_info()
        assert info["algo"] == d.VOTE_FUSED_ARGMAX and info["packed"] == packed
        if band_rows:
            assert info["band_rows"] == band_rows and info["bands"] == -(-ny // band_rows)
        for g, w, name in zip(got, want, ("depth", "confidence", "index")):
            assert np.array_equal(g, w), "op %d %s differs at %d pixels" % (op, name, (g != w).sum())
        assert want[1].max() > 1.0
    # one camera: vote -> arg-max (`out` may be the voting mapper itself)
    want = unfused(ctx, ref_m[:1], batches[:1], 0)
    fus_m[0].computeDepthMapOfEvents(fus_m[:1], batches[:1], 0)
    got = fus_m[0].fetchDepthMap()
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    for o in ref_m + fus_m + [out] + batches:
        o.close()


def test_fused_against_the_oracle(ctx):
    """Not only self-consistent: indices equal the CPU oracle's wherever its top-2 gap exceeds the DSI
    tolerance, and everywhere else the GPU's plane is a provable near-tie of the oracle's fused DSI."""
    nx, ny, nz = 120, 90, 40
    rig = syn.stereo_rig(60_000, width=nx, height=ny, duration=0.3, seed=8, n_points=1500)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    batches = rig_batches(ctx, rig)
    ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(2)]
    ms[0].computeDepthMapOfEvents(ms, batches, d.FUSE_HM)
    depth, conf, idx = ms[0].fetchDepthMap()
    refs = []
    for c in range(2):
        r = OracleMapper(rig["cam"], dimZ=nz, min_depth=4.0, max_depth=200.0)
        assert r.evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
        refs.append(r.dsi)
    rf = orc.fuse2(refs[0], refs[1], 2)
    rconf, ridx = orc.collapse_max_z(rf)
    tol = 2e-4 * np.maximum(1.0, rconf)
    picked = np.take_along_axis(rf, idx[None].astype(np.int64), axis=0)[0]
    assert np.all((idx == ridx) | (picked >= rconf - tol)), "a GPU arg-max that is not a near-tie of the oracle's"
    assert np.allclose(conf, rconf, rtol=1e-4, atol=1e-4)
    assert np.array_equal(depth, ms[0].raw_depths_vec_[idx])
    assert (idx == ridx).mean() > 0.97
    for o in ms + batches:
        o.close()


def test_fused_camera_without_packets(ctx):
    """evaluateDSI returns false for < 1024 events (mapper_emvs_stereo.cpp:71-75): that camera's DSI is
    all zero; HM with a zero volume is zero everywhere -> confidence 0, index 0."""
    nx, ny, nz = 96, 72, 16
    rig = syn.stereo_rig(20_000, width=nx, height=ny, duration=0.2, seed=3)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    b0 = rig_batches(ctx, rig, 1)[0]
    empty = d.EventBatch(ctx, np.zeros(0, np.uint16), np.zeros(0, np.uint16), np.zeros((0, 12), np.float32),
                         np.zeros(0, np.uint32))
    ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(2)]
    ms[0].computeDepthMapOfEvents(ms, [b0, empty], d.FUSE_HM)
    depth, conf, idx = ms[0].fetchDepthMap()
    assert not conf.any() and not idx.any() and np.all(depth == ms[0].raw_depths_vec_[0])
    # max(dsi0, 0) = dsi0: the single-camera arg-max
    ms[0].computeDepthMapOfEvents(ms, [b0, empty], d.FUSE_MAX)
    got = ms[0].fetchDepthMap()
    ms[1].evaluateDSI_batch(b0)
    ms[1].computeDepthMap()
    want = ms[1].fetchDepthMap()
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    # the empty camera first
    ms[0].computeDepthMapOfEvents(ms, [empty, b0], d.FUSE_MAX)
    for g, w in zip(ms[0].fetchDepthMap(), want):
        assert np.array_equal(g, w)
    # both empty: an all-zero fused DSI
    ms[0].computeDepthMapOfEvents(ms, [empty, empty], d.FUSE_AM)
    depth, conf, idx = ms[0].fetchDepthMap()
    assert not conf.any() and not idx.any()
    for o in ms + [b0, empty]:
        o.close()


@pytest.mark.parametrize("packed", [1, 5])
def test_fused_with_slow_planes_unequal_cameras_and_a_lut(ctx, packed):
    """The paths a plain rig does not reach: a rectification LUT with huge and NaN entries (locations above 2^40
    make every plane of their packet take the IEEE-divide stream; NaN locations are dropped), cameras with
    different packet counts, inverse depth planes, a virtual camera from a field of view.  Still bit-identical to
    vote -> fuse -> collapse."""
    nx, ny, nz = 120, 90, 24
    rig = syn.stereo_rig(50_000, width=nx, height=ny, duration=0.3, seed=5, n_points=700)
    lut = syn.radial_lut(rig["cam"])
    lut[7 * nx + 11] = (3.0e13, 5.0)          # |x0| > 2^40 after the warp: the "slow" stream
    lut[20 * nx + 40] = (np.nan, 1.0)
    lut[33 * nx + 3] = (2.0, -np.inf)
    shape = d.ShapeDSI(100, 80, nz, 3.0, 150.0, 70.0)
    ev = [rig["events"][0], tuple(a[:31_000] for a in rig["events"][1])]
    for e in ev:                               # make sure the special pixels fire in several packets
        e[0][::997] = 11
        e[1][::997] = 7
        e[0][5::1013] = 40
        e[1][5::1013] = 20
    batches = []
    for c in range(2):
        first, Rt = d.packetize(ev[c][2], rig["trajectories"][c], rig["T_rv_w"])
        batches.append(d.EventBatch(ctx, ev[c][0], ev[c][1], Rt, first))
    assert batches[0].n_packets != batches[1].n_packets
    mk = lambda: [d.MapperEMVS(ctx, rig["cam"], shape, lut=lut, inverse_depth=True) for _ in range(2)]
    ref_m, fus_m = mk(), mk()
    for m in fus_m:
        m.set_packed_lanes(packed)
    for op in (d.FUSE_HM, d.FUSE_GM, d.FUSE_MAX):
        want = unfused(ctx, ref_m, batches, op)
        fus_m[1].computeDepthMapOfEvents(fus_m, batches, op)
        got = fus_m[1].fetchDepthMap()
        for g, w, name in zip(got, want, ("depth", "confidence", "index")):
            assert np.array_equal(g, w), "op %d %s differs at %d pixels" % (op, name, (g != w).sum())
    assert want[1].max() > 1.0 and np.isfinite(want[1]).all()
    for o in ref_m + fus_m + batches:
        o.close()


def _reference_sequence(ctx, mappers, out, batches, op):
    """evaluateDSI per camera, then process1.cpp:126-191 with Grid3D calls, then the arg-max -- the unfused path."""
    for m, bt in zip(mappers, batches):
        if bt.n_packets:
            m.evaluateDSI_batch(bt)
        else:
            m.dsi_.resetGrid()          # fewer than 1024 events: evaluateDSI returns false, the DSI stays all zero (:71-75)
    if len(mappers) == 1:
        out.computeDepthMap(mappers[0].dsi_)
        return out.fetchDepthMap()
    fused = out.dsi_
    fused.resetGrid()
    fused.addTwoGrids(mappers[0].dsi_)
    proc._fuse_cameras(fused, mappers[1].dsi_, op)
    if len(mappers) == 3:
        if op == 1:
            fused.minTwoGrids(mappers[2].dsi_)
        elif op == 2:
            fused.harmonicMeanTwoGrids(mappers[2].dsi_, 3)
        elif op == 6:
            fused.maxTwoGrids(mappers[2].dsi_)
    out.computeDepthMap(fused)
    return out.fetchDepthMap()


@pytest.mark.parametrize("seed", list(range(16)))
def test_fused_fuzz_over_shapes_bands_and_mappings(ctx, seed):
    """Random grids (odd widths, one-row bands, a single plane, 256 planes, more rows than events), band heights,
    lane mappings, fusion ops, event counts per camera, one to three cameras, each with its own intrinsics and depth
    range: the fused kernel always gives the bits of the reference's sequence of calls."""
    rng = np.random.default_rng(900 + seed)
    nx = int(rng.choice([2, 3, 17, 64, 129, 346, 700]))
    ny = int(rng.choice([2, 5, 48, 97, 260]))
    nz = int(rng.choice([1, 2, 7, 33, 256]))
    if nx * ny * nz > 6_000_000:
        nz = 7
    n_ev = int(rng.choice([1025, 3000, 20_000, 70_000]))
    n_cams = int(rng.choice([1, 2, 2, 3]))
    rig = syn.stereo_rig(n_ev, width=max(nx, 8), height=max(ny, 8), duration=0.2, seed=50 + seed, n_points=300, n_cams=max(n_cams, 2))
    far, fov = float(rng.uniform(20.0, 200.0)), float(rng.choice([0.0, 60.0]))
    packed = int(rng.choice([-1, 1, 3, 5, 6]))
    band_rows = int(rng.choice([0, 1, 2, 3, 9, 40]))
    op = int(rng.integers(1, 7))
    w, h, fx, fy, cx, cy = rig["cam"]
    cams, shapes, batches = [], [], []
    for c in range(n_cams):
        own = bool(rng.integers(0, 2)) and c > 0          # this camera has a calibration of its own
        cams.append((w, h, fx * float(rng.uniform(0.95, 1.05)), fy * float(rng.uniform(0.95, 1.05)),
                     cx + float(rng.uniform(-3, 3)), cy + float(rng.uniform(-3, 3))) if own else rig["cam"])
        shapes.append(d.ShapeDSI(nx, ny, nz, 2.0 if not own else 2.5, far if not own else far * 0.8, fov))
        keep = n_ev if c == 0 else int(rng.choice([n_ev, max(1025, n_ev // 2), 700]))   # 700 < 1024: evaluateDSI returns false
        ev = tuple(a[:keep] for a in rig["events"][c])
        pk = d.packetize(ev[2], rig["trajectories"][c], rig["T_rv_w"])
        first, Rt = pk if pk is not None else (np.zeros(0, np.uint32), np.zeros((0, 12), np.float32))
        batches.append(d.EventBatch(ctx, ev[0], ev[1], Rt, first))
    mk = lambda: [d.MapperEMVS(ctx, cams[c], shapes[c]) for c in range(n_cams)]
    ref_m, fus_m = mk(), mk()
    ref_out, fus_out = d.MapperEMVS(ctx, cams[0], shapes[0]), d.MapperEMVS(ctx, cams[0], shapes[0])
    for m in fus_m + [fus_out]:
        m.set_packed_lanes(packed)
        if band_rows:
            m.set_band_params(band_rows, 0, 0)
    want = _reference_sequence(ctx, ref_m, ref_out, batches, op)
    fus_out.computeDepthMapOfEvents(fus_m, batches, op)
    got = fus_out.fetchDepthMap()
    for g, w_, name in zip(got, want, ("depth", "confidence", "index")):
        assert np.array_equal(g, w_), "seed %d (%dx%dx%d, %d events, %d cameras, mapping %d, band %d, op %d): %s differs at %d pixels" % (
            seed, nx, ny, nz, n_ev, n_cams, packed, band_rows, op, name, (g != w_).sum())
    for o in ref_m + fus_m + batches + [ref_out, fus_out]:
        o.close()


def test_fused_rejects_bad_arguments(ctx):
    rig = syn.stereo_rig(5_000, width=64, height=48, duration=0.1, seed=1)
    a = d.MapperEMVS(ctx, rig["cam"], d.ShapeDSI(0, 0, 8, 4.0, 100.0, 0.0))
    b = d.MapperEMVS(ctx, rig["cam"], d.ShapeDSI(0, 0, 9, 4.0, 100.0, 0.0))
    batches = rig_batches(ctx, rig)
    with pytest.raises(d.DsiError) as e:
        a.computeDepthMapOfEvents([a, b], batches, d.FUSE_HM)
    assert e.value.code == engine.ERR_SHAPE
    with pytest.raises(d.DsiError) as e:
        a.computeDepthMapOfEvents([a, a], batches, d.FUSE_HM)
    assert e.value.code == engine.ERR_INVALID
    c = d.MapperEMVS(ctx, rig["cam"], d.ShapeDSI(0, 0, 8, 4.0, 100.0, 0.0))
    with pytest.raises(d.DsiError) as e:
        a.computeDepthMapOfEvents([a, c], batches, 9)
    assert e.value.code == engine.ERR_BAD_OP
    for o in [a, b, c] + batches:
        o.close()


@pytest.mark.parametrize("packed", [0, 1, 5])
def test_dsi_bits_do_not_depend_on_bands_or_chunks(ctx, packed):
    """Every voxel is fl(exact 64-bit sum of its votes): on the seam rows between bands the sums of the
    two neighbours are added as integers (k_seam_rows), and several packet chunks leave raw 64-bit partial
    volumes that are added as integers (k_reduce_partials) -- one rounding, whatever the decomposition."""
    nx, ny, nz = 130, 97, 12
    rig = syn.stereo_rig(40_000, width=nx, height=ny, duration=0.25, seed=17, n_points=400)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    batch = rig_batches(ctx, rig, 1)[0]
    base = None
    for chunks in (1, 3, 8):
        for band_rows in (97, 3, 10, 31):
            m = d.MapperEMVS(ctx, rig["cam"], shape)
            m.set_vote_algo(d.VOTE_LDS_BANDS)
            m.set_packed_lanes(packed)
            m.set_band_params(band_rows, chunks, 0)
            m.evaluateDSI_batch(batch)
            info = m.last_vote_info()
            assert info["band_rows"] == band_rows and info["chunks"] == chunks
            got = m.dsi_.download()
            m.close()
            if base is None:
                base = got                              # one band, one chunk: no seam, no partial volume
                assert base.max() > 4.0
            assert np.array_equal(got, base), (chunks, band_rows, int((got != base).sum()))
    batch.close()


def test_window_stream_fused_vote_is_bit_identical(ctx):
    """BASELINE configs[2] shape (512x512x200, 2 x 500 k events per 50 ms window): the stream with
    fused_vote=True returns the depth maps of the materialising stream bit for bit."""
    NX, NY, NZ, EV, DUR, NWIN = 512, 512, 200, 500_000, 0.05, 4
    t0 = 10.0
    rig = syn.stereo_rig(NWIN * EV, width=640, height=480, t0=t0, duration=NWIN * DUR, seed=77, n_points=6000)
    shape = d.ShapeDSI(NX, NY, NZ, 4.0, 200.0, 0.0)
    bounds = proc.window_bounds(t0, t0 + NWIN * DUR + 1e-9, DUR, DUR)
    a = proc.WindowStream(ctx, (rig["cam"],) * 2, shape, d.FUSE_HM)
    b = proc.WindowStream(ctx, (rig["cam"],) * 2, shape, d.FUSE_HM, fused_vote=True)
    for (lo, hi) in bounds:
        ev = [proc.window_events(rig["events"][c], lo, hi) for c in range(2)]
        want = a.fetch(a.submit(ev, rig["trajectories"], hi))
        got = b.fetch(b.submit(ev, rig["trajectories"], hi))
        for g, w, name in zip(got, want, ("depth", "confidence", "index")):
            assert np.array_equal(g, w), "%s differs at %d pixels" % (name, (g != w).sum())
        assert want[1].max() > 1.0
    info = b.mappers[0].last_vote_info()
    assert info["algo"] == d.VOTE_FUSED_ARGMAX and info["bands"] == 14 and info["band_rows"] == 37
    a.close()
    b.close()


def test_window_stream_filtered_outputs_without_a_dsi(ctx):
    """main.cpp:281: every window ends in getDepthMapFromDSI(depth, confidence, mask, options).  The
    fused-vote stream has no DSI to hand to it; fetch(slot, options) applies the same filters
    (mapper_emvs_stereo.cpp:390-437) to the arg-max the fused kernel produced -- identical to
    getDepthMapFromDSI on the materialised fused DSI, for the concurrent stream too."""
    NX, NY, NZ, EV, DUR, NWIN = 346, 260, 100, 120_000, 0.05, 3
    t0 = 3.0
    rig = syn.stereo_rig(NWIN * EV, width=346, height=260, t0=t0, duration=NWIN * DUR, seed=5, n_points=3000)
    shape = d.ShapeDSI(NX, NY, NZ, 4.0, 120.0, 0.0)
    opts = d.OptionsDepthMap()
    opts.max_confidence = 0
    bounds = proc.window_bounds(t0, t0 + NWIN * DUR + 1e-9, DUR, DUR)
    a = proc.WindowStream(ctx, (rig["cam"],) * 2, shape, d.FUSE_HM)
    streams = [proc.WindowStream(ctx, (rig["cam"],) * 2, shape, d.FUSE_HM, fused_vote=True, concurrent=conc)
               for conc in (False, True)]
    for (lo, hi) in bounds:
        ev = [proc.window_events(rig["events"][c], lo, hi) for c in range(2)]
        slot = a.submit(ev, rig["trajectories"], hi)
        want = a.extract[slot].getDepthMapFromDSI(a.fused_grid(slot), opts)
        assert want[2].sum() > 100                      # the mask keeps something
        for b in streams:
            got = b.fetch(b.submit(ev, rig["trajectories"], hi), opts)
            for g, w, name in zip(got, want, ("depth", "confidence", "mask")):
                assert np.array_equal(g, w, equal_nan=True), "%s differs at %d pixels" % (name, (g != w).sum())
    # the raw map was consumed by the filters: a second pass has nothing to work on
    with pytest.raises(d.DsiError) as e:
        streams[0].extract[(streams[0].k - 1) % 2].filterDepthMap(opts)
    assert e.value.code == engine.ERR_INVALID
    a.close()
    for b in streams:
        b.close()


def test_fused_cameras_with_their_own_calibration(ctx):
    """process1.cpp:73-110 builds one mapper per camera from that camera's own calibration: sensor intrinsics K_
    (mapper_emvs_stereo.cpp:46-48), the virtual camera derived from its fx (:219-239), its LUT, its depth planes.
    The fused call takes all of that per camera -- not from camera 0 -- and converts indices to depths with the
    OUTPUT mapper's planes, like mapper_fused.getDepthMapFromDSI does."""
    nx, ny, nz = 160, 120, 40
    rig = syn.stereo_rig(60_000, width=nx, height=ny, duration=0.3, seed=9, n_points=900)
    w, h, fx, fy, cx, cy = rig["cam"]
    cams = [rig["cam"], (w, h, fx * 1.04, fy * 0.97, cx + 3.5, cy - 2.25)]
    shapes = [d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0), d.ShapeDSI(0, 0, nz, 3.0, 120.0, 0.0)]   # own z0 and planes too
    out_shape = d.ShapeDSI(0, 0, nz, 5.0, 90.0, 0.0)
    batches = rig_batches(ctx, rig)
    mk = lambda: [d.MapperEMVS(ctx, cams[c], shapes[c]) for c in range(2)] + [d.MapperEMVS(ctx, cams[0], out_shape)]
    ref_m, fus_m = mk(), mk()
    for op in (d.FUSE_HM, d.FUSE_MIN, d.FUSE_AM):
        for m, b in zip(ref_m[:2], batches):
            m.evaluateDSI_batch(b)
        ref_m[2].computeDepthMapOfFusion(ref_m[0].dsi_, ref_m[1].dsi_, op)
        want = ref_m[2].fetchDepthMap()
        fus_m[2].computeDepthMapOfEvents(fus_m[:2], batches, op)
        got = fus_m[2].fetchDepthMap()
        for g, w_, name in zip(got, want, ("depth", "confidence", "index")):
            assert np.array_equal(g, w_), "op %d: %s differs at %d pixels" % (op, name, (g != w_).sum())
        assert want[1].max() > 1.0
    # and the calibration matters: camera 1 voted with camera 0's intrinsics gives another map
    same = [d.MapperEMVS(ctx, cams[0], shapes[c]) for c in range(2)]
    fus_m[2].computeDepthMapOfEvents(same, batches, d.FUSE_AM)
    assert not np.array_equal(fus_m[2].fetchDepthMap()[1], want[1])
    for o in ref_m + fus_m + same + batches:
        o.close()


@pytest.mark.parametrize("packed", [-1, 1, 3, 5, 6])
@pytest.mark.parametrize("shape,band_rows", [((140, 100, 36), 9), ((96, 72, 20), 0), ((346, 260, 12), 0)])
def test_fused_four_cameras_geometric_mean_tree(ctx, packed, shape, band_rows):
    """BASELINE configs[4]'s rig (round 6, VERDICT r05 "missing" 3): four cameras fused by the balanced tree of the
    reference's 2-ary geometric mean (cartesian3dgrid.h:150-156: sqrt(sqrt(c0 c1) sqrt(c2 c3))) and the arg-max, in the one
    kernel that never writes a DSI -- bit for bit the depth map of evaluateDSI x 4 + computeDepthMapOfFusionN(GM tree), for
    every lane mapping, unequal packet counts, a camera without packets; with two cameras the tree is FUSE_GM."""
    nx, ny, nz = shape
    rig = syn.stereo_rig(60_000, width=nx, height=ny, duration=0.3, seed=33, n_points=900, n_cams=4)
    rig["events"][2] = tuple(a[:41_000] for a in rig["events"][2])
    rig["events"][3] = tuple(a[:13_000] for a in rig["events"][3])
    sh = d.ShapeDSI(0, 0, nz, 4.0, 150.0, 0.0)
    batches = rig_batches(ctx, rig, 4)
    ref_m = [d.MapperEMVS(ctx, rig["cam"], sh) for _ in range(4)]
    fus_m = [d.MapperEMVS(ctx, rig["cam"], sh) for _ in range(5)]       # fus_m[4]: the output mapper
    for m in fus_m:
        m.set_packed_lanes(packed)
        m.set_band_params(band_rows, 0, 0)
    for m, b in zip(ref_m, batches):
        m.evaluateDSI_batch(b)
    ref_m[0].computeDepthMapOfFusionN([m.dsi_ for m in ref_m], d.ACC_GM_TREE)
    want = ref_m[0].fetchDepthMap()
    fus_m[4].computeDepthMapOfEventsN(fus_m[:4], batches)
    got = fus_m[4].fetchDepthMap()
    for g, w, name in zip(got, want, ("depth", "confidence", "index")):
        assert np.array_equal(g, w), "%s differs at %d pixels" % (name, (g != w).sum())
    assert want[1].max() > 0.5 and fus_m[4].last_vote_info()["algo"] == d.VOTE_FUSED_ARGMAX
    # again (the keys are self-clearing), then two cameras: the tree is the reference's 2-ary op
    fus_m[4].computeDepthMapOfEventsN(fus_m[:4], batches)
    assert np.array_equal(fus_m[4].fetchDepthMap()[2], want[2])
    fus_m[4].computeDepthMapOfEventsN(fus_m[:2], batches[:2])
    two = fus_m[4].fetchDepthMap()
    fus_m[4].computeDepthMapOfEvents(fus_m[:2], batches[:2], d.FUSE_GM)
    assert all(np.array_equal(a, b) for a, b in zip(two, fus_m[4].fetchDepthMap()))
    # a camera without packets: its DSI is all zero, so is the geometric mean
    empty = d.EventBatch(ctx, np.zeros(0, np.uint16), np.zeros(0, np.uint16), np.zeros((0, 12), np.float32),
                         np.zeros(0, np.uint32))
    fus_m[4].computeDepthMapOfEventsN(fus_m[:4], batches[:3] + [empty])
    assert not fus_m[4].fetchDepthMap()[1].any()
    for n_bad in (1, 3):
        with pytest.raises(d.DsiError):
            fus_m[4].computeDepthMapOfEventsN(fus_m[:n_bad], batches[:n_bad])
    with pytest.raises(d.DsiError):
        fus_m[4].computeDepthMapOfEventsN(fus_m[:4], batches, d.ACC_LOG_SUM)
    for o in ref_m + fus_m + batches + [empty]:
        o.close()


@pytest.mark.parametrize("n_cams", [2, 4])
def test_fused_kernel_with_the_pairs_taken_in_turn(ctx, n_cams):
    """Round 6: when the bands an XCD works on at once hold more records than its L2 (here 4 / 2 cameras x 1.2 M events at
    512 x 512 x 24: 5 bands x 1.8 / 0.9 MB), the XCD's workgroups take the (band, plane) pairs of its stretch IN TURN -- all 32 on
    consecutive planes of one band -- instead of a contiguous piece each: the first pair fixed, every further pair DRAWN from
    the XCD's counter behind the arg-max keys (and from the other XCDs' once that one is dry).  Which workgroup votes a
    pair changes no bit: the depth map is that of evaluateDSI x n + the fusion inside the arg-max.  Called twice: the
    counters must be zero again for the second call (k_unpack_argmax clears them with the keys)."""
    nx, ny, nz = 512, 512, 24
    rig = syn.stereo_rig(1_200_000, width=nx, height=ny, duration=0.3, seed=91, n_points=4000, n_cams=4)
    sh = d.ShapeDSI(0, 0, nz, 4.0, 150.0, 0.0)
    batches = rig_batches(ctx, rig, n_cams)
    ref_m = [d.MapperEMVS(ctx, rig["cam"], sh) for _ in range(n_cams)]
    fus_m = [d.MapperEMVS(ctx, rig["cam"], sh) for _ in range(n_cams + 1)]
    for m, b in zip(ref_m, batches):
        m.evaluateDSI_batch(b)
    if n_cams == 4:
        ref_m[0].computeDepthMapOfFusionN([m.dsi_ for m in ref_m], d.ACC_GM_TREE)
        fus_m[-1].computeDepthMapOfEventsN(fus_m[:4], batches)
    else:
        ref_m[0].computeDepthMapOfFusion(ref_m[0].dsi_, ref_m[1].dsi_, d.FUSE_HM)
        fus_m[-1].computeDepthMapOfEvents(fus_m[:2], batches, d.FUSE_HM)
    want = ref_m[0].fetchDepthMap()
    for call in range(2):
        if call:
            if n_cams == 4:
                fus_m[-1].computeDepthMapOfEventsN(fus_m[:4], batches)
            else:
                fus_m[-1].computeDepthMapOfEvents(fus_m[:2], batches, d.FUSE_HM)
        got = fus_m[-1].fetchDepthMap()
        for g, w, name in zip(got, want, ("depth", "confidence", "index")):
            assert np.array_equal(g, w), "call %d: %s differs at %d pixels" % (call, name, (g != w).sum())
    assert want[1].max() > 0.5
    for o in ref_m + fus_m + batches:
        o.close()


@pytest.mark.parametrize("packed", [-1, 3, 5])
def test_fused_three_cameras_follow_process_1(ctx, packed):
    """The trinocular rig (EVIMO2; process1.cpp:105-117, :169-191): fused = op(dsi0, dsi1), then min /
    harmonicMeanTwoGrids(dsi2, 3) / max with the third camera -- and nothing for ops 3, 4, 5, which the reference's
    switch lets fall through.  The fused kernel with three cameras gives the depth map of that sequence + arg-max bit
    for bit; different packet counts per camera, one band height that leaves a ragged last band."""
    nx, ny, nz = 140, 100, 36
    rig = syn.stereo_rig(45_000, width=nx, height=ny, duration=0.3, seed=21, n_points=800, n_cams=3)
    rig["events"][2] = tuple(a[:28_000] for a in rig["events"][2])
    shape = d.ShapeDSI(0, 0, nz, 4.0, 150.0, 0.0)
    batches = rig_batches(ctx, rig, 3)
    assert batches[2].n_packets < batches[0].n_packets
    ref_m = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(4)]
    fus_m = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(4)]
    for m in fus_m:                  # fus_m[3] is the output mapper of the calls below
        m.set_packed_lanes(packed)
        m.set_band_params(9, 0, 0)
    for m, b in zip(ref_m[:3], batches):
        m.evaluateDSI_batch(b)
    two = {}
    for op in range(1, 7):
        fused = ref_m[3].dsi_
        fused.resetGrid()                                   # process1.cpp:126-127
        fused.addTwoGrids(ref_m[0].dsi_)
        proc._fuse_cameras(fused, ref_m[1].dsi_, op)        # :136-158
        if op == 1:                                         # :169-191
            fused.minTwoGrids(ref_m[2].dsi_)
        elif op == 2:
            fused.harmonicMeanTwoGrids(ref_m[2].dsi_, 3)
        elif op == 6:
            fused.maxTwoGrids(ref_m[2].dsi_)
        want = ref_m[3].getDepthMapFromDSI()
        fus_m[3].computeDepthMapOfEvents(fus_m[:3], batches, op)
        got = fus_m[3].fetchDepthMap()
        for g, w, name in zip(got, want, ("depth", "confidence", "index")):
            assert np.array_equal(g, w), "op %d: %s differs at %d pixels" % (op, name, (g != w).sum())
        assert want[1].max() > 0.5
        fus_m[3].computeDepthMapOfEvents(fus_m[:2], batches[:2], op)
        two[op] = fus_m[3].fetchDepthMap()[1]
        # the third camera changes the map exactly for the ops whose third step exists
        assert np.array_equal(two[op], want[1]) == (op in (3, 4, 5))
    # a third camera without packets (evaluateDSI returned false: an all-zero DSI): min wipes the map, max keeps it
    empty = d.EventBatch(ctx, np.zeros(0, np.uint16), np.zeros(0, np.uint16), np.zeros((0, 12), np.float32),
                         np.zeros(0, np.uint32))
    fus_m[3].computeDepthMapOfEvents(fus_m[:3], batches[:2] + [empty], d.FUSE_MIN)
    assert not fus_m[3].fetchDepthMap()[1].any()
    fus_m[3].computeDepthMapOfEvents(fus_m[:3], batches[:2] + [empty], d.FUSE_MAX)
    assert np.array_equal(fus_m[3].fetchDepthMap()[1], two[d.FUSE_MAX])
    with pytest.raises(d.DsiError) as e:
        fus_m[3].computeDepthMapOfEvents(fus_m[:3] + [ref_m[0]], batches + [batches[0]], d.FUSE_HM)
    assert e.value.code == engine.ERR_INVALID
    for o in ref_m + fus_m + batches + [empty]:
        o.close()


def test_distinct_contexts_from_distinct_threads(ctx):
    """include/dsi_engine.h: "Distinct contexts may be used from distinct threads" (the reference's evaluateDSI is not
    re-entrant: function-static scratch, mapper_emvs_stereo.cpp:79-83).  Four host threads, each with its own context,
    mappers and batches, run the unfused and the fused path concurrently; every thread gets the bits the main thread
    got alone."""
    import threading
    nx, ny, nz = 200, 150, 48
    rigs = [syn.stereo_rig(40_000 + 5_000 * k, width=nx, height=ny, duration=0.25, seed=70 + k, n_points=900) for k in range(4)]
    shape = d.ShapeDSI(0, 0, nz, 4.0, 180.0, 0.0)

    def work(context, rig, rounds):
        batches = rig_batches(context, rig)
        ms = [d.MapperEMVS(context, rig["cam"], shape) for _ in range(2)]
        out = d.MapperEMVS(context, rig["cam"], shape)
        res = []
        for it in range(rounds):
            op = (d.FUSE_HM, d.FUSE_MIN, d.FUSE_AM)[it % 3]
            a = unfused(context, ms, batches, op)
            out.computeDepthMapOfEvents(ms, batches, op)
            b = out.fetchDepthMap()
            res.append((a, b))
        for o in ms + [out] + batches:
            o.close()
        return res

    want = [work(ctx, rig, 3) for rig in rigs]
    got, errors = [None] * 4, []

    def run(k):
        try:
            c = d.Context(0)
            got[k] = work(c, rigs[k], 3)
            c.close()
        except Exception as e:      # noqa: BLE001 -- reported below, from the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=run, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(4):
        for it, ((wa, wb), (ga, gb)) in enumerate(zip(want[k], got[k])):
            for w_, g_, name in zip(wa + wb, ga + gb, ("depth", "confidence", "index") * 2):
                assert np.array_equal(w_, g_), "thread %d round %d: %s differs" % (k, it, name)
            for x, y in zip(ga, gb):
                assert np.array_equal(x, y)     # and fused == unfused there too
