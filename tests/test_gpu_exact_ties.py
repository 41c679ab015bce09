"""The exact tie resolver (dsi_mapper_resolve_near_ties): "depth map equal to the CPU reference" on EVERY pixel.

The engine sums a voxel's votes exactly and rounds once; the reference adds them in fp32 in event order
(cartesian3dgrid.h:261-270 inside mapper_emvs_stereo.cpp:197-201).  Where a column's best planes are closer than
that difference the first-maximum plane (cartesian3dgrid.cpp:132-134) can differ -- by a whole plane of depth.  The
resolver re-sums the contending voxels in the reference's order; afterwards the index map must equal the oracle's
with `array_equal`, no "safe pixel" mask."""
import numpy as np
import pytest

import dvs_mcemvs_amd as d
from dvs_mcemvs_amd import process as proc, synthetic as syn
from oracle import oracle as orc
from oracle_pipeline import OracleMapper

pytestmark = pytest.mark.gpu


def _batches(ctx, rig, n):
    out = []
    for c in range(n):
        first, Rt = d.packetize(rig["events"][c][2], rig["trajectories"][c], rig["T_rv_w"])
        out.append(d.EventBatch(ctx, rig["events"][c][0], rig["events"][c][1], Rt, first))
    return out


def _oracle_fused(rig, n, op, dims, lut=None, inverse=False, depths=(4.0, 200.0)):
    nx, ny, nz = dims
    dsis = []
    for c in range(n):
        r = OracleMapper(rig["cam"], dimX=nx, dimY=ny, dimZ=nz, min_depth=depths[0], max_depth=depths[1], lut=lut,
                         inverse_depth=inverse)
        assert r.evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
        dsis.append(r.dsi)
    ref = dsis[0] if n == 1 else orc.fuse2(dsis[0].copy(), dsis[1], op)
    return ref, r.planes


@pytest.mark.parametrize("rel_gap,min_ranks", [(0.05, 3_000), (0.45, 31_000)])
def test_resolver_with_very_many_contenders(ctx, rel_gap, min_ranks):
    """A gap far beyond rounding size makes (nearly) every voxel of a column a contender: tens of thousands of (camera, voxel)
    ranks.  Below 30,720 ranks the recorded votes are partitioned through the per-stretch table in LDS (round 6: counting
    launch -> k_tie_colscan -> scattering launch, no global atomic), above it with one global atomic per vote; short and empty
    runs, runs that are not a multiple of the addition kernel's 64 weights per turn, and rows of a wave whose runs end at
    different turns all occur.  Whatever the gap, re-summing MORE voxels in the reference's order must give the oracle's map."""
    nx, ny, nz = 64, 48, 24
    rig = syn.stereo_rig(60_000, width=nx, height=ny, duration=0.3, seed=77, n_points=500)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    batches = _batches(ctx, rig, 2)
    ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(2)]
    for m, b in zip(ms, batches):
        m.evaluateDSI_batch(b)
    out = d.MapperEMVS(ctx, rig["cam"], shape)
    out.computeDepthMapOfFusion(ms[0].dsi_, ms[1].dsi_, d.FUSE_HM)
    ref, planes = _oracle_fused(rig, 2, d.FUSE_HM, (nx, ny, nz))
    rconf, ridx = orc.collapse_max_z(ref)
    info = out.resolveNearTies(ms, batches, d.FUSE_HM, rel_gap=rel_gap)
    depth, conf, idx = out.fetchDepthMap()
    assert 2 * info["candidate_voxels"] >= min_ranks, info
    assert info["gap_widenings"] == 0 and info["premise_ok"] == 1, info
    assert np.array_equal(idx, ridx), "%d pixels differ; %r" % ((idx != ridx).sum(), info)
    assert np.array_equal(depth, planes[ridx])
    assert np.allclose(conf, rconf, rtol=1e-4, atol=1e-6)
    for o in ms + [out] + batches:
        o.close()


@pytest.mark.parametrize("n_cams,op", [(1, 0), (2, d.FUSE_HM), (2, d.FUSE_MIN), (2, d.FUSE_GM), (2, d.FUSE_AM),
                                       (2, d.FUSE_RMS), (2, d.FUSE_MAX)])
@pytest.mark.parametrize("events", [12_000, 150_000])
def test_resolved_index_map_equals_the_oracles(ctx, n_cams, op, events):
    """Few events: most columns hold a handful of votes and many planes tie exactly or nearly; many events: ties
    are rare but real.  Either way the resolved map is the oracle's, pixel for pixel."""
    nx, ny, nz = 96, 72, 32
    rig = syn.stereo_rig(events, width=nx, height=ny, duration=0.3, seed=5 + events % 7, n_points=700)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    batches = _batches(ctx, rig, n_cams)
    ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(n_cams)]
    for m, b in zip(ms, batches):
        m.evaluateDSI_batch(b)
    out = d.MapperEMVS(ctx, rig["cam"], shape)
    if n_cams == 1:
        out.computeDepthMap(ms[0].dsi_)
    else:
        out.computeDepthMapOfFusion(ms[0].dsi_, ms[1].dsi_, op)
    _, conf0, idx0 = out.fetchDepthMap()
    ref, planes = _oracle_fused(rig, n_cams, op, (nx, ny, nz))
    rconf, ridx = orc.collapse_max_z(ref)
    info = out.resolveNearTies(ms, batches, op)
    depth, conf, idx = out.fetchDepthMap()
    assert np.array_equal(idx, ridx), "%d of %d pixels differ after the resolver (%d before); %r" % (
        (idx != ridx).sum(), idx.size, (idx0 != ridx).sum(), info)
    assert np.array_equal(depth, planes[ridx])
    assert info["changed_pixels"] == int((idx0 != idx).sum())
    assert info["near_tie_pixels"] > 0 and info["candidate_voxels"] >= 2 * info["near_tie_pixels"]
    # the patched pixels carry the reference-order confidence (exactly the oracle's); the others the exact-sum one
    changed = idx0 != idx
    assert np.array_equal(conf[changed], rconf[changed])
    assert np.allclose(conf, rconf, rtol=1e-4, atol=1e-6)
    # the call's own check of its premise: the two summation orders differ by far less than the gap it re-sums
    assert 8 * info["max_order_diff"] < info["rel_gap"], info
    for o in ms + [out] + batches:
        o.close()


def test_resolver_with_lut_inverse_depth_and_the_fused_vote_kernel(ctx):
    """Distortion LUT + inverse depth planes; the depth map to resolve comes from the fused vote -> fusion -> arg-max
    kernel (no DSI written by it; the camera DSIs the resolver reads are built by evaluateDSI_batch)."""
    nx, ny, nz = 120, 90, 40
    rig = syn.stereo_rig(80_000, width=nx, height=ny, duration=0.3, seed=3, n_points=900)
    lut = syn.radial_lut(rig["cam"])
    shape = d.ShapeDSI(0, 0, nz, 3.0, 60.0, 0.0)
    batches = _batches(ctx, rig, 2)
    ms = [d.MapperEMVS(ctx, rig["cam"], shape, lut=lut, inverse_depth=True) for _ in range(2)]
    for m, b in zip(ms, batches):
        m.evaluateDSI_batch(b)
    out = d.MapperEMVS(ctx, rig["cam"], shape, lut=lut, inverse_depth=True)
    out.computeDepthMapOfEvents(ms, batches, d.FUSE_HM)
    ref, planes = _oracle_fused(rig, 2, d.FUSE_HM, (nx, ny, nz), lut=lut, inverse=True, depths=(3.0, 60.0))
    rconf, ridx = orc.collapse_max_z(ref)
    info = out.resolveNearTies(ms, batches, d.FUSE_HM)
    depth, conf, idx = out.fetchDepthMap()
    assert np.array_equal(idx, ridx), info
    assert np.array_equal(depth, planes[ridx])
    # a second call finds the same contenders and changes nothing
    again = out.resolveNearTies(ms, batches, d.FUSE_HM)
    assert again["changed_pixels"] == 0 and again["candidate_voxels"] == info["candidate_voxels"]
    for o in ms + [out] + batches:
        o.close()


@pytest.mark.parametrize("mode", ["gm_tree", "min", "max", "am"])
def test_exact_depth_map_of_four_cameras(ctx, mode):
    """configs[4]'s topology (4 cameras, n-ary fusion) at a size the oracle covers whole: after
    process.exact_depth_map_nary the index map equals the arg-max of the oracle's n-ary fusion on every pixel."""
    nx, ny, nz = 96, 72, 24
    rig = syn.stereo_rig(60_000, width=nx, height=ny, duration=0.3, seed=13, n_cams=4, n_points=600)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    batches = _batches(ctx, rig, 4)
    ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(4)]
    for m, b in zip(ms, batches):
        m.evaluateDSI_batch(b)
    fused = d.MapperEMVS(ctx, rig["cam"], shape)
    acc = {"gm_tree": d.ACC_GM_TREE, "min": d.ACC_MIN, "max": d.ACC_MAX, "am": d.ACC_SUM}[mode]
    fused.dsi_.setToFusionOfN([m.dsi_ for m in ms], acc)
    fused.computeDepthMap()
    _, _, idx0 = fused.fetchDepthMap()
    dsis = []
    for c in range(4):
        r = OracleMapper(rig["cam"], dimX=nx, dimY=ny, dimZ=nz, min_depth=4.0, max_depth=200.0)
        assert r.evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
        dsis.append(r.dsi)
    ref = orc.fuse_gm_tree(dsis) if mode == "gm_tree" else orc.fuse_nary(dsis, acc)
    ridx = ref.argmax(axis=0)
    info = process_nary(fused, ms, batches, acc)
    _, conf, idx = fused.fetchDepthMap()
    assert np.array_equal(idx, ridx), "%d pixels differ (%d before); %r" % ((idx != ridx).sum(), (idx0 != ridx).sum(), info)
    assert info["changed_pixels"] == int((idx0 != idx).sum()) and info["near_tie_pixels"] > 0
    for o in ms + [fused] + batches:
        o.close()


def process_nary(fused, ms, batches, acc):
    return proc.exact_depth_map_nary(fused, ms, batches, acc)


@pytest.mark.parametrize("mode", ["gm_tree", "min", "max", "am"])
def test_proven_mode_of_four_cameras(ctx, mode):
    """dsi_mapper_prove_near_ties_n (ABI 10): the per-column proof for configs[4]'s topology.  With a gap of rounding size the
    contested columns are unproven and gap_needed is rounding-sized; process.exact_depth_map_nary_proven settles every column
    (wider gap, the hard columns on all planes), and the map is the oracle's.  At configs[4]'s own size the oracle only covers a
    strip -- there the proof is what covers the rest of the image."""
    nx, ny, nz = 96, 72, 24
    rig = syn.stereo_rig(60_000, width=nx, height=ny, duration=0.3, seed=13, n_cams=4, n_points=600)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    batches = _batches(ctx, rig, 4)
    ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(4)]
    for m, b in zip(ms, batches):
        m.evaluateDSI_batch(b)
    fused = d.MapperEMVS(ctx, rig["cam"], shape)
    acc = {"gm_tree": d.ACC_GM_TREE, "min": d.ACC_MIN, "max": d.ACC_MAX, "am": d.ACC_SUM}[mode]
    fused.dsi_.setToFusionOfN([m.dsi_ for m in ms], acc)
    fused.computeDepthMap()
    tiny = fused.proveNearTiesN(ms, batches, acc, rel_gap=1e-7)
    assert tiny["columns"] == nx * ny and tiny["columns_unproven"] > 0 and 1e-7 < tiny["gap_needed"] <= 1.0, tiny
    for c in range(4):      # the counters of every camera against the resolver's own event pass
        vox = np.random.default_rng(c).integers(0, nx * ny * nz, 2000).astype(np.uint32)
        assert np.array_equal(fused.proofVotes(c, vox), ms[c].exactVoxels(batches[c], vox)[1])
    info, proof = proc.exact_depth_map_nary_proven(fused, ms, batches, acc)
    assert proof["columns_proven"] + proof["columns_unproven"] == nx * ny
    assert proof["columns_unproven"] == proof["columns_resolved_fully"], (info, proof)
    dsis = []
    for c in range(4):
        r = OracleMapper(rig["cam"], dimX=nx, dimY=ny, dimZ=nz, min_depth=4.0, max_depth=200.0)
        assert r.evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
        dsis.append(r.dsi)
    ref = orc.fuse_gm_tree(dsis) if mode == "gm_tree" else orc.fuse_nary(dsis, acc)
    _, _, idx = fused.fetchDepthMap()
    assert np.array_equal(idx, ref.argmax(axis=0)), "%d pixels differ; %r" % ((idx != ref.argmax(axis=0)).sum(), proof)
    with pytest.raises(d.DsiError):
        fused.proveNearTiesN(ms[:3], batches[:3], d.ACC_GM_TREE)       # the tree needs 2, 4 or 8 cameras
    with pytest.raises(d.DsiError):
        fused.proveNearTiesN(ms, batches, d.ACC_LOG_SUM)               # exp(mean(log)): no reference counterpart to bound
    for o in ms + [fused] + batches:
        o.close()


def test_resolver_checks_its_premise_and_widens_the_gap(ctx):
    """the gap is a premise (the rigorous per-voxel bound (votes - 1) * 2^-24 can exceed it), so the call measures
    the difference between the two summation orders on the voxels it re-sums and repeats the pass with a 4 x wider gap while
    8 * max_order_diff >= rel_gap.  Forced here by asking for a gap of 4 x the measured difference: one widening, then ok."""
    nx, ny, nz = 96, 72, 32
    rig = syn.stereo_rig(150_000, width=nx, height=ny, duration=0.3, seed=8, n_points=700)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    batches = _batches(ctx, rig, 2)
    ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(2)]
    for m, b in zip(ms, batches):
        m.evaluateDSI_batch(b)
    out = d.MapperEMVS(ctx, rig["cam"], shape)
    out.computeDepthMapOfFusion(ms[0].dsi_, ms[1].dsi_, d.FUSE_HM)
    first = out.resolveNearTies(ms, batches, d.FUSE_HM)
    assert first["premise_ok"] == 1 and first["gap_widenings"] == 0
    diff = first["max_order_diff"]
    if diff <= 0:
        pytest.skip("the two summation orders agree exactly on this input")
    out.computeDepthMapOfFusion(ms[0].dsi_, ms[1].dsi_, d.FUSE_HM)
    ref, _ = _oracle_fused(rig, 2, d.FUSE_HM, (nx, ny, nz))
    _, ridx = orc.collapse_max_z(ref)
    _, _, idx = out.fetchDepthMap()
    assert np.array_equal(idx, ridx), first
    asked = 4.0 * diff
    forced = out.resolveNearTies(ms, batches, d.FUSE_HM, rel_gap=asked)
    # (a narrower gap re-sums fewer voxels and may measure a smaller difference: the mechanics are what is asserted)
    assert 0 <= forced["gap_widenings"] <= 3
    assert forced["rel_gap"] == pytest.approx(asked * 4 ** forced["gap_widenings"], rel=1e-5)
    assert forced["premise_ok"] == int(8 * forced["max_order_diff"] < forced["rel_gap"]), forced
    if forced["max_order_diff"] >= 0.99 * diff:
        assert forced["gap_widenings"] >= 1, forced
    # a gap that three widenings cannot bring above 8 x the difference: reported, not hidden
    out.computeDepthMapOfFusion(ms[0].dsi_, ms[1].dsi_, d.FUSE_HM)
    hopeless = out.resolveNearTies(ms, batches, d.FUSE_HM, rel_gap=diff / 64.0)
    if hopeless["candidate_voxels"]:
        assert hopeless["gap_widenings"] == 3 and hopeless["premise_ok"] == 0, hopeless
    for o in ms + [out] + batches:
        o.close()


@pytest.mark.parametrize("n_cams,op", [(1, 0), (2, d.FUSE_HM), (2, d.FUSE_MIN), (2, d.FUSE_GM), (2, d.FUSE_AM),
                                       (2, d.FUSE_RMS), (2, d.FUSE_MAX)])
def test_the_resolvers_premise_is_proven_column_by_column(ctx, n_cams, op):
    """dsi_mapper_prove_near_ties (round 6): every voxel's votes are COUNTED, the reference's fp32 event-order sum is bounded
    from them ((n - 1) u / (1 - (n - 1) u) of the weights' real sum), and a column is proven when no plane below the
    resolver's threshold can reach the maximum's plane under those bounds.  (1) The counts are the resolver's own -- the
    inverted event pass reports the same number for any voxel (an independent kernel).  (2) With a gap of rounding size
    nothing is proven for the contested columns, and gap_needed says how far to widen; (3) resolving with that gap and
    proving again proves EVERY column, and the resolved map is the oracle's -- now not by comparison alone."""
    nx, ny, nz = 96, 72, 32
    rig = syn.stereo_rig(150_000, width=nx, height=ny, duration=0.3, seed=11, n_points=700)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    batches = _batches(ctx, rig, n_cams)
    ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(n_cams)]
    for m, b in zip(ms, batches):
        m.evaluateDSI_batch(b)
    out = d.MapperEMVS(ctx, rig["cam"], shape)
    if n_cams == 1:
        out.computeDepthMap(ms[0].dsi_)
    else:
        out.computeDepthMapOfFusion(ms[0].dsi_, ms[1].dsi_, op)
    tiny = out.proveNearTies(ms, batches, op, rel_gap=1e-7)
    assert tiny["columns"] == nx * ny and tiny["columns_proven"] + tiny["columns_unproven"] == nx * ny
    assert tiny["columns_unproven"] > 0 and tiny["gap_needed"] > 1e-7, tiny
    assert tiny["gap_needed"] < 1e-2, tiny  # rounding-sized: the bounds are not vacuous
    # (1) the counters against the resolver's event pass, on random voxels and on the heaviest column
    rng = np.random.default_rng(3)
    vox = rng.integers(0, nx * ny * nz, 4000).astype(np.uint32)
    for c in range(n_cams):
        got = out.proofVotes(c, vox)
        _, want = ms[c].exactVoxels(batches[c], vox)
        assert np.array_equal(got, want), "camera %d: %d of %d vote counts differ" % (c, (got != want).sum(), vox.size)
        assert got.max() > 10
    # (2) + (3)
    gap = float(tiny["gap_needed"]) * 1.05 + 1e-9
    info = out.resolveNearTies(ms, batches, op, rel_gap=gap)
    assert info["gap_widenings"] == 0
    proof = out.proveNearTies(ms, batches, op, rel_gap=gap)
    assert proof["columns_unproven"] == 0 and proof["columns_proven"] == nx * ny and proof["gap_needed"] == 0.0, proof
    ref, planes = _oracle_fused(rig, n_cams, op, (nx, ny, nz))
    _, ridx = orc.collapse_max_z(ref)
    _, _, idx = out.fetchDepthMap()
    assert np.array_equal(idx, ridx)
    # a much wider gap stays proven (more planes re-summed, fewer to bound)
    wide = out.proveNearTies(ms, batches, op, rel_gap=min(0.4, 50 * gap))
    assert wide["columns_unproven"] == 0
    for o in ms + [out] + batches:
        o.close()


@pytest.mark.parametrize("events", [12_000, 400_000])
def test_the_proofs_interval_holds_the_reference_order_value(ctx, events):
    """The bound k_tie_prove relies on, checked voxel by voxel against values it never sees: for 30,000 random voxels the
    engine's exact value E (the DSI), the counted votes n (dsi_mapper_proof_votes) and the value R the reference's summation
    order gives (dsi_mapper_exact_voxels: fp32, event order, an independent kernel).  R must lie in
    [max(0, E(1 - 2u) - n 2^-31) (1 - g), (E(1 + 2u) + n 2^-31) (1 + g)], g = (n - 1) u / (1 - (n - 1) u), u = 2^-24 -- the
    interval of tie_reference_interval, restated here in numpy -- and must be positive wherever E is; and the truncation of the
    weights to the 2^-31 grid is one-sided: the engine's sum never exceeds the reference-order value by more than rounding."""
    nx, ny, nz = 96, 72, 32
    rig = syn.stereo_rig(events, width=nx, height=ny, duration=0.3, seed=31, n_points=300)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    b = _batches(ctx, rig, 1)[0]
    m = d.MapperEMVS(ctx, rig["cam"], shape)
    m.evaluateDSI_batch(b)
    m.computeDepthMap(m.dsi_)
    m.proveNearTies([m], [b], 0)
    rng = np.random.default_rng(8)
    dsi = m.dsi_.download().reshape(-1)
    heavy = np.argsort(dsi)[-5000:].astype(np.uint32)                     # the most-voted voxels, where the bound matters
    vox = np.concatenate([rng.integers(0, dsi.size, 25000).astype(np.uint32), heavy])
    E = dsi[vox].astype(np.float64)
    n = m.proofVotes(0, vox).astype(np.float64)
    R, n2 = m.exactVoxels(b, vox)
    assert np.array_equal(n2.astype(np.float64), n)
    R = R.astype(np.float64)
    u, q = 2.0 ** -24, n * 2.0 ** -31
    g = np.where(n > 0, (n - 1) * u / (1 - (n - 1) * u), 0.0)
    hi = (E * (1 + 2 * u) + q) * (1 + g)
    lo = np.maximum(0.0, E * (1 - 2 * u) - q) * (1 - g)
    lo = np.where(E > 0, np.maximum(lo, 2.0 ** -33), lo)
    assert np.all(R <= hi) and np.all(R >= lo), "%d voxels outside their interval" % int(((R > hi) | (R < lo)).sum())
    assert np.all((n > 0) | ((E == 0) & (R == 0)))
    assert n.max() > (50 if events < 100_000 else 2000)
    # how much of the interval the real difference uses: far less than all of it (worst-case bound), but not nothing
    used = np.abs(R - E) / np.maximum(hi - lo, 1e-300)
    assert used.max() < 0.5 and used.max() > 1e-4
    m.close()
    b.close()


def test_interval_grids_enclose_the_reference_order_values(ctx):
    """The interval grids of the general proof (dsi_mapper_reference_interval, dsi_grid_widen_interval): per camera lo <= R <= hi
    for the reference-order value R of 20,000 voxels (dsi_mapper_exact_voxels); after the camera fusion applied to lo and to hi
    and widened by its roundings, lo_f <= op(R0, R1) <= hi_f with the reference's scalar op on the host; widening only widens;
    and dsi_grid_prove_columns on those grids agrees with dsi_mapper_prove_near_ties, which computes the same bounds in double."""
    from dvs_mcemvs_amd import engine
    nx, ny, nz = 96, 72, 32
    rig = syn.stereo_rig(300_000, width=nx, height=ny, duration=0.3, seed=23, n_points=400)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    batches = _batches(ctx, rig, 2)
    ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(2)]
    for m, b in zip(ms, batches):
        m.evaluateDSI_batch(b)
    out = d.MapperEMVS(ctx, rig["cam"], shape)
    g = {k: d.Grid3D(ctx, nx, ny, nz) for k in ("lo0", "hi0", "lo1", "hi1", "fused")}
    rng = np.random.default_rng(1)
    heavy = np.argsort(ms[0].dsi_.download().reshape(-1))[-4000:].astype(np.uint32)
    vox = np.concatenate([rng.integers(0, nx * ny * nz, 16000).astype(np.uint32), heavy])
    R = []
    for c in range(2):
        out.referenceInterval(ms[c], batches[c], g["lo%d" % c], g["hi%d" % c])
        r = ms[c].exactVoxels(batches[c], vox)[0]
        lo, hi = (g[k % c].download().reshape(-1)[vox] for k in ("lo%d", "hi%d"))
        assert np.all(lo <= r) and np.all(r <= hi), "camera %d: %d voxels outside" % (c, int(((r < lo) | (r > hi)).sum()))
        assert np.all(lo >= 0) and (hi - lo).max() > 0
        R.append(r)
    for op in (d.FUSE_HM, d.FUSE_GM, d.FUSE_MIN, d.FUSE_RMS):
        lo_f, hi_f = d.Grid3D(ctx, nx, ny, nz), d.Grid3D(ctx, nx, ny, nz)
        lo_f.setToFusionOf(g["lo0"], g["lo1"], op)
        hi_f.setToFusionOf(g["hi0"], g["hi1"], op)
        before = (lo_f.download().copy(), hi_f.download().copy())
        engine.widen_interval(lo_f, hi_f, 8)
        lo_w, hi_w = lo_f.download(), hi_f.download()
        assert np.all(lo_w <= before[0]) and np.all(hi_w >= before[1]) and np.all(lo_w >= 0)
        rf = engine.reference_fuse2(op, R[0], R[1])
        assert np.all(lo_w.reshape(-1)[vox] <= rf) and np.all(rf <= hi_w.reshape(-1)[vox]), "op %d" % op
        # the same columns proven as by the dedicated two-camera proof (same bounds, there in double precision throughout:
        # the fp32 grids can only be a little wider)
        g["fused"].setToFusionOf(ms[0].dsi_, ms[1].dsi_, op)
        general = out.proveColumns(g["fused"], lo_f, hi_f, rel_gap=1e-5)
        direct = out.proveNearTies(ms, batches, op, rel_gap=1e-5)
        assert general["columns"] == direct["columns"] == nx * ny
        assert direct["columns_proven"] >= general["columns_proven"] >= direct["columns_proven"] - 0.02 * nx * ny, (general, direct)
        assert general["max_votes"] == direct["max_votes"] or general["max_votes"] == 0
        for o in (lo_f, hi_f):
            o.close()
    for o in ms + [out] + batches + list(g.values()):
        o.close()


@pytest.mark.parametrize("events,op", [(12_000, d.FUSE_HM), (12_000, d.FUSE_GM), (150_000, d.FUSE_HM), (150_000, d.FUSE_MIN)])
def test_proven_mode_settles_every_column(ctx, events, op):
    """process.resolve_near_ties_proven: resolve, prove; a moderately wider gap for the columns whose bounds ask for one; and the
    columns no gap of reasonable size settles -- with few events many columns' maxima are a handful of tiny weights, where the
    engine's 2^-31 weight grid is as coarse as the values -- re-summed on ALL their planes.  Afterwards every column is either
    proven by bounds or exact by construction, and the map is the oracle's."""
    nx, ny, nz = 96, 72, 32
    rig = syn.stereo_rig(events, width=nx, height=ny, duration=0.3, seed=5 + events % 7, n_points=700)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    batches = _batches(ctx, rig, 2)
    ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(2)]
    for m, b in zip(ms, batches):
        m.evaluateDSI_batch(b)
    out = d.MapperEMVS(ctx, rig["cam"], shape)
    out.computeDepthMapOfFusion(ms[0].dsi_, ms[1].dsi_, op)
    info, proof = proc.resolve_near_ties_proven(out, ms, batches, op)
    assert proof["columns"] == nx * ny and proof["columns_proven"] + proof["columns_unproven"] == nx * ny
    assert proof["columns_unproven"] == proof["columns_resolved_fully"], (info, proof)
    ref, planes = _oracle_fused(rig, 2, op, (nx, ny, nz))
    rconf, ridx = orc.collapse_max_z(ref)
    depth, conf, idx = out.fetchDepthMap()
    assert np.array_equal(idx, ridx), "%d pixels differ; %r" % ((idx != ridx).sum(), proof)
    assert np.array_equal(depth, planes[ridx])
    if proof["columns_resolved_fully"]:
        # a fully re-summed column carries the oracle's own confidence, bit for bit
        pix, _ = out.proofUnproven()
        assert np.array_equal(conf.reshape(-1)[pix], rconf.reshape(-1)[pix])
    for o in ms + [out] + batches:
        o.close()


def test_in_order_fetch_gives_the_same_depth_map(ctx):
    """dsi_mapper_fetch_depth_map_in_order (the maps stored by a kernel into mapped page-locked memory on the compute stream,
    or copied there when the destination is pageable) = dsi_mapper_fetch_depth_map."""
    nx, ny, nz = 96, 72, 32
    rig = syn.stereo_rig(40_000, width=nx, height=ny, duration=0.2, seed=4, n_points=500)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    b = _batches(ctx, rig, 1)[0]
    m = d.MapperEMVS(ctx, rig["cam"], shape)
    m.evaluateDSI_batch(b)
    m.computeDepthMap()
    want = m.fetchDepthMap()
    got = m.fetchDepthMap(in_order=True)                       # pageable destinations
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    pins = [d.PinnedArray((ny, nx), t) for t in (np.float32, np.float32, np.uint8)]
    for p in pins:
        p.a[...] = 0
    m.fetchDepthMapInOrder(*[p.a for p in pins])               # mapped page-locked destinations: k_store_depth_map
    m.fetchWait()
    for p, w in zip(pins, want):
        assert np.array_equal(p.a, w)
    for o in pins + [m, b]:
        o.close()


def test_resolver_argument_checks(ctx):
    rig = syn.stereo_rig(5_000, width=64, height=48, duration=0.1, seed=1)
    shape = d.ShapeDSI(0, 0, 8, 4.0, 100.0, 0.0)
    batches = _batches(ctx, rig, 2)
    ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(2)]
    out = d.MapperEMVS(ctx, rig["cam"], shape)
    with pytest.raises(d.DsiError):                      # no depth map to resolve yet
        out.resolveNearTies(ms, batches, d.FUSE_HM)
    for m, b in zip(ms, batches):
        m.evaluateDSI_batch(b)
    out.computeDepthMapOfFusion(ms[0].dsi_, ms[1].dsi_, d.FUSE_HM)
    with pytest.raises(d.DsiError):
        out.resolveNearTies(ms, batches, 9)
    with pytest.raises(d.DsiError):
        out.resolveNearTies(ms, batches, d.FUSE_HM, rel_gap=0.9)
    with pytest.raises(d.DsiError):
        out.resolveNearTies(ms + ms[:1], batches + batches[:1], d.FUSE_HM)
    other = d.MapperEMVS(ctx, rig["cam"], d.ShapeDSI(0, 0, 9, 4.0, 100.0, 0.0))
    with pytest.raises(d.DsiError):
        out.resolveNearTies([ms[0], other], batches, d.FUSE_HM)
    # the proof: same argument rules; no list of unproven columns before a proof has run; and a DSI that is not the exact
    # sums its bounds describe (fp32 global atomics, the paired 32-bit cells) is refused, not "proven"
    with pytest.raises(d.DsiError):
        out.proofUnproven()
    with pytest.raises(d.DsiError):
        out.proveNearTies(ms, batches, 9)
    with pytest.raises(d.DsiError):
        out.proveNearTies([ms[0], other], batches, d.FUSE_HM)
    assert out.proveNearTies(ms, batches, d.FUSE_HM)["columns"] == 64 * 48
    with pytest.raises(d.DsiError):
        out.proofVotes(0, np.array([64 * 48 * 8], np.uint32))      # outside the grid
    for knob in ("paired", "global"):
        if knob == "paired":
            ms[1].set_packed_lanes(8)
        else:
            ms[1].set_packed_lanes(-1)
            ms[1].set_vote_algo(d.VOTE_GLOBAL_ATOMIC)
        ms[1].evaluateDSI_batch(batches[1])
        with pytest.raises(d.DsiError):
            out.proveNearTies(ms, batches, d.FUSE_HM)
    for o in ms + [out, other] + batches:
        o.close()


def test_window_stream_with_exact_ties_equals_the_oracle_on_every_pixel(ctx):
    """BASELINE configs[2] shape (512x512x200, 2 x 500 k events per 50 ms window, camera HM): with
    WindowStream(exact_ties=True) the plane index map of a window equals the oracle's arg-max of ITS fused volume
    on every one of the 262,144 pixels."""
    NX, NY, NZ, EV, DUR = 512, 512, 200, 500_000, 0.05
    t0 = 10.0
    rig = syn.stereo_rig(2 * EV, width=640, height=480, t0=t0, duration=2 * DUR, seed=77, n_points=6000)
    shape = d.ShapeDSI(NX, NY, NZ, 4.0, 200.0, 0.0)
    ws = proc.WindowStream(ctx, (rig["cam"],) * 2, shape, d.FUSE_HM, materialize_fused=False, exact_ties=True)
    lo, hi = proc.window_bounds(t0, t0 + 2 * DUR + 1e-9, DUR, DUR)[1]
    ev = [proc.window_events(rig["events"][c], lo, hi) for c in range(2)]
    depth, conf, idx = ws.fetch(ws.submit(ev, rig["trajectories"], hi))
    info = ws.last_resolve
    T_rv_w = proc.reference_view_process1(rig["trajectories"][0], hi)
    dsis = []
    for c in range(2):
        r = OracleMapper(rig["cam"], dimX=NX, dimY=NY, dimZ=NZ, min_depth=4.0, max_depth=200.0)
        assert r.evaluateDSI(ev[c], rig["trajectories"][c], T_rv_w)
        dsis.append(r.dsi)
    ref = orc.fuse2(dsis[0].copy(), dsis[1], 2)
    rconf, ridx = orc.collapse_max_z(ref)
    print("configs[2] window, exact ties: %r" % (info,))
    assert np.array_equal(idx, ridx), "%d pixels differ; %r" % ((idx != ridx).sum(), info)
    assert np.array_equal(depth, r.planes[ridx])
    assert np.allclose(conf, rconf, rtol=1e-4, atol=1e-6)
    ws.close()
