"""Post-arg-max part of MapperEMVS::getDepthMapFromDSI (mapper_emvs_stereo.cpp:390-436):
confidence normalisation, Gaussian adaptive threshold, masked Huang median, border removal,
index -> depth.  CPU: known answers for the oracle's restatement; GPU: HIP == oracle, bit exact
(everything is integer or exactly representable float work)."""
import numpy as np
import pytest

from oracle import oracle as orc


def test_uniform_confidence_gives_empty_mask():
    conf = np.full((20, 30), 3.0, np.float32)
    idx = np.full((20, 30), 7, np.uint8)
    planes = orc.depth_planes(1.0, 5.0, 16)
    r = orc.depth_map_filters(conf, idx, planes, 5, 5.0, 5, max_confidence=0.0)
    # (0,0) <- max_confidence = 0 is the minimum, everything else maps to 255; (0,0) -> 0
    assert r["confidence"][0, 0] == 0.0
    assert r["conf8"][0, 0] == 0 and (r["conf8"].reshape(-1)[1:] == 255).all()
    # nothing exceeds its neighbourhood mean by C -- except next to the zeroed pixel (0,0), whose
    # neighbours see a lowered mean (a quirk of :393-396), and those lie in the removed border
    assert not r["mask"].any()
    assert not r["idx_filtered"][6:, :].any() and not r["idx_filtered"][:, 6:].any()
    assert r["idx_filtered"][1, 1] == 7             # the median ran on the mask BEFORE border removal
    assert (r["depth"][6:, 6:] == planes[0]).all()  # empty windows -> median 0 (median_filtering.cpp:7-18)


def test_isolated_peak_is_selected_and_border_removed():
    conf = np.zeros((21, 21), np.float32)
    conf[10, 10] = 100.0
    conf[1, 10] = 100.0                              # inside the removed boundary (border = 2)
    idx = np.full((21, 21), 3, np.uint8)
    idx[10, 10] = 9
    planes = orc.depth_planes(1.0, 5.0, 16)
    r = orc.depth_map_filters(conf, idx, planes, 5, 4.0, 5, max_confidence=100.0)
    assert r["conf8"][10, 10] == 255 and r["conf8"][0, 0] == 0
    # mean at the peak = 255 * 0.375^2 = 35.86 -> 36; 255 - 36 > 4
    assert r["mask"][10, 10] == 1 and r["mask"].sum() == 1
    assert r["mask"][1, 10] == 0                     # x<=2 | y<=2 | ... cleared (:316-329)
    # the median used the mask BEFORE the border removal: windows containing (1,10) or (10,10)
    assert r["idx_filtered"][10, 10] == 9 and r["idx_filtered"][12, 12] == 9
    assert r["idx_filtered"][1, 10] == 3 and r["idx_filtered"][20, 20] == 0
    assert r["depth"][10, 10] == planes[9]


def test_median_definition_even_count():
    """compute_median_histogram: smallest v with cumulative count >= (num+1)/2 (lower median)."""
    conf = np.zeros((9, 9), np.float32)
    idx = np.zeros((9, 9), np.uint8)
    for (y, x, c, v) in [(4, 3, 50, 10), (4, 4, 60, 20), (4, 5, 70, 30), (3, 4, 80, 40)]:
        conf[y, x], idx[y, x] = c, v
    r = orc.depth_map_filters(conf, idx, orc.depth_planes(1, 5, 64), 3, 1.0, 3, max_confidence=80.0)
    assert r["mask"][4, 3] == r["mask"][4, 4] == r["mask"][4, 5] == r["mask"][3, 4] == 1
    assert r["idx_filtered"][4, 4] == 20             # values {10,20,30,40}: middle = 2 -> 20


def test_max_confidence_caps_the_range():
    rng = np.random.default_rng(2)
    conf = rng.gamma(1.0, 3.0, (30, 40)).astype(np.float32)
    idx = rng.integers(0, 50, (30, 40)).astype(np.uint8)
    lo = orc.depth_map_filters(conf, idx, orc.depth_planes(1, 5, 64), 5, 4.0, 5, max_confidence=0.0)
    hi = orc.depth_map_filters(conf, idx, orc.depth_planes(1, 5, 64), 5, 4.0, 5, max_confidence=10 * conf.max())
    assert hi["conf8"].max() <= 26                   # 255 / 10
    assert hi["mask"].sum() < lo["mask"].sum()       # "pixels with few votes" no longer look confident


@pytest.mark.gpu
@pytest.mark.parametrize("ksize,C_,med", [(5, 4.0, 5), (3, 2.0, 3), (7, 5.0, 9), (5, 4.5, 1), (9, 3.0, 5)])
def test_hip_filters_match_oracle(ctx, ksize, C_, med):
    import dvs_mcemvs_amd as d
    from dvs_mcemvs_amd import synthetic as syn
    rig = syn.stereo_rig(60000, width=120, height=90, duration=0.3, seed=5)
    m = d.MapperEMVS(ctx, rig["cam"], d.ShapeDSI(0, 0, 40, 4.0, 200.0, 0.0))
    assert m.evaluateDSI(rig["events"][0], rig["trajectories"][0], rig["T_rv_w"])
    _, conf, idx = m.getDepthMapFromDSI()
    for max_conf in (0.0, 60.0):
        ref = orc.depth_map_filters(conf, idx, m.raw_depths_vec_, ksize, C_, med, max_conf)
        depth, conf2, mask = m.getDepthMapFromDSI(options_depth_map=d.OptionsDepthMap(ksize, C_, med, max_conf))
        assert np.array_equal(conf2, ref["confidence"])
        assert np.array_equal(mask, ref["mask"]), "mask differs at %d pixels" % (mask != ref["mask"]).sum()
        assert np.array_equal(m.depth_cell_indices_filtered, ref["idx_filtered"])
        assert np.array_equal(depth, ref["depth"])
        if ksize <= 7:
            assert mask.sum() > 0
    with pytest.raises(d.DsiError):
        m.getDepthMapFromDSI(options_depth_map=d.OptionsDepthMap(4, 4.0, 5, 0.0))   # even kernel
    m.close()
