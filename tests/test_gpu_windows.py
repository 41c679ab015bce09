"""BASELINE configs[2] as a workload: "Stereo DSEC full sequence streamed in 50 ms event windows,
512x512x200 DSI" -- the --full_seq loop of the reference (main.cpp:174-302: a window of `duration`
seconds every `out_skip` seconds, process_1 on each: reset + vote x2, camera HM, arg-max) through
the engine's streaming path (pooled device blocks for the events, uploads on the copy stream, depth
map of window w fetched while window w+1 is queued).

Sensor 640x480 with the DSEC intrinsics (calib.cpp:463-470), DSI 512x512x200 (--dimX / --dimY),
10 Mev/s per camera => 500 k events per 50 ms window, 24 windows.  Three windows are checked voxel
for voxel against the CPU oracle (both camera DSIs, the fused DSI, the depth map); every window is
checked for the properties that do not need the oracle."""
import numpy as np
import pytest

import dvs_mcemvs_amd as d
from dvs_mcemvs_amd import process as proc, synthetic as syn
from oracle import oracle as orc
from oracle_pipeline import OracleMapper

pytestmark = pytest.mark.gpu

NX, NY, NZ = 512, 512, 200
N_WIN, EV_WIN, DUR = 24, 500_000, 0.05


def test_stream_of_50ms_windows_configs2(ctx):
    t0 = 10.0
    rig = syn.stereo_rig(N_WIN * EV_WIN, width=640, height=480, t0=t0, duration=N_WIN * DUR, seed=77,
                         n_points=6000)
    cam = rig["cam"]
    shape = d.ShapeDSI(NX, NY, NZ, 4.0, 200.0, 0.0)       # fov < 10: the camera's focal length (:220-224)
    bounds = proc.window_bounds(t0, t0 + N_WIN * DUR + 1e-9, DUR, DUR)
    assert len(bounds) >= 20
    ws = proc.WindowStream(ctx, (cam, cam), shape, d.FUSE_HM)
    assert ws.mappers[0].dsi_.getDimensions() == (NX, NY, NZ)
    planes = ws.mappers[0].raw_depths_vec_
    checked = {0, len(bounds) // 2, len(bounds) - 1}
    results, pending = {}, None
    voted_total = 0

    def collect(w, slot):
        depth, conf, idx = ws.fetch(slot)
        results[w] = (depth, conf, idx)
        # properties every window must have
        assert np.all(conf >= 0) and np.isfinite(conf).all()
        assert np.array_equal(depth, planes[idx])                       # mapper_emvs_stereo.cpp:302-313
        assert conf.max() > 1.0                                         # the scene produced a peak

    for w, (a, b) in enumerate(bounds):
        ev = [proc.window_events(rig["events"][c], a, b) for c in range(2)]
        assert all(250_000 < e[0].shape[0] < 1_000_000 for e in ev)   # ~500 k (the scene thins out as the rig advances)
        slot = ws.submit(ev, rig["trajectories"], b)                     # forward looking: ts = stop
        if w in checked:
            # voxel for voxel against the oracle (the next window would overwrite the camera DSIs)
            T_rv_w = proc.reference_view_process1(rig["trajectories"][0], b)
            got = [ws.mappers[c].dsi_.download() for c in range(2)]
            fused = ws.fused_grid(slot).download()
            refs = []
            for c in range(2):
                r = OracleMapper(cam, dimX=NX, dimY=NY, dimZ=NZ, min_depth=4.0, max_depth=200.0)
                assert r.evaluateDSI(ev[c], rig["trajectories"][c], T_rv_w)
                err = np.abs(got[c].astype(np.float64) - r.dsi) / np.maximum(1.0, np.abs(r.dsi))
                assert err.max() <= 1e-4, "window %d camera %d: %g" % (w, c, err.max())
                refs.append(r.dsi)
            assert np.array_equal(fused, orc.fuse2(got[0], got[1], 2))    # fusion is exact on the GPU's inputs
            rf = orc.fuse2(refs[0], refs[1], 2)
            assert (np.abs(fused.astype(np.float64) - rf) / np.maximum(1.0, np.abs(rf))).max() <= 1e-4
            rconf, ridx = orc.collapse_max_z(fused)
            depth, conf, idx = ws.fetch(slot)
            assert np.array_equal(idx, ridx) and np.array_equal(conf, rconf)
            info = ws.mappers[0].last_vote_info()
            assert info["algo"] == d.VOTE_LDS_BANDS
        if pending is not None:
            collect(*pending)
        pending = (w, slot)
    collect(*pending)
    voted_total = ws.voted
    assert voted_total >= 2 * len(bounds) * 240 * 1024
    # determinism through the pooled blocks: window 0 again, after 23 other windows went through the
    # same device blocks, gives the same bits
    a, b = bounds[0]
    ev = [proc.window_events(rig["events"][c], a, b) for c in range(2)]
    slot = ws.submit(ev, rig["trajectories"], b)
    again = ws.fetch(slot)
    for x, y in zip(again, results[0]):
        assert np.array_equal(x, y)
    # consecutive windows look at (almost) the same scene from (almost) the same place: their depth
    # maps agree on most confident pixels -- a sanity check that no window was fed another's poses
    d0, c0, _ = results[0]
    d1, c1, _ = results[1]
    strong = (c0 > np.percentile(c0, 90)) & (c1 > np.percentile(c1, 90))
    assert strong.sum() > 1000
    assert np.median(np.abs(d0[strong] - d1[strong]) / d0[strong]) < 0.4
    ws.close()


def test_full_sequence_generator_small(ctx):
    """full_sequence() = main.cpp:177 loop bounds + pipelined fetch, on a small grid; midpoint
    reference view (forward_looking = false, main.cpp:188)."""
    rig = syn.stereo_rig(60_000, width=96, height=72, t0=3.0, duration=0.6, seed=5)
    shape = d.ShapeDSI(0, 0, 24, 4.0, 100.0, 0.0)
    out = list(proc.full_sequence(ctx, (rig["cam"],) * 2, shape, rig["events"], rig["trajectories"], 3.0, 3.6,
                                  0.2, 0.1, forward_looking=False))
    assert [round(o[0], 6) for o in out] == [3.1, 3.2, 3.3, 3.4, 3.5][:len(out)] and len(out) >= 4
    for ts, depth, conf, idx in out:
        a, b = ts - 0.1, ts + 0.1
        T_rv_w = proc.reference_view_process1(rig["trajectories"][0], ts)
        dsis = []
        for c in range(2):
            r = OracleMapper(rig["cam"], dimZ=24, min_depth=4.0, max_depth=100.0)
            assert r.evaluateDSI(proc.window_events(rig["events"][c], a, b), rig["trajectories"][c], T_rv_w)
            dsis.append(r.dsi)
        rf = orc.fuse2(dsis[0], dsis[1], 2)
        rconf, ridx = orc.collapse_max_z(rf)
        srt = np.sort(rf, axis=0)
        safe = (srt[-1] - srt[-2]) > 2e-4 * np.maximum(1.0, srt[-1])
        assert safe.mean() > 0.5 and np.array_equal(idx[safe], ridx[safe])
        assert np.allclose(conf, rconf, rtol=1e-4, atol=1e-4)
