"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/dsi_engine.h declares, refuses to run without a GPU (no CPU fallback), and its
host-side logic (packetisation + pose pipeline, mapper_emvs_stereo.cpp:67-105,
trajectory.hpp:92-126) agrees with the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

import dvs_mcemvs_amd as d
from dvs_mcemvs_amd import engine, synthetic as syn
from oracle import oracle as orc
from oracle_pipeline import OracleMapper

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "dsi_engine.h")).read()
    return sorted(set(re.findall(r"DSI_API[^;(]*?\b(dsi_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built):
    syms = header_symbols()
    assert len(syms) >= 40
    lib = ctypes.CDLL(d.library_path())
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, "not exported: %s" % missing
    assert lib.dsi_abi_version() == 5
    # the Python binding declares a signature for every exported entry point
    L = d.load_library()
    unbound = [s for s in syms if getattr(L, s).argtypes is None]
    assert not unbound, "no ctypes signature: %s" % unbound


def test_no_cpu_fallback(built):
    """Without a gfx950 device the constructors fail loudly (DSI_ERR_NO_DEVICE)."""
    if d.device_count() > 0:
        pytest.skip("a GPU is visible on this box")
    with pytest.raises(d.DsiError) as e:
        d.Context(0)
    assert e.value.code == engine.ERR_NO_DEVICE
    assert "no HIP device" in str(e.value) or "gfx950" in str(e.value)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under dvs_mcemvs_amd/ may import, link or
    load it (a product path through the oracle would void every parity claim)."""
    pkg = os.path.join(ROOT, "dvs_mcemvs_amd")
    banned = ("import oracle", "from oracle", "dsi_oracle", "libdsi_oracle", "oracle_pipeline", "orc_")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                hits = [b for b in banned if b in txt]
                assert not hits, "%s references the oracle: %s" % (os.path.join(dirpath, f), hits)
    # and the shared library does not link it
    import subprocess
    out = subprocess.run(["ldd", d.library_path()], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_host_pose_at_matches_oracle(built):
    rig = syn.stereo_rig(2048, width=32, height=24, duration=0.3, seed=5)
    times, poses = rig["trajectories"][1]
    for t in np.linspace(times[0] - 0.05, times[-1] + 0.05, 41):
        a = d.pose_at((times, poses), t)
        b = orc.pose_at(times, poses, t)
        assert (a is None) == (b is None)
        if a is not None:
            assert np.allclose(a, b, rtol=0, atol=1e-13)
    # exact hits on control poses: upper_bound semantics (trajectory.hpp:98)
    assert d.pose_at((times, poses), times[0]) is not None
    assert d.pose_at((times, poses), times[-1]) is None


def test_host_packetize_matches_oracle(built):
    rig = syn.stereo_rig(7000, width=48, height=36, duration=0.3, seed=6)
    x, y, ts = rig["events"][0]
    times, poses = rig["trajectories"][0]
    r = OracleMapper(rig["cam"], dimZ=4, min_depth=1.0, max_depth=5.0)
    for n in (1023, 1024, 1025, 2048, 2049, 7000):
        got = d.packetize(ts[:n], (times, poses), rig["T_rv_w"])
        ref = r.packetize(ts[:n], (times, poses), rig["T_rv_w"])
        assert (got is None) == (ref is None)
        if got is not None:
            assert np.array_equal(got[0], ref[0])
            assert np.allclose(got[1], ref[1], rtol=0, atol=1e-7)
    # pose lookups that fail slide the packet start one event at a time (:95-99)
    keep = times > ts[800]
    got = d.packetize(ts, (times[keep], poses[keep]), rig["T_rv_w"])
    ref = r.packetize(ts, (times[keep], poses[keep]), rig["T_rv_w"])
    assert got[0][0] % 1024 != 0
    assert np.array_equal(got[0], ref[0]) and np.allclose(got[1], ref[1], atol=1e-7)
    # no pose at all: true with zero packets (the reference returns true, votes nothing)
    got = d.packetize(ts, (times + 100.0, poses), rig["T_rv_w"])
    assert got is not None and got[0].shape[0] == 0


def test_argument_validation_without_gpu(built):
    L = d.load_library()
    assert L.dsi_context_synchronize(None) == engine.ERR_INVALID
    assert L.dsi_grid_reset(None) == engine.ERR_INVALID
    assert L.dsi_mapper_set_vote_algo(None, 1) == engine.ERR_INVALID
    assert b"null" in L.dsi_last_error()
    out = np.zeros(7)
    t = np.array([0.0, 1.0])
    p = np.array([[0, 0, 0, 1, 0, 0, 0], [1, 0, 0, 1, 0, 0, 0]], float)
    rc = L.dsi_pose_at(t.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                       p.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 2, 5.0,
                       out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    assert rc == engine.ERR_INVALID and b"extrapolate" in L.dsi_last_error()
