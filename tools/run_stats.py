"""How tight are the runs the banded kernel walks?  total run length vs accepted votes."""
import ctypes as C, sys, os
os.environ["DSI_ENGINE_EXPERIMENTS"] = "1"   # the hooks used below exist only in the experiments flavour (build.py --experiments)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dvs_mcemvs_amd as d
from dvs_mcemvs_amd import synthetic as syn
L = d.load_library()
L.dsi_test_run_length_total.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
ctx = d.Context(0)
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for (nx, ny, nz, ev) in [(346, 260, 100, 1000000), (512, 512, 200, 500000), (1024, 1024, 256, 300000)]:
    rig = syn.stereo_rig(ev, width=nx, height=ny, seed=1234)
    m = d.MapperEMVS(ctx, rig["cam"], d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0))
    m.set_packed_lanes(MODE)
    m.evaluateDSI(rig["events"][0], rig["trajectories"][0], rig["T_rv_w"])
    t, n = C.c_ulonglong(), C.c_ulonglong()
    assert L.dsi_test_run_length_total(m._h, C.byref(t), C.byref(n)) == 0
    info = m.last_vote_info()
    v = m.dsi_.download()
    accepted = float(v.sum(dtype=np.float64))
    print("%dx%dx%d: voted events*planes %.4g  run total %.4g  accepted votes %.4g  bands %d rows %d  mapping %d (S=%d)  run/accepted %.3f  avg run %.1f"
          % (nx, ny, nz, m.n_voted * nz, t.value, accepted, info["bands"], info["band_rows"], info["packed"], info["group_packets"], t.value / accepted, t.value / n.value))
    m.close()
