"""GPU test of distributed.PipelinedTemporalFusion (two HIP streams, double-buffered accumulator,
torch events) at world size 1: five rounds with different DSIs must each give the reference's
temporal harmonic fusion of that round (process2.cpp:217-226) and its arg-max.

Runs in a subprocess that imports torch BEFORE the engine, like bench.py does on the distributed
path: libdsi_engine.so then binds to the HIP runtime torch loaded, so torch streams / events and
the engine's streams belong to one runtime."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import torch
import dvs_mcemvs_amd as d
from dvs_mcemvs_amd import distributed as dd
from oracle import oracle as orc

nx, ny, nz = 96, 64, 20
ctx, side = d.Context(0), d.Context(0)
cam = (nx, ny, 70.0, 70.0, 48.0, 32.0)
mf = d.MapperEMVS(side, cam, d.ShapeDSI(0, 0, nz, 1.0, 5.0, 0.0))
rounds = 12
snaps = [torch.empty((nz, ny, nx), dtype=torch.float32, device="cuda") for _ in range(rounds)]
state = {"k": 0}
def extract(g):
    # still on the side stream, right after the round's finalize: arg-max, then keep a device copy
    mf.computeDepthMap(g)
    slot = [s for s in pipe.slots if s["acc_side"] is g][0]
    with torch.cuda.stream(pipe.streams[1]):
        snaps[state["k"]].copy_(slot["tensor"], non_blocking=True)
    state["k"] += 1
pipe = dd.PipelinedTemporalFusion.on_gpu(ctx, side, (nx, ny, nz), d.ACC_INV_SUM, 1, extract=extract)
rng = np.random.default_rng(0)
vols = [rng.uniform(0, 4, (nz, ny, nx)).astype(np.float32) for _ in range(rounds)]
# stand-ins for "vote + camera fusion of round k": uploads queued on the main stream, one grid per
# round so that the host never has to wait inside the loop
fused = [d.Grid3D(ctx, nx, ny, nz) for _ in range(rounds)]
for f, v in zip(fused, vols):
    f.upload(v)
for f in fused:
    pipe.submit(f)                  # no host synchronisation in here
pipe.drain()
torch.cuda.synchronize()
for k in range(rounds):
    ref = orc.finalize(orc.accumulate(np.zeros_like(vols[k]), vols[k], 1), 1, 1)
    got = snaps[k].cpu().numpy()
    assert np.array_equal(got, ref), "round %%d differs: %%g" %% (k, np.abs(got - ref).max())
depth, conf, idx = mf.fetchDepthMap()   # of the last round
rconf, ridx = orc.collapse_max_z(ref)
assert np.array_equal(conf, rconf) and np.array_equal(idx, ridx)
print("PIPELINED_OK")
'''


def test_pipelined_temporal_fusion_on_two_streams(built):
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "PIPELINED_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
