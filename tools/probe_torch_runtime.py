"""Does libdsi_engine.so (hipcc 7.2 code objects) work when it binds to the HIP runtime that
torch (ROCm 7.0 wheels) loaded first?  Needed for the N>1 path (torch.distributed nccl)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
print("torch", torch.__version__, torch.version.hip, torch.cuda.get_device_name(0))
import numpy as np
import dvs_mcemvs_amd as d
ctx = d.Context(0)
t = torch.zeros((4, 6, 8), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
g = d.Grid3D(ctx, 8, 6, 4, device_ptr=t.data_ptr())
h = d.Grid3D(ctx, 8, 6, 4)
h.upload(np.full((4, 6, 8), 2.5, np.float32))
g.addTwoGrids(h)
ctx.synchronize()
print("tensor sum after engine add:", float(t.sum().item()), "expected", 2.5 * 192)
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "libamdhip64" in l or "libdsi_engine" in l})
print("\n".join(libs))
