"""Builds dvs_mcemvs_amd/libdsi_engine.so (hand-written HIP, gfx950 only) in-tree.

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the reference
CPU path has no FMA, and the coordinate arithmetic must round where it rounds.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libdsi_engine.so")
SOURCES = ["dsi_kernels.hip", "dsi_engine.cpp"]
HEADERS = ["dsi_kernels.h", "dsi_host.hpp", os.path.join("..", "..", "include", "dsi_engine.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fvisibility=hidden", "-Wall", "-Wextra", "-Wno-unused-parameter", "-x", "hip"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the gfx950 DSI engine")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, experiments=False):
    """experiments=True (`--experiments`): also compile the environment knobs of the timing experiments quoted in
    DESIGN.md (DSI_EXPERIMENT, DSI_PERSISTENT, DSI_PASS_LG, DSI_GROUP_PACKETS, DSI_PREP_OVERLAP).  The production
    library does not read the environment."""
    if not force and not experiments and not needs_build():
        return OUT
    cmd = ([hipcc()] + FLAGS + (["-DDSI_TIMING_EXPERIMENTS"] if experiments else []) +
           [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT + ".tmp"])
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, experiments="--experiments" in sys.argv))
