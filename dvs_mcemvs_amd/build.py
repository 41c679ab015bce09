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
# the EXPERIMENTS flavour (-DDSI_TIMING_EXPERIMENTS: environment knobs + dsi_test_* hooks, some of which corrupt the
# DSIs on purpose) is a DIFFERENT file, loaded only by an explicit opt-in (engine.load_library: DSI_ENGINE_EXPERIMENTS=1)
OUT_EXPERIMENTS = os.path.join(HERE, "libdsi_engine_experiments.so")
SOURCES = ["dsi_kernels.hip", "dsi_engine.cpp"]
HEADERS = ["dsi_kernels.h", "dsi_vote_asm.h", "dsi_host.hpp", os.path.join("..", "..", "include", "dsi_engine.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fvisibility=hidden", "-Wall", "-Wextra", "-Wno-unused-parameter", "-x", "hip"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the gfx950 DSI engine")


def needs_build(out=OUT):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, experiments=False):
    """Builds the production library; experiments=True (`--experiments`) builds the experiments flavour INSTEAD, into
    its own file (libdsi_engine_experiments.so): the environment knobs of the timing experiments quoted in NOTEBOOK.md
    (DSI_EXPERIMENT, DSI_PERSISTENT, DSI_PASS_LG, DSI_GROUP_PACKETS, DSI_PREP_OVERLAP) and the dsi_test_* hooks exist
    only there.  The production library reads no environment and exports no test hook."""
    out = OUT_EXPERIMENTS if experiments else OUT
    if not force and not needs_build(out):
        return out
    cmd = ([hipcc()] + FLAGS + (["-DDSI_TIMING_EXPERIMENTS"] if experiments else []) +
           [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out + ".tmp"])
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, experiments="--experiments" in sys.argv))
