#!/usr/bin/env python
"""bench.py -- DSI build + fuse + arg-max throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload stereo|windows|cameras4]

Workloads (a "step" = one pass of the hot path over one batch of synthetic input already
resident in HBM):

  stereo   (default; BASELINE.json configs[1], the configuration `metric` is quoted on)
           2 cameras x 10 M events, 346x260x100: per camera evaluateDSI past the pose lookup
           (per-packet homography, z0 warp, reset, voting; mapper_emvs_stereo.cpp:108-146), camera
           fusion (harmonic mean, process1.cpp:126-141), arg-max + depth
           (cartesian3dgrid.cpp:115-137, mapper_emvs_stereo.cpp:302-313).
           N > 1 (one process per GPU): every rank owns an independent time slice of the same size
           (weak scaling, configs[3]); the slices are fused across time with the reference's
           harmonic accumulator (process2.cpp:217-226): local 1/(0.01+v), ONE RCCL all-reduce(sum)
           of the volume over xGMI -- issued by the ENGINE (C ABI, dsi_grid_allreduce) on a second
           HIP stream so that it overlaps the next step's voting -- local n/acc, arg-max.

N > 1 is always N processes, one per GPU, whoever starts them: under a launcher (`python -m
torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: RANK / WORLD_SIZE in the environment) this
process is one rank; without one (`python bench.py --gpus N ...`) it starts the N ranks itself
(dvs_mcemvs_amd/launch.py) and relays rank 0's JSON line.  Fewer than N devices, a WORLD_SIZE that contradicts
--gpus, or an RCCL communicator that cannot be formed are ERRORS (non-zero exit): there is no fallback to fewer
ranks or to another collective.  The line proves its rank count: `rccl_ranks` is ncclCommCount of the engine's
communicator, `ranks` lists every rank's device as RCCL reports it.
  windows  configs[2]: stream of 50 ms windows, 2 cameras x 500 k events each, sensor 640x480, DSI
           512x512x200 (main.cpp:174-302 with process_method 1): a step = one window = reset +
           vote x2 + HM + arg-max; the depth map of window w is fetched while w+1 is queued.
           N > 1: independent windows round-robin over the ranks (replicas, no collective).
  cameras4 configs[4] shape: 4 cameras, 1024x1024x256, n-ary geometric-mean camera fusion
           (--events per camera, default 2 M).  N > 1: plane sharding -- every rank owns a plane
           range of every camera's DSI, the only exchange is ONE all-reduce(MAX) of packed arg-max
           keys (strong scaling: the total work is fixed).

value = events voted by all ranks / max-over-ranks time.  Prints ONE JSON line (rank 0).
Data: synthetic (dvs_mcemvs_amd/synthetic.py).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CU, LANES, CLK = 256, 64, 2.4e9
LDS_CLK_CONFLICT_FREE = 6.2     # clocks per ds_add_u64 wave instruction, conflict-free addresses
LDS_CLK_RANDOM = 11.2           # ... random cells of a band (profiles/r01_microbench_lds_atomic_bench2.txt)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["stereo", "windows", "cameras4"], default="stereo")
    ap.add_argument("--events", type=int, default=None, help="events per camera per GPU (per window for `windows`)")
    ap.add_argument("--dims", type=int, nargs=3, default=None)
    ap.add_argument("--algo", type=int, default=0, help="0 auto, 1 global atomics, 2 LDS bands")
    ap.add_argument("--band", type=int, nargs=3, default=[0, 0, 0], help="band_rows chunks block")
    ap.add_argument("--points", type=int, default=5000, help="scene points of the synthetic rig (SURVEY 8d: 2000-20000)")
    ap.add_argument("--packed", type=int, default=-1,
                    help="lane mapping: -1 auto, 0 per-packet waves, 1 packed (hand-scheduled), 2 packet groups, "
                         "3 packed (compiled), 4 groups (hand-scheduled), 5 packed + vector fill, 6 = 5 compiled, 7 = 1 with dealt passes")
    ap.add_argument("--pass-lg", type=int, default=0, help="tuning: log2 of the packets per pass of the voting streams (0 = automatic)")
    ap.add_argument("--inline-cuts", type=int, default=-1,
                    help="tuning (lane mappings 5 / 6): packets per call from which the voting kernel derives its runs itself "
                         "instead of reading a cut table (-1 = the engine's default 8192, 0 = always)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true",
                    help="skip the host-fed (PCIe-inclusive) measurements and the 512x512x200 stream-kernel timings "
                         "(profiling runs: only the timed steps launch kernels)")
    ap.add_argument("--materialize-fused", action="store_true",
                    help="windows / cameras4: also write the fused DSI (mapper_fused.dsi_) instead of fusing the "
                         "cameras inside the arg-max kernel")
    ap.add_argument("--fused-vote", dest="fused_vote", action="store_true", default=None,
                    help="windows: vote, fuse the cameras and keep the arg-max in ONE kernel, no DSI written "
                         "(dsi_mapper_depth_map_of_events); default for the windows workload")
    ap.add_argument("--no-fused-vote", dest="fused_vote", action="store_false")
    ap.add_argument("--window-depth", type=int, default=2,
                    help="windows: windows in flight (each on its own stream when not --serial-windows); 1, 2, 3 or 4")
    ap.add_argument("--serial-windows", action="store_true",
                    help="windows: all windows on ONE stream (no overlap of consecutive windows); for rocprofv3 runs, "
                         "whose per-kernel durations otherwise include the time a kernel waits for the previous "
                         "window's workgroups to leave the CUs")
    ap.add_argument("--temporal-collective", choices=["allreduce", "reduce_scatter"], default="allreduce",
                    help="stereo, N > 1: the time slices' temporal fusion as ONE all-reduce of the "
                         "accumulator + finalize + arg-max on every rank (default), or as a reduce-scatter by planes + "
                         "finalize / arg-max of the owned planes + all-reduce(MAX) of 8-byte keys (half the xGMI bytes; "
                         "the fused DSI is not completed on any rank)")
    ap.add_argument("--gm", choices=["tree", "log"], default="tree",
                    help="cameras4: n-ary geometric mean as the balanced tree of the reference's 2-ary sqrt(a*b) "
                         "(DSI_ACC_GM_TREE, default) or as exp(mean(log)) (DSI_ACC_LOG_SUM)")
    ap.add_argument("--no-extra", action="store_true",
                    help="default (stereo, 1 GPU) run only: do not append the one-GPU lines of the windows and "
                         "cameras4 workloads (each is a sub-run of this script) to the JSON line")
    ap.add_argument("--camera-streams", type=int, default=2,
                    help="cameras4: contexts (HIP streams) the cameras are dealt over, 1 to 4.  With 2, camera c+1's packet "
                         "sort / coefficient tables and the first workgroups of its voting kernel start on the CUs camera c's "
                         "persistent voting kernel has already left (its workgroups fill a CU's LDS, so nothing else runs "
                         "beside them)")
    ap.add_argument("--tile", type=int, default=1,
                    help="cameras4: the events of every camera are --events / --tile generated events, repeated --tile times "
                         "with the rig moved on between the repeats (different poses: different votes) -- how BASELINE "
                         "configs[4]'s 100 M events per camera are built in seconds of host time (--events 100000000 --tile 10), "
                         "like tests/test_gpu_parity.py::test_configs4_end_to_end_at_full_size does")
    ap.add_argument("--clock-ramp", type=int, default=None,
                    help="untimed steps run BEFORE the --warmup steps to bring the GPU's clocks up after the idle stretch in "
                         "which the host generated the inputs (default: ~0.25 s worth: 100 stereo steps, 600 windows, 40 "
                         "cameras4 steps; 0 = none).  The same count on every rank (steps contain collectives)")
    ap.add_argument("--no-sensitivity", action="store_true",
                    help="default (stereo, 1 GPU) run only: skip the input-sensitivity sub-records (the voting kernel on "
                         "uniformly random pixels, on a dense scene and on the recorded zurich_city_04 trajectory)")
    a = ap.parse_args()
    defaults = {"stereo": ((346, 260, 100), 10_000_000, 100, 5), "windows": ((512, 512, 200), 500_000, 100, 8),
                "cameras4": ((1024, 1024, 256), 2_000_000, 5, 1)}[a.workload]
    a.dims = a.dims or list(defaults[0])
    a.events = a.events or defaults[1]
    a.steps = a.steps if a.steps is not None else defaults[2]
    a.warmup = a.warmup if a.warmup is not None else defaults[3]
    if a.clock_ramp is None:
        a.clock_ramp = {"stereo": 100, "windows": 600, "cameras4": 40}[a.workload]
    return a


def make_comm(d, dd, ctx, D):
    """(allreduce callable, engine communicator or None, description).  N > 1: the engine's own RCCL communicator
    (dsi_comm_create_rank; the 128-byte unique id travels over the gloo side channel) or an error -- there is no
    other collective backend."""
    if D.world == 1:
        return dd.engine_allreduce(None), None, "none (1 rank)"
    uid, uid_err = None, ""
    if D.rank == 0:
        try:
            uid = d.Comm.unique_id()
        except d.DsiError as e:          # librccl missing / broken: every rank must learn it, or the others wait in the broadcast
            uid_err = str(e)
    uid, uid_err = D.broadcast((uid, uid_err))
    if uid is None:
        raise SystemExit("bench.py: rank 0 could not create the RCCL unique id (%s); there is no other collective backend" % uid_err)
    try:
        comm = d.Comm(ctx, uid, D.world, D.rank)
        ok, err = 1, ""
    except d.DsiError as e:
        comm, ok, err = None, 0, str(e)
    if D.min(ok) < 1.0:
        if comm is not None:
            comm.close()
        raise SystemExit("bench.py rank %d: the engine's RCCL communicator could not be formed on every rank (%s); "
                         "no fallback collective exists" % (D.rank, err or "see the other ranks"))
    return dd.engine_allreduce(comm), comm, "RCCL from the engine's C ABI (dsi_comm_create_rank, dsi_grid_allreduce)"


def kernel_source_sha16():
    import hashlib
    path = os.path.join(ROOT, "dvs_mcemvs_amd", "csrc", "dsi_kernels.hip")
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def profiled_counters(workload_key, kernel):
    """The PMC-derived figures of the dominant kernel -- HBM-side traffic per launch, LDS-busy, parked and bank-conflict
    shares -- from profiles/counters.json (tools/make_counters_json.py over the round's rocprofv3 --pmc summaries), quoted
    ONLY when that file was made from this very kernel source and names the kernel that ran here (a stale constant
    would look like a measurement).  None otherwise."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    if not os.path.exists(path):
        return None
    try:
        cj = json.load(open(path))
        if cj.get("kernel_source_sha16") != kernel_source_sha16():
            return None
        w = (cj.get("workloads") or {}).get(workload_key)
        if not w or not str(w.get("kernel", "")).startswith(kernel):
            return None
        return w
    except Exception:
        return None


def lds_block(accepted, kern_ms):
    adds = 4.0 * accepted / (kern_ms * 1e-3)
    lane_rate = CU * LANES * CLK
    return adds, lane_rate / LDS_CLK_CONFLICT_FREE, lane_rate / LDS_CLK_RANDOM


def roofline_block(info, kern_ms, kt_n, accepted, ev_per_launch, nz, traffic, records=None):
    """The voting kernel is an LDS-privatised scatter-add: what bounds it is the rate of 64-bit LDS
    atomic adds (4 per accepted event-plane).  peak = 256 CU x 64 lanes x 2.4 GHz / 6.2 clocks per
    ds_add_u64 wave instruction (conflict-free addresses; guide LDS section ~6, measured 6.1-6.2).
    The HBM view is reported beside it: measured fabric traffic / kernel time against 8 TB/s, and
    the SURVEY 8(d) "algorithmic bytes" figure (32*Nz+8 B per event as if every vote were an HBM
    read-modify-write), which is NOT a bound for this kernel (it exceeds the HBM peak)."""
    kernel = {0: "k_vote_bands", 1: "k_vote_bands_packed", 2: "k_vote_groups", 3: "k_vote_bands_packed",
              4: "k_vote_groups", 5: "k_vote_bands_vfill", 6: "k_vote_bands_vfill", 7: "k_vote_bands_packed",
              8: "k_vote_bands_packed"}[info["packed"]] \
        if info["algo"] == 2 else ("k_vote_fuse_argmax" if info["algo"] == 3 else "k_vote_global")
    if not kt_n:
        return {"bound": "lds_atomic", "achieved": None, "peak": None, "unit": "G adds/s", "frac": None,
                "traffic": traffic, "kernel": kernel}
    adds, peak, rnd = lds_block(accepted, kern_ms)
    alg_bytes = (32.0 * nz + 8.0) * ev_per_launch
    out = {"bound": "lds_atomic", "achieved": adds / 1e9, "peak": peak / 1e9,
           "unit": "G 64-bit LDS atomic adds/s", "frac": adds / peak, "traffic": traffic,
           "kernel": kernel, "kernel_avg_ms": kern_ms, "kernel_launches": kt_n,
           "accepted_event_planes_per_launch": accepted,
           "rate_random_cells": rnd / 1e9, "frac_of_random_cell_rate": adds / rnd,
           "kernel_Mevents_per_s": ev_per_launch / (kern_ms * 1e-3) / 1e6,
           "hbm_algorithmic_equiv": {"bytes_per_launch": alg_bytes,
                                     "GBps": alg_bytes / (kern_ms * 1e-3) / 1e9,
                                     "note": "SURVEY 8(d) byte model; not a bound (votes never leave LDS)"}}
    if records:
        # `achieved` counts ALGORITHMIC adds (4 per accepted event-plane, i.e. with the multiplicity of merged
        # duplicates); the hardware issues 4 per accepted RECORD.  Both fractions of the same roof:
        issued, _, _ = lds_block(records, kern_ms)
        out["accepted_records_per_launch"] = records
        out["records_per_accepted_event_plane"] = records / accepted if accepted else None   # duplicate-merge ratio
        out["achieved_issued"] = issued / 1e9
        out["frac_issued"] = issued / peak
    if traffic:
        gbps = traffic / (kern_ms * 1e-3) / 1e9
        out["hbm"] = {"achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                      "traffic_over_algorithmic": traffic / alg_bytes,
                      "source": "profiles/traffic.json (2*FETCH_SIZE + WRITE_SIZE PMC passes of this kernel)"}
    return out


def stream_kernels(d, ctx):
    """DSI-fuse and arg-max as HBM-bandwidth kernels: on a 512x512x200 pair (210 MB per volume, above
    the 256 MiB Infinity Cache for the 3 volumes of a fuse) whatever the workload's grid is."""
    nx, ny, nz = 512, 512, 200
    a, b = d.Grid3D(ctx, nx, ny, nz), d.Grid3D(ctx, nx, ny, nz)
    cam = (nx, ny, 400.0, 400.0, 256.0, 256.0)
    m = d.MapperEMVS(ctx, cam, d.ShapeDSI(0, 0, nz, 1.0, 10.0, 0.0))
    rng = np.random.default_rng(1)
    vol = rng.random((nz, ny, nx), dtype=np.float32)
    a.upload(vol)
    b.upload(vol[::-1].copy())
    reps = 30
    out = {}
    nvox = nx * ny * nz
    for name, fn, nbytes in (("dsi_fuse", lambda: a.harmonicMeanTwoGrids(b), 12.0 * nvox),
                             ("dsi_fuse_into", lambda: m.dsi_.setToFusionOf(a, b, d.FUSE_HM), 12.0 * nvox),
                             ("nary_accumulate_log", lambda: a.accumulate(b, d.ACC_LOG_SUM), 12.0 * nvox)):
        fn()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        ms = ctx.timer_stop() / reps
        out[name] = {"ms": ms, "GBps": nbytes / (ms * 1e-3) / 1e9,
                     "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    # the 4-camera geometric mean as the tree of the reference's 2-ary op: 4 reads + 1 write per voxel, and
    # inside the arg-max kernel (4 reads)
    c4 = [a, b, d.Grid3D(ctx, nx, ny, nz), d.Grid3D(ctx, nx, ny, nz)]
    c4[2].upload(vol[:, ::-1].copy())
    c4[3].upload(vol[:, :, ::-1].copy())
    for name, fn, nbytes in (("gm_tree_4", lambda: m.dsi_.setToFusionOfN(c4, d.ACC_GM_TREE), 20.0 * nvox),
                             ("gm_log_4", lambda: m.dsi_.setToFusionOfN(c4, d.ACC_LOG_SUM), 20.0 * nvox),
                             ("argmax_of_gm_tree_4", lambda: m.computeDepthMapOfFusionN(c4, d.ACC_GM_TREE),
                              16.0 * nvox + 9.0 * nx * ny)):
        fn()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        ms = ctx.timer_stop() / reps
        out[name] = {"ms": ms, "GBps": nbytes / (ms * 1e-3) / 1e9,
                     "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    # arg-max: ONE volume (210 MB) fits the 256 MiB Infinity Cache, so re-reading it measures the cache (round 4 did:
    # 5.8 TB/s); here four different volumes take turns (840 MB between two reads of the same one)
    turn = {"i": 0}

    def argmax_next():
        m.computeDepthMap(c4[turn["i"] & 3])
        turn["i"] += 1
    for _ in range(4):
        argmax_next()
    ctx.timer_start()
    for _ in range(reps + 2):
        argmax_next()
    ms = ctx.timer_stop() / (reps + 2)
    nbytes = (4.0 * nz + 9.0) * nx * ny
    out["argmax"] = {"ms": ms, "GBps": nbytes / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "note": "four 210 MB volumes in turn: served from HBM, not from the Infinity Cache"}
    out["grid"] = "%dx%dx%d (%.0f MB per volume)" % (nx, ny, nz, 4.0 * nvox / 1e6)
    out["resident_in"] = "HBM (three 210 MB volumes of a fuse do not fit the 256 MiB Infinity Cache)"
    for o in [m] + c4:
        o.close()
    # the same 2-ary fuse at the METRIC's own shape (346x260x100: 36 MB per volume, 108 MB per fuse): it runs out of the
    # Infinity Cache, so its GB/s is a cache figure, NOT an HBM figure -- reported beside, labelled
    mx, my, mz = 346, 260, 100
    a2, b2 = d.Grid3D(ctx, mx, my, mz), d.Grid3D(ctx, mx, my, mz)
    v2 = rng.random((mz, my, mx), dtype=np.float32)
    a2.upload(v2)
    b2.upload(v2[::-1].copy())
    a2.harmonicMeanTwoGrids(b2)
    ctx.timer_start()
    for _ in range(reps):
        a2.harmonicMeanTwoGrids(b2)
    ms = ctx.timer_stop() / reps
    nb2 = 12.0 * mx * my * mz
    out["dsi_fuse_at_metric_shape"] = {"grid": "%dx%dx%d (%.0f MB per volume)" % (mx, my, mz, 4.0 * mx * my * mz / 1e6), "ms": ms,
                                       "GBps": nb2 / (ms * 1e-3) / 1e9, "resident_in": "Infinity Cache (108 MB per fuse < 256 MiB)",
                                       "note": "a cache-bandwidth figure: not comparable with the HBM peak"}
    a2.close()
    b2.close()
    return out


def cpu_baseline(d, rig, dims, n_sample):
    """The oracle (a port of the reference's CPU path, oracle/) on the GPU box's host cores, same
    workload: stage A + reset + fillVoxelGrid of camera 0 (OpenMP over planes like the reference,
    mapper_emvs_stereo.cpp:168), built with the reference's flags (-O3, no -march=native) and as
    "tuned" (-O3 -march=native); best of 5 after a warm-up; one thread on a 1/16 sample; plus the CPU
    fusion (harmonic mean) and arg-max of the same grid."""
    from oracle import oracle as orc
    from oracle_pipeline import OracleMapper
    nx, ny, nz = dims
    n_s = min(n_sample, rig["events"][0][0].shape[0])
    x, y, ts = (a[:n_s] for a in rig["events"][0])
    first, Rt = d.packetize(ts, rig["trajectories"][0], rig["T_rv_w"])
    first = first.astype(np.int64)
    out = {}

    def run(tag):
        r = OracleMapper(rig["cam"], dimX=nx, dimY=ny, dimZ=nz, min_depth=4.0, max_depth=200.0)
        r.evaluate_packets(x, y, first[:64], Rt[:64])      # warm-up (page in, spin up OpenMP)
        best = float("inf")
        for _ in range(5):
            t1 = time.perf_counter()
            r.evaluate_packets(x, y, first, Rt)             # stage A + reset + fillVoxelGrid
            best = min(best, time.perf_counter() - t1)
        out[tag] = first.shape[0] * 1024 / best / 1e6
        return r, best

    r, tc = run("value")
    n1 = max(64, first.shape[0] // 16)
    all_threads = orc.num_threads()
    orc.set_num_threads(1)
    t1 = time.perf_counter()
    r.evaluate_packets(x, y, first[:n1], Rt[:n1])
    out["one_thread_value"] = n1 * 1024 / (time.perf_counter() - t1) / 1e6
    orc.set_num_threads(all_threads)
    # CPU fusion + arg-max on the same grid (the reference runs them single-threaded,
    # cartesian3dgrid.h:63 "do not use parallelization yet")
    g2 = r.dsi[::-1].copy()
    best_f = best_a = float("inf")
    for _ in range(5):
        t1 = time.perf_counter()
        orc.fuse2(r.dsi, g2, 2)
        best_f = min(best_f, time.perf_counter() - t1)
        t1 = time.perf_counter()
        orc.collapse_max_z(r.dsi)
        best_a = min(best_a, time.perf_counter() - t1)
    nvox = nx * ny * nz
    out["fuse_ms"] = best_f * 1e3
    out["fuse_GBps"] = 12.0 * nvox / best_f / 1e9
    out["argmax_ms"] = best_a * 1e3
    out["argmax_GBps"] = (4.0 * nz + 9.0) * nx * ny / best_a / 1e9
    try:
        orc.use_native(True)                                # -O3 -march=native build, compiled on this box
        _, tn = run("tuned_value")
        out["tuned_flags"] = "-O3 -march=native -fopenmp -ffp-contract=off"
    except Exception as e:                                  # no compiler on the box: report, do not fail
        out["tuned_value"] = None
        out["tuned_error"] = str(e)[:200]
    finally:
        orc.use_native(False)
    out.update({"unit": "Mevents/s", "cores": min(all_threads, nz), "kind": "port",
                "sample": "camera 0, first %d events (%d packets) of the same workload, %dx%dx%d DSI; oracle stage A + "
                          "fillVoxelGrid, OpenMP over planes (reference strategy: at most dimZ threads busy of %d "
                          "visible), -O3 no -march=native (reference flags); best of 5 after warm-up, %.2f s per run"
                          % (n_s, first.shape[0], nx, ny, nz, all_threads, tc)})
    return out


def parity_block(d, rig, dims, mappers, batches, fused):
    """Closes the depth-map statement on the bench's own inputs (rank 0, N = 1, stereo): the CPU oracle
    (oracle/, the checker -- never the thing measured) builds both camera DSIs from the same events, fuses
    them and takes the arg-max; reported: the worst voxel error of the GPU DSIs and of the fused DSI, the
    fraction of pixels whose plane index equals the oracle's, the fraction where it differs but is a provable
    near-tie of the oracle's column, and the number of pixels that are neither (0 = "depth map equal to the CPU
    reference wherever that is defined")."""
    from oracle import oracle as orc
    from oracle_pipeline import OracleMapper, argmax_report
    nx, ny, nz = dims
    t1 = time.perf_counter()
    refs, errs = [], []
    for c in range(2):
        mappers[c].evaluateDSI_batch(batches[c])
        r = OracleMapper(rig["cam"], dimX=nx, dimY=ny, dimZ=nz, min_depth=4.0, max_depth=200.0)
        r.evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
        got = mappers[c].dsi_.download()
        errs.append(float((np.abs(got.astype(np.float64) - r.dsi) / np.maximum(1.0, np.abs(r.dsi))).max()))
        refs.append(r.dsi)
    fused.setToFusionOf(mappers[0].dsi_, mappers[1].dsi_, d.FUSE_HM)
    mappers[0].computeDepthMap(fused)
    depth, conf, idx = mappers[0].fetchDepthMap()
    ref = orc.fuse2(refs[0].copy(), refs[1], 2)
    gf = fused.download()
    ferr = float((np.abs(gf.astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))).max())
    tol = 3e-4                                               # HM of two volumes each within 1e-4
    unresolved = argmax_report(idx, ref, tol)
    # the exact tie resolver (dsi_mapper_resolve_near_ties, optional, off the timed path): the contending voxels of the
    # near-tie columns re-summed in the reference's order, after which the index map must BE the oracle's
    res = mappers[0].resolveNearTies(mappers, batches, d.FUSE_HM)
    depth, conf, idx = mappers[0].fetchDepthMap()
    rep = argmax_report(idx, ref, tol)
    rep["index_map_equals_oracle"] = bool(np.array_equal(idx, ref.argmax(axis=0)))
    # its cost in a stream of steps: the first call also allocates the scratch it keeps (grow-only), so the arg-max is
    # taken again and resolved again, and that call's time is `elapsed_ms`
    first_ms = res["elapsed_ms"]
    mappers[0].computeDepthMap(fused)
    res = mappers[0].resolveNearTies(mappers, batches, d.FUSE_HM)
    idx2 = mappers[0].fetchDepthMap()[2]
    res["first_call_ms"] = first_ms
    res["repeatable"] = bool(np.array_equal(idx2, idx))
    rep["exact_tie_resolver"] = res
    # the resolver's premise as a per-column PROOF (dsi_mapper_prove_near_ties, a verification pass off every timed path):
    # every voxel's votes counted, the reference's fp32 event-order sums bounded from the counts; with the default gap first,
    # then -- if some columns' bounds need a wider gap -- resolved and proven again with the gap the proof asks for
    try:
        from dvs_mcemvs_amd import process as proc
        at_default = mappers[0].proveNearTies(mappers, batches, d.FUSE_HM, rel_gap=res["rel_gap"])
        mappers[0].computeDepthMap(fused)
        info_p, proof_p = proc.resolve_near_ties_proven(mappers[0], mappers, batches, d.FUSE_HM)
        idx_p = mappers[0].fetchDepthMap()[2]
        settled = proof_p["columns_unproven"] == proof_p["columns_resolved_fully"]
        # the resolver's steady cost at the gap the proof settled on (the call inside the helper may have grown the scratch)
        mappers[0].computeDepthMap(fused)
        steady = mappers[0].resolveNearTies(mappers, batches, d.FUSE_HM, rel_gap=info_p["rel_gap"])
        rep["proof"] = {
            "at_default_gap": at_default,
            "proven_mode": {"rel_gap": info_p["rel_gap"], "candidate_voxels": steady["candidate_voxels"], "votes": steady["votes"],
                            "resolver_elapsed_ms": steady["elapsed_ms"], "proof": proof_p,
                            "index_map_equals_oracle": bool(np.array_equal(idx_p, ref.argmax(axis=0)))},
            "every_column_settled": bool(settled),
            "resolver_ms_proven": float(steady["elapsed_ms"]),
            "note": ("a column is proven when no plane outside the re-summed gap can reach the maximum's plane under rigorous "
                     "bounds of the reference's fp32 event-order sums ((n-1)u/(1-(n-1)u) of the weights' sum, n = the voxel's "
                     "counted votes); process.resolve_near_ties_proven widens the gap where the bounds ask for it and re-sums "
                     "the columns no moderate gap settles on all their planes: the resolved index is then the reference's by "
                     "proof, not by comparison with the oracle.  The proof passes are verification passes, off every timed path")}
    except d.DsiError as e:
        rep["proof"] = {"error": str(e)}
    rep["without_resolver"] = {k: unresolved[k] for k in ("argmax_agree_frac", "near_tie_frac", "violations")}
    planes = mappers[0].raw_depths_vec_
    rep.update({"dsi_max_rel_err": errs, "fused_max_rel_err": ferr, "dsi_tolerance": 1e-4, "fused_tolerance": tol,
                "depth_is_plane_of_index": bool(np.array_equal(depth, planes[idx])),
                "checker": "oracle/ (CPU restatement, parity unpinned), both cameras, all events; %.1f s" %
                           (time.perf_counter() - t1)})
    return rep


def paired_mode_block(d, ctx, rig, dims, batches, exact_mappers, fused, steps, voted_per_step):
    """The same stereo step with the OPT-IN lane mapping 8 (paired 32-bit Q.19 cells: two LDS atomics per vote instead of
    four, rounded weights instead of the exact Q33.31 sums -- VERDICT r05 item 3).  Reported BESIDE `value`, which stays
    the exact default: step and kernel time, the roofline fractions of the same roof (algorithmic adds: still 4 per
    accepted event-plane; issued: 2 per record), the worst voxel against the exact mapping's DSI, the overflow report."""
    nx, ny, nz = dims
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    ms = []
    for c in range(2):
        m = d.MapperEMVS(ctx, rig["cam"], shape)
        m.set_packed_lanes(8)
        ms.append(m)

    def step():
        for c in range(2):
            ms[c].evaluateDSI_batch(batches[c])
        fused.setToFusionOf(ms[0].dsi_, ms[1].dsi_, d.FUSE_HM)
        ms[0].computeDepthMap(fused)
    for _ in range(5):
        step()
    ctx.synchronize()
    for m in ms:
        m.set_kernel_timing(True)
        m.vote_kernel_time()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.synchronize()
    ms_step = 1e3 * (time.perf_counter() - t0) / steps
    kt, kn = 0.0, 0
    for m in ms:
        a, n = m.vote_kernel_time()
        kt += a
        kn += n
        m.set_kernel_timing(False)
    kern_ms = kt / max(1, kn)
    accepted, records = exact_mappers[0].vote_statistics(batches[0])
    adds, peak, _ = lds_block(accepted, kern_ms)
    errs = []
    for c in range(2):
        exact_mappers[c].evaluateDSI_batch(batches[c])
        a = exact_mappers[c].dsi_.download().astype(np.float64)
        b = ms[c].dsi_.download().astype(np.float64)
        errs.append(float((np.abs(a - b) / np.maximum(1.0, np.abs(a))).max()))
    out = {"lane_mapping": 8, "value": voted_per_step / (ms_step * 1e-3) / 1e6, "unit": "Mevents/s", "ms_per_step": ms_step,
           "kernel_avg_ms": kern_ms, "kernel_launches": kn, "frac": adds / peak,
           "frac_issued": (2.0 * records / (kern_ms * 1e-3)) / peak if records else None,
           "lds_atomics_per_record": 2,
           "max_rel_diff_vs_exact_mapping": errs, "tolerance": 1e-4,
           "overflow_reported": [bool(m.paired_overflow()) for m in ms],
           "note": "OPT-IN (dsi_mapper_set_packed_lanes(m, 8)); rounded Q.19 weights on paired 32-bit cells, not the exact Q33.31 "
                   "sums; `value` above is the exact default mapping"}
    for m in ms:
        m.close()
    return out


def sensitivity_block(d, syn, ctx, args, dims, tune):
    """How much the dominant kernel's time depends on the INPUT, at the headline size (one camera, `--events` events,
    the headline grid): the scene decides how many events of a packet share a pixel (merged into one record by the
    packet sort: fewer atomics) and how often two lanes of a wave instruction hit the same voxel.  Three inputs beside
    the headline's (5,000 scene points + 10 % noise):
      uniform_pixels      every event on a uniformly random pixel (no structure, hardly any duplicate: the most
                          atomics per event),
      dense_scene         200,000 scene points (SURVEY 8d's range is 2,000-20,000),
      zurich_city_04      the recorded vehicle trajectory of DSEC zurich_city_04, 10-15 s
                          (tests/golden/zurich_city_04_poses_9_16s.npz, from the reference's pose.bag), 5 s window as
                          in cfg/DSEC/zurich_04_a_full/dsec.conf:13-14.
    Each sub-record carries its own kernel time (HIP events, `reps` launches after `warm` warm-up launches), fractions of
    the LDS roof and duplicate-merge ratio; `quote` names the slowest -- the number to quote for this kernel."""
    nx, ny, nz = dims
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    n_ev, reps, warm = args.events, 25, 25
    cases = []

    def uniform():
        rig = syn.stereo_rig(2048, width=nx, height=ny, t0=10.0, duration=0.5, seed=4321, n_cams=1)   # poses only
        rng = np.random.default_rng(4321)
        ev = (rng.integers(0, nx, n_ev).astype(np.uint16), rng.integers(0, ny, n_ev).astype(np.uint16),
              np.sort(rng.uniform(rig["t0"], rig["t1"], n_ev)))
        return rig, ev, "uniformly random pixels, analytic rig trajectory"

    def dense():
        rig = syn.stereo_rig(n_ev, width=nx, height=ny, t0=10.0, duration=0.5, seed=4322, n_cams=1, n_points=200_000)
        return rig, rig["events"][0], "200,000 scene points + 10 % noise, analytic rig trajectory"

    def zurich():
        z = np.load(os.path.join(ROOT, "tests", "golden", "zurich_city_04_poses_9_16s.npz"))
        rig = syn.stereo_rig(n_ev, width=nx, height=ny, t0=10.0, duration=5.0, seed=4323, n_cams=1,
                             n_points=args.points, pose_fn=syn.recorded_rig(z["times"], z["poses"]))
        return rig, rig["events"][0], ("%d scene points + 10 %% noise along the recorded zurich_city_04 trajectory, 10-15 s"
                                       % args.points)

    for name, make in (("uniform_pixels", uniform), ("dense_scene", dense), ("zurich_city_04", zurich)):
        t1 = time.perf_counter()
        rig, ev, what = make()
        first, Rt = d.packetize(ev[2], rig["trajectories"][0], rig["T_rv_w"])
        batch = d.EventBatch(ctx, ev[0], ev[1], Rt, first)
        m = tune(d.MapperEMVS(ctx, rig["cam"], shape))
        # warm-up: the allocations, and the clocks -- the GPU idles while the host generates the case's events, and the first
        # launches after an idle stretch run slow (8 timed launches after ONE warm-up read 1.56 ms for the dense scene
        # that a sustained loop runs in 1.32 ms)
        for _ in range(warm):
            m.evaluateDSI_batch(batch)
        ctx.synchronize()
        m.set_kernel_timing(True)
        m.vote_kernel_time()
        for _ in range(reps):
            m.evaluateDSI_batch(batch)
        ms, n = m.vote_kernel_time()
        m.set_kernel_timing(False)
        accepted, records = m.vote_statistics(batch)
        rb = roofline_block(m.last_vote_info(), ms / max(1, n), n, accepted, first.shape[0] * d.PACKET_SIZE, nz, None, records)
        cases.append({"case": name, "input": what, "events": int(first.shape[0] * d.PACKET_SIZE),
                      "kernel": rb["kernel"], "kernel_avg_ms": rb["kernel_avg_ms"], "kernel_launches": n,
                      "kernel_Mevents_per_s": rb["kernel_Mevents_per_s"], "frac": rb["frac"],
                      "frac_issued": rb.get("frac_issued"),
                      "records_per_accepted_event_plane": rb.get("records_per_accepted_event_plane"),
                      "accepted_event_planes_per_launch": accepted, "prepare_s": time.perf_counter() - t1})
        m.close()
        batch.close()
    return cases


def other_workloads():
    """The one-GPU lines of BASELINE configs[2] (stream of 50 ms windows) and of the configs[4] shape (4 cameras,
    1024x1024x256, n-ary GM), each a sub-run of this script with its own roofline block, so that the driver's
    default run carries them too.  Run after this process has released the GPU."""
    import subprocess
    res = {}
    runs = (("windows", ["--workload", "windows"]), ("cameras4", ["--workload", "cameras4"]),
            # BASELINE configs[4] at ITS OWN size on one GPU: 4 x 99,993,600 events, 1024 x 1024 x 256, GM tree + arg-max
            ("cameras4_full", ["--workload", "cameras4", "--events", "100000000", "--tile", "10", "--steps", "3", "--warmup", "1",
                               "--clock-ramp", "1"]))
    for name, flags in runs:
        # (the windows line carries its host-fed rates: Python from pageable / page-locked arrays, C++ from std::vector<Event>)
        cmd = [sys.executable, os.path.abspath(__file__)] + flags + ["--no-cpu", "--no-extra"] + ([] if name == "windows" else ["--no-host-fed"])
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1]
            j = json.loads(line)
            res[name] = {k: j.get(k) for k in ("value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline",
                                                 "step_ms", "windows_per_s", "x_real_time", "timed_region_s",
                                                 "device_memory", "input_gen_s", "host_fed", "exact_ties", "ms_per_window_exact")}
        except Exception as e:                                  # report, do not fail the headline
            res[name] = {"error": str(e)[:300]}
    return res


def main():
    args = parse()
    from dvs_mcemvs_amd import launch
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if not launch.launched_by_a_launcher():
        if args.gpus > 1:
            # nobody started the ranks: start them here, one process per GPU (never a smaller job)
            import __graft_entry__ as ge
            ge.build()
            import dvs_mcemvs_amd as d0
            n_dev = d0.device_count()
            sys.exit(launch.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                        n_devices=n_dev))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d contradicts WORLD_SIZE=%s of the launcher; refusing to guess"
                         % (args.gpus, os.environ["WORLD_SIZE"]))
    # (plumbing tests on a one-GPU box: DSI_BENCH_DEVICE=0 puts every rank on GPU 0 -- RCCL then refuses the
    #  communicator and the run fails, which is the point of such a test)
    D = launch.Dist(device=os.environ.get("DSI_BENCH_DEVICE"))
    world, rank = D.world, D.rank

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    D.barrier()
    import dvs_mcemvs_amd as d
    from dvs_mcemvs_amd import distributed as dd, process as proc, synthetic as syn

    nx, ny, nz = args.dims
    n_dev = d.device_count()
    if D.local_rank >= n_dev:
        raise SystemExit("bench.py rank %d: device %d requested but this node has %d GPU device(s); one rank per "
                         "device, no fallback" % (rank, D.local_rank, n_dev))
    ctx = d.Context(D.local_rank)
    allreduce, comm, collective = make_comm(d, dd, ctx, D)
    # the proof of the rank count: what RCCL itself reports for the communicator, gathered from every rank
    rccl = comm.query() if comm is not None else None
    ranks = D.gather({"rank": rank, "pid": os.getpid(), "hip_device": ctx.device,
                      "rccl_nranks": rccl[0] if rccl else None, "rccl_rank": rccl[1] if rccl else None,
                      "rccl_device": rccl[2] if rccl else None})

    def tune(m):
        m.set_vote_algo(args.algo)
        m.set_band_params(*args.band)
        m.set_packed_lanes(args.packed)
        if args.inline_cuts >= 0:
            m.set_inline_cuts(args.inline_cuts)
        if args.pass_lg:
            from dvs_mcemvs_amd import engine as eng
            if not eng.experiments_requested():
                raise SystemExit("--pass-lg is a knob of the experiments flavour of the engine: build it (python -m "
                                 "dvs_mcemvs_amd.build --experiments) and run with DSI_ENGINE_EXPERIMENTS=1")
            d.load_library().dsi_test_pass_lg(m._h, args.pass_lg)
        return m

    t_gen = time.time()
    extra = {}
    closers = []
    vote_mappers = []

    if args.workload == "stereo":
        # ---- one stereo time slice per rank, generated and uploaded before timing ----
        rig = syn.stereo_rig(args.events, width=nx, height=ny, t0=10.0 + 0.5 * rank, duration=0.5,
                             seed=1234 + 100 * rank, n_points=args.points)
        shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)   # cfg/DSEC/zurich_04_a_full/dsec.conf:11-12,17
        mappers, batches, voted = [], [], 0
        for c in range(2):
            m = tune(d.MapperEMVS(ctx, rig["cam"], shape))
            first, Rt = d.packetize(rig["events"][c][2], rig["trajectories"][c], rig["T_rv_w"])
            batches.append(d.EventBatch(ctx, rig["events"][c][0], rig["events"][c][1], Rt, first))
            voted += first.shape[0] * d.PACKET_SIZE
            mappers.append(m)
        vote_mappers = mappers
        extra["batch0"] = batches[0]
        fused = d.Grid3D(ctx, nx, ny, nz)
        closers += mappers + batches + [fused]
        temporal = None
        if world > 1:
            ctx_side = d.Context(D.local_rank)
            mapper_fused = d.MapperEMVS(ctx_side, rig["cam"], shape)
            temporal = dd.EnginePipelinedTemporalFusion(
                ctx, ctx_side, (nx, ny, nz), d.ACC_INV_SUM, world, allreduce, extract=mapper_fused.computeDepthMap,
                scattered=(mapper_fused, comm) if args.temporal_collective == "reduce_scatter" else None)
            fused.resetGrid()
            temporal.submit(fused)      # one un-timed round: set-up problems show up here on every rank
            temporal.drain()

        def step():
            for c in range(2):
                mappers[c].evaluateDSI_batch(batches[c])
            # process1.cpp:126-141 (resetGrid; addTwoGrids(dsi0); harmonicMeanTwoGrids(dsi1)) in one pass
            fused.setToFusionOf(mappers[0].dsi_, mappers[1].dsi_, d.FUSE_HM)
            if temporal is None:
                mappers[0].computeDepthMap(fused)
            else:
                temporal.submit(fused)   # process2.cpp:220 + ONE all-reduce(sum) + n/acc + arg-max, side stream

        def sync():
            ctx.synchronize()
            if temporal is not None:
                temporal.drain()

        voted_per_step = voted
        workload = ("stereo (2-cam) synthetic DSEC-like rig, %d events/cam per GPU, %dx%dx%d DSI, harmonic camera "
                    "fusion + arg-max%s" % (args.events, nx, ny, nz, "" if world == 1 else
                                            ", %d time slices (one per GPU) fused by ONE all-reduce of inverse sums"
                                            % world))
        parallelism, scaling = ("1 GPU" if world == 1 else "time-slice x%d" % world), "weak"
        ev_per_launch = voted / 2.0

    elif args.workload == "windows":
        # ---- configs[2]: 8 distinct 50 ms windows (cycled), resident in HBM as packetised batches ----
        depth = max(1, min(4, args.window_depth))
        n_distinct = 12 if depth == 3 else 8          # (a multiple of the depth: window wi always lands in slot wi % depth)
        dur = args.events / 10.0e6                    # 10 Mev/s per camera
        rig = syn.stereo_rig(n_distinct * args.events, width=640, height=480, t0=10.0 + 10.0 * rank,
                             duration=n_distinct * dur, seed=77 + rank, n_points=max(args.points, 6000))
        shape = d.ShapeDSI(nx, ny, nz, 4.0, 200.0, 0.0)
        fused_vote = (args.fused_vote is not False) and not args.materialize_fused and args.algo in (0, 2)
        concurrent = fused_vote and not args.serial_windows
        ws = proc.WindowStream(ctx, (rig["cam"],) * 2, shape, d.FUSE_HM, materialize_fused=args.materialize_fused,
                               fused_vote=fused_vote, concurrent=concurrent, depth=depth)
        for m in [m for ms_ in ws.mapper_sets for m in ms_] + ws.extract:
            tune(m)
        # (the fused kernel reads its knobs and its timer from the OUTPUT mapper of the call)
        vote_mappers = ws.extract if fused_vote else [m for ms_ in ws.mapper_sets for m in ms_]
        bounds = proc.window_bounds(rig["t0"], rig["t1"] + 1e-9, dur, dur)[:n_distinct]
        wins, host_wins = [], []
        for wi, (a, b) in enumerate(bounds):
            T_rv_w = proc.reference_view_process1(rig["trajectories"][0], b)
            per_cam, host = [], []
            for c in range(2):
                ev = proc.window_events(rig["events"][c], a, b)
                first, Rt = d.packetize(ev[2], rig["trajectories"][c], T_rv_w)
                # (its batches live in its slot's context)
                per_cam.append(d.EventBatch(ws.context_of_slot(wi % depth), ev[0], ev[1], Rt, first))
                host.append(ev)
            wins.append((per_cam, b))
            host_wins.append((host, b))
            closers += per_cam
        closers.append(ws)
        state = {"w": 0, "pending": []}

        def step():
            per_cam, ts = wins[state["w"] % len(wins)]
            state["pending"].append(ws.submit(None, rig["trajectories"], ts, batches=per_cam))
            while len(state["pending"]) >= depth:
                ws.fetch(state["pending"].pop(0))     # depth map of the oldest window in flight (device -> host)
            state["w"] += 1

        def sync():
            while state["pending"]:
                ws.fetch(state["pending"].pop(0))
            for c_ in ws.contexts:
                c_.synchronize()

        extra["concurrent"] = concurrent
        extra["serial_step"] = (step, sync)
        voted_per_step = float(np.mean([sum(b.n_packets for b in pc) for pc, _ in wins])) * d.PACKET_SIZE
        workload = ("stream of %.0f ms windows (main.cpp:177 loop), 2 cameras x %d events per window, sensor 640x480, "
                    "%dx%dx%d DSI, per window: reset + vote x2 + harmonic camera fusion + arg-max + depth-map fetch (%s)"
                    % (dur * 1e3, args.events, nx, ny, nz,
                       "fused DSI written" if args.materialize_fused else
                       "ONE kernel votes both cameras band by band in LDS, fuses them and keeps the running arg-max: "
                       "no DSI is written" + ("; consecutive windows on two streams" if concurrent else "") if fused_vote else
                       "camera fusion computed inside the arg-max kernel, fused DSI not written"))
        parallelism = "1 GPU" if world == 1 else "replicas x%d (independent windows, no collective)" % world
        scaling = "weak"
        ev_per_launch = voted_per_step if fused_vote else voted_per_step / 2.0
        extra["host_windows"] = host_wins
        extra["resident_windows"] = wins
        extra["depth"] = depth
        extra["fused_vote"] = fused_vote
        extra["last_window"] = wins[-1][0]
        extra["last_mappers"] = ws.mapper_sets[(len(wins) - 1) % len(ws.mapper_sets)]

    else:
        # ---- configs[4] shape: 4 cameras, n-ary GM; N > 1: plane sharding ----
        tile = max(1, args.tile)
        rig = syn.stereo_rig(args.events // tile, width=nx, height=ny, t0=10.0, duration=0.5, seed=1234, n_cams=4,
                             n_points=args.points)
        shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
        begin, count = dd.plane_ranges(nz, world)[rank]
        mappers, batches, voted = [], [], 0
        # --fused-vote: the four cameras through ONE kernel that votes, fuses (GM tree) and keeps the running arg-max: no DSI
        # is written (dsi_mapper_depth_map_of_events_n; one GPU, the tree form, one context)
        # Default: on up to 4 M events per camera (4 x 2 M: 6.04 vs 6.24 ms per step), off beyond -- the DSI-less kernel votes the two
        # seam rows of every band twice (14-row bands at 1024 wide: +14 % votes), which at 100 M events per camera costs far more
        # than the 4 GiB of DSI traffic it saves
        want_fused4 = args.fused_vote if args.fused_vote is not None else args.events <= 4_000_000
        fused4 = bool(want_fused4) and world == 1 and args.gm == "tree" and not args.materialize_fused and args.algo in (0, 2)
        # cameras dealt over one or two contexts of this GPU (the arg-max / fusion waits for both and releases them)
        cam_ctxs = [ctx] + [d.Context(D.local_rank) for _ in range(0 if fused4 else max(0, min(4, args.camera_streams) - 1))]
        for c in range(4):
            cc = cam_ctxs[c % len(cam_ctxs)]
            m = tune(d.MapperEMVS(cc, rig["cam"], shape, plane_range=(begin, count)))
            first, Rt = d.packetize(rig["events"][c][2], rig["trajectories"][c], rig["T_rv_w"])
            ex, ey = rig["events"][c][0], rig["events"][c][1]
            if tile > 1:
                # the same pixels seen from a rig that has moved on: `tile` different stretches of the trajectory
                n = first.shape[0] * d.PACKET_SIZE
                assert np.array_equal(first, np.arange(first.shape[0], dtype=np.uint32) * d.PACKET_SIZE)
                rts = []
                for i in range(tile):
                    r_ = Rt.copy()
                    r_[:, 9] += 0.02 * i
                    r_[:, 11] += 0.01 * i
                    rts.append(r_)
                ex, ey, Rt, first = np.tile(ex[:n], tile), np.tile(ey[:n], tile), np.concatenate(rts), None
            batches.append(d.EventBatch(cc, ex, ey, Rt, first))
            voted += batches[-1].n_packets * d.PACKET_SIZE
            mappers.append(m)
        vote_mappers = mappers
        extra["batch0"] = batches[0]
        fused = d.Grid3D(ctx, nx, ny, count)
        closers += mappers + batches + [fused] + cam_ctxs[1:]
        gm_mode = d.ACC_GM_TREE if args.gm == "tree" else d.ACC_LOG_SUM

        if fused4:
            out_mapper = tune(d.MapperEMVS(ctx, rig["cam"], shape))
            closers.append(out_mapper)
            vote_mappers = [out_mapper]           # (the fused kernel reads its knobs and its timer from the OUTPUT mapper)
            extra["fused_vote"] = True
            extra["last_mappers"], extra["last_window"] = mappers, batches

        def step():
            if fused4:
                out_mapper.computeDepthMapOfEventsN(mappers, batches)
                return
            for c in range(4):
                mappers[c].evaluateDSI_batch(batches[c])
            if comm is None and not args.materialize_fused:
                # n-ary GM inside the arg-max kernel: same bits, the fused volume is never written
                mappers[0].computeDepthMapOfFusionN([m.dsi_ for m in mappers], gm_mode)
                return
            for cc in cam_ctxs[1:]:
                ctx.wait_for(cc)                                                 # the fusion reads grids voted on the other stream
            fused.setToFusionOfN([m.dsi_ for m in mappers], gm_mode)           # n-ary GM, voxel-wise, local
            for cc in cam_ctxs[1:]:
                cc.wait_for(ctx)                                                 # ... whose next votes must not overtake it
            if comm is None:
                mappers[0].computeDepthMap(fused)
            else:
                mappers[0].computeDepthMapSharded(fused, comm)                  # ONE all-reduce(MAX) of keys

        def sync():
            for cc in cam_ctxs:
                cc.synchronize()

        if len(cam_ctxs) > 1 and not fused4:
            # an event pair around a kernel also times its wait for the other stream's workgroups to leave the CUs: the
            # dominant kernel's duration is measured on un-overlapped steps after the timed region (like the windows)
            extra["concurrent"] = True

            def serial_step():
                for c in range(4):
                    mappers[c].evaluateDSI_batch(batches[c])
                    sync()
            extra["serial_step"] = (serial_step, sync)
            extra["camera_streams"] = len(cam_ctxs)

        # every rank votes ALL events into its plane range: the job's events are counted once
        voted_per_step = voted if rank == 0 else 0.0
        workload = ("4-camera synthetic rig, %d events/cam, %dx%dx%d DSI, n-ary geometric-mean camera fusion (%s) + arg-max%s"
                    % (args.events, nx, ny, nz,
                       "tree of the reference's 2-ary sqrt(a*b)" if args.gm == "tree" else "exp(mean(log))",
                       (" (fused DSI written)" if args.materialize_fused else
                        " -- ONE kernel votes the four cameras band by band in LDS, fuses them and keeps the running arg-max: no DSI "
                        "is written" if fused4 else " in one kernel (fused DSI not written)")
                       if world == 1 else ", planes sharded over %d GPUs" % world) +
                    ("; cameras dealt over %d streams" % len(cam_ctxs) if len(cam_ctxs) > 1 else ""))
        parallelism, scaling = ("1 GPU" if world == 1 else "plane-shard x%d" % world), "strong"
        ev_per_launch = voted if fused4 else voted / 4.0
    t_gen = time.time() - t_gen

    def barrier():
        sync()
        D.barrier()
        sync()

    # clock ramp (untimed, before the contract's W warm-up steps): the GPU idled for seconds while the host generated the
    # inputs, and the first launches after an idle stretch run 1-2 % slow (measured: stereo 2.623 -> 2.589 ms per step,
    # windows 0.451 -> 0.443, cameras4 6.51 -> 6.37 with a long warm-up); a stream of real windows never idles
    # ... and the same K steps timed WITHOUT it first (after the W warm-up steps only, the contract's letter), reported
    # beside `value` as value_no_ramp
    no_ramp = None
    if args.clock_ramp > 0 and world == 1:
        for _ in range(args.warmup):
            step()
        barrier()
        t_nr = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        t_nr = time.perf_counter() - t_nr
        no_ramp = {"value": voted_per_step * args.steps / t_nr / 1e6, "ms_per_step": 1e3 * t_nr / args.steps,
                   "note": "the same K steps after the W warm-up steps only, before the clock-ramp steps"}
    t_ramp = time.perf_counter()
    for i in range(args.clock_ramp):
        step()
        if i % 16 == 15:
            sync()
    sync()
    t_ramp = time.perf_counter() - t_ramp
    for _ in range(args.warmup):
        step()
    barrier()
    overlapped = bool(extra.get("concurrent"))
    for m in vote_mappers:
        m.set_kernel_timing(not overlapped)
        m.vote_kernel_time()
    t0 = time.perf_counter()
    ctx.timer_start()
    ctx.timeline_mark()
    for _ in range(args.steps):
        step()
        ctx.timeline_mark()             # a HIP event per step on the compute stream (asynchronous)
    gpu_ms = ctx.timer_stop()
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = ctx.timeline_read()
    if overlapped:
        # consecutive windows overlap on the device (two streams), so an event pair around a kernel also times
        # its wait for the previous window's workgroups: the dominant kernel's duration is measured on extra,
        # un-overlapped launches AFTER the timed region (one window at a time)
        for m in vote_mappers:
            m.set_kernel_timing(True)
        s_step, s_sync = extra["serial_step"]
        for _ in range(max(8, min(args.steps, 32))):
            s_step()
            s_sync()
    kt_ms, kt_n = 0.0, 0
    for m in vote_mappers:
        ms, n = m.vote_kernel_time()
        kt_ms += ms
        kt_n += n
        m.set_kernel_timing(False)
    elapsed = D.max(elapsed)
    voted_all = D.sum(voted_per_step)

    info = vote_mappers[0].last_vote_info()
    ms_per_step = 1e3 * elapsed / args.steps
    value = voted_all * args.steps / elapsed / 1e6      # Mevents/s, whole job
    kern_ms = kt_ms / max(1, kt_n)
    # accepted event-planes of one launch = sum of the DSI it writes (the 4 bilinear weights of a vote sum to 1);
    # accepted RECORDS = the same after the packet sort merged same-pixel events of a packet (one record = 4 LDS
    # atomics actually issued): dsi_mapper_vote_statistics, outside the timed region
    if extra.get("fused_vote"):
        # the fused kernel writes no DSI: the two camera DSIs of one window are built here, once, to count the work
        # of a launch (both cameras are voted by one launch)
        accepted = records = 0.0
        for m, b in zip(extra["last_mappers"], extra["last_window"]):
            a_, r_ = m.vote_statistics(b)
            accepted += a_
            records += r_
    elif "batch0" in extra:
        accepted, records = vote_mappers[0].vote_statistics(extra["batch0"])
    else:
        accepted, records = float(np.sum(vote_mappers[0].dsi_.download(), dtype=np.float64)), None

    # the collective alone (N > 1, stereo): K un-overlapped all-reduces of the accumulator-sized volume on the compute
    # stream, all ranks in step -- the xGMI cost the pipelined run hides behind the next step's voting
    collective_block = None
    if comm is not None and args.workload == "stereo":
        reps = 10
        allreduce(fused, d.REDUCE_SUM)
        barrier()
        ctx.timer_start()
        for _ in range(reps):
            allreduce(fused, d.REDUCE_SUM)
        coll_ms = D.max(ctx.timer_stop() / reps)
        nbytes = 4.0 * nx * ny * nz
        collective_block = {"op": "all-reduce(sum) of the inverse-sum accumulator, fp32 [%d][%d][%d]" % (nz, ny, nx),
                            "bytes_per_rank": nbytes, "avg_ms_alone": coll_ms, "reps": reps,
                            "algbw_GBps": nbytes / (coll_ms * 1e-3) / 1e9,
                            "busbw_GBps": nbytes / (coll_ms * 1e-3) / 1e9 * 2.0 * (world - 1) / world,
                            "per_step": "1 (on the side stream, overlapped with the next step's voting)",
                            "form": args.temporal_collective}

    out = None
    if rank == 0:
        # PMC traffic of the dominant kernel is a separate rocprofv3 run (tools/profile_round.sh ->
        # profiles/traffic.json); it is quoted here only if that run profiled THIS kernel source on THIS
        # workload shape -- otherwise null (a stale constant would look like a measurement)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.workload == "stereo":
            try:
                tj = json.load(open(tpath))
                if (tj.get("kernel_source_sha16") == kernel_source_sha16() and tj.get("dims") == [nx, ny, nz]
                        and tj.get("events_per_launch") == int(ev_per_launch) and args.points == 5000):   # (measured on the default input)
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # the same for the other workloads' kernels, and the "why" counters of every workload (profiles/counters.json)
        wkey = {"stereo": "stereo", "windows": "windows", "cameras4": "cameras4_full" if args.events >= 50_000_000 else "cameras4"}[args.workload]
        if args.workload == "stereo" and ([nx, ny, nz] != [346, 260, 100] or args.points != 5000):
            wkey = "%dx%dx%d" % (nx, ny, nz)
        kname = {0: "k_vote_bands", 1: "k_vote_bands_packed", 3: "k_vote_bands_packed", 5: "k_vote_bands_vfill",
                 6: "k_vote_bands_vfill", 7: "k_vote_bands_packed"}.get(info["packed"], "k_vote") if info["algo"] == 2 else \
            ("k_vote_fuse_argmax" if info["algo"] == 3 else "k_vote_global")
        pc = profiled_counters(wkey, kname)
        if traffic is None and pc and pc.get("events_per_launch") in (None, int(ev_per_launch)):
            traffic = pc.get("hbm_bytes_per_launch")
        roofline = roofline_block(info, kern_ms, kt_n, accepted, ev_per_launch, info_nz(vote_mappers[0]), traffic,
                                  records)
        if pc:
            roofline["counters"] = {k: pc.get(k) for k in ("kernel", "lds_busy_frac", "parked_frac", "bank_conflict_frac",
                                                           "wait_inst_lds_frac", "valu_busy_frac", "kernel_avg_us_profiled",
                                                           "source")}
        if overlapped:
            roofline["kernel_timing"] = ("HIP events around %d un-overlapped launches after the timed region (in the "
                                         "timed region consecutive %s overlap on two streams)"
                                         % (kt_n, "windows" if args.workload == "windows" else "cameras' kernels"))
        streams = stream_kernels(d, ctx) if not args.no_host_fed else None

        # ---- host-buffer (PCIe-inclusive) rates, reported beside `value`, never as it ----
        h2d = {}
        if args.no_host_fed:
            pass
        elif args.workload == "stereo":
            pk = [d.packetize(rig["events"][c][2], rig["trajectories"][c], rig["T_rv_w"]) for c in range(2)]
            best = best2 = float("inf")
            for _ in range(3):
                t1 = time.perf_counter()
                bt = d.EventBatch(ctx, rig["events"][0][0], rig["events"][0][1], pk[0][1], pk[0][0])
                mappers[0].evaluateDSI_batch(bt)
                ctx.synchronize()
                best = min(best, time.perf_counter() - t1)
                bt.close()
            for _ in range(3):
                t1 = time.perf_counter()
                bts = []
                for c in range(2):     # camera 1's upload overlaps camera 0's voting (copy stream)
                    bts.append(d.EventBatch(ctx, rig["events"][c][0], rig["events"][c][1], pk[c][1], pk[c][0]))
                    mappers[c].evaluateDSI_batch(bts[-1])
                fused.setToFusionOf(mappers[0].dsi_, mappers[1].dsi_, d.FUSE_HM)
                mappers[0].computeDepthMap(fused)
                ctx.synchronize()
                best2 = min(best2, time.perf_counter() - t1)
                for b in bts:
                    b.close()
            h2d = {"one_camera_Mevents_per_s": pk[0][0].shape[0] * d.PACKET_SIZE / best / 1e6,
                   "stereo_step_Mevents_per_s": sum(p_[0].shape[0] for p_ in pk) * d.PACKET_SIZE / best2 / 1e6,
                   "source": "pageable host memory"}
        elif args.workload == "windows":
            hw = extra["host_windows"]
            nrep = 40
            pend = None
            t1 = time.perf_counter()
            for w in range(nrep):       # host packetisation + upload + compute + depth-map fetch per window
                ev, ts = hw[w % len(hw)]
                slot = ws.submit(ev, rig["trajectories"], ts)
                if pend is not None:
                    ws.fetch(pend)
                pend = slot
            ws.fetch(pend)
            dt = (time.perf_counter() - t1) / nrep
            h2d = {"ms_per_window": dt * 1e3, "windows_per_s": 1.0 / dt,
                   "x_real_time": (args.events / 10.0e6) / dt, "Mevents_per_s": voted_per_step / dt / 1e6,
                   "source": "pageable host memory; includes the host packetisation + pose interpolation"}
            # the same stream with the events in page-locked memory (the event source writes there):
            # uploads are plain DMAs on the copy stream, the host never waits for them
            pins = []
            for ev, ts in hw:
                cams_ = []
                for c in range(2):
                    px, py = d.PinnedArray(ev[c][0].shape, np.uint16), d.PinnedArray(ev[c][1].shape, np.uint16)
                    px.a[:] = ev[c][0]
                    py.a[:] = ev[c][1]
                    cams_.append((px, py, ev[c][2]))
                pins.append((cams_, ts))
            pend = None
            for rep in range(2):                     # first pass warms the staging buffers
                t1 = time.perf_counter()
                for w in range(nrep):
                    cams_, ts = pins[w % len(pins)]
                    slot = ws.submit([(cx.a, cy.a, cts) for cx, cy, cts in cams_], rig["trajectories"], ts,
                                     asynchronous=True)
                    if pend is not None:
                        ws.fetch(pend)
                    pend = slot
                ws.fetch(pend)
                pend = None
                dtp = (time.perf_counter() - t1) / nrep
            h2d["pinned"] = {"ms_per_window": dtp * 1e3, "windows_per_s": 1.0 / dtp,
                             "x_real_time": (args.events / 10.0e6) / dtp, "Mevents_per_s": voted_per_step / dtp / 1e6,
                             "source": "page-locked host memory (dsi_host_alloc), asynchronous uploads; includes the "
                                       "host packetisation + pose interpolation"}
            for cams_, _ in pins:
                for cx, cy, _ in cams_:
                    cx.close()
                    cy.close()
            h2d["cpp_stream"] = cpp_stream_line()

        # ---- windows: what the parity-complete depth map costs (the exact tie resolver in the loop: needs the camera DSIs,
        # so the unfused path + dsi_mapper_resolve_near_ties per window), one stream, resident batches ----
        exact_ties = None
        if args.workload == "windows" and world == 1 and not args.no_host_fed:
            per = {}
            res_ms = []
            for name, kw in (("fused_vote", dict(fused_vote=True, materialize_fused=False)),
                             ("exact_ties", dict(materialize_fused=False, exact_ties=True))):
                ws2 = proc.WindowStream(ctx, (rig["cam"],) * 2, shape, d.FUSE_HM, depth=1, **kw)
                for m2 in [m2 for ms_ in ws2.mapper_sets for m2 in ms_] + ws2.extract:
                    tune(m2)
                own = [w_ for i_, w_ in enumerate(extra["resident_windows"]) if i_ % extra["depth"] == 0]   # (slot 0's context = ctx)
                nrep = 24
                for r_ in range(nrep + 6):
                    if r_ == 6:
                        ctx.synchronize()
                        t1 = time.perf_counter()
                    pc_, ts_ = own[r_ % len(own)]
                    ws2.fetch(ws2.submit(None, rig["trajectories"], ts_, batches=pc_))
                    if name == "exact_ties" and r_ >= 6 and ws2.last_resolve is not None:
                        res_ms.append(ws2.last_resolve)
                ctx.synchronize()
                per[name] = 1e3 * (time.perf_counter() - t1) / nrep
                ws2.close()
            exact_ties = {"ms_per_window_one_stream": per["exact_ties"], "fused_vote_ms_per_window_one_stream": per["fused_vote"],
                          "plus_ms_per_window": per["exact_ties"] - per["fused_vote"],
                          "resolver_elapsed_ms_mean": float(np.mean([r_["elapsed_ms"] for r_ in res_ms])) if res_ms else None,
                          "near_tie_pixels_mean": float(np.mean([r_["near_tie_pixels"] for r_ in res_ms])) if res_ms else None,
                          "premise_ok_all": bool(all(r_["premise_ok"] for r_ in res_ms)) if res_ms else None,
                          "note": "depth 1, one stream, fetch included; the resolver needs the camera DSIs: unfused votes + fusion "
                                  "inside the arg-max + dsi_mapper_resolve_near_ties"}

        cpu = None
        if not args.no_cpu and world == 1:      # (the CPU baseline is a rank-0, N = 1 figure; at N > 1 the other ranks would only wait for it)
            cpu = cpu_baseline(d, rig, (nx, ny, nz) if args.workload != "windows" else (nx, ny, nz),
                               10_000_000 if args.workload == "stereo" else 1_000_000)

        parity = None
        if not args.no_cpu and args.workload == "stereo" and world == 1:
            parity = parity_block(d, rig, (nx, ny, nz), mappers, batches, fused)

        sensitivity = None
        if args.workload == "stereo" and world == 1 and not args.no_sensitivity and not args.no_host_fed:
            cases = sensitivity_block(d, syn, ctx, args, (nx, ny, nz), tune)
            head = {"case": "headline", "input": "%d scene points + 10 %% noise, analytic rig trajectory" % args.points,
                    "kernel_avg_ms": roofline.get("kernel_avg_ms"), "frac": roofline.get("frac"),
                    "frac_issued": roofline.get("frac_issued"),
                    "records_per_accepted_event_plane": roofline.get("records_per_accepted_event_plane")}
            slowest = max(cases + [head], key=lambda c: c["kernel_avg_ms"] or 0.0)
            sensitivity = {"cases": cases, "headline": head,
                           "quote": {"case": slowest["case"], "kernel_avg_ms": slowest["kernel_avg_ms"],
                                     "frac": slowest["frac"], "frac_issued": slowest["frac_issued"],
                                     "note": "the slowest input: the figure to quote for this kernel"}}

        paired = None
        if args.workload == "stereo" and world == 1 and not args.no_extra and args.packed != 8 and info["algo"] == 2:
            try:
                paired = paired_mode_block(d, ctx, rig, (nx, ny, nz), batches, mappers, fused, args.steps, voted_per_step)
            except Exception as e:                  # report, do not fail the headline
                paired = {"error": str(e)[:300]}
        others = args.workload == "stereo" and world == 1 and not args.no_extra and not args.no_host_fed
        out = {
            "metric": "Mevents/s into DSI (346x260x100) + DSI-fuse GB/s",
            "value": value, "unit": "Mevents/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "events_voted_per_step": voted_all, "vote_algo": info["algo"],
                       "bands": info["bands"], "band_rows": info["band_rows"], "chunks": info["chunks"],
                       "block_threads": info["block_threads"], "lds_bytes": info["lds_bytes"],
                       "packed_lanes": info["packed"], "parallelism": parallelism, "collective": collective,
                       "launched_by": ("a launcher (RANK / WORLD_SIZE in the environment)" if not D.spawned else
                                       "bench.py itself (dvs_mcemvs_amd.launch.spawn_ranks), one process per GPU")
                       if world > 1 else "single process"},
            "launch_env": launch.environment_report() if world > 1 else None,
            "rccl_ranks": rccl[0] if rccl else 1,
            "ranks": ranks,
            "collective": collective_block,
            "dsi_fuse_GBps": streams["dsi_fuse"]["GBps"] if streams else None,
            "dsi_fuse_ms": streams["dsi_fuse"]["ms"] if streams else None,
            "dsi_fuse_frac_of_hbm_peak": streams["dsi_fuse"]["frac_of_hbm_peak"] if streams else None,
            "argmax_GBps": streams["argmax"]["GBps"] if streams else None,
            "argmax_ms": streams["argmax"]["ms"] if streams else None,
            "stream_kernels": streams,
            "gpu_ms_per_step_hip_events": gpu_ms / args.steps,
            "timed_region_s": elapsed,
            "value_no_ramp": no_ramp["value"] if no_ramp else None,
            "no_ramp": no_ramp,
            "clock_ramp": {"steps": args.clock_ramp, "seconds": t_ramp,
                           "note": "untimed steps before the warm-up steps: brings the clocks up after the idle input generation"},
            "host_fed": h2d, "roofline": roofline, "cpu_baseline": cpu, "input_gen_s": t_gen,
            "step_ms": {"min": float(step_ms.min()), "median": float(np.median(step_ms)), "max": float(step_ms.max()),
                        "n": int(step_ms.shape[0]), "source": "HIP events between consecutive steps on the compute stream" +
                        (" of slot 0 (with two streams a step's events bracket parts of two windows)" if overlapped else "")}
            if step_ms.shape[0] else None,
            "parity": parity,
            "paired_mode": paired,
            "exact_ties": exact_ties if args.workload == "windows" else None,
            "sensitivity": sensitivity,
            "device_memory": device_memory(),
        }
        if others:
            out["other_workloads_pending"] = True
        out["config"]["n_points"] = int(max(args.points, 6000) if args.workload == "windows" else args.points)
        out["config"]["noise_frac"] = 0.10
        if parity:
            out["argmax_agree_frac"] = parity["argmax_agree_frac"]
            out["near_tie_frac"] = parity["near_tie_frac"]
            # the PARITY-COMPLETE figure: the same step followed by the exact tie resolver, after which the index map IS the
            # oracle's on every pixel (`value` is the DSI build + fusion + arg-max, BASELINE's metric; this stands beside it)
            rz = parity.get("exact_tie_resolver") or {}
            if rz.get("elapsed_ms") is not None:
                ms_exact = ms_per_step + float(rz["elapsed_ms"])
                out["value_exact"] = voted_all / (ms_exact * 1e-3) / 1e6
                out["ms_per_step_exact"] = ms_exact
                out["value_exact_note"] = ("events / (step + dsi_mapper_resolve_near_ties %.3f ms in a stream of calls): depth map "
                                           "equal to the CPU oracle's on every pixel (parity.index_map_equals_oracle = %s)"
                                           % (float(rz["elapsed_ms"]), parity.get("index_map_equals_oracle")))
            pf = parity.get("proof") or {}
            if pf.get("every_column_settled"):
                ms_proven = ms_per_step + float(pf["resolver_ms_proven"])
                out["value_proven"] = voted_all / (ms_proven * 1e-3) / 1e6
                out["ms_per_step_proven"] = ms_proven
                out["value_proven_note"] = ("events / (step + the resolver with the gap under which dsi_mapper_prove_near_ties proves "
                                            "EVERY column, %.3f ms): the index map is the reference's by a per-call proof "
                                            "(parity.proof; the proof pass itself is a verification pass and is not in this time)"
                                            % float(pf["resolver_ms_proven"]))
        if streams:
            out["dsi_fuse_shape"] = streams.get("grid")
            out["dsi_fuse_resident_in"] = streams.get("resident_in")
            out["dsi_fuse_at_metric_shape"] = streams.get("dsi_fuse_at_metric_shape")
        if args.workload == "windows" and exact_ties:
            out["ms_per_window_exact"] = ms_per_step + exact_ties["plus_ms_per_window"]
        if args.workload == "windows":
            out["windows_per_s"] = world * args.steps / elapsed
            out["x_real_time"] = (args.events / 10.0e6) * world * args.steps / elapsed
    for o in closers:
        o.close()
    if args.workload == "stereo" and world > 1:
        temporal.close()
        mapper_fused.close()
        ctx_side.close()
    if comm is not None:
        comm.close()
    ctx.close()
    D.close()
    if rank == 0 and out is not None and out.get("other_workloads_pending"):
        out.pop("other_workloads_pending")
        out["other_workloads"] = other_workloads()
    if rank == 0:
        # RCCL prints a version banner through C stdio; push it out first so that the JSON
        # line is the last line on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out))
        sys.stdout.flush()


def cpp_stream_line():
    """The same stream through the C++ adapter (dsi::full_sequence_depth_maps, include/dsi_process.hpp) fed from
    std::vector<dsi::Event> -- 24-byte structs in pageable memory, like the reference holds its events
    (main.cpp:177-302): tools/window_stream_bench in its own process, after this one's work."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "window_stream_bench")
    if not os.path.exists(exe):
        return {"error": "tools/window_stream_bench is not built (__graft_entry__.build())"}
    try:
        r = subprocess.run([exe, "4.0"], capture_output=True, text=True, timeout=300)
        rows = [json.loads(ln[5:]) for ln in r.stdout.splitlines() if ln.startswith("JSON ")]
        if not rows:
            return {"error": (r.stdout + r.stderr)[-300:]}
        return {"by_depth": rows, "source": "std::vector<dsi::Event> (24-byte structs, pageable), host threads turn a window into the "
                "engine's arrays in page-locked staging; 2 x 500 k events per 50 ms window, 512x512x200; steady state"}
    except Exception as e:
        return {"error": str(e)[:300]}


def device_memory():
    """Bytes of HBM in use on the current device at the end of the timed region (hipMemGetInfo: everything the engine
    holds -- the allocations only grow --, plus the runtime's own few hundred MB)."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        free_b, total_b = ctypes.c_size_t(0), ctypes.c_size_t(0)
        if hip.hipMemGetInfo(ctypes.byref(free_b), ctypes.byref(total_b)) != 0:
            return None
        return {"used_GB": (total_b.value - free_b.value) / 1e9, "total_GB": total_b.value / 1e9,
                "source": "hipMemGetInfo after the timed region (grow-only allocations: the high-water mark)"}
    except OSError:
        return None


def info_nz(mapper):
    return mapper.dsi_.getDimensions()[2]


if __name__ == "__main__":
    main()
