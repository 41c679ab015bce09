#!/usr/bin/env python
"""bench.py -- DSI build + fuse + arg-max throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one stereo batch that is already resident
in HBM:  for each of the 2 cameras  evaluateDSI past the pose lookup (per-packet
homography, z0 warp, reset, voting; mapper_emvs_stereo.cpp:108-146), then camera fusion
(harmonic mean, process1.cpp:126-141), then arg-max + depth (cartesian3dgrid.cpp:115-137,
mapper_emvs_stereo.cpp:302-313).  At N = 1 this is BASELINE.json configs[1]:
"Stereo (2-cam) DSEC, 10 M events/cam, 346x260x100 DSI, harmonic fusion, 1xMI355X".

N > 1 (one process per GPU, launched by torch.distributed.run): every rank owns an
independent time slice of the same size (weak scaling, configs[3] shape), builds and
camera-fuses its DSI, then the slices are fused across time with the reference's
harmonic accumulator (process2.cpp:217-226): local 1/(0.01+v), ONE RCCL all-reduce(sum)
of the 36 MB volume over xGMI, local n/acc, arg-max -- issued on a second HIP stream so that it
overlaps the next step's voting (every step's fusion is complete before the closing barrier).
value = events voted by all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).  Data: synthetic (dvs_mcemvs_amd/synthetic.py).
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

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--events", type=int, default=10_000_000, help="events per camera per GPU")
    ap.add_argument("--dims", type=int, nargs=3, default=[346, 260, 100])
    ap.add_argument("--algo", type=int, default=0, help="0 auto, 1 global atomics, 2 LDS bands")
    ap.add_argument("--band", type=int, nargs=3, default=[0, 0, 0], help="band_rows chunks block")
    ap.add_argument("--points", type=int, default=5000, help="scene points of the synthetic rig (SURVEY 8d: 2000-20000)")
    ap.add_argument("--packed", type=int, default=-1, help="-1 auto (= 1), 0 per-packet waves, 1 packed lanes (hand-scheduled), 2 packet groups, 3 packed (compiled loop), 4 packet groups (hand-scheduled), 5 packed with vector fill")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000,
                    help="events of camera 0 the CPU oracle is timed on (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="2: each camera's mapper on its own HIP stream (context), meeting at the fusion")
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    use_dist = world > 1 or os.environ.get("DSI_BENCH_FORCE_DIST") == "1"  # the env knob lets a
    # 1-GPU box exercise the torch.distributed/RCCL code path (world_size 1)
    if use_dist:
        # torch first: libdsi_engine.so then binds to the HIP runtime torch already loaded,
        # so that RCCL (torch.distributed "nccl") and the engine share one runtime.
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world),
              file=sys.stderr)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    import dvs_mcemvs_amd as d
    from dvs_mcemvs_amd import synthetic as syn

    nx, ny, nz = args.dims
    ctx = d.Context(local_rank)

    # ---- inputs: one stereo time slice per rank, generated and uploaded before timing ----
    t_gen = time.time()
    rig = syn.stereo_rig(args.events, width=nx, height=ny, t0=10.0 + 0.5 * rank, duration=0.5,
                         seed=1234 + 100 * rank, n_points=args.points)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)  # cfg/DSEC/zurich_04_a_full/dsec.conf:11-12,17
    mappers, batches, voted = [], [], 0
    ctx_cam1 = d.Context(local_rank) if args.streams == 2 else ctx
    cam_ctx = [ctx, ctx_cam1]
    for c in range(2):
        m = d.MapperEMVS(cam_ctx[c], rig["cam"], shape)
        m.set_vote_algo(args.algo)
        m.set_band_params(*args.band)
        m.set_packed_lanes(args.packed)
        first, Rt = d.packetize(rig["events"][c][2], rig["trajectories"][c], rig["T_rv_w"])
        batches.append(d.EventBatch(cam_ctx[c], rig["events"][c][0], rig["events"][c][1], Rt, first))
        voted += first.shape[0] * d.PACKET_SIZE
        mappers.append(m)
    fused = d.Grid3D(ctx, nx, ny, nz)
    t_gen = time.time() - t_gen

    temporal = None
    if use_dist:
        # temporal fusion across ranks, pipelined: round k's all-reduce + finalize + arg-max run on
        # a second HIP stream (its own context + the reference's "mapper_fused") while the main
        # stream already votes round k+1
        from dvs_mcemvs_amd import distributed as dd
        ctx_side = d.Context(local_rank)
        mapper_fused = d.MapperEMVS(ctx_side, rig["cam"], shape)
        temporal = dd.PipelinedTemporalFusion.on_gpu(ctx, ctx_side, (nx, ny, nz), d.ACC_INV_SUM, world,
                                                     extract=mapper_fused.computeDepthMap)
        # one un-timed round now: any problem with the two-stream / RCCL set-up shows up here, on
        # every rank, before the warm-up
        fused.resetGrid()
        temporal.submit(fused)
        temporal.drain()

    def step():
        if ctx_cam1 is not ctx:
            ctx_cam1.wait_for(ctx)        # the previous step's fusion has read camera 1's DSI
        for c in range(2):
            mappers[c].evaluateDSI_batch(batches[c])
        if ctx_cam1 is not ctx:
            ctx.wait_for(ctx_cam1)        # fusion after both cameras
        # process1.cpp:126-141 (resetGrid; addTwoGrids(dsi0); harmonicMeanTwoGrids(dsi1)) in one pass
        fused.setToFusionOf(mappers[0].dsi_, mappers[1].dsi_, d.FUSE_HM)
        if temporal is None:
            mappers[0].computeDepthMap(fused)
        else:
            # process2.cpp:220: acc += 1/(0.01 + fused); ONE RCCL all-reduce(sum) over xGMI; n/acc; arg-max
            temporal.submit(fused)

    def barrier():
        ctx.synchronize()
        ctx_cam1.synchronize()
        if temporal is not None:
            temporal.drain()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    for m in mappers:
        m.set_kernel_timing(True)
        m.vote_kernel_time()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    gpu_ms = ctx.timer_stop()
    barrier()
    elapsed = time.perf_counter() - t0
    kt_ms, kt_n = 0.0, 0
    for m in mappers:
        ms, n = m.vote_kernel_time()
        kt_ms += ms
        kt_n += n
        m.set_kernel_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        v = torch.tensor([float(voted)], dtype=torch.float64, device="cuda")
        dist.all_reduce(v)
        voted_all = float(v.item())
    else:
        voted_all = float(voted)

    # ---- fusion + arg-max kernels on their own (DSI-fuse GB/s, 12 B/voxel algorithmic) ----
    reps = 50
    ctx.timer_start()
    for _ in range(reps):
        fused.harmonicMeanTwoGrids(mappers[1].dsi_)
    fuse_ms = ctx.timer_stop() / reps
    ctx.timer_start()
    for _ in range(reps):
        mappers[0].computeDepthMap(fused)
    argmax_ms = ctx.timer_stop() / reps
    nvox = nx * ny * nz
    fuse_gbps = 12.0 * nvox / (fuse_ms * 1e-3) / 1e9
    argmax_gbps = (4.0 * nz + 9.0) * nx * ny / (argmax_ms * 1e-3) / 1e9

    # ---- host-buffer (PCIe-inclusive) rate, reported beside `value`, never as it: upload the raw
    # events + poses of camera 0 from pageable host memory, evaluate, wait
    h2d_rate = h2d_stereo_rate = None
    if rank == 0:
        ev0 = rig["events"][0]
        first0, Rt0 = d.packetize(ev0[2], rig["trajectories"][0], rig["T_rv_w"])
        best = float("inf")
        for _ in range(3):
            t1 = time.perf_counter()
            bt = d.EventBatch(ctx, ev0[0], ev0[1], Rt0, first0)
            mappers[0].evaluateDSI_batch(bt)
            ctx.synchronize()
            best = min(best, time.perf_counter() - t1)
            bt.close()
        h2d_rate = first0.shape[0] * d.PACKET_SIZE / best / 1e6
        # the whole stereo step from host memory: camera 1's upload overlaps camera 0's voting
        # (evaluate returns as soon as its host buffers are consumed), then fusion + arg-max
        pk = [d.packetize(rig["events"][c][2], rig["trajectories"][c], rig["T_rv_w"]) for c in range(2)]
        best2 = float("inf")
        for _ in range(3):
            t1 = time.perf_counter()
            bts = []
            for c in range(2):
                bts.append(d.EventBatch(cam_ctx[c], rig["events"][c][0], rig["events"][c][1], pk[c][1], pk[c][0]))
                mappers[c].evaluateDSI_batch(bts[-1])
            if ctx_cam1 is not ctx:
                ctx.wait_for(ctx_cam1)
            fused.setToFusionOf(mappers[0].dsi_, mappers[1].dsi_, d.FUSE_HM)
            mappers[0].computeDepthMap(fused)
            ctx.synchronize()
            best2 = min(best2, time.perf_counter() - t1)
            for b in bts:
                b.close()
        h2d_stereo_rate = sum(p_[0].shape[0] for p_ in pk) * d.PACKET_SIZE / best2 / 1e6

    info = mappers[0].last_vote_info()
    ms_per_step = 1e3 * elapsed / args.steps
    value = voted_all * args.steps / elapsed / 1e6  # Mevents/s, whole job

    # ---- roofline of the dominant kernel (the voting kernel) ----
    # algorithmic bytes per event = Nz * 4 voxels * 8 B (fp32 read+write) + 8 B (x0,y0)
    # (SURVEY.md 8d); one launch votes one camera's events of this rank.
    bytes_per_event = 32.0 * nz + 8.0
    ev_per_launch = voted / 2.0
    kern_ms = kt_ms / max(1, kt_n)
    achieved = bytes_per_event * ev_per_launch / (kern_ms * 1e-3) / 1e9 if kt_n else None
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                "kernel": {0: "k_vote_bands", 1: "k_vote_bands_packed", 2: "k_vote_groups",
                           3: "k_vote_bands_packed", 4: "k_vote_groups", 5: "k_vote_bands_packed", 6: "k_vote_bands_packed"}[info["packed"]] if info["algo"] == 2 else "k_vote_global",
                "kernel_avg_ms": kern_ms, "kernel_launches": kt_n,
                "algorithmic_bytes_per_launch": bytes_per_event * ev_per_launch,
                "kernel_Mevents_per_s": ev_per_launch / (kern_ms * 1e-3) / 1e6 if kt_n else None}

    # ---- what actually bounds the voting kernel: 64-bit LDS atomic adds (4 per accepted event-plane;
    # the sum of a DSI = accepted event-planes because the 4 bilinear weights of a vote sum to 1).
    # Rates per wave instruction measured with tools/lds_atomic_bench2.hip on this chip (16 waves/CU
    # issuing back to back): 6.2 clk conflict-free, 11.2 clk with random cells of a band.
    accepted = float(np.sum(mappers[0].dsi_.download(), dtype=np.float64))
    adds = 4.0 * accepted / (kern_ms * 1e-3) if kt_n else None
    cu_lane_rate = 256 * 64 * 2.4e9
    lds_atomics = {"adds_per_s": adds, "unit": "64-bit LDS atomic adds/s (one camera launch)",
                   "accepted_event_planes_per_launch": accepted,
                   "peak_conflict_free": cu_lane_rate / 6.2, "rate_random_cells": cu_lane_rate / 11.2,
                   "frac_of_conflict_free_peak": adds / (cu_lane_rate / 6.2) if adds else None,
                   "frac_of_random_cell_rate": adds / (cu_lane_rate / 11.2) if adds else None}

    # ---- CPU baseline: the oracle (a port of the reference's CPU path) on a bounded sample ----
    cpu = None
    if rank == 0 and not args.no_cpu and args.cpu_sample >= 2048:
        from oracle import oracle as orc
        from oracle_pipeline import OracleMapper
        n_s = min(args.cpu_sample, rig["events"][0][0].shape[0])
        x, y, ts = (a[:n_s] for a in rig["events"][0])
        r = OracleMapper(rig["cam"], dimZ=nz, min_depth=4.0, max_depth=200.0)
        first, Rt = d.packetize(ts, rig["trajectories"][0], rig["T_rv_w"])
        first = first.astype(np.int64)
        r.evaluate_packets(x, y, first[:64], Rt[:64])  # warm-up (page in, spin up OpenMP)
        tc = float("inf")
        for _ in range(3):                            # best of 3
            t1 = time.perf_counter()
            r.evaluate_packets(x, y, first, Rt)       # stage A + reset + fillVoxelGrid
            tc = min(tc, time.perf_counter() - t1)
        # one thread on a 1/16 sample (SURVEY 8d asks for both)
        n1 = max(64, first.shape[0] // 16)
        all_threads = orc.num_threads()
        orc.set_num_threads(1)
        t1 = time.perf_counter()
        r.evaluate_packets(x, y, first[:n1], Rt[:n1])
        t_one = time.perf_counter() - t1
        orc.set_num_threads(all_threads)
        cpu = {"value": first.shape[0] * 1024 / tc / 1e6, "unit": "Mevents/s",
               "one_thread_value": n1 * 1024 / t_one / 1e6,
               "cores": min(orc.num_threads(), nz), "kind": "port",  # OpenMP over planes: at most nz threads work
               "sample": "camera 0, first %d events (%d packets) of the same workload, %dx%dx%d DSI; "
                         "oracle stage A + fillVoxelGrid, OpenMP over planes (reference strategy: at most dimZ threads busy), -O3 no -march=native; best of 3, %.2f s wall"
                         % (n_s, first.shape[0], nx, ny, nz, tc)}

    if rank == 0:
        out = {
            "metric": "Mevents/s into DSI (346x260x100) + DSI-fuse GB/s",
            "value": value, "unit": "Mevents/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "stereo (2-cam) synthetic DSEC-like rig, %d events/cam per GPU, %dx%dx%d DSI, "
                            "harmonic camera fusion + arg-max%s" % (
                                args.events, nx, ny, nz,
                                "" if world == 1 else
                                ", %d time slices (one per GPU) fused by RCCL all-reduce of inverse sums" % world),
                "events_voted_per_step": voted_all, "vote_algo": info["algo"], "bands": info["bands"],
                "band_rows": info["band_rows"], "chunks": info["chunks"],
                "block_threads": info["block_threads"], "lds_bytes": info["lds_bytes"], "packed_lanes": info["packed"],
                "parallelism": "1 GPU" if world == 1 else "time-slice x%d" % world},
            "dsi_fuse_GBps": fuse_gbps, "dsi_fuse_ms": fuse_ms, "dsi_fuse_frac_of_hbm_peak": fuse_gbps / HBM_PEAK_GBPS,
            "argmax_GBps": argmax_gbps, "argmax_ms": argmax_ms,
            "gpu_ms_per_step_hip_events": gpu_ms / args.steps,
            "h2d_inclusive_Mevents_per_s": h2d_rate, "h2d_inclusive_stereo_step_Mevents_per_s": h2d_stereo_rate,
            "roofline": roofline, "lds_atomics": lds_atomics, "cpu_baseline": cpu, "input_gen_s": t_gen,
        }
    for o in mappers + batches + [fused]:
        o.close()
    if temporal is not None:
        temporal.close()
        mapper_fused.close()
        ctx_side.close()
    if ctx_cam1 is not ctx:
        ctx_cam1.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
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


if __name__ == "__main__":
    main()
