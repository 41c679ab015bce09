"""N > 1 orchestration with the REAL engine (BASELINE configs[3] and configs[4]) on a one-GPU box,
and the engine's own RCCL communicator (C ABI) at the sizes one GPU allows.

RCCL admits one rank per device, so the 2-rank tests put both ranks on GPU 0 and carry the
collective over gloo through host memory (`distributed.host_staged_allreduce`); everything else --
slice / plane partitioning, the engine's accumulate / finalize / n-ary fusion / arg-max kernels, the
orchestration classes -- is exactly what runs over RCCL on an 8-GPU node, where only the transport
of the one collective differs (`distributed.engine_allreduce(comm)`).

  configs[3]  time slices -> ranks, per-rank camera HM, temporal HM = ONE all-reduce(sum) of
              1/(0.01+v) + local n/acc (process2.cpp:211-242, cartesian3dgrid.h:72-86)
  configs[4]  4 cameras, 1024x1024x256, n-ary GM camera fusion, plane-sharded: each rank owns a
              plane range of every camera's DSI, ONE all-reduce(MAX) of packed arg-max keys
"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    return dist


# ------------------------------------------------------------------ configs[3]: time slices
N_SLICES, EV_PER_SLICE = 8, 160_000
DIMS3 = (346, 260, 100)


def _slices_rig():
    from dvs_mcemvs_amd import synthetic as syn
    nx, ny, _ = DIMS3
    return syn.stereo_rig(N_SLICES * EV_PER_SLICE, width=nx, height=ny, duration=0.4, seed=33, n_points=3000)


def _time_slice_job(d, dd, rig, world, rank, allreduce, pipelined):
    """Slices k = rank, rank + world, ... of BOTH cameras through the engine; returns the temporally
    fused DSI, its depth map and the per-slice camera-fused checksums."""
    nx, ny, nz = DIMS3
    ctx = d.Context(0)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    mappers = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(2)]
    sub = d.Grid3D(ctx, nx, ny, nz)               # mapper_fused_subinterval.dsi_
    bounds = dd.subinterval_bounds(rig["events"][0][0].shape[0], N_SLICES)   # process2.cpp:46-47
    mine = dd.slices_of_rank(N_SLICES, world, rank)
    if pipelined:
        ctx_side = d.Context(0)
        mapper_fused = d.MapperEMVS(ctx_side, rig["cam"], shape)
        # every "round" here is one complete temporal fusion over the ranks' current slice
        tf = dd.EnginePipelinedTemporalFusion(ctx, ctx_side, DIMS3, d.ACC_INV_SUM, world, allreduce,
                                              extract=mapper_fused.computeDepthMap)
    else:
        tf = dd.EngineTemporalFusion(ctx, DIMS3, d.ACC_INV_SUM, N_SLICES, allreduce)
    outs = []
    for k in mine:
        a, b = bounds[k]
        for c in range(2):
            x, y, ts = (arr[a:b] for arr in rig["events"][c])
            assert mappers[c].evaluateDSI((x, y, ts), rig["trajectories"][c], rig["T_rv_w"])   # :119, :146
        sub.setToFusionOf(mappers[0].dsi_, mappers[1].dsi_, d.FUSE_HM)                         # :159-189
        if pipelined:
            side = tf.submit(sub)
            tf.drain()
            outs.append((side.download(),) + mapper_fused.fetchDepthMap())
        else:
            tf.add(sub)                                                                            # :218-220
    if not pipelined:
        fused = tf.finish()                                                                        # :221-225
        depth, conf, idx = mappers[0].getDepthMapFromDSI(fused)
        outs.append((fused.download(), depth, conf, idx))
    tf.close()
    for o in mappers + [sub]:
        o.close()
    if pipelined:
        mapper_fused.close()
        ctx_side.close()
    ctx.close()
    return outs


def _worker_slices(rank, world, port, out_dir, pipelined):
    dist = _init(rank, world, port)
    import dvs_mcemvs_amd as d
    from dvs_mcemvs_amd import distributed as dd
    rig = _slices_rig()
    outs = _time_slice_job(d, dd, rig, world, rank, dd.host_staged_allreduce(), pipelined)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
             **{"%s%d" % (n, i): o[j] for i, o in enumerate(outs) for j, n in enumerate(("dsi", "depth", "conf", "idx"))})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pipelined", [False, True])
def test_two_ranks_real_engine_time_slices_equal_single_process(ctx, tmp_path, pipelined):
    """configs[3] at N = 2: both ranks drive the real engine; result = the single-process temporal
    fusion of all 8 slices (process_2, temporal_fusion = 2)."""
    import torch.multiprocessing as mp
    import dvs_mcemvs_amd as d
    from dvs_mcemvs_amd import distributed as dd
    from oracle import oracle as orc
    mp.spawn(_worker_slices, args=(2, _free_port(), str(tmp_path), pipelined), nprocs=2, join=True)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(2)]
    rig = _slices_rig()
    nx, ny, nz = DIMS3
    if not pipelined:
        # single process, all 8 slices in order, engine kernels: the reference's loop
        ref = _time_slice_job(d, dd, rig, 1, 0, dd.engine_allreduce(None), False)[0]
        for k in range(2):
            assert np.array_equal(r[0]["dsi0"], r[k]["dsi0"])             # every rank holds the same volume
            assert np.array_equal(r[0]["idx0"], r[k]["idx0"])
        # the sum of inverses is taken in another order (0,2,4,6 | 1,3,5,7 vs 0..7): equal to rounding
        err = np.abs(r[0]["dsi0"].astype(np.float64) - ref[0]) / np.maximum(1.0, np.abs(ref[0]))
        assert err.max() <= 1e-5
        # arg-max of the rank's own volume is exact, and equals the single-process one where its top-2
        # gap exceeds the rounding difference
        conf, idx = orc.collapse_max_z(r[0]["dsi0"])
        assert np.array_equal(idx, r[0]["idx0"]) and np.array_equal(conf, r[0]["conf0"])
        srt = np.sort(ref[0], axis=0)
        safe = (srt[-1] - srt[-2]) > 1e-4 * np.maximum(1.0, srt[-1])
        assert safe.mean() > 0.5 and np.array_equal(ref[3][safe], r[0]["idx0"][safe])
        assert ref[0].max() > 1.0
    else:
        # 4 rounds; round i fuses slice 2i (rank 0) with slice 2i+1 (rank 1): n = 2 temporal HM
        ctx1 = d.Context(0)
        shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
        mappers = [d.MapperEMVS(ctx1, rig["cam"], shape) for _ in range(2)]
        bounds = dd.subinterval_bounds(rig["events"][0][0].shape[0], N_SLICES)
        subs = []
        for a, b in bounds:
            for c in range(2):
                assert mappers[c].evaluateDSI(tuple(arr[a:b] for arr in rig["events"][c]), rig["trajectories"][c],
                                              rig["T_rv_w"])
            g = d.Grid3D(ctx1, nx, ny, nz)
            g.setToFusionOf(mappers[0].dsi_, mappers[1].dsi_, d.FUSE_HM)
            subs.append(g.download())
            g.close()
        for i in range(N_SLICES // 2):
            acc = orc.accumulate(orc.accumulate(np.zeros_like(subs[0]), subs[2 * i], 1), subs[2 * i + 1], 1)
            want = orc.finalize(acc, 1, 2)
            for k in range(2):
                assert np.array_equal(r[k]["dsi%d" % i], want), "round %d rank %d" % (i, k)   # a + b commutes: bit-equal
                conf, idx = orc.collapse_max_z(want)
                assert np.array_equal(r[k]["idx%d" % i], idx) and np.array_equal(r[k]["conf%d" % i], conf)
        for o in mappers + [ctx1]:
            o.close()


# ------------------------------------------------------------------ configs[4]: 4 cameras, plane shards
DIMS4 = (1024, 1024, 256)
EV4 = 120_000


def _rig4():
    from dvs_mcemvs_amd import synthetic as syn
    return syn.stereo_rig(EV4, width=DIMS4[0], height=DIMS4[1], duration=0.2, seed=51, n_cams=4, n_points=4000)


def _worker_planes(rank, world, port, out_dir):
    dist = _init(rank, world, port)
    import dvs_mcemvs_amd as d
    from dvs_mcemvs_amd import distributed as dd
    rig = _rig4()
    nx, ny, nz = DIMS4
    ctx = d.Context(0)
    begin, count = dd.plane_ranges(nz, world)[rank]
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    mappers = [d.MapperEMVS(ctx, rig["cam"], shape, plane_range=(begin, count)) for _ in range(4)]
    for c in range(4):        # every rank votes ALL events of every camera into its plane range
        assert mappers[c].evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
    fused = d.Grid3D(ctx, nx, ny, count)
    fused.setToFusionOfN([m.dsi_ for m in mappers], d.ACC_LOG_SUM)      # n-ary GM, local: voxel-wise
    depth, conf, gidx = dd.plane_sharded_depth_map(mappers[0], fused)    # ONE all-reduce(MAX) (gloo here)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), depth=depth, conf=conf, idx=gidx,
             shard=fused.download()[:: max(1, count // 8)], begin=begin, count=count,
             info=np.array([mappers[0].last_vote_info()["packed"], mappers[0].last_vote_info()["bands"]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_real_engine_four_cameras_plane_sharded_gm(ctx, tmp_path):
    """configs[4] at N = 2 on one GPU (the shape and the whole call sequence; 120 k events per
    camera instead of 100 M): against the CPU oracle's unsharded result."""
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    from oracle_pipeline import OracleMapper
    mp.spawn(_worker_planes, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(2)]
    rig = _rig4()
    nx, ny, nz = DIMS4
    dsis = []
    for c in range(4):
        m = OracleMapper(rig["cam"], dimZ=nz, min_depth=4.0, max_depth=200.0)
        assert m.evaluateDSI(rig["events"][c], rig["trajectories"][c], rig["T_rv_w"])
        dsis.append(m.dsi)
    gm = orc.fuse_nary(dsis, 2)                       # exp(mean(log)), 0 if any camera has 0
    assert gm.max() > 0
    for k in range(2):                                # the shards the ranks fused (a sample of planes)
        b, cnt = int(r[k]["begin"]), int(r[k]["count"])
        want = gm[b:b + cnt][:: max(1, cnt // 8)]
        err = np.abs(r[k]["shard"].astype(np.float64) - want) / np.maximum(1.0, np.abs(want))
        assert err.max() <= 1e-4
        assert int(r[k]["info"][0]) == 5              # the wide-grid lane mapping ran
    assert np.array_equal(r[0]["idx"], r[1]["idx"]) and np.array_equal(r[0]["conf"], r[1]["conf"])
    assert np.array_equal(r[0]["depth"], r[1]["depth"])
    conf, idx = orc.collapse_max_z(gm)
    srt = np.sort(gm, axis=0)
    # indices agree wherever the oracle's top-2 gap exceeds the DSI tolerance (GM of 4 values each
    # within 1e-4: relative 1e-4 as well); all-zero columns (conf 0) give index 0 on both sides
    safe = ((srt[-1] - srt[-2]) > 4e-4 * np.maximum(1.0, srt[-1])) | (srt[-1] == 0)
    assert safe.mean() > 0.9
    assert np.array_equal(r[0]["idx"][safe], idx[safe])
    assert np.allclose(r[0]["conf"][safe], conf[safe], rtol=2e-4, atol=2e-4)
    planes = orc.depth_planes(4.0, 200.0, nz)
    assert np.array_equal(r[0]["depth"], planes[r[0]["idx"]])


# ------------------------------------------------------------------ reduce-scatter form, the RCCL path's own code
DIMS_RS = (128, 96, 21)      # 21 planes over 2 ranks: q = 10 and ONE remainder plane, owned by both


def _worker_scattered(rank, world, port, out_dir):
    dist = _init(rank, world, port)
    import dvs_mcemvs_amd as d
    from dvs_mcemvs_amd import distributed as dd, synthetic as syn
    nx, ny, nz = DIMS_RS
    rig = syn.stereo_rig(world * 60_000, width=nx, height=ny, duration=0.2, seed=61, n_points=900)
    per = rig["events"][0][0].shape[0] // world
    ctx = d.Context(0)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    ms = [d.MapperEMVS(ctx, rig["cam"], shape) for _ in range(2)]
    for c in range(2):
        ev = tuple(a[rank * per:(rank + 1) * per] for a in rig["events"][c])
        assert ms[c].evaluateDSI(ev, rig["trajectories"][c], rig["T_rv_w"])
    sub = d.Grid3D(ctx, nx, ny, nz)
    sub.setToFusionOf(ms[0].dsi_, ms[1].dsi_, d.FUSE_HM)
    out = {}
    for variant in ("allreduce", "reduce_scatter"):
        acc = d.Grid3D(ctx, nx, ny, nz)
        acc.accumulateBegin(d.ACC_INV_SUM)
        acc.accumulate(sub, d.ACC_INV_SUM)
        if variant == "allreduce":
            dd.host_staged_allreduce()(acc, d.REDUCE_SUM)
            acc.finalize(d.ACC_INV_SUM, world)
            ms[0].computeDepthMap(acc)
        else:
            # dsi_scatter_plan + dsi_mapper_depth_map_scattered_local + dsi_mapper_depth_map_from_keys: what
            # dsi_mapper_depth_map_reduce_scattered runs around its two nccl calls
            dd.host_staged_depth_map_reduce_scattered(ms[0], acc, world, rank, d.ACC_INV_SUM, world)
            out["acc_after"] = acc.download()
        out[variant] = ms[0].fetchDepthMap()
        acc.close()
    np.savez(os.path.join(out_dir, "rs_rank%d.npz" % rank), acc_after=out["acc_after"],
             **{"%s_%s" % (v, n): out[v][j] for v in ("allreduce", "reduce_scatter") for j, n in enumerate(("depth", "conf", "idx"))})
    for o in ms + [sub]:
        o.close()
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reduce_scattered_depth_map_runs_the_rccl_paths_code(ctx, tmp_path):
    """the reduce-scatter form of the temporal fusion at N = 2 with the real engine, both
    exchanges carried through host memory; the plane partition (remainder plane included), the local finalize /
    arg-max / key packing and the unpacking are the functions the RCCL entry point calls.  Bit-equal to
    all-reduce + finalize + arg-max (a two-term sum has one order), identical on both ranks."""
    import torch.multiprocessing as mp
    from dvs_mcemvs_amd import engine
    mp.spawn(_worker_scattered, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [np.load(tmp_path / ("rs_rank%d.npz" % k)) for k in range(2)]
    for k in range(2):
        for n in ("depth", "conf", "idx"):
            assert np.array_equal(r[k]["reduce_scatter_" + n], r[k]["allreduce_" + n]), (k, n)
            assert np.array_equal(r[k]["reduce_scatter_" + n], r[0]["reduce_scatter_" + n])
    assert r[0]["allreduce_conf"].max() > 0.5
    # only the owned planes (and the shared remainder plane) were finalised on a rank: the accumulator is consumed
    sp = [engine.scatter_plan(DIMS_RS[2], 2, k) for k in range(2)]
    assert sp[0]["tail_count"] == 1 and sp[0]["own_count"] == 10
    tb = sp[0]["tail_begin"]
    assert np.array_equal(r[0]["acc_after"][tb], r[1]["acc_after"][tb])
    b0, b1 = sp[0]["own_begin"], sp[1]["own_begin"]
    assert not np.array_equal(r[0]["acc_after"][b0:b0 + 10], r[1]["acc_after"][b0:b0 + 10])
    assert not np.array_equal(r[0]["acc_after"][b1:b1 + 10], r[1]["acc_after"][b1:b1 + 10])


def test_bench_gpus_2_is_two_rccl_ranks_or_an_error(built):
    """`python bench.py --gpus 2` with no launcher: on a box with >= 2 GPUs it starts two ranks itself and the JSON
    line proves it (rccl_ranks = ncclCommCount = 2, two distinct devices); on the one-GPU lease it exits non-zero
    with a message naming the device count -- never a one-rank number labelled 2."""
    import json
    import subprocess
    import dvs_mcemvs_amd as d
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DSI_BENCH_DEVICE")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu",
           "--no-host-fed", "--events", "2000000"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200)
    if d.device_count() < 2:
        assert r.returncode == 2, r.stderr[-2000:]
        assert "refusing to run a smaller job" in r.stderr and "1 GPU device" in r.stderr
        assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        # and two ranks forced onto ONE device (a launcher's doing) fail in RCCL, loudly, instead of falling back
        env2 = dict(env, DSI_BENCH_DEVICE="0")
        r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                             "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + cmd[1:],
                            capture_output=True, text=True, env=env2, timeout=1200)
        assert r2.returncode != 0 and "no fallback collective" in r2.stderr, r2.stderr[-3000:]
        return
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2
    assert sorted(k["rccl_device"] for k in line["ranks"]) == [0, 1] and len({k["pid"] for k in line["ranks"]}) == 2
    assert line["collective"]["avg_ms_alone"] > 0 and "spawn_ranks" in line["config"]["launched_by"]


# ------------------------------------------------------------------ the engine's RCCL communicator
def test_engine_rccl_communicator_single_rank(ctx):
    """dsi_comm_* / dsi_grid_allreduce* / dsi_mapper_depth_map_sharded* with one rank: RCCL is loaded
    by the engine, initialised both ways (ncclCommInitAll over contexts, unique id + rank), and the
    collectives run on the engine's streams.  (More ranks need more GPUs: the driver's 8-GPU run.)"""
    import dvs_mcemvs_amd as d
    from dvs_mcemvs_amd import distributed as dd
    from oracle import oracle as orc
    rng = np.random.default_rng(2)
    nx, ny, nz = 96, 64, 20
    vol = rng.gamma(2.0, 4.0, (nz, ny, nx)).astype(np.float32)
    g = d.Grid3D(ctx, nx, ny, nz)
    comms = d.Comm.create_all([ctx])
    assert comms[0].size == 1 and comms[0].rank == 0
    for op in (d.REDUCE_SUM, d.REDUCE_MIN, d.REDUCE_MAX):
        g.upload(vol)
        d.allreduce_all(comms, [g], op)
        assert np.array_equal(g.download(), vol)
    comms[0].close()
    comm = d.Comm(ctx, d.Comm.unique_id(), 1, 0)
    assert comm.query() == (1, 0, ctx.device)     # ncclCommCount / UserRank / CuDevice: what bench.py prints as proof
    g.upload(vol)
    g.allReduce(comm, d.REDUCE_SUM)
    assert np.array_equal(g.download(), vol)
    # temporal fusion through the engine communicator (what bench.py --gpus N runs)
    tf = dd.EngineTemporalFusion(ctx, (nx, ny, nz), d.ACC_INV_SUM, 3, dd.engine_allreduce(comm))
    ref = np.zeros_like(vol)
    for _ in range(3):
        tf.add(g)
        ref = orc.accumulate(ref, vol, 1)
    assert np.array_equal(tf.finish().download(), orc.finalize(ref, 1, 3))
    tf.close()
    # plane-sharded arg-max on the device: a shard of planes [5, 5+9) of a 20-plane depth vector
    cam = (nx, ny, 80.0, 80.0, 48.0, 32.0)
    m = d.MapperEMVS(ctx, cam, d.ShapeDSI(0, 0, nz, 1.0, 9.0, 0.0), plane_range=(5, 9))
    shard = d.Grid3D(ctx, nx, ny, 9)
    sv = vol[5:14].copy()
    sv[:, 3, 4] = 0.0                          # an all-zero column: confidence 0, local index 0 -> global 5
    sv[2, 7, 7] = sv[6, 7, 7] = 1000.0         # a tie: the first (smaller index) wins
    shard.upload(sv)
    depth, conf, gidx = dd.plane_sharded_depth_map(m, shard, comm=comm)
    rconf, ridx = orc.collapse_max_z(sv)
    assert np.array_equal(conf, rconf) and np.array_equal(gidx, ridx + 5)
    assert gidx[3, 4] == 5 and gidx[7, 7] == 7
    full = orc.depth_planes(1.0, 9.0, nz)
    assert np.array_equal(depth, full[gidx])
    d.depth_map_sharded_all([m], [shard], [comm])
    d2, c2, i2 = m.fetchDepthMap()
    assert np.array_equal(i2, gidx) and np.array_equal(c2, conf) and np.array_equal(d2, depth)
    # the host-key path used by the gloo tests gives the same
    d3, c3, i3 = dd.plane_sharded_depth_map(m, shard)
    assert np.array_equal(i3, gidx) and np.array_equal(c3, conf) and np.array_equal(d3, depth)
    for o in (m, shard, g, comm):
        o.close()


def test_engine_collectives_over_every_device(ctx):
    """One process, EVERY GPU of the box (dsi_device_count(): 1 on a single-GPU lease, 8 on the scaling node --
    there this is the first correctness run of RCCL with more than one rank): dsi_comm_create_all, then
      * dsi_grid_allreduce_all (sum / min / max) against numpy on integer-valued volumes (exact in any order),
      * time-slice sharding: every device builds the stereo DSI of its own slice, camera HM, temporal HM by
        all-reduce + finalize + arg-max  AND  by reduce-scatter + owned-plane finalize / arg-max + key all-reduce;
        both against the single-device loop over the same slices,
      * plane sharding: every device votes all events into its plane range, dsi_mapper_depth_map_sharded_all
        against the unsharded single-device depth map, bit for bit."""
    import dvs_mcemvs_amd as d
    from dvs_mcemvs_amd import distributed as dd, synthetic as syn
    n = d.device_count()
    assert n >= 1
    ctxs = [ctx] + [d.Context(i) for i in range(1, n)]
    comms = d.Comm.create_all(ctxs)
    assert [c.rank for c in comms] == list(range(n)) and all(c.size == n for c in comms)
    # ---- all-reduce
    rng = np.random.default_rng(5)
    nx, ny, nz = 96, 72, 21                     # 21 planes: not a multiple of 2, 4 or 8 (remainder planes)
    vols = [rng.integers(0, 1000, (nz, ny, nx)).astype(np.float32) for _ in range(n)]
    grids = [d.Grid3D(c, nx, ny, nz) for c in ctxs]
    for op, ref in ((d.REDUCE_SUM, np.sum(vols, axis=0)), (d.REDUCE_MIN, np.min(vols, axis=0)),
                    (d.REDUCE_MAX, np.max(vols, axis=0))):
        for g, v in zip(grids, vols):
            g.upload(v)
        d.allreduce_all(comms, grids, op)
        for g in grids:
            assert np.array_equal(g.download(), ref.astype(np.float32))
    # ---- time slices: slice k on device k
    rig = syn.stereo_rig(n * 40_000, width=nx, height=ny, duration=0.1 * n, seed=21, n_points=800)
    shape = d.ShapeDSI(0, 0, nz, 4.0, 200.0, 0.0)
    per = rig["events"][0][0].shape[0] // n
    T = rig["T_rv_w"]

    def slice_fused(c_, k):
        ms = [d.MapperEMVS(c_, rig["cam"], shape) for _ in range(2)]
        for cam in range(2):
            ev = tuple(a[k * per:(k + 1) * per] for a in rig["events"][cam])
            assert ms[cam].evaluateDSI(ev, rig["trajectories"][cam], T)
        f = d.Grid3D(c_, nx, ny, nz)
        f.setToFusionOf(ms[0].dsi_, ms[1].dsi_, d.FUSE_HM)
        for m_ in ms:
            m_.close()
        return f

    single = dd.EngineTemporalFusion(ctx, (nx, ny, nz), d.ACC_INV_SUM, n, dd.engine_allreduce(None))
    for k in range(n):
        f = slice_fused(ctx, k)
        single.add(f)
        f.close()
    ref_mapper = d.MapperEMVS(ctx, rig["cam"], shape)
    ref_fused = single.finish()
    ref_mapper.computeDepthMap(ref_fused)
    ref_depth, ref_conf, ref_idx = ref_mapper.fetchDepthMap()
    ref_vol = ref_fused.download()
    assert ref_conf.max() > 0.5
    for variant in ("allreduce", "reduce_scatter"):
        accs, mappers = [], []
        for r, c_ in enumerate(ctxs):
            acc = d.Grid3D(c_, nx, ny, nz)
            acc.accumulateBegin(d.ACC_INV_SUM)
            f = slice_fused(c_, r)
            acc.accumulate(f, d.ACC_INV_SUM)
            f.close()
            accs.append(acc)
            mappers.append(d.MapperEMVS(c_, rig["cam"], shape))
        if variant == "allreduce":
            d.allreduce_all(comms, accs, d.REDUCE_SUM)
            for acc, m_ in zip(accs, mappers):
                acc.finalize(d.ACC_INV_SUM, n)
                m_.computeDepthMap(acc)
        else:
            d.depth_map_reduce_scattered_all(mappers, accs, comms, d.ACC_INV_SUM, n)
        outs = [m_.fetchDepthMap() for m_ in mappers]
        for r in range(1, n):                                   # every rank has the same map
            for a_, b_ in zip(outs[r], outs[0]):
                assert np.array_equal(a_, b_)
        depth, conf, idx = outs[0]
        if n <= 2:                                              # a two-term sum has one order
            assert np.array_equal(idx, ref_idx) and np.array_equal(conf, ref_conf) and np.array_equal(depth, ref_depth)
        else:                                                   # ring order: sums differ in the last bits
            assert np.allclose(conf, ref_conf, rtol=1e-5, atol=1e-6)
            picked = np.take_along_axis(ref_vol, idx[None].astype(np.int64), axis=0)[0]
            assert np.all(picked >= ref_conf * (1 - 1e-5) - 1e-6)
        assert np.array_equal(depth, mappers[0].raw_depths_vec_[idx])
        for o in accs + mappers:
            o.close()
    # ---- plane sharding
    ranges = dd.plane_ranges(nz, n)
    ms, shards = [], []
    ev0 = rig["events"][0]
    for r, c_ in enumerate(ctxs):
        m_ = d.MapperEMVS(c_, rig["cam"], shape, plane_range=ranges[r])
        assert m_.evaluateDSI(ev0, rig["trajectories"][0], T)
        ms.append(m_)
        shards.append(m_.dsi_)
    d.depth_map_sharded_all(ms, shards, comms)
    whole = d.MapperEMVS(ctx, rig["cam"], shape)
    assert whole.evaluateDSI(ev0, rig["trajectories"][0], T)
    whole.computeDepthMap()
    want = whole.fetchDepthMap()
    for m_ in ms:
        for a_, b_ in zip(m_.fetchDepthMap(), want):
            assert np.array_equal(a_, b_)
    for o in ms + [whole, ref_mapper] + grids + comms:
        o.close()
    single.close()
    for c_ in ctxs[1:]:
        c_.close()
