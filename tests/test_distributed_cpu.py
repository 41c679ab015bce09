"""world_size-2 gloo test (CPU) of the multi-GPU orchestration: slice partitioning +
the single all-reduce(sum) that implements the reference's temporal fusion
(process2.cpp:211-242).  Per-slice DSIs come from the CPU oracle here (tests may use it);
on GPUs they come from the engine and the tensor aliases a Grid3D."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from dvs_mcemvs_amd import distributed as dd, synthetic as syn
    from oracle import oracle as orc
    from oracle_pipeline import OracleMapper

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rig = syn.stereo_rig(9000, width=40, height=30, duration=0.3, seed=17)
    x, y, ts = rig["events"][0]
    n_slices = 4
    bounds = dd.subinterval_bounds(x.shape[0], n_slices)
    acc = np.zeros((8, 30, 40), np.float32)
    for k in dd.slices_of_rank(n_slices, world, rank):
        a, b = bounds[k]
        m = OracleMapper(rig["cam"], dimZ=8, min_depth=4.0, max_depth=100.0)
        assert m.evaluateDSI((x[a:b], y[a:b], ts[a:b]), rig["trajectories"][0], rig["T_rv_w"])
        acc = orc.accumulate(acc, m.dsi, mode)
    t = torch.from_numpy(acc)
    dd.allreduce_sum_(t)
    fused = orc.finalize(t.numpy(), mode, n_slices)
    tmax = dd.allreduce_max_scalar(1.0 + rank)
    assert tmax == float(world)
    np.save(os.path.join(out_dir, "fused_rank%d.npy" % rank), fused)
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", [0, 1])
def test_two_rank_temporal_fusion_equals_single_process(tmp_path, mode):
    import torch.multiprocessing as mp
    from dvs_mcemvs_amd import distributed as dd, synthetic as syn
    from oracle import oracle as orc
    from oracle_pipeline import OracleMapper

    port = _free_port()
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    # single-process reference: all 4 slices in order (process2.cpp:98-242)
    rig = syn.stereo_rig(9000, width=40, height=30, duration=0.3, seed=17)
    x, y, ts = rig["events"][0]
    acc = np.zeros((8, 30, 40), np.float32)
    for a, b in dd.subinterval_bounds(x.shape[0], 4):
        m = OracleMapper(rig["cam"], dimZ=8, min_depth=4.0, max_depth=100.0)
        assert m.evaluateDSI((x[a:b], y[a:b], ts[a:b]), rig["trajectories"][0], rig["T_rv_w"])
        acc = orc.accumulate(acc, m.dsi, mode)
    ref = orc.finalize(acc, mode, 4)
    r0 = np.load(tmp_path / "fused_rank0.npy")
    r1 = np.load(tmp_path / "fused_rank1.npy")
    assert np.array_equal(r0, r1)                   # every rank holds the same fused DSI
    # summation order differs (slices 0,2 | 1,3 vs 0,1,2,3): equal to rounding
    assert np.allclose(r0, ref, rtol=1e-6, atol=1e-6)
    assert r0.any()


def test_partitioning():
    from dvs_mcemvs_amd import distributed as dd
    assert dd.subinterval_bounds(10, 4) == [(0, 2), (2, 4), (4, 6), (6, 8)]  # remainder dropped
    assert dd.slices_of_rank(8, 8, 3) == [3]
    assert dd.slices_of_rank(8, 2, 1) == [1, 3, 5, 7]
    owned = sorted(k for r in range(3) for k in dd.slices_of_rank(7, 3, r))
    assert owned == list(range(7))


class _NumpyGrid:
    """Stand-in for an engine Grid3D that aliases a torch CPU tensor (oracle arithmetic)."""

    def __init__(self, tensor):
        self.a = tensor.numpy()

    def resetGrid(self):
        self.a[...] = 0

    def addInverseOfTwoGrids(self, g):
        from oracle import oracle as orc
        self.a[...] = orc.accumulate(self.a.copy(), g, 1)

    def addTwoGrids(self, g):
        from oracle import oracle as orc
        self.a[...] = orc.accumulate(self.a.copy(), g, 0)

    def computeHMfromSumOfInv(self, n):
        from oracle import oracle as orc
        self.a[...] = orc.finalize(self.a.copy(), 1, n)

    def computeAMfromSum(self, n):
        from oracle import oracle as orc
        self.a[...] = orc.finalize(self.a.copy(), 0, n)


def _pipelined_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from dvs_mcemvs_amd import distributed as dd

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rng = np.random.default_rng(5)                      # same stream on every rank
    rounds = [rng.uniform(0, 3, (world, 4, 6, 5)).astype(np.float32) for _ in range(5)]
    slots = []
    for _ in range(2):
        t = torch.zeros((4, 6, 5), dtype=torch.float32)
        g = _NumpyGrid(t)
        slots.append({"tensor": t, "acc_main": g, "acc_side": g})
    seen = []
    pipe = dd.PipelinedTemporalFusion(slots, 1, world, streams=None,
                                      extract=lambda grid: seen.append(grid.a.copy()))
    for r in rounds:
        pipe.submit(r[rank])                            # this rank's slice of the round
    pipe.drain()
    assert len(seen) == 5 and pipe.k == 5
    np.save(os.path.join(out_dir, "pipe_rank%d.npy" % rank), np.stack(seen))
    dist.destroy_process_group()


def test_pipelined_temporal_fusion_rounds(tmp_path):
    """Five fusion rounds through the double-buffered PipelinedTemporalFusion (program-order mode
    on CPU): every round equals the single-process harmonic fusion of that round's slices."""
    import torch.multiprocessing as mp
    from oracle import oracle as orc

    port = _free_port()
    mp.spawn(_pipelined_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "pipe_rank0.npy")
    r1 = np.load(tmp_path / "pipe_rank1.npy")
    assert np.array_equal(r0, r1)
    rng = np.random.default_rng(5)
    for k in range(5):
        slices = rng.uniform(0, 3, (2, 4, 6, 5)).astype(np.float32)
        acc = np.zeros((4, 6, 5), np.float32)
        for s in slices:
            acc = orc.accumulate(acc, s, 1)
        assert np.allclose(r0[k], orc.finalize(acc, 1, 2), rtol=1e-6, atol=1e-7)


def test_plane_ranges_and_argmax_keys():
    from dvs_mcemvs_amd import distributed as dd
    from oracle import oracle as orc
    assert dd.plane_ranges(256, 8) == [(32 * r, 32) for r in range(8)]
    assert dd.plane_ranges(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    rng = np.random.default_rng(9)
    dsi = rng.integers(0, 4, (37, 9, 11)).astype(np.float32)    # many ties -> "first maximum wins" matters
    dsi[:, 0, 0] = 0.0                                            # all-zero column -> index 0
    dsi[5, 1, 1] = dsi[30, 1, 1] = 99.0                           # equal maxima in different shards
    conf, idx = orc.collapse_max_z(dsi)
    keys = None
    for b, c in dd.plane_ranges(37, 5):
        cl, il = orc.collapse_max_z(dsi[b:b + c])
        k = dd.pack_argmax_keys(cl, il, b)
        keys = k if keys is None else np.maximum(keys, k)
    c2, i2 = dd.unpack_argmax_keys(keys)
    assert np.array_equal(c2, conf) and np.array_equal(i2, idx) and i2[1, 1] == 5 and i2[0, 0] == 0


def _plane_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from dvs_mcemvs_amd import distributed as dd, synthetic as syn
    from oracle import oracle as orc
    from oracle_pipeline import OracleMapper

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rig = syn.stereo_rig(6000, width=40, height=30, duration=0.3, seed=23)
    nz = 13
    b, c = dd.plane_ranges(nz, world)[rank]
    fused = None
    for cam in range(2):                                 # every rank reads ALL events, owns planes [b, b+c)
        m = OracleMapper(rig["cam"], dimZ=nz, min_depth=4.0, max_depth=100.0)
        assert m.evaluateDSI(rig["events"][cam], rig["trajectories"][cam], rig["T_rv_w"])
        shard = m.dsi[b:b + c]                           # planes are independent (mapper_emvs_stereo.cpp:168)
        fused = shard.copy() if fused is None else orc.fuse2(fused, shard, 3)   # GM, voxel-wise: local
    conf_l, idx_l = orc.collapse_max_z(fused)
    conf, idx = dd.allreduce_argmax(conf_l, idx_l, b)    # the only collective
    np.savez(os.path.join(out_dir, "plane_rank%d.npz" % rank), conf=conf, idx=idx)
    dist.destroy_process_group()


def test_plane_sharded_argmax_equals_single_process(tmp_path):
    """configs[4]-style sharding (one big DSI, planes split over ranks, voxel-wise camera fusion local,
    ONE all-reduce(MAX) of packed (confidence, index) keys) gives the unsharded depth-map inputs."""
    import torch.multiprocessing as mp
    from dvs_mcemvs_amd import synthetic as syn
    from oracle import oracle as orc
    from oracle_pipeline import OracleMapper

    port = _free_port()
    mp.spawn(_plane_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    rig = syn.stereo_rig(6000, width=40, height=30, duration=0.3, seed=23)
    fused = None
    for cam in range(2):
        m = OracleMapper(rig["cam"], dimZ=13, min_depth=4.0, max_depth=100.0)
        assert m.evaluateDSI(rig["events"][cam], rig["trajectories"][cam], rig["T_rv_w"])
        fused = m.dsi.copy() if fused is None else orc.fuse2(fused, m.dsi, 3)
    conf, idx = orc.collapse_max_z(fused)
    for r in range(3):
        z = np.load(tmp_path / ("plane_rank%d.npz" % r))
        assert np.array_equal(z["conf"], conf) and np.array_equal(z["idx"], idx)
    assert conf.max() > 0
