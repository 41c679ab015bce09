"""One rank of the CPU multi-rank tests (tests/test_distributed_cpu.py), started by launch.spawn_ranks:
    python distributed_cpu_worker.py <job> <out_dir> [args...]
Joins the ranks with launch.Dist (gloo), runs the product's orchestration (dvs_mcemvs_amd.distributed) on stand-in
grids whose voxel arithmetic is the CPU oracle's, with the host-staged test transport as the collective."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from dvs_mcemvs_amd import distributed as dd, engine as E, launch, synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle_pipeline import OracleMapper  # noqa: E402


class NumpyGrid:
    """Stand-in for an engine Grid3D (same method names), oracle arithmetic on a numpy volume."""

    def __init__(self, nx, ny, nz):
        self.a = np.zeros((nz, ny, nx), np.float32)

    @property
    def shape(self):
        return self.a.shape

    def accumulateBegin(self, mode):
        self.a[...] = {E.ACC_MIN: np.inf, E.ACC_MAX: -np.inf}.get(int(mode), 0.0)

    def accumulate(self, g, mode):
        src = g.a if isinstance(g, NumpyGrid) else np.asarray(g, np.float32)
        self.a[...] = orc.accumulate(self.a.copy(), src, int(mode))

    def finalize(self, mode, n):
        self.a[...] = orc.finalize(self.a.copy(), int(mode), int(n))

    def download(self):
        return self.a.copy()

    def upload(self, host):
        self.a[...] = host

    def collapseMaxZSlice(self):
        return orc.collapse_max_z(self.a)

    def close(self):
        pass


class NullContext:
    def wait_for(self, other):
        pass

    def synchronize(self):
        pass


class NumpyMapper:
    """Stand-in for the three transport-free steps of MapperEMVS.computeDepthMapReduceScattered: the partition and
    the key word are the engine's (dsi_scatter_plan, dsi_argmax_keys_*), the voxel arithmetic the oracle's."""

    def computeDepthMapScatteredLocal(self, acc, nranks, rank, mode, n_maps):
        sp = E.scatter_plan(acc.shape[0], nranks, rank)
        keys = np.zeros(acc.shape[1:], np.uint64)                     # a rank that owns no plane contributes zeros
        for b, c in ((sp["own_begin"], sp["own_count"]), (sp["tail_begin"], sp["tail_count"])):
            if c <= 0:
                continue
            part = orc.finalize(acc.a[b:b + c].copy(), int(mode), int(n_maps))
            acc.a[b:b + c] = part
            conf, idx = orc.collapse_max_z(part)
            keys = np.maximum(keys, dd.pack_argmax_keys(conf, idx, b))
        self.keys = keys

    def argmaxKeys(self):
        return self.keys

    def setArgmaxKeys(self, keys):
        self.keys = keys

    def computeDepthMapFromKeys(self):
        self.conf, self.idx = dd.unpack_argmax_keys(self.keys)

    def computeDepthMapReduceScattered(self, acc, comm, mode, n_maps):
        """MapperEMVS.computeDepthMapReduceScattered with the two nccl calls replaced by the host-staged transport
        (comm = (world, rank)): what EnginePipelinedTemporalFusion(scattered=...) calls once per round."""
        world, rank = comm
        dd.host_staged_depth_map_reduce_scattered(self, acc, world, rank, mode, n_maps)
        self.rounds.append((self.conf.copy(), self.idx.copy()))


def job_temporal(D, out_dir, mode):
    rig = syn.stereo_rig(9000, width=40, height=30, duration=0.3, seed=17)
    x, y, ts = rig["events"][0]
    n_slices = 4
    bounds = dd.subinterval_bounds(x.shape[0], n_slices)
    tf = dd.EngineTemporalFusion(None, (40, 30, 8), mode, n_slices, dd.host_staged_allreduce(),
                                 make_grid=lambda: NumpyGrid(40, 30, 8))
    t0 = time.perf_counter()
    mine = dd.slices_of_rank(n_slices, D.world, D.rank)
    for k in mine:
        a, b = bounds[k]
        m = OracleMapper(rig["cam"], dimZ=8, min_depth=4.0, max_depth=100.0)
        assert m.evaluateDSI((x[a:b], y[a:b], ts[a:b]), rig["trajectories"][0], rig["T_rv_w"])
        tf.add(m.dsi)
    fused = tf.finish()
    elapsed = time.perf_counter() - t0
    np.save(os.path.join(out_dir, "fused_rank%d.npy" % D.rank), fused.a)
    # the bench's aggregation over the ranks (launch.aggregate): units summed, time = max over ranks
    each = D.gather(elapsed)
    value, elapsed_max, units = launch.aggregate(D, len(mine), elapsed, 1)
    if D.rank == 0:
        print(json.dumps({"world": D.world, "units_all_ranks": units, "elapsed_max": elapsed_max, "elapsed_each": each,
                          "value": value, "spawned": D.spawned}))


def job_pipelined(D, out_dir):
    rng = np.random.default_rng(5)                      # same stream on every rank
    rounds = [rng.uniform(0, 3, (D.world, 4, 6, 5)).astype(np.float32) for _ in range(5)]
    seen = []

    def slot():
        g = NumpyGrid(5, 6, 4)
        return (g, g)
    pipe = dd.EnginePipelinedTemporalFusion(NullContext(), NullContext(), (5, 6, 4), E.ACC_INV_SUM, D.world,
                                            dd.host_staged_allreduce(), extract=lambda g: seen.append(g.a.copy()),
                                            make_slot=slot)
    for r in rounds:
        pipe.submit(r[D.rank])                          # this rank's slice of the round
    pipe.drain()
    assert len(seen) == 5 and pipe.k == 5
    np.save(os.path.join(out_dir, "pipe_rank%d.npy" % D.rank), np.stack(seen))


def job_planes(D, out_dir):
    rig = syn.stereo_rig(6000, width=40, height=30, duration=0.3, seed=23)
    nz = 13
    b, c = dd.plane_ranges(nz, D.world)[D.rank]
    fused = None
    for cam in range(2):                                 # every rank reads ALL events, owns planes [b, b+c)
        m = OracleMapper(rig["cam"], dimZ=nz, min_depth=4.0, max_depth=100.0)
        assert m.evaluateDSI(rig["events"][cam], rig["trajectories"][cam], rig["T_rv_w"])
        shard = m.dsi[b:b + c]                           # planes are independent (mapper_emvs_stereo.cpp:168)
        fused = shard.copy() if fused is None else orc.fuse2(fused, shard, 3)   # GM, voxel-wise: local
    conf_l, idx_l = orc.collapse_max_z(fused)
    keys = dd.host_staged_allreduce_keys(dd.pack_argmax_keys(conf_l, idx_l, b))     # the only collective
    conf, idx = dd.unpack_argmax_keys(keys)
    np.savez(os.path.join(out_dir, "plane_rank%d.npz" % D.rank), conf=conf, idx=idx)


def job_scattered(D, out_dir, nz):
    rng = np.random.default_rng(31)
    slices = rng.uniform(0, 3, (D.world, nz, 6, 7)).astype(np.float32)
    slices[:, :, 0, 0] = 0.0
    acc = NumpyGrid(7, 6, nz)
    acc.accumulateBegin(E.ACC_INV_SUM)
    acc.accumulate(slices[D.rank], E.ACC_INV_SUM)
    m = NumpyMapper()
    dd.host_staged_depth_map_reduce_scattered(m, acc, D.world, D.rank, E.ACC_INV_SUM, D.world)
    np.savez(os.path.join(out_dir, "scattered_rank%d.npz" % D.rank), conf=m.conf, idx=m.idx)


def job_pipelined_scattered(D, out_dir, nz):
    """EnginePipelinedTemporalFusion with scattered= (the reduce-scatter form of the round's collective) over a stream
    of rounds: world 3 with dimZ 8 gives q = 2 owned planes per rank and 2 tail planes every rank finalizes."""
    rng = np.random.default_rng(77)                      # same stream on every rank
    rounds = [rng.uniform(0, 3, (D.world, nz, 6, 5)).astype(np.float32) for _ in range(4)]
    for r in rounds:
        r[:, :, 0, 0] = 0.0                              # an all-zero column: index 0 on every rank
    m = NumpyMapper()
    m.rounds = []

    def slot():
        g = NumpyGrid(5, 6, nz)
        return (g, g)
    pipe = dd.EnginePipelinedTemporalFusion(NullContext(), NullContext(), (5, 6, nz), E.ACC_INV_SUM, D.world, None,
                                            scattered=(m, (D.world, D.rank)), make_slot=slot)
    for r in rounds:
        pipe.submit(r[D.rank])
    pipe.drain()
    assert len(m.rounds) == 4 and pipe.k == 4
    np.savez(os.path.join(out_dir, "pipe_scattered_rank%d.npz" % D.rank), conf=np.stack([c for c, _ in m.rounds]),
             idx=np.stack([i for _, i in m.rounds]))


def main():
    job, out_dir = sys.argv[1], sys.argv[2]
    D = launch.Dist()
    if job == "temporal":
        job_temporal(D, out_dir, int(sys.argv[3]))
    elif job == "pipelined":
        job_pipelined(D, out_dir)
    elif job == "planes":
        job_planes(D, out_dir)
    elif job == "scattered":
        job_scattered(D, out_dir, int(sys.argv[3]))
    elif job == "pipelined_scattered":
        job_pipelined_scattered(D, out_dir, int(sys.argv[3]))
    else:
        raise SystemExit("unknown job %r" % job)
    D.close()


if __name__ == "__main__":
    main()
