"""Multi-GPU sharding of the DSI path: one process per GPU.

The data-path collective is the ENGINE's (include/dsi_engine.h "multi-GPU": RCCL called from the C ABI,
`Comm` / `Grid3D.allReduce` / `MapperEMVS.computeDepthMapSharded` / `...ReduceScattered`).  The orchestration
classes below take it as a callable `allreduce(grid, op)`, so that the same code runs over RCCL (production,
`engine_allreduce(comm)`), over a host-staged gloo exchange (`host_staged_allreduce()`: tests that put two
ranks on ONE GPU, which RCCL refuses, and CPU tests with stand-in grids) or over nothing (one rank).  There is
no second GPU collective backend: torch.distributed appears only as the side channel of tests and of bench.py
(rendezvous, barriers, the RCCL unique id, a few scalars) -- see launch.py.

What shards:

* time slices -- Alg. 2 of the reference (process2.cpp:98-249) splits an event stream into
  `num_subintervals` sub-intervals BY EVENT COUNT, builds one DSI per sub-interval and fuses them voxel-wise
  over time with
      HM:  acc += 1/(0.01 + dsi_k)  for every k,  then  n/acc      (cartesian3dgrid.h:72-86)
      AM:  acc += dsi_k,                          then  acc/n      (cartesian3dgrid.h:64-70, 87-93)
  Both accumulators are plain sums, so with slice k on rank k % world the whole temporal fusion is: local
  accumulate -> ONE all-reduce(sum) of the volume -> local finalize (or: reduce-scatter by planes -> finalize
  + arg-max of the owned planes -> all-reduce(MAX) of packed keys).  Events never move between GPUs.
* planes -- one DSI too big for one GPU (configs[4]): rank r owns a contiguous plane range of every camera's
  DSI, votes ALL events into it, fuses the cameras voxel-wise locally; the only exchange is one
  all-reduce(MAX) of packed (confidence, plane) keys.

The partition arithmetic (which planes / slices a rank owns, the key word) lives in the engine
(csrc/dsi_host.hpp, exported as dsi_scatter_plan / dsi_plane_range / dsi_argmax_keys_*): the functions here
call it, they do not restate it.
"""
import numpy as np

from . import engine as E


def subinterval_bounds(n_events, num_subintervals):
    """Event index ranges of process_2's sub-intervals (process2.cpp:46-47, :105-107):
    num_events_per_subinterval = n / k (integer division); the remainder is dropped."""
    per = int(n_events) // int(num_subintervals)
    return [(k * per, (k + 1) * per) for k in range(int(num_subintervals))]


def slices_of_rank(num_slices, world_size, rank):
    """Round-robin assignment slice k -> rank k % world_size."""
    return [k for k in range(int(num_slices)) if k % int(world_size) == int(rank)]


# ---------------------------------------------------------------- plane sharding (one big DSI)
def plane_ranges(num_planes, world_size):
    """Contiguous, balanced plane ranges [(begin, count)] per rank (dsi_plane_range; the reference layout
    [z][y][x] makes a plane range a contiguous slab of the volume)."""
    return [E.plane_range(num_planes, world_size, r) for r in range(int(world_size))]


def pack_argmax_keys(conf, idx_local, plane_begin):
    """(confidence, plane index) of a shard's collapseMaxZSlice -> one 64-bit word per pixel whose MAX over
    shards is the unsharded result: Grid3D::collapseMaxZSlice takes the FIRST maximum (std::max_element,
    cartesian3dgrid.cpp:115-137), i.e. the larger confidence wins and, on equal confidence, the smaller global
    plane index.  dsi_argmax_keys_pack -- the word k_pack_argmax builds on the device."""
    return E.argmax_keys_pack(conf, idx_local, plane_begin)


def unpack_argmax_keys(keys):
    return E.argmax_keys_unpack(keys)


# ------------------------------------------------------------------ the collective as a callable
def engine_allreduce(comm):
    """allreduce(grid, op) over an engine communicator (RCCL from the C ABI, on the grid's stream)."""
    def f(grid, op):
        if comm is not None and comm.size > 1:
            grid.allReduce(comm, op)
    return f


def _gloo_op(op):
    import torch.distributed as dist
    return {E.REDUCE_SUM: dist.ReduceOp.SUM, E.REDUCE_MIN: dist.ReduceOp.MIN, E.REDUCE_MAX: dist.ReduceOp.MAX}[int(op)]


def host_staged_allreduce(group=None):
    """allreduce(grid, op) through host memory and a torch.distributed (gloo) group: the TEST transport, for
    several ranks on one GPU (RCCL admits one rank per device) and for CPU stand-in grids.  `grid` needs
    download() / upload(); synchronises the grid's stream."""
    def f(grid, op):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        t = torch.from_numpy(np.ascontiguousarray(grid.download()))
        dist.all_reduce(t, op=_gloo_op(op), group=group)
        grid.upload(t.numpy())
    return f


def host_staged_allreduce_keys(keys, group=None):
    """MAX over the ranks of packed arg-max keys (uint64 words below 2^40, so int64 orders them alike)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return keys
    t = torch.from_numpy(np.ascontiguousarray(keys).view(np.int64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t.numpy().view(np.uint64)


def host_staged_reduce_scatter(acc, world, rank, op, group=None):
    """The exchange of dsi_mapper_depth_map_reduce_scattered's first step over the test transport, with the RCCL
    path's own partition (dsi_scatter_plan): afterwards planes [own_begin, +own_count) and the tail planes of
    `acc` hold the reduced values on this rank -- and, like after ncclReduceScatter in place, the other planes
    hold whatever this rank had (here: its own partial sums)."""
    import torch
    import torch.distributed as dist
    sp = E.scatter_plan(acc.shape[0], world, rank)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return sp
    mine = acc.download()
    t = torch.from_numpy(mine.copy())
    dist.all_reduce(t, op=_gloo_op(op), group=group)
    full = t.numpy()
    for b, c in ((sp["own_begin"], sp["own_count"]), (sp["tail_begin"], sp["tail_count"])):
        mine[b:b + c] = full[b:b + c]
    acc.upload(mine)
    return sp


def host_staged_depth_map_reduce_scattered(mapper, acc, world, rank, mode, n_maps, group=None):
    """MapperEMVS.computeDepthMapReduceScattered with both exchanges carried through host memory: the SAME partition
    (dsi_scatter_plan), local step (dsi_mapper_depth_map_scattered_local) and key unpacking
    (dsi_mapper_depth_map_from_keys) the RCCL path runs, so that on a multi-GPU node only the nccl* calls are new."""
    host_staged_reduce_scatter(acc, world, rank, E.acc_reduce_op(mode), group)
    mapper.computeDepthMapScatteredLocal(acc, world, rank, mode, n_maps)
    mapper.setArgmaxKeys(host_staged_allreduce_keys(mapper.argmaxKeys(), group))
    mapper.computeDepthMapFromKeys()


# ------------------------------------------------------------------ engine-native orchestration
class EngineTemporalFusion:
    """Time-slice sharding (configs[3]; process2.cpp:211-242 across GPUs): every rank accumulates the
    slices it owns, ONE all-reduce of the accumulator, local finalize.  `mode` is any
    dsi_acc_mode_t (ACC_INV_SUM = the reference's temporal HM, ACC_SUM = its temporal AM; the n-ary
    GM / RMS / min / max modes reduce the same way).  The accumulator starts at the mode's identity,
    so a rank that owns no slice contributes nothing.
    make_grid: factory of the accumulator (default: an engine Grid3D in `ctx`; CPU tests pass a stand-in)."""

    def __init__(self, ctx, dims, mode, num_slices, allreduce, make_grid=None):
        nx, ny, nz = dims
        self.acc = make_grid() if make_grid is not None else E.Grid3D(ctx, nx, ny, nz)
        self.mode, self.n, self.allreduce = int(mode), int(num_slices), allreduce
        self.op = E.acc_reduce_op(self.mode)
        self.reset()

    def reset(self):
        self.acc.accumulateBegin(self.mode)

    def add(self, dsi):
        self.acc.accumulate(dsi, self.mode)

    def finish(self):
        self.allreduce(self.acc, self.op)
        self.acc.finalize(self.mode, self.n)
        return self.acc

    def finish_depth_map(self, mapper, comm):
        """The depth map of the fused DSI on every rank WITHOUT completing the fused DSI anywhere: reduce-scatter
        by planes, local finalize + arg-max of the owned planes, one all-reduce(MAX) of packed keys
        (dsi_mapper_depth_map_reduce_scattered: half the xGMI bytes of finish() + computeDepthMap).  The
        accumulator is consumed; results via mapper.fetchDepthMap()."""
        mapper.computeDepthMapReduceScattered(self.acc, comm, self.mode, self.n)

    def close(self):
        self.acc.close()


class EnginePipelinedTemporalFusion:
    """EngineTemporalFusion for a stream of rounds (bench steps, sliding windows, main.cpp:177): round k's
    all-reduce + finalize + depth-map extraction run on the side context's stream while the main
    context already votes round k+1.  Ordering is by dsi_context_wait_for only; the host never
    blocks in submit().  The accumulator is double-buffered and aliased in both contexts.

        main stream :  ... vote, camera-fuse(k) | acc[k%2] = f(fused)      | vote, camera-fuse(k+1) ...
        side stream :                            wait | all-reduce(acc[k%2]), finalize, arg-max |

    make_slot: factory of one (main, side) pair of accumulator views (default: an engine Grid3D in ctx_main
    and an alias of its memory in ctx_side; CPU tests pass stand-ins and contexts whose wait_for is a no-op)."""

    def __init__(self, ctx_main, ctx_side, dims, mode, num_slices, allreduce, extract=None, depth=2, scattered=None,
                 make_slot=None):
        """scattered = (mapper_in_side_context, comm): the round's collective is the reduce-scatter form
        (MapperEMVS.computeDepthMapReduceScattered: reduce-scatter by planes, finalize + arg-max of the owned
        planes, all-reduce(MAX) of keys) instead of all-reduce + finalize + extract; the depth map lands in that
        mapper's buffers, the fused DSI is not completed on any rank."""
        nx, ny, nz = dims
        self.ctx_main, self.ctx_side = ctx_main, ctx_side
        self.mode, self.n, self.allreduce, self.extract = int(mode), int(num_slices), allreduce, extract
        self.scattered = scattered
        self.op = E.acc_reduce_op(self.mode)
        self.slots = []
        for _ in range(depth):
            if make_slot is not None:
                self.slots.append(make_slot())
                continue
            main = E.Grid3D(ctx_main, nx, ny, nz)
            side = E.Grid3D(ctx_side, nx, ny, nz, device_ptr=main.device_ptr)
            self.slots.append((main, side))
        self.k = 0

    def submit(self, fused):
        main, side = self.slots[self.k % len(self.slots)]
        if self.k >= len(self.slots):
            self.ctx_main.wait_for(self.ctx_side)   # the round that used this buffer has left it
        main.accumulateBegin(self.mode)
        main.accumulate(fused, self.mode)
        self.ctx_side.wait_for(self.ctx_main)
        if self.scattered is not None:
            mapper, comm = self.scattered
            mapper.computeDepthMapReduceScattered(side, comm, self.mode, self.n)
        else:
            self.allreduce(side, self.op)
            side.finalize(self.mode, self.n)
            if self.extract is not None:
                self.extract(side)
        self.k += 1
        return side

    def drain(self):
        self.ctx_main.synchronize()
        self.ctx_side.synchronize()

    def close(self):
        for main, side in self.slots:
            if side is not main:
                side.close()
            main.close()


def plane_sharded_depth_map(mapper, fused_shard, comm=None, group=None):
    """collapseMaxZSlice of a plane-sharded DSI (configs[4]).  With an engine communicator: ONE
    RCCL all-reduce(MAX) of packed keys on the device (dsi_mapper_depth_map_sharded).  Without:
    the same keys over the test transport (host-staged gloo `group`).  Returns (depth, conf, global idx) --
    identical on every rank."""
    if comm is not None:
        mapper.computeDepthMapSharded(fused_shard, comm)
        return mapper.fetchDepthMap()
    conf, idx = fused_shard.collapseMaxZSlice()
    keys = host_staged_allreduce_keys(pack_argmax_keys(conf, idx, mapper.plane_begin), group)
    conf, gidx = unpack_argmax_keys(keys)
    return mapper.full_depths_[gidx], conf, gidx
