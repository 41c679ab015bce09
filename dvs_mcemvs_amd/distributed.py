"""Multi-GPU sharding of the DSI path: one process per GPU.

The collective itself is the engine's (include/dsi_engine.h "multi-GPU": RCCL called from the C
ABI, `Comm` / `Grid3D.allReduce` here); the Engine* classes below take the collective as a callable
`allreduce(grid, op)` so that the same orchestration runs over RCCL (production), over a
host-staged gloo all-reduce (2-rank tests on one GPU: RCCL allows one rank per device) or over
nothing (one rank).  The older torch.distributed classes (tensor aliases + ExternalStream) remain
for callers that already live in torch.

What shards: time slices.  Alg. 2 of the reference (process2.cpp:98-249)
splits an event stream into `num_subintervals` sub-intervals BY EVENT COUNT, builds one
DSI per sub-interval and fuses them voxel-wise over time with

    HM:  acc += 1/(0.01 + dsi_k)  for every k,  then  n/acc      (cartesian3dgrid.h:72-86)
    AM:  acc += dsi_k,                          then  acc/n      (cartesian3dgrid.h:64-70, 87-93)

Both accumulators are plain sums, so with slice k on rank k % world the whole temporal
fusion is: local accumulate -> ONE all-reduce(sum) of the volume -> local finalize.  No
other collective is on the data path (events never move between GPUs).

This module holds only the partitioning arithmetic and the collective call; the
accumulate / finalize maps are the engine's kernels (Grid3D.addInverseOfTwoGrids, ...).
"""
import numpy as np


def subinterval_bounds(n_events, num_subintervals):
    """Event index ranges of process_2's sub-intervals (process2.cpp:46-47, :105-107):
    num_events_per_subinterval = n / k (integer division); the remainder is dropped."""
    per = int(n_events) // int(num_subintervals)
    return [(k * per, (k + 1) * per) for k in range(int(num_subintervals))]


def slices_of_rank(num_slices, world_size, rank):
    """Round-robin assignment slice k -> rank k % world_size."""
    return [k for k in range(int(num_slices)) if k % int(world_size) == int(rank)]


def allreduce_sum_(tensor, group=None):
    """In-place all-reduce(sum) of a volume accumulator.  `tensor` is a torch tensor on the
    process's device (for the GPU path it aliases a Grid3D through Grid3D(device_ptr=...))."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tensor
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor


def allreduce_max_scalar(value, device=None, group=None):
    """max over ranks of a python float (bench timing)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


# ---------------------------------------------------------------- plane sharding (one big DSI)
def plane_ranges(num_planes, world_size):
    """Contiguous, balanced plane ranges [(begin, count)] per rank (the reference layout
    [z][y][x] makes a plane range a contiguous slab of the volume)."""
    base, extra = divmod(int(num_planes), int(world_size))
    out, b = [], 0
    for r in range(int(world_size)):
        c = base + (1 if r < extra else 0)
        out.append((b, c))
        b += c
    return out


def pack_argmax_keys(conf, idx_local, plane_begin):
    """(confidence, plane index) of a shard's collapseMaxZSlice -> one int64 per pixel whose MAX over
    shards is the unsharded result: Grid3D::collapseMaxZSlice takes the FIRST maximum
    (std::max_element, cartesian3dgrid.cpp:115-137), i.e. the larger confidence wins and, on equal
    confidence, the smaller global plane index.  DSI values are >= 0, so their IEEE bit patterns
    order like the floats."""
    conf = np.ascontiguousarray(conf, np.float32)
    bits = conf.view(np.uint32).astype(np.int64)
    gidx = np.asarray(idx_local).astype(np.int64) + int(plane_begin)
    return (bits << 8) | (255 - gidx)


def unpack_argmax_keys(keys):
    keys = np.asarray(keys, np.int64)
    conf = (keys >> 8).astype(np.uint32).view(np.float32)
    idx = (255 - (keys & 255)).astype(np.uint8)
    return conf, idx


def allreduce_argmax(conf, idx_local, plane_begin, device=None, group=None):
    """The one collective of plane sharding: all-reduce(MAX) of the packed (confidence, index)
    keys (8 B per pixel; 8 MB at 1024x1024).  Returns (conf f32, global idx u8) on every rank."""
    import torch
    import torch.distributed as dist
    keys = pack_argmax_keys(conf, idx_local, plane_begin)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = torch.from_numpy(keys)
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        keys = t.cpu().numpy()
    return unpack_argmax_keys(keys)


class TemporalFusion:
    """Temporal fusion of per-slice DSIs across ranks.

    acc_grid : engine Grid3D that aliases `acc_tensor` (same device memory)
    mode     : ACC_INV_SUM (temporal_fusion = 2, HM) or ACC_SUM (temporal_fusion = 4, AM)
    """

    def __init__(self, ctx, acc_grid, acc_tensor, mode, num_slices, group=None):
        self.ctx, self.acc, self.tensor = ctx, acc_grid, acc_tensor
        self.mode, self.n, self.group = int(mode), int(num_slices), group

    def reset(self):
        self.acc.resetGrid()

    def add(self, dsi):
        if self.mode == 1:
            self.acc.addInverseOfTwoGrids(dsi)   # process2.cpp:218-220
        else:
            self.acc.addTwoGrids(dsi)            # process2.cpp:231-233

    def finish(self):
        """all-reduce the accumulator and finalize (process2.cpp:221-225 / :234-238).
        On a GPU the collective is issued with the engine's HIP stream as torch's current stream,
        so it is ordered after the accumulate kernels and before the finalize kernel on the
        device; the host does not wait."""
        if getattr(self.tensor, "is_cuda", False):
            import torch
            with torch.cuda.stream(torch.cuda.ExternalStream(self.ctx.stream)):
                allreduce_sum_(self.tensor, self.group)
        else:
            allreduce_sum_(self.tensor, self.group)
        if self.mode == 1:
            self.acc.computeHMfromSumOfInv(self.n)
        else:
            self.acc.computeAMfromSum(self.n)
        return self.acc


class PipelinedTemporalFusion:
    """TemporalFusion for a STREAM of fusion rounds (sliding windows, main.cpp:177; bench steps):
    round k's all-reduce + finalize + depth-map extraction run on a second HIP stream while the
    main stream already votes round k+1.  The volume accumulator is double-buffered.

        main stream :  ... vote, camera-fuse(k) | acc[k%2] = f(fused)      | vote, camera-fuse(k+1) ...
        side stream :                            wait | all-reduce(acc[k%2]), finalize, arg-max |

    Device-side ordering only (events); the host never blocks in submit().

    slots   : list of dicts {"tensor": torch tensor [Nz][Ny][Nx], "acc_main": Grid3D aliasing it in
              the main context, "acc_side": Grid3D aliasing it in the side context}
    streams : (main, side) torch streams wrapping the two contexts' HIP streams, or None to run
              everything in program order (CPU tests)
    extract : optional callable(acc_side_grid) run on the side stream after finalize
              (e.g. mapper_fused.computeDepthMap)
    """

    def __init__(self, slots, mode, num_slices, streams=None, extract=None, group=None):
        self.slots, self.mode, self.n = slots, int(mode), int(num_slices)
        self.streams, self.extract, self.group = streams, extract, group
        self.k = 0
        for s in self.slots:
            s["used"] = False
            if streams is not None:
                import torch
                s["filled"], s["free"] = torch.cuda.Event(), torch.cuda.Event()

    @classmethod
    def on_gpu(cls, ctx_main, ctx_side, dims, mode, num_slices, extract=None, depth=2, group=None):
        import torch
        from .engine import Grid3D
        nx, ny, nz = dims
        slots = []
        for _ in range(depth):
            t = torch.empty((nz, ny, nx), dtype=torch.float32, device="cuda")
            slots.append({"tensor": t,
                          "acc_main": Grid3D(ctx_main, nx, ny, nz, device_ptr=t.data_ptr()),
                          "acc_side": Grid3D(ctx_side, nx, ny, nz, device_ptr=t.data_ptr())})
        streams = (torch.cuda.ExternalStream(ctx_main.stream), torch.cuda.ExternalStream(ctx_side.stream))
        obj = cls(slots, mode, num_slices, streams, extract, group)
        obj._ctxs = (ctx_main, ctx_side)
        return obj

    def submit(self, fused):
        """Round k: accumulate `fused` (main stream), then all-reduce / finalize / extract on the
        side stream.  Returns the slot's side-context grid (valid after drain() or after the
        slot's "free" event)."""
        s = self.slots[self.k % len(self.slots)]
        main, side = self.streams if self.streams is not None else (None, None)
        if main is not None and s["used"]:
            main.wait_event(s["free"])            # round k - depth has left this buffer
        s["acc_main"].resetGrid()
        if self.mode == 1:
            s["acc_main"].addInverseOfTwoGrids(fused)   # process2.cpp:218-220
        else:
            s["acc_main"].addTwoGrids(fused)            # process2.cpp:231-233
        if main is not None:
            import torch
            s["filled"].record(main)
            side.wait_event(s["filled"])
            with torch.cuda.stream(side):
                allreduce_sum_(s["tensor"], self.group)
        else:
            allreduce_sum_(s["tensor"], self.group)
        if self.mode == 1:
            s["acc_side"].computeHMfromSumOfInv(self.n)
        else:
            s["acc_side"].computeAMfromSum(self.n)
        if self.extract is not None:
            self.extract(s["acc_side"])
        if main is not None:
            s["free"].record(side)
        s["used"] = True
        self.k += 1
        return s["acc_side"]

    def drain(self):
        """Host waits for both streams."""
        if self.streams is not None:
            for c in self._ctxs:
                c.synchronize()

    def close(self):
        for s in self.slots:
            for key in ("acc_main", "acc_side"):
                if hasattr(s[key], "close"):
                    s[key].close()


# ------------------------------------------------------------------ engine-native orchestration
def engine_allreduce(comm):
    """allreduce(grid, op) over an engine communicator (RCCL from the C ABI, on the grid's stream)."""
    def f(grid, op):
        if comm is not None and comm.size > 1:
            grid.allReduce(comm, op)
    return f


def host_staged_allreduce(group=None):
    """allreduce(grid, op) through host memory and torch.distributed (gloo): for tests that put
    several ranks on ONE GPU, which RCCL refuses.  Synchronises the grid's stream."""
    def f(grid, op):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        t = torch.from_numpy(grid.download())
        dist.all_reduce(t, op={0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MIN, 2: dist.ReduceOp.MAX}[int(op)],
                        group=group)
        grid.upload(t.numpy())
    return f


class EngineTemporalFusion:
    """Time-slice sharding (configs[3]; process2.cpp:211-242 across GPUs): every rank accumulates the
    slices it owns, ONE all-reduce of the accumulator, local finalize.  `mode` is any
    dsi_acc_mode_t (ACC_INV_SUM = the reference's temporal HM, ACC_SUM = its temporal AM; the n-ary
    GM / RMS / min / max modes reduce the same way).  The accumulator starts at the mode's identity,
    so a rank that owns no slice contributes nothing."""

    def __init__(self, ctx, dims, mode, num_slices, allreduce):
        from .engine import Grid3D, acc_reduce_op
        nx, ny, nz = dims
        self.acc = Grid3D(ctx, nx, ny, nz)
        self.mode, self.n, self.allreduce = int(mode), int(num_slices), allreduce
        self.op = acc_reduce_op(self.mode)
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
    """EngineTemporalFusion for a stream of rounds (bench steps, sliding windows): round k's
    all-reduce + finalize + depth-map extraction run on the side context's stream while the main
    context already votes round k+1.  Ordering is by dsi_context_wait_for only; the host never
    blocks in submit().  The accumulator is double-buffered and aliased in both contexts."""

    def __init__(self, ctx_main, ctx_side, dims, mode, num_slices, allreduce, extract=None, depth=2, scattered=None):
        """scattered = (mapper_in_side_context, comm): the round's collective is the reduce-scatter form
        (MapperEMVS.computeDepthMapReduceScattered: reduce-scatter by planes, finalize + arg-max of the owned
        planes, all-reduce(MAX) of keys) instead of all-reduce + finalize + extract; the depth map lands in that
        mapper's buffers, the fused DSI is not completed on any rank."""
        from .engine import Grid3D, acc_reduce_op
        nx, ny, nz = dims
        self.ctx_main, self.ctx_side = ctx_main, ctx_side
        self.mode, self.n, self.allreduce, self.extract = int(mode), int(num_slices), allreduce, extract
        self.scattered = scattered
        self.op = acc_reduce_op(self.mode)
        self.slots = []
        for _ in range(depth):
            main = Grid3D(ctx_main, nx, ny, nz)
            side = Grid3D(ctx_side, nx, ny, nz, device_ptr=main.device_ptr)
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
            side.close()
            main.close()


def plane_sharded_depth_map(mapper, fused_shard, comm=None, group=None):
    """collapseMaxZSlice of a plane-sharded DSI (configs[4]).  With an engine communicator: ONE
    RCCL all-reduce(MAX) of packed keys on the device (dsi_mapper_depth_map_sharded).  Without:
    the same keys through torch.distributed `group` on the host (gloo tests).  Returns
    (depth, conf, global idx) -- identical on every rank."""
    if comm is not None:
        mapper.computeDepthMapSharded(fused_shard, comm)
        return mapper.fetchDepthMap()
    conf, idx = fused_shard.collapseMaxZSlice()
    conf, gidx = allreduce_argmax(conf, idx, mapper.plane_begin, group=group)
    return mapper.full_depths_[gidx], conf, gidx
