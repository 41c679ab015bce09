"""Multi-GPU sharding of the DSI path: one process per GPU, torch.distributed
("nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

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
        """all-reduce the accumulator and finalize (process2.cpp:221-225 / :234-238)."""
        import torch
        self.ctx.synchronize()                   # engine stream -> torch stream
        allreduce_sum_(self.tensor, self.group)
        if self.tensor.is_cuda:
            torch.cuda.current_stream().synchronize()
        if self.mode == 1:
            self.acc.computeHMfromSumOfInv(self.n)
        else:
            self.acc.computeAMfromSum(self.n)
        return self.acc
