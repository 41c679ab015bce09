"""ctypes binding of include/dsi_engine.h.  Host-side mirror of the reference's
operator interface on the hot path:

    Grid3D       cartesian3dgrid/include/cartesian3dgrid/cartesian3dgrid.h:22-247
    MapperEMVS   mapper_emvs_stereo/include/mapper_emvs_stereo/mapper_emvs_stereo.hpp:94-155
    ShapeDSI     mapper_emvs_stereo.hpp:40-65

Method names, argument meaning and error behaviour follow the reference; where the
reference aborts (glog CHECK) or throws std::out_of_range this binding raises
DsiError carrying the C status code.  numpy arrays cross the boundary as plain
pointers; nothing here computes on the CPU.
"""
import atexit
import ctypes as C
import os
import sys
import weakref

import numpy as np

PACKET_SIZE = 1024

FUSE_MIN, FUSE_HM, FUSE_GM, FUSE_AM, FUSE_RMS, FUSE_MAX = 1, 2, 3, 4, 5, 6
ACC_SUM, ACC_INV_SUM, ACC_LOG_SUM, ACC_SQ_SUM, ACC_MIN, ACC_MAX, ACC_GM_TREE = 0, 1, 2, 3, 4, 5, 6
REDUCE_SUM, REDUCE_MIN, REDUCE_MAX = 0, 1, 2
VOTE_AUTO, VOTE_GLOBAL_ATOMIC, VOTE_LDS_BANDS, VOTE_FUSED_ARGMAX = 0, 1, 2, 3

(OK, ERR_INVALID, ERR_TOO_FEW_EVENTS, ERR_HIP, ERR_SHAPE, ERR_BAD_OP, ERR_NO_DEVICE,
 ERR_CONTEXT, ERR_COMM) = range(9)
COMM_ID_BYTES = 128

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class DsiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("dsi_engine error %d: %s" % (code, message))
        self.code = code


class _MapperConfig(C.Structure):
    _fields_ = [("sensor_width", C.c_int), ("sensor_height", C.c_int), ("K", C.c_float * 4),
                ("dim_x", C.c_int), ("dim_y", C.c_int), ("dim_z", C.c_int),
                ("min_depth", C.c_float), ("max_depth", C.c_float), ("fov_deg", C.c_float),
                ("inverse_depth", C.c_int), ("lut", C.POINTER(C.c_float)),
                ("plane_begin", C.c_int), ("plane_count", C.c_int)]


class _DepthMapOptions(C.Structure):
    _fields_ = [("adaptive_threshold_kernel_size", C.c_int), ("adaptive_threshold_c", C.c_double),
                ("median_filter_size", C.c_int), ("max_confidence", C.c_double)]


class _ScatterPlan(C.Structure):
    _fields_ = [("q", C.c_int), ("own_begin", C.c_int), ("own_count", C.c_int), ("tail_begin", C.c_int),
                ("tail_count", C.c_int)]


class _ProveInfo(C.Structure):
    _fields_ = [("rel_gap", C.c_float), ("columns", C.c_longlong), ("columns_proven", C.c_longlong),
                ("columns_unproven", C.c_longlong), ("gap_needed", C.c_double), ("max_votes", C.c_longlong),
                ("elapsed_ms", C.c_float), ("columns_resolved_fully", C.c_longlong)]


class _ResolveInfo(C.Structure):
    _fields_ = [("rel_gap", C.c_float), ("near_tie_pixels", C.c_int), ("candidate_voxels", C.c_int),
                ("candidate_planes", C.c_int), ("votes", C.c_longlong), ("changed_pixels", C.c_int),
                ("max_rel_bound", C.c_double), ("max_order_diff", C.c_double), ("elapsed_ms", C.c_float),
                ("gap_widenings", C.c_int), ("premise_ok", C.c_int), ("columns_bounded", C.c_int)]


class _VoteInfo(C.Structure):
    _fields_ = [("algo", C.c_int), ("bands", C.c_int), ("band_rows", C.c_int),
                ("chunks", C.c_int), ("block_threads", C.c_int), ("lds_bytes", C.c_size_t),
                ("n_packets", C.c_size_t), ("packed", C.c_int), ("group_packets", C.c_int)]


def experiments_requested():
    """DSI_ENGINE_EXPERIMENTS=1 in the environment of THIS process: load the experiments flavour of the library
    (libdsi_engine_experiments.so: environment knobs + dsi_test_* hooks of the timing experiments, some of which make
    the DSIs wrong on purpose).  An explicit opt-in of development tools and of the tests of those hooks; nothing in the
    product sets it."""
    return os.environ.get("DSI_ENGINE_EXPERIMENTS", "0") not in ("", "0")


def library_path():
    return os.path.join(_HERE, "libdsi_engine_experiments.so" if experiments_requested() else "libdsi_engine.so")


def load_library():
    """Load libdsi_engine.so (building it is __graft_entry__.build()'s job).
    Raises when it is absent -- there is no fallback implementation."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing: build the HIP engine first (python -m dvs_mcemvs_amd.build%s). "
            "dvs_mcemvs_amd has no CPU fallback." % (path, " --experiments" if experiments_requested() else ""))
    L = C.CDLL(path)
    L.dsi_build_flavour.restype = C.c_int
    if bool(L.dsi_build_flavour()) != experiments_requested():
        raise ImportError("%s is the %s flavour of the engine but the %s one was asked for: rebuild it "
                          "(python -m dvs_mcemvs_amd.build --force)" %
                          (path, "experiments" if L.dsi_build_flavour() else "production",
                           "experiments" if experiments_requested() else "production"))
    vp, f32p, u8p, u16p, u32p, f64p = (C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8),
                                       C.POINTER(C.c_uint16), C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_double))
    intp, szp, u64p = C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)
    sig = {
        "dsi_last_error": (C.c_char_p, []),
        "dsi_abi_version": (C.c_int, []),
        "dsi_mapper_plane_begin": (C.c_int, [vp]),
        "dsi_mapper_full_depths": (C.c_int, [vp, f32p, intp]),
        "dsi_device_count": (C.c_int, []),
        "dsi_context_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "dsi_context_destroy": (C.c_int, [vp]),
        "dsi_context_synchronize": (C.c_int, [vp]),
        "dsi_context_stream": (vp, [vp]),
        "dsi_context_wait_for": (C.c_int, [vp, vp]),
        "dsi_context_device": (C.c_int, [vp]),
        "dsi_context_timer_start": (C.c_int, [vp]),
        "dsi_context_timer_stop": (C.c_int, [vp, f32p]),
        "dsi_context_timeline_mark": (C.c_int, [vp]),
        "dsi_context_timeline_read": (C.c_int, [vp, f32p, C.c_size_t, szp]),
        "dsi_build_flavour": (C.c_int, []),
        "dsi_mapper_vote_statistics": (C.c_int, [vp, vp, f64p, f64p]),
        "dsi_grid_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
        "dsi_grid_wrap": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]),
        "dsi_grid_destroy": (C.c_int, [vp]),
        "dsi_grid_dims": (C.c_int, [vp, intp, intp, intp]),
        "dsi_grid_reset": (C.c_int, [vp]),
        "dsi_grid_device_ptr": (vp, [vp]),
        "dsi_grid_upload": (C.c_int, [vp, f32p]),
        "dsi_grid_download": (C.c_int, [vp, f32p]),
        "dsi_grid_fuse2": (C.c_int, [vp, vp, C.c_int]),
        "dsi_grid_fuse2_into": (C.c_int, [vp, vp, vp, C.c_int]),
        "dsi_grid_fuse_hm_n": (C.c_int, [vp, vp, C.c_int]),
        "dsi_grid_accumulate": (C.c_int, [vp, vp, C.c_int]),
        "dsi_grid_finalize": (C.c_int, [vp, C.c_int, C.c_int]),
        "dsi_grid_accumulate_begin": (C.c_int, [vp, C.c_int]),
        "dsi_grid_fuse_n": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.c_int]),
        "dsi_acc_reduce_op": (C.c_int, [C.c_int]),
        "dsi_grid_collapse_max_z": (C.c_int, [vp, f32p, u8p]),
        "dsi_grid_collapse_max_z_dev": (C.c_int, [vp, vp, vp, vp, vp]),
        "dsi_grid_mean_square": (C.c_int, [vp, f64p]),
        "dsi_mapper_create": (C.c_int, [vp, C.POINTER(_MapperConfig), C.POINTER(vp)]),
        "dsi_mapper_destroy": (C.c_int, [vp]),
        "dsi_mapper_grid": (vp, [vp]),
        "dsi_mapper_geometry": (C.c_int, [vp, f32p, f32p, intp, intp, intp]),
        "dsi_mapper_set_vote_algo": (C.c_int, [vp, C.c_int]),
        "dsi_mapper_set_band_params": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "dsi_mapper_set_packed_lanes": (C.c_int, [vp, C.c_int]),
        "dsi_mapper_set_inline_cuts": (C.c_int, [vp, C.c_longlong]),
        "dsi_mapper_paired_overflow": (C.c_int, [vp, C.POINTER(C.c_int)]),
        "dsi_mapper_fill_voxel_grid": (C.c_int, [vp, f32p, f32p, C.c_size_t]),
        "dsi_batch_create": (C.c_int, [vp, u16p, u16p, C.c_size_t, u32p, f32p, C.c_size_t,
                                       C.POINTER(vp)]),
        "dsi_batch_destroy": (C.c_int, [vp]),
        "dsi_batch_create_async": (C.c_int, [vp, u16p, u16p, C.c_size_t, u32p, f32p, C.c_size_t,
                                             C.POINTER(vp)]),
        "dsi_batch_uploaded": (C.c_int, [vp]),
        "dsi_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(vp)]),
        "dsi_host_free": (C.c_int, [vp]),
        "dsi_mapper_fetch_depth_map_async": (C.c_int, [vp, f32p, f32p, u8p]),
        "dsi_mapper_fetch_depth_map_in_order": (C.c_int, [vp, f32p, f32p, u8p]),
        "dsi_mapper_fetch_wait": (C.c_int, [vp]),
        "dsi_batch_num_packets": (C.c_size_t, [vp]),
        "dsi_mapper_evaluate_batch": (C.c_int, [vp, vp]),
        "dsi_mapper_evaluate": (C.c_int, [vp, u16p, u16p, f64p, C.c_size_t, f64p, f64p,
                                          C.c_size_t, f64p, szp]),
        "dsi_packetize": (C.c_int, [f64p, C.c_size_t, f64p, f64p, C.c_size_t, f64p, u32p, f32p,
                                    szp]),
        "dsi_packetize_strided": (C.c_int, [vp, C.c_size_t, C.c_size_t, f64p, f64p, C.c_size_t, f64p, u32p, f32p, szp]),
        "dsi_pose_at": (C.c_int, [f64p, f64p, C.c_size_t, C.c_double, f64p]),
        "dsi_mapper_depth_map": (C.c_int, [vp, f32p, f32p, u8p]),
        "dsi_mapper_depth_map_of": (C.c_int, [vp, vp]),
        "dsi_mapper_fetch_depth_map": (C.c_int, [vp, f32p, f32p, u8p]),
        "dsi_mapper_depth_map_of_fusion": (C.c_int, [vp, vp, vp, C.c_int]),
        "dsi_mapper_depth_map_of_fusion_n": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.c_int]),
        "dsi_mapper_depth_map_of_events": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int]),
        "dsi_mapper_depth_map_of_events_n": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int]),
        "dsi_mapper_get_depth_map_from_dsi": (C.c_int, [vp, vp, C.POINTER(_DepthMapOptions), f32p, f32p,
                                                       u8p, u8p]),
        "dsi_mapper_filter_depth_map": (C.c_int, [vp, C.POINTER(_DepthMapOptions), f32p, f32p, u8p, u8p]),
        "dsi_mapper_last_vote_info": (C.c_int, [vp, C.POINTER(_VoteInfo)]),
        "dsi_mapper_set_kernel_timing": (C.c_int, [vp, C.c_int]),
        "dsi_mapper_vote_kernel_time": (C.c_int, [vp, f32p, intp]),
        "dsi_comm_unique_id": (C.c_int, [u8p]),
        "dsi_comm_create_rank": (C.c_int, [vp, u8p, C.c_int, C.c_int, C.POINTER(vp)]),
        "dsi_comm_create_all": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(vp)]),
        "dsi_comm_destroy": (C.c_int, [vp]),
        "dsi_comm_rank": (C.c_int, [vp]),
        "dsi_comm_size": (C.c_int, [vp]),
        "dsi_grid_allreduce": (C.c_int, [vp, vp, C.c_int]),
        "dsi_grid_allreduce_all": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int]),
        "dsi_mapper_depth_map_sharded": (C.c_int, [vp, vp, vp]),
        "dsi_mapper_depth_map_sharded_all": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int]),
        "dsi_mapper_depth_map_reduce_scattered": (C.c_int, [vp, vp, vp, C.c_int, C.c_int]),
        "dsi_mapper_depth_map_reduce_scattered_all": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int,
                                                              C.c_int, C.c_int]),
        "dsi_comm_query": (C.c_int, [vp, intp, intp, intp]),
        "dsi_scatter_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_ScatterPlan)]),
        "dsi_plane_range": (C.c_int, [C.c_int, C.c_int, C.c_int, intp, intp]),
        "dsi_argmax_keys_pack": (C.c_int, [f32p, u8p, C.c_size_t, C.c_int, u64p]),
        "dsi_argmax_keys_unpack": (C.c_int, [u64p, C.c_size_t, f32p, u8p]),
        "dsi_mapper_depth_map_scattered_local": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
        "dsi_mapper_argmax_keys_download": (C.c_int, [vp, u64p]),
        "dsi_mapper_argmax_keys_upload": (C.c_int, [vp, u64p]),
        "dsi_mapper_depth_map_from_keys": (C.c_int, [vp]),
        "dsi_mapper_resolve_near_ties": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int,
                                                  C.POINTER(_ResolveInfo)]),
        "dsi_mapper_prove_near_ties": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.POINTER(_ProveInfo)]),
        "dsi_mapper_prove_near_ties_n": (C.c_int, [vp, vp, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.POINTER(_ProveInfo)]),
        "dsi_mapper_reference_interval": (C.c_int, [vp, vp, vp, vp, vp]),
        "dsi_grid_widen_interval": (C.c_int, [vp, vp, C.c_int]),
        "dsi_grid_prove_columns": (C.c_int, [vp, vp, vp, vp, C.POINTER(_ProveInfo)]),
        "dsi_mapper_proof_votes": (C.c_int, [vp, C.c_int, u32p, C.c_size_t, u32p]),
        "dsi_mapper_proof_unproven": (C.c_int, [vp, u32p, f32p, C.c_size_t, szp]),
        "dsi_grid_near_tie_voxels": (C.c_int, [vp, vp, C.c_float, u32p, C.c_size_t, szp, szp]),
        "dsi_mapper_exact_voxels": (C.c_int, [vp, vp, u32p, C.c_size_t, f32p, u32p]),
        "dsi_reference_fuse2": (C.c_int, [C.c_int, f32p, f32p, C.c_size_t, f32p]),
        "dsi_reference_accumulate": (C.c_int, [C.c_int, f32p, f32p, C.c_size_t]),
        "dsi_reference_finalize": (C.c_int, [C.c_int, f32p, C.c_size_t, C.c_int]),
        "dsi_mapper_patch_depth_map": (C.c_int, [vp, u32p, u8p, f32p, C.c_size_t]),
    }
    if experiments_requested():     # hooks that exist only in the experiments flavour
        sig.update({
            "dsi_test_div_probe": (C.c_int, [vp, f32p, f32p, C.c_size_t, f32p, f32p]),
            "dsi_test_pass_lg": (C.c_int, [vp, C.c_int]),
            "dsi_test_fused_fixed_cost": (C.c_int, [vp, C.c_int]),
            "dsi_test_fused_trace_enable": (C.c_int, [vp, szp]),
            "dsi_test_fused_trace_read": (C.c_int, [vp, vp, C.c_size_t]),
            "dsi_test_run_length_total": (C.c_int, [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
        })
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the .so does not export the ABI
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


EXPORTED_SYMBOLS = None  # filled by tests from include/dsi_engine.h


# Device objects must be released while the HIP runtime is still alive: close whatever is
# still open at interpreter exit (children before contexts), and never touch the runtime
# from __del__ once the interpreter is finalizing.
_LIVE = weakref.WeakSet()


def _track(obj):
    _LIVE.add(obj)


@atexit.register
def _close_all():
    objs = list(_LIVE)
    for o in objs:
        if not isinstance(o, Context):
            try:
                o.close()
            except Exception:
                pass
    for o in objs:
        if isinstance(o, Context):
            try:
                o.close()
            except Exception:
                pass


def _safe_del(obj):
    if sys is None or sys.is_finalizing():
        return
    try:
        obj.close()
    except Exception:
        pass


def _check(rc):
    if rc != OK:
        msg = load_library().dsi_last_error()
        raise DsiError(rc, msg.decode() if msg else "")


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def acc_reduce_op(mode):
    """REDUCE_SUM / REDUCE_MIN / REDUCE_MAX: how an accumulator of `mode` combines across GPUs."""
    r = load_library().dsi_acc_reduce_op(int(mode))
    if r < 0:
        raise DsiError(ERR_BAD_OP, "bad accumulate mode %r" % (mode,))
    return r


def device_count():
    return load_library().dsi_device_count()


class Context:
    """One GPU + one HIP stream.  All objects created from a context share its stream."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(load_library().dsi_context_create(int(device), C.byref(self._h)))
        _track(self)

    def close(self):
        if self._h:
            # objects created from this context hold a pointer to it on the C side: release whatever is
            # still open (e.g. left behind by an exception) BEFORE the context goes
            for o in list(_LIVE):
                if o is not self and getattr(o, "ctx", None) is self:
                    try:
                        o.close()
                    except Exception:
                        pass
            # (the C side refuses while children are alive: that would be a permanent leak of the stream, the pool and
            #  the scratch -- raise, and keep the handle so that a later close() can succeed)
            _check(load_library().dsi_context_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        _safe_del(self)

    def synchronize(self):
        _check(load_library().dsi_context_synchronize(self._h))

    @property
    def stream(self):
        return load_library().dsi_context_stream(self._h)

    def wait_for(self, other):
        """Device-side: work queued on this context from now on starts after everything already
        queued on `other` (no host wait)."""
        _check(load_library().dsi_context_wait_for(self._h, other._h))

    @property
    def device(self):
        return load_library().dsi_context_device(self._h)

    def timer_start(self):
        _check(load_library().dsi_context_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        _check(load_library().dsi_context_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def timeline_mark(self):
        """Record a HIP event on the stream (asynchronous); see timeline_read()."""
        _check(load_library().dsi_context_timeline_mark(self._h))

    def timeline_read(self, capacity=1 << 16):
        """Milliseconds between consecutive marks (waits for the last one; clears the timeline)."""
        out = np.zeros(capacity, np.float32)
        n = C.c_size_t()
        _check(load_library().dsi_context_timeline_read(self._h, _ptr(out, C.c_float), capacity, C.byref(n)))
        return out[:min(n.value, capacity)].copy()


class Comm:
    """One rank of an RCCL communicator created by the engine itself (include/dsi_engine.h,
    "multi-GPU"): nothing here needs torch.  One process per GPU: rank 0 makes
    `uid = Comm.unique_id()`, ships the 128 bytes to the other ranks, every rank builds
    `Comm(ctx, uid, nranks, rank)`.  One process, several GPUs: `Comm.create_all(contexts)`."""

    def __init__(self, ctx, uid, nranks, rank, _handle=None):
        self.ctx = ctx
        if _handle is not None:
            self._h = _handle
        else:
            buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(uid))
            self._h = C.c_void_p()
            _check(load_library().dsi_comm_create_rank(ctx._h, buf, int(nranks), int(rank), C.byref(self._h)))
        _track(self)

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        _check(load_library().dsi_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def create_all(cls, contexts):
        n = len(contexts)
        hs = (C.c_void_p * n)(*[c._h for c in contexts])
        out = (C.c_void_p * n)()
        _check(load_library().dsi_comm_create_all(hs, n, out))
        return [cls(contexts[i], None, n, i, _handle=C.c_void_p(out[i])) for i in range(n)]

    @property
    def rank(self):
        return load_library().dsi_comm_rank(self._h)

    @property
    def size(self):
        return load_library().dsi_comm_size(self._h)

    def query(self):
        """(nranks, rank, device) as RCCL itself reports them (ncclCommCount / ncclCommUserRank / ncclCommCuDevice)."""
        n, r, dev = C.c_int(), C.c_int(), C.c_int()
        _check(load_library().dsi_comm_query(self._h, C.byref(n), C.byref(r), C.byref(dev)))
        return n.value, r.value, dev.value

    def close(self):
        if self._h:
            load_library().dsi_comm_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        _safe_del(self)


def scatter_plan(dim_z, nranks, rank):
    """dsi_scatter_plan: which planes rank `rank` of `nranks` owns when the temporal fusion's collective is a
    reduce-scatter: dict(q, own_begin, own_count, tail_begin, tail_count) -- the arithmetic the RCCL path uses."""
    sp = _ScatterPlan()
    _check(load_library().dsi_scatter_plan(int(dim_z), int(nranks), int(rank), C.byref(sp)))
    return {k: getattr(sp, k) for k, _ in _ScatterPlan._fields_}


def plane_range(dim_z, nranks, rank):
    """dsi_plane_range: (begin, count) of rank's planes under plane sharding."""
    b, c = C.c_int(), C.c_int()
    _check(load_library().dsi_plane_range(int(dim_z), int(nranks), int(rank), C.byref(b), C.byref(c)))
    return b.value, c.value


def argmax_keys_pack(conf, idx_local, plane_begin):
    conf = _arr(conf, np.float32)
    idx_local = _arr(idx_local, np.uint8)
    keys = np.empty(conf.shape, np.uint64)
    _check(load_library().dsi_argmax_keys_pack(_ptr(conf, C.c_float), _ptr(idx_local, C.c_uint8), conf.size,
                                               int(plane_begin), _ptr(keys, C.c_uint64)))
    return keys


def argmax_keys_unpack(keys):
    keys = _arr(keys, np.uint64)
    conf = np.empty(keys.shape, np.float32)
    idx = np.empty(keys.shape, np.uint8)
    _check(load_library().dsi_argmax_keys_unpack(_ptr(keys, C.c_uint64), keys.size, _ptr(conf, C.c_float),
                                                 _ptr(idx, C.c_uint8)))
    return conf, idx


def widen_interval(lo, hi, roundings):
    """dsi_grid_widen_interval: lo <- lo (1 - k u) rounded down, hi <- hi (1 + k u) rounded up, u = 2^-24 -- after a grid op
    that performs k fp32 roundings per voxel was applied to both grids of an interval."""
    _check(load_library().dsi_grid_widen_interval(lo._h, hi._h, int(roundings)))


def reference_fuse2(op, a, g):
    """The reference's 2-ary camera-fusion op on host arrays (dsi_reference_fuse2: cartesian3dgrid.h:111-190 applied to
    a grid initialised by resetGrid(); addTwoGrids(a)), bit for bit what the device computes."""
    a, g = _arr(a, np.float32), _arr(g, np.float32)
    out = np.empty(a.shape, np.float32)
    _check(load_library().dsi_reference_fuse2(int(op), _ptr(a, C.c_float), _ptr(g, C.c_float), a.size, _ptr(out, C.c_float)))
    return out


def reference_accumulate(mode, acc, g):
    """acc + g (ACC_SUM) / acc + 1/(0.01 + g) (ACC_INV_SUM) on host arrays (cartesian3dgrid.h:64-78); returns a new array."""
    acc, g = np.array(acc, np.float32, order="C"), _arr(g, np.float32)
    _check(load_library().dsi_reference_accumulate(int(mode), _ptr(acc, C.c_float), _ptr(g, C.c_float), acc.size))
    return acc


def reference_finalize(mode, acc, n_maps):
    """acc / n (ACC_SUM) / n / acc (ACC_INV_SUM) on a host array (cartesian3dgrid.h:80-93); returns a new array."""
    acc = np.array(acc, np.float32, order="C")
    _check(load_library().dsi_reference_finalize(int(mode), _ptr(acc, C.c_float), acc.size, int(n_maps)))
    return acc


def allreduce_all(comms, grids, op):
    """All ranks of one process at once (comms from Comm.create_all; grids[i] on comms[i]'s GPU)."""
    n = len(comms)
    cs = (C.c_void_p * n)(*[c._h for c in comms])
    gs = (C.c_void_p * n)(*[g._h for g in grids])
    _check(load_library().dsi_grid_allreduce_all(cs, gs, n, int(op)))


def depth_map_reduce_scattered_all(mappers, accs, comms, mode, n_maps):
    """One process, several GPUs: MapperEMVS.computeDepthMapReduceScattered on every rank in one group call."""
    n = len(comms)
    hm = (C.c_void_p * n)(*[m._h for m in mappers])
    hg = (C.c_void_p * n)(*[g._h for g in accs])
    hc = (C.c_void_p * n)(*[c._h for c in comms])
    _check(load_library().dsi_mapper_depth_map_reduce_scattered_all(hm, hg, hc, n, int(mode), int(n_maps)))


def depth_map_sharded_all(mappers, grids, comms):
    n = len(comms)
    ms = (C.c_void_p * n)(*[m._h for m in mappers])
    gs = (C.c_void_p * n)(*[g._h for g in grids])
    cs = (C.c_void_p * n)(*[c._h for c in comms])
    _check(load_library().dsi_mapper_depth_map_sharded_all(ms, gs, cs, n))


class Grid3D:
    """Device-resident counterpart of the reference's Grid3D (cartesian3dgrid.h:22-247).

    Layout volume[x + dimX*(y + dimY*z)]; host views are numpy [dimZ][dimY][dimX].
    Fusion methods keep the reference's names and in-place semantics.
    """

    def __init__(self, ctx, dimX, dimY, dimZ, _handle=None, _owner=None, device_ptr=None):
        self.ctx = ctx
        self._owner = _owner
        self._keep = None
        L = load_library()
        if _handle is not None:
            self._h = _handle
            self._owned = False
        elif device_ptr is not None:
            self._h = C.c_void_p()
            _check(L.dsi_grid_wrap(ctx._h, dimX, dimY, dimZ, C.c_void_p(device_ptr),
                                   C.byref(self._h)))
            self._owned = True
        else:
            self._h = C.c_void_p()
            _check(L.dsi_grid_create(ctx._h, dimX, dimY, dimZ, C.byref(self._h)))
            self._owned = True
        if self._owned:
            _track(self)

    def close(self):
        if getattr(self, "_owned", False) and self._h:
            load_library().dsi_grid_destroy(self._h)
        self._h = C.c_void_p()
        self._owned = False

    def __del__(self):
        _safe_del(self)

    def getDimensions(self):
        nx, ny, nz = C.c_int(), C.c_int(), C.c_int()
        _check(load_library().dsi_grid_dims(self._h, C.byref(nx), C.byref(ny), C.byref(nz)))
        return nx.value, ny.value, nz.value

    @property
    def shape(self):
        nx, ny, nz = self.getDimensions()
        return nz, ny, nx

    @property
    def device_ptr(self):
        return load_library().dsi_grid_device_ptr(self._h)

    def resetGrid(self):
        _check(load_library().dsi_grid_reset(self._h))

    def upload(self, host):
        host = _arr(host, np.float32)
        if host.shape != self.shape:
            raise DsiError(ERR_SHAPE, "host array shape %s != grid shape %s" % (host.shape, self.shape))
        _check(load_library().dsi_grid_upload(self._h, _ptr(host, C.c_float)))

    def download(self):
        out = np.empty(self.shape, np.float32)
        _check(load_library().dsi_grid_download(self._h, _ptr(out, C.c_float)))
        return out

    # -- voxel-wise fusion, cartesian3dgrid.h:64-192 --------------------------------
    def _fuse(self, other, op):
        _check(load_library().dsi_grid_fuse2(self._h, other._h, op))

    def minTwoGrids(self, grid2):
        self._fuse(grid2, FUSE_MIN)

    def harmonicMeanTwoGrids(self, grid2, n=None):
        if n is None:
            self._fuse(grid2, FUSE_HM)
        else:
            _check(load_library().dsi_grid_fuse_hm_n(self._h, grid2._h, int(n)))

    def geometricMeanTwoGrids(self, grid2):
        self._fuse(grid2, FUSE_GM)

    def arithmeticMeanTwoGrids(self, grid2):
        self._fuse(grid2, FUSE_AM)

    def rmsTwoGrids(self, grid2):
        self._fuse(grid2, FUSE_RMS)

    def maxTwoGrids(self, grid2):
        self._fuse(grid2, FUSE_MAX)

    def fuseTwoGrids(self, grid2, fusion_method):
        """The switch(fusion_method) of process1.cpp:136-158."""
        self._fuse(grid2, int(fusion_method))

    def setToFusionOf(self, grid_a, grid_b, fusion_method):
        """self = op(grid_a, grid_b): the result of the reference's
        `resetGrid(); addTwoGrids(a); <op>TwoGrids(b)` (process1.cpp:126-158) in one pass."""
        _check(load_library().dsi_grid_fuse2_into(self._h, grid_a._h, grid_b._h, int(fusion_method)))

    def addTwoGrids(self, grid2):
        _check(load_library().dsi_grid_accumulate(self._h, grid2._h, ACC_SUM))

    def addInverseOfTwoGrids(self, grid2):
        _check(load_library().dsi_grid_accumulate(self._h, grid2._h, ACC_INV_SUM))

    def computeAMfromSum(self, n):
        _check(load_library().dsi_grid_finalize(self._h, ACC_SUM, int(n)))

    def computeHMfromSumOfInv(self, n):
        _check(load_library().dsi_grid_finalize(self._h, ACC_INV_SUM, int(n)))

    # -- n-ary accumulate / finalize (dsi_acc_mode_t): the reference's temporal accumulators
    #    (modes 0, 1) and the n-ary forms of its 2-ary camera-fusion ops (modes 2..5), which it
    #    does not have (process1.cpp:169-191 drops camera 3 for GM / AM / RMS)
    def accumulateBegin(self, mode):
        _check(load_library().dsi_grid_accumulate_begin(self._h, int(mode)))

    def accumulate(self, grid2, mode):
        _check(load_library().dsi_grid_accumulate(self._h, grid2._h, int(mode)))

    def finalize(self, mode, n):
        _check(load_library().dsi_grid_finalize(self._h, int(mode), int(n)))

    def allReduce(self, comm, op):
        """In-place RCCL all-reduce over `comm` on this grid's stream (op: REDUCE_SUM/MIN/MAX, or
        acc_reduce_op(mode) for an accumulator)."""
        _check(load_library().dsi_grid_allreduce(comm._h, self._h, int(op)))

    def setToFusionOfN(self, grids, mode):
        """self = n-ary mean of `grids`: ACC_SUM arithmetic, ACC_LOG_SUM geometric (0 where any
        grid is 0), ACC_SQ_SUM root mean square, ACC_MIN / ACC_MAX."""
        if len(grids) <= 8 and all(g is not self for g in grids):
            hs = (C.c_void_p * len(grids))(*[g._h for g in grids])   # one pass over the volumes, same bits
            _check(load_library().dsi_grid_fuse_n(self._h, hs, len(grids), int(mode)))
            return
        self.accumulateBegin(mode)
        for g in grids:
            self.accumulate(g, mode)
        self.finalize(mode, len(grids))

    # -- cartesian3dgrid.cpp:115-137, :164-174 --------------------------------------
    def collapseMaxZSlice(self):
        """Returns (max_val float32 [dimY][dimX], max_pos uint8 [dimY][dimX])."""
        nz, ny, nx = self.shape
        conf = np.empty((ny, nx), np.float32)
        idx = np.empty((ny, nx), np.uint8)
        _check(load_library().dsi_grid_collapse_max_z(self._h, _ptr(conf, C.c_float),
                                                      _ptr(idx, C.c_uint8)))
        return conf, idx

    def computeMeanSquare(self):
        out = C.c_double()
        _check(load_library().dsi_grid_mean_square(self._h, C.byref(out)))
        return out.value


class ShapeDSI:
    """mapper_emvs_stereo.hpp:40-65"""

    def __init__(self, dimX=0, dimY=0, dimZ=100, min_depth=0.3, max_depth=5.0, fov=0.0):
        self.dimX_, self.dimY_, self.dimZ_ = int(dimX), int(dimY), int(dimZ)
        self.min_depth_, self.max_depth_, self.fov_ = float(min_depth), float(max_depth), float(fov)


class OptionsDepthMap:
    """mapper_emvs_stereo.hpp:68-82 (the fields getDepthMapFromDSI reads; defaults of main.cpp:73-75,97)."""

    def __init__(self, adaptive_threshold_kernel_size=5, adaptive_threshold_c=5.0, median_filter_size=5,
                 max_confidence=0.0):
        self.adaptive_threshold_kernel_size_ = int(adaptive_threshold_kernel_size)
        self.adaptive_threshold_c_ = float(adaptive_threshold_c)
        self.median_filter_size_ = int(median_filter_size)
        self.max_confidence = float(max_confidence)


class PinnedArray:
    """numpy array in page-locked host memory (dsi_host_alloc): the source / destination of
    asynchronous uploads and depth-map fetches.  `a` is the array; close() frees the memory."""

    def __init__(self, shape, dtype):
        dtype = np.dtype(dtype)
        n = int(np.prod(shape))
        self._p = C.c_void_p()
        _check(load_library().dsi_host_alloc(max(1, n * dtype.itemsize), C.byref(self._p)))
        buf = (C.c_uint8 * (n * dtype.itemsize)).from_address(self._p.value)
        self.a = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
        _track(self)

    def close(self):
        if self._p:
            self.a = None
            load_library().dsi_host_free(self._p)
        self._p = C.c_void_p()

    def __del__(self):
        _safe_del(self)


class EventBatch:
    """Device-resident events of one evaluateDSI call + their packetisation
    (mapper_emvs_stereo.cpp:88-105).  asynchronous=True: x, y, Rt, packet_first are views of
    PinnedArray memory that the caller leaves alone until uploaded() (or a later synchronisation of
    the context that evaluated the batch); the constructor then returns without waiting."""

    def __init__(self, ctx, x, y, Rt, packet_first=None, asynchronous=False):
        x = _arr(x, np.uint16)
        y = _arr(y, np.uint16)
        Rt = _arr(Rt, np.float32).reshape(-1, 12)
        pf = None
        if packet_first is not None:
            packet_first = _arr(packet_first, np.uint32)
            pf = _ptr(packet_first, C.c_uint32)
        self.ctx = ctx
        self.n_packets = Rt.shape[0]
        self.n_events = x.shape[0]
        self._h = C.c_void_p()
        L = load_library()
        fn = L.dsi_batch_create_async if asynchronous else L.dsi_batch_create
        if asynchronous:
            self._keep = (x, y, Rt, packet_first)   # the views (not the pinned memory itself)
        _check(fn(ctx._h, _ptr(x, C.c_uint16), _ptr(y, C.c_uint16), x.shape[0], pf, _ptr(Rt, C.c_float),
                  Rt.shape[0], C.byref(self._h)))
        _track(self)

    def uploaded(self):
        return bool(load_library().dsi_batch_uploaded(self._h))

    def close(self):
        if self._h:
            load_library().dsi_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        _safe_del(self)


class MapperEMVS:
    """Device-resident counterpart of EMVS::MapperEMVS (mapper_emvs_stereo.hpp:94-155).

    cam = (width, height, fx, fy, cx, cy): full resolution and the projection-matrix
    intrinsics the reference reads from image_geometry::PinholeCameraModel
    (mapper_emvs_stereo.cpp:34-48).  lut = undistortion table of
    precomputeRectifiedPoints (mapper_emvs_stereo.cpp:256-299), shape [H*W][2] or None.
    """

    def __init__(self, ctx, cam, dsi_shape, lut=None, inverse_depth=False, plane_range=None):
        """plane_range = (begin, count): own only those planes of dsi_shape's dimZ planes (plane
        sharding across GPUs; the DSI then has `count` planes and equals that slice of the full one)."""
        w, h, fx, fy, cx, cy = cam
        cfg = _MapperConfig()
        if plane_range is not None:
            cfg.plane_begin, cfg.plane_count = int(plane_range[0]), int(plane_range[1])
        cfg.sensor_width, cfg.sensor_height = int(w), int(h)
        cfg.K = (C.c_float * 4)(fx, fy, cx, cy)
        cfg.dim_x, cfg.dim_y, cfg.dim_z = dsi_shape.dimX_, dsi_shape.dimY_, dsi_shape.dimZ_
        cfg.min_depth, cfg.max_depth, cfg.fov_deg = (dsi_shape.min_depth_, dsi_shape.max_depth_,
                                                     dsi_shape.fov_)
        cfg.inverse_depth = 1 if inverse_depth else 0
        self._lut = None
        if lut is not None:
            self._lut = _arr(lut, np.float32).reshape(int(w) * int(h), 2)
            cfg.lut = _ptr(self._lut, C.c_float)
        self.ctx = ctx
        self._h = C.c_void_p()
        L = load_library()
        _check(L.dsi_mapper_create(ctx._h, C.byref(cfg), C.byref(self._h)))
        _track(self)
        nx, ny, nz = C.c_int(), C.c_int(), C.c_int()
        kv = (C.c_float * 4)()
        _check(L.dsi_mapper_geometry(self._h, kv, None, C.byref(nx), C.byref(ny), C.byref(nz)))
        self.dimX, self.dimY, self.dimZ = nx.value, ny.value, nz.value
        planes = np.empty(self.dimZ, np.float32)
        _check(L.dsi_mapper_geometry(self._h, None, _ptr(planes, C.c_float), None, None, None))
        self.raw_depths_vec_ = planes
        self.plane_begin = int(L.dsi_mapper_plane_begin(self._h))
        nfull = C.c_int()
        _check(L.dsi_mapper_full_depths(self._h, None, C.byref(nfull)))
        self.full_depths_ = np.empty(nfull.value, np.float32)   # whole depth vector (plane shards index into it)
        _check(L.dsi_mapper_full_depths(self._h, _ptr(self.full_depths_, C.c_float), None))
        self.virtual_cam_ = tuple(float(v) for v in kv)  # fx, fy, cx, cy
        # public member dsi_ (mapper_emvs_stereo.hpp:116)
        self.dsi_ = Grid3D(ctx, 0, 0, 0, _handle=C.c_void_p(L.dsi_mapper_grid(self._h)), _owner=self)
        self.name = ""

    def close(self):
        if self._h:
            load_library().dsi_mapper_destroy(self._h)
            self._h = C.c_void_p()
            self.dsi_._h = C.c_void_p()

    def __del__(self):
        _safe_del(self)

    def set_vote_algo(self, algo):
        _check(load_library().dsi_mapper_set_vote_algo(self._h, int(algo)))

    def set_band_params(self, band_rows=0, chunks=0, block_threads=0):
        _check(load_library().dsi_mapper_set_band_params(self._h, int(band_rows), int(chunks),
                                                         int(block_threads)))

    def set_packed_lanes(self, mode=-1):
        _check(load_library().dsi_mapper_set_packed_lanes(self._h, int(mode)))

    def paired_overflow(self):
        """Lane mapping 8 (paired 32-bit cells): did a cell of the last vote reach half its capacity?  (Then vote again
        with an exact mapping.)"""
        flag = C.c_int(0)
        _check(load_library().dsi_mapper_paired_overflow(self._h, C.byref(flag)))
        return bool(flag.value)

    def set_inline_cuts(self, min_packets=-1):
        """Lane mappings 5 / 6: from how many packets per call on the voting kernel derives the packets' runs itself
        instead of reading a cut table (-1 = default 8192, 0 = always)."""
        _check(load_library().dsi_mapper_set_inline_cuts(self._h, int(min_packets)))

    def last_vote_info(self):
        info = _VoteInfo()
        _check(load_library().dsi_mapper_last_vote_info(self._h, C.byref(info)))
        return {k: getattr(info, k) for k, _ in _VoteInfo._fields_}

    def vote_statistics(self, batch):
        """(accepted event-planes, accepted records) of voting `batch`: the DSI's sum, and the same count after the
        packet sort merged same-pixel events of a packet (= LDS atomics issued / 4).  Votes twice; the DSI is left as
        evaluateDSI_batch leaves it."""
        a, r = C.c_double(), C.c_double()
        _check(load_library().dsi_mapper_vote_statistics(self._h, batch._h, C.byref(a), C.byref(r)))
        return a.value, r.value

    def set_kernel_timing(self, enable=True):
        _check(load_library().dsi_mapper_set_kernel_timing(self._h, 1 if enable else 0))

    def vote_kernel_time(self):
        """(total ms, launches) of the voting kernel since the last call (HIP events)."""
        ms, n = C.c_float(), C.c_int()
        _check(load_library().dsi_mapper_vote_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def evaluateDSI(self, events, trajectory, T_rv_w):
        """mapper_emvs_stereo.cpp:67-148.  events = (x uint16[n], y uint16[n], ts float64[n]);
        trajectory = (times float64[m], poses float64[m][7] as tx,ty,tz,qw,qx,qy,qz);
        T_rv_w = 7 doubles.  Returns False where the reference returns false."""
        x, y, ts = events
        x, y, ts = _arr(x, np.uint16), _arr(y, np.uint16), _arr(ts, np.float64)
        times, poses = trajectory
        times, poses = _arr(times, np.float64), _arr(poses, np.float64).reshape(-1, 7)
        T = _arr(T_rv_w, np.float64)
        voted = C.c_size_t()
        rc = load_library().dsi_mapper_evaluate(
            self._h, _ptr(x, C.c_uint16), _ptr(y, C.c_uint16), _ptr(ts, C.c_double), x.shape[0],
            _ptr(times, C.c_double), _ptr(poses, C.c_double), times.shape[0], _ptr(T, C.c_double),
            C.byref(voted))
        if rc == ERR_TOO_FEW_EVENTS:
            return False
        _check(rc)
        self.n_voted = voted.value
        return True

    def evaluateDSI_batch(self, batch):
        """evaluateDSI past the pose lookup, on a device-resident EventBatch (asynchronous)."""
        _check(load_library().dsi_mapper_evaluate_batch(self._h, batch._h))

    def fillVoxelGrid(self, event_locations_z0, camera_centers):
        """mapper_emvs_stereo.cpp:151-205 (private in the reference; the parity test point).
        Accumulates into dsi_ without resetting it."""
        xy = _arr(event_locations_z0, np.float32).reshape(-1, 2)
        cc = _arr(camera_centers, np.float32).reshape(-1, 3)
        if xy.shape[0] != cc.shape[0] * PACKET_SIZE:
            raise DsiError(ERR_INVALID, "need 1024 event locations per camera centre")
        _check(load_library().dsi_mapper_fill_voxel_grid(self._h, _ptr(xy, C.c_float),
                                                         _ptr(cc, C.c_float), cc.shape[0]))

    def getDepthMapFromDSI(self, grid=None, options_depth_map=None):
        """mapper_emvs_stereo.cpp:339-437.  With options_depth_map (OptionsDepthMap): the full
        extraction -- arg-max, confidence normalisation, Gaussian adaptive threshold, masked
        median, border removal -- returning (depth_map, confidence_map, mask) like the
        reference's signature (dense inpainted map excluded).  Without: the raw arg-max
        (:368) + convertDepthIndicesToValues (:302-313), returning (depth, confidence, indices)."""
        L = load_library()
        if options_depth_map is not None:
            o = options_depth_map
            opts = _DepthMapOptions(o.adaptive_threshold_kernel_size_, o.adaptive_threshold_c_,
                                    o.median_filter_size_, o.max_confidence)
            depth = np.empty((self.dimY, self.dimX), np.float32)
            conf = np.empty((self.dimY, self.dimX), np.float32)
            mask = np.empty((self.dimY, self.dimX), np.uint8)
            self.depth_cell_indices_filtered = np.empty((self.dimY, self.dimX), np.uint8)
            _check(L.dsi_mapper_get_depth_map_from_dsi(
                self._h, (grid or self.dsi_)._h, C.byref(opts), _ptr(depth, C.c_float), _ptr(conf, C.c_float),
                _ptr(mask, C.c_uint8), _ptr(self.depth_cell_indices_filtered, C.c_uint8)))
            return depth, conf, mask
        _check(L.dsi_mapper_depth_map_of(self._h, (grid or self.dsi_)._h))
        return self.fetchDepthMap()

    def filterDepthMap(self, options_depth_map):
        """mapper_emvs_stereo.cpp:390-437 -- everything getDepthMapFromDSI does after
        collapseMaxZSlice -- on the raw depth map this mapper holds from the last computeDepthMap /
        computeDepthMapOfFusion / computeDepthMapOfEvents call: (depth_map, confidence_map, mask).
        The way to the reference's filtered per-window outputs when no DSI is materialised."""
        o = options_depth_map
        opts = _DepthMapOptions(o.adaptive_threshold_kernel_size_, o.adaptive_threshold_c_,
                                o.median_filter_size_, o.max_confidence)
        depth = np.empty((self.dimY, self.dimX), np.float32)
        conf = np.empty((self.dimY, self.dimX), np.float32)
        mask = np.empty((self.dimY, self.dimX), np.uint8)
        self.depth_cell_indices_filtered = np.empty((self.dimY, self.dimX), np.uint8)
        _check(load_library().dsi_mapper_filter_depth_map(
            self._h, C.byref(opts), _ptr(depth, C.c_float), _ptr(conf, C.c_float), _ptr(mask, C.c_uint8),
            _ptr(self.depth_cell_indices_filtered, C.c_uint8)))
        return depth, conf, mask

    def computeDepthMap(self, grid=None):
        """Asynchronous half of getDepthMapFromDSI; pair with fetchDepthMap()."""
        _check(load_library().dsi_mapper_depth_map_of(self._h, (grid or self.dsi_)._h))

    def computeDepthMapOfFusion(self, grid_a, grid_b, fusion_method):
        """computeDepthMap of op(grid_a, grid_b) without materialising the fused DSI (bit-identical to
        setToFusionOf + computeDepthMap; one pass over the two volumes)."""
        _check(load_library().dsi_mapper_depth_map_of_fusion(self._h, grid_a._h, grid_b._h, int(fusion_method)))

    def computeDepthMapOfFusionN(self, grids, mode):
        """computeDepthMap of the n-ary fusion of up to 8 grids (mode = ACC_*) without materialising
        the fused DSI: bit-identical to Grid3D.setToFusionOfN + computeDepthMap."""
        hs = (C.c_void_p * len(grids))(*[g._h for g in grids])
        _check(load_library().dsi_mapper_depth_map_of_fusion_n(self._h, hs, len(grids), int(mode)))

    def computeDepthMapOfEvents(self, mappers, batches, fusion_method=FUSE_HM):
        """Depth map of 1, 2 or 3 cameras' events without building their DSIs: one kernel votes each
        (band, plane) of every camera into LDS, fuses the cameras per voxel and keeps the running
        arg-max in registers (dsi_mapper_depth_map_of_events).  Bit-identical to evaluateDSI_batch on
        every mapper + computeDepthMapOfFusion (or computeDepthMap for one camera; for three, the
        trinocular sequence of process1.cpp:126-191: the 2-ary op, then min / harmonicMeanTwoGrids(g, 3) /
        max with camera 2 -- ops 3, 4, 5 ignore it like the reference); the mappers' DSIs
        are not touched.  Results land in THIS mapper's depth-map buffers (fetchDepthMap)."""
        hm = (C.c_void_p * len(mappers))(*[m._h for m in mappers])
        hb = (C.c_void_p * len(batches))(*[b._h for b in batches])
        _check(load_library().dsi_mapper_depth_map_of_events(self._h, hm, hb, len(mappers), int(fusion_method)))

    def computeDepthMapOfEventsN(self, mappers, batches, mode=None):
        """Depth map of four (or two) cameras' events fused by the geometric-mean tree (ACC_GM_TREE) without building their
        DSIs (dsi_mapper_depth_map_of_events_n): bit-identical to evaluateDSI_batch on every mapper +
        computeDepthMapOfFusionN([m.dsi_ ...], ACC_GM_TREE).  Results in THIS mapper's depth-map buffers."""
        hm = (C.c_void_p * len(mappers))(*[m._h for m in mappers])
        hb = (C.c_void_p * len(batches))(*[b._h for b in batches])
        _check(load_library().dsi_mapper_depth_map_of_events_n(self._h, hm, hb, len(mappers),
                                                               int(ACC_GM_TREE if mode is None else mode)))

    def resolveNearTies(self, mappers, batches, fusion_method=FUSE_HM, rel_gap=0.0):
        """Exact tie resolver (dsi_mapper_resolve_near_ties): after the mappers' DSIs were built from `batches`
        (evaluateDSI_batch) and this mapper holds the depth map of their fusion, re-sum the contending voxels of the
        near-tie columns in the REFERENCE's summation order (fp32, event order) and re-pick the first maximum, so that
        the plane index map equals the CPU reference's on every pixel.  Returns the call's statistics (dict)."""
        n = len(mappers)
        hm = (C.c_void_p * n)(*[m._h for m in mappers])
        hb = (C.c_void_p * n)(*[b._h for b in batches])
        info = _ResolveInfo()
        info.rel_gap = float(rel_gap)
        _check(load_library().dsi_mapper_resolve_near_ties(self._h, hm, hb, n, int(fusion_method), C.byref(info)))
        return {k: getattr(info, k) for k, _ in _ResolveInfo._fields_}

    def proveNearTies(self, mappers, batches, fusion_method=FUSE_HM, rel_gap=0.0):
        """The resolver's premise as a per-column proof (dsi_mapper_prove_near_ties; a verification pass): counts the votes of
        every voxel of the mappers' DSIs (built from `batches`) and checks, with rigorous bounds of the reference's fp32
        event-order sums, that in every column no plane outside `rel_gap` of the maximum can be the reference's first
        maximum.  Returns {columns, columns_proven, columns_unproven, gap_needed, max_votes, elapsed_ms, rel_gap}."""
        n = len(mappers)
        hm = (C.c_void_p * n)(*[m._h for m in mappers])
        hb = (C.c_void_p * n)(*[b._h for b in batches])
        info = _ProveInfo()
        info.rel_gap = float(rel_gap)
        _check(load_library().dsi_mapper_prove_near_ties(self._h, hm, hb, n, int(fusion_method), C.byref(info)))
        return {k: getattr(info, k) for k, _ in _ProveInfo._fields_}

    def proveNearTiesN(self, mappers, batches, mode, fused_grid=None, rel_gap=0.0):
        """dsi_mapper_prove_near_ties_n: the per-column proof for n <= 8 cameras fused by an n-ary mode (ACC_GM_TREE with
        2 / 4 / 8 cameras, ACC_MIN, ACC_MAX, ACC_SUM); fused_grid: the grid that holds the fusion (default: this mapper's
        dsi_), whose values decided the near-tie columns."""
        n = len(mappers)
        hm = (C.c_void_p * n)(*[m._h for m in mappers])
        hb = (C.c_void_p * n)(*[b._h for b in batches])
        info = _ProveInfo()
        info.rel_gap = float(rel_gap)
        g = fused_grid if fused_grid is not None else self.dsi_
        _check(load_library().dsi_mapper_prove_near_ties_n(self._h, g._h, hm, hb, n, int(mode), C.byref(info)))
        return {k: getattr(info, k) for k, _ in _ProveInfo._fields_}

    def referenceInterval(self, mapper, batch, lo, hi):
        """dsi_mapper_reference_interval: lo, hi (Grid3D) <- voxel-wise bounds of the value the reference holds in the DSI
        `mapper` built from `batch` (its votes counted; this mapper lends the scratch)."""
        _check(load_library().dsi_mapper_reference_interval(self._h, mapper._h, batch._h, lo._h, hi._h))

    def proveColumns(self, fused, lo, hi, rel_gap=0.0):
        """dsi_grid_prove_columns: every column of `fused` (the engine's values) against the interval grids lo / hi."""
        info = _ProveInfo()
        info.rel_gap = float(rel_gap)
        _check(load_library().dsi_grid_prove_columns(self._h, fused._h, lo._h, hi._h, C.byref(info)))
        return {k: getattr(info, k) for k, _ in _ProveInfo._fields_}

    def proofUnproven(self):
        """dsi_mapper_proof_unproven: (pixels uint32[n], gaps float32[n]) -- the columns the last proveNearTies could not
        prove and the rel_gap each of them alone would need."""
        n = C.c_size_t(0)
        lib = load_library()
        _check(lib.dsi_mapper_proof_unproven(self._h, None, None, 0, C.byref(n)))
        pixels, gaps = np.empty(n.value, np.uint32), np.empty(n.value, np.float32)
        if n.value:
            _check(lib.dsi_mapper_proof_unproven(self._h, _ptr(pixels, C.c_uint32), _ptr(gaps, C.c_float), n.value, C.byref(n)))
        return pixels, gaps

    def proofVotes(self, camera, voxels):
        """dsi_mapper_proof_votes: the votes of the listed voxels of camera 0 / 1 as the last proveNearTies counted them."""
        voxels = _arr(voxels, np.uint32)
        votes = np.empty(voxels.shape, np.uint32)
        _check(load_library().dsi_mapper_proof_votes(self._h, int(camera), _ptr(voxels, C.c_uint32), voxels.size,
                                                     _ptr(votes, C.c_uint32)))
        return votes

    def nearTieVoxels(self, grid=None, rel_gap=0.0):
        """dsi_grid_near_tie_voxels of `grid` (default: this mapper's DSI): the voxels (linear indices z*Ny*Nx + y*Nx + x)
        within rel_gap of their column's maximum, for columns with >= 2 of them; a column's run contiguous, planes
        ascending.  Returns (voxels uint32[n], n_columns)."""
        g = self.dsi_ if grid is None else grid
        cap = 1 << 16
        while True:
            vox = np.empty(cap, np.uint32)
            n, cols = C.c_size_t(), C.c_size_t()
            _check(load_library().dsi_grid_near_tie_voxels(self._h, g._h, float(rel_gap), _ptr(vox, C.c_uint32), cap,
                                                           C.byref(n), C.byref(cols)))
            if n.value <= cap:
                return vox[:n.value].copy(), cols.value
            cap = n.value

    def exactVoxels(self, batch, voxels):
        """dsi_mapper_exact_voxels: (values float32[n], votes uint32[n]) of the listed voxels of the DSI this mapper
        builds from `batch`, summed the way the reference sums (fp32, event order)."""
        voxels = _arr(voxels, np.uint32)
        values = np.empty(voxels.shape, np.float32)
        votes = np.empty(voxels.shape, np.uint32)
        _check(load_library().dsi_mapper_exact_voxels(self._h, batch._h, _ptr(voxels, C.c_uint32), voxels.size,
                                                      _ptr(values, C.c_float), _ptr(votes, C.c_uint32)))
        return values, votes

    def patchDepthMap(self, pixels, idx, conf):
        """dsi_mapper_patch_depth_map: overwrite pixels (y*Nx + x) of the raw depth map held on the device."""
        pixels, idx, conf = _arr(pixels, np.uint32), _arr(idx, np.uint8), _arr(conf, np.float32)
        _check(load_library().dsi_mapper_patch_depth_map(self._h, _ptr(pixels, C.c_uint32), _ptr(idx, C.c_uint8),
                                                         _ptr(conf, C.c_float), pixels.size))

    def computeDepthMapSharded(self, grid, comm):
        """Plane-sharded arg-max: local collapse of this rank's plane range, ONE all-reduce(MAX) of
        packed (confidence, index) keys, index -> depth over the full depth vector; fetchDepthMap()
        then returns the unsharded result on every rank."""
        _check(load_library().dsi_mapper_depth_map_sharded(self._h, (grid or self.dsi_)._h, comm._h))

    def computeDepthMapReduceScattered(self, acc, comm, mode, n_maps):
        """Temporal fusion's last step across GPUs with a reduce-scatter instead of an all-reduce: `acc` (this
        rank's accumulated slices, mode = ACC_*) is reduced by plane ranges, every rank finalises and arg-maxes
        its planes, one all-reduce(MAX) of packed keys -> the fused depth map on every rank (fetchDepthMap).
        `acc` is consumed."""
        _check(load_library().dsi_mapper_depth_map_reduce_scattered(self._h, acc._h, comm._h, int(mode), int(n_maps)))

    def computeDepthMapScatteredLocal(self, acc, nranks, rank, mode, n_maps):
        """The local step of computeDepthMapReduceScattered for a caller-provided transport: finalize + arg-max of
        the planes scatter_plan(dimZ, nranks, rank) gives this rank -> packed keys on the device (argmaxKeys)."""
        _check(load_library().dsi_mapper_depth_map_scattered_local(self._h, acc._h, int(nranks), int(rank), int(mode),
                                                                   int(n_maps)))

    def argmaxKeys(self):
        ny, nx = self.dsi_.shape[1:]
        keys = np.empty((ny, nx), np.uint64)
        _check(load_library().dsi_mapper_argmax_keys_download(self._h, _ptr(keys, C.c_uint64)))
        return keys

    def setArgmaxKeys(self, keys):
        keys = _arr(keys, np.uint64)
        ny, nx = self.dsi_.shape[1:]
        assert keys.shape == (ny, nx)
        _check(load_library().dsi_mapper_argmax_keys_upload(self._h, _ptr(keys, C.c_uint64)))

    def computeDepthMapFromKeys(self):
        """keys (after the caller's MAX over the ranks) -> confidence / index / depth (fetchDepthMap)."""
        _check(load_library().dsi_mapper_depth_map_from_keys(self._h))

    def fetchDepthMapAsync(self, depth, conf, idx):
        """Queue the device -> host copies into PinnedArray-backed arrays (any may be None) on the copy
        stream; they are valid after fetchWait()."""
        _check(load_library().dsi_mapper_fetch_depth_map_async(
            self._h, None if depth is None else _ptr(depth, C.c_float),
            None if conf is None else _ptr(conf, C.c_float), None if idx is None else _ptr(idx, C.c_uint8)))

    def fetchWait(self):
        _check(load_library().dsi_mapper_fetch_wait(self._h))

    def fetchDepthMapInOrder(self, depth, conf, idx):
        """Like fetchDepthMapAsync, but on the context's COMPUTE stream behind whatever it holds (no second stream per
        context: for pipelines with one context per window in flight); valid after fetchWait()."""
        _check(load_library().dsi_mapper_fetch_depth_map_in_order(
            self._h, None if depth is None else _ptr(depth, C.c_float),
            None if conf is None else _ptr(conf, C.c_float), None if idx is None else _ptr(idx, C.c_uint8)))

    def fetchDepthMap(self, in_order=False):
        """(depth, confidence, indices) on the host (synchronises).  in_order: the copies go on the compute stream (see
        fetchDepthMapInOrder) instead of the context's copy stream."""
        depth = np.empty((self.dimY, self.dimX), np.float32)
        conf = np.empty((self.dimY, self.dimX), np.float32)
        idx = np.empty((self.dimY, self.dimX), np.uint8)
        if in_order:
            self.fetchDepthMapInOrder(depth, conf, idx)
            self.fetchWait()                       # (pageable destinations: the runtime has copied by then)
            _check(load_library().dsi_context_synchronize(self.ctx._h))
            return depth, conf, idx
        _check(load_library().dsi_mapper_fetch_depth_map(self._h, _ptr(depth, C.c_float),
                                                         _ptr(conf, C.c_float), _ptr(idx, C.c_uint8)))
        return depth, conf, idx


def packetize(ts, trajectory, T_rv_w):
    """Host packetisation + pose pipeline of evaluateDSI (mapper_emvs_stereo.cpp:67-105).
    Returns (packet_first uint32[np], Rt float32[np][12]) or None when the reference
    would return false."""
    ts = _arr(ts, np.float64)
    times, poses = trajectory
    times, poses = _arr(times, np.float64), _arr(poses, np.float64).reshape(-1, 7)
    T = _arr(T_rv_w, np.float64)
    cap = ts.shape[0] // PACKET_SIZE + 1
    first = np.empty(cap, np.uint32)
    Rt = np.empty((cap, 12), np.float32)
    n = C.c_size_t()
    rc = load_library().dsi_packetize(_ptr(ts, C.c_double), ts.shape[0], _ptr(times, C.c_double),
                                      _ptr(poses, C.c_double), times.shape[0], _ptr(T, C.c_double),
                                      _ptr(first, C.c_uint32), _ptr(Rt, C.c_float), C.byref(n))
    if rc == ERR_TOO_FEW_EVENTS:
        return None
    _check(rc)
    return first[:n.value].copy(), Rt[:n.value].copy()


def packetize_strided(ts_view, trajectory, T_rv_w):
    """packetize() with the timestamps read where they lie: ts_view is a 1-D float64 view with any stride -- a field of
    a structured array of events, say -- and is not copied (dsi_packetize_strided: one timestamp per packet is read)."""
    if ts_view.dtype != np.float64 or ts_view.ndim != 1:
        raise ValueError("packetize_strided: a 1-D float64 view is required")
    n = int(ts_view.shape[0])
    stride = int(ts_view.strides[0]) if n else 8
    if stride < 8:
        raise ValueError("packetize_strided: stride %d is smaller than a timestamp" % stride)
    times, poses = trajectory
    times, poses = _arr(times, np.float64), _arr(poses, np.float64).reshape(-1, 7)
    T = _arr(T_rv_w, np.float64)
    cap = n // PACKET_SIZE + 1
    first = np.empty(cap, np.uint32)
    Rt = np.empty((cap, 12), np.float32)
    npk = C.c_size_t()
    rc = load_library().dsi_packetize_strided(C.c_void_p(ts_view.ctypes.data if n else 0), stride, n, _ptr(times, C.c_double),
                                              _ptr(poses, C.c_double), times.shape[0], _ptr(T, C.c_double),
                                              _ptr(first, C.c_uint32), _ptr(Rt, C.c_float), C.byref(npk))
    if rc == ERR_TOO_FEW_EVENTS:
        return None
    _check(rc)
    return first[:npk.value].copy(), Rt[:npk.value].copy()


def pose_at(trajectory, t):
    """LinearTrajectory::getPoseAt (trajectory.hpp:92-126); None when it returns false."""
    times, poses = trajectory
    times, poses = _arr(times, np.float64), _arr(poses, np.float64).reshape(-1, 7)
    out = np.empty(7, np.float64)
    rc = load_library().dsi_pose_at(_ptr(times, C.c_double), _ptr(poses, C.c_double), times.shape[0],
                                    float(t), _ptr(out, C.c_double))
    if rc == ERR_INVALID:
        return None
    _check(rc)
    return out
