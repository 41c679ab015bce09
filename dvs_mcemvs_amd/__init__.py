"""dvs_mcemvs_amd -- MI355X-native DSI construction / fusion / arg-max engine.

Thin ctypes binding over the C ABI in include/dsi_engine.h (libdsi_engine.so,
hand-written HIP for gfx950).  Class and method names follow the reference
(tub-rip/dvs_mcemvs): ``Grid3D`` (cartesian3dgrid.h) and ``MapperEMVS``
(mapper_emvs_stereo.hpp).  There is no CPU or PyTorch fallback anywhere in this
package: if the shared library is missing or no gfx950 GPU is visible, the
constructors raise.
"""
from .engine import (  # noqa: F401
    ACC_GM_TREE,
    ACC_INV_SUM,
    ACC_LOG_SUM,
    ACC_MAX,
    ACC_MIN,
    ACC_SQ_SUM,
    ACC_SUM,
    FUSE_AM,
    FUSE_GM,
    FUSE_HM,
    FUSE_MAX,
    FUSE_MIN,
    FUSE_RMS,
    PACKET_SIZE,
    REDUCE_MAX,
    REDUCE_MIN,
    REDUCE_SUM,
    VOTE_AUTO,
    VOTE_GLOBAL_ATOMIC,
    VOTE_LDS_BANDS,
    VOTE_FUSED_ARGMAX,
    Comm,
    Context,
    DsiError,
    EventBatch,
    Grid3D,
    MapperEMVS,
    OptionsDepthMap,
    PinnedArray,
    ShapeDSI,
    acc_reduce_op,
    allreduce_all,
    depth_map_reduce_scattered_all,
    depth_map_sharded_all,
    device_count,
    library_path,
    load_library,
    packetize,
    packetize_strided,
    pose_at,
)

__all__ = [
    "Comm", "allreduce_all", "depth_map_sharded_all", "depth_map_reduce_scattered_all", "Context", "Grid3D", "MapperEMVS", "ShapeDSI", "OptionsDepthMap", "EventBatch", "PinnedArray", "DsiError", "device_count",
    "library_path", "load_library", "packetize", "packetize_strided", "pose_at", "PACKET_SIZE",
    "FUSE_MIN", "FUSE_HM", "FUSE_GM", "FUSE_AM", "FUSE_RMS", "FUSE_MAX", "ACC_SUM", "ACC_INV_SUM", "ACC_LOG_SUM", "ACC_SQ_SUM", "ACC_MIN", "ACC_MAX", "ACC_GM_TREE",
    "REDUCE_SUM", "REDUCE_MIN", "REDUCE_MAX", "acc_reduce_op",
    "VOTE_AUTO", "VOTE_GLOBAL_ATOMIC", "VOTE_LDS_BANDS", "VOTE_FUSED_ARGMAX",
]
