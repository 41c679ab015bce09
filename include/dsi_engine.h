/*
 * dsi_engine.h -- C ABI of the MI355X-native DSI construction / fusion / arg-max
 * engine (libdsi_engine.so, hand-written HIP for gfx950).
 *
 * This is the drop-in boundary for the hot path of tub-rip/dvs_mcemvs:
 *     MapperEMVS::evaluateDSI -> fillVoxelGrid -> Grid3D::accumulateGridValueAt,
 *     Grid3D voxel-wise fusion, Grid3D::collapseMaxZSlice.
 * The reference has no FFI layer (it is one C++ process); each entry point below
 * names the reference C++ member it replaces (file:line relative to the
 * reference checkout).  include/dsi_engine.hpp re-creates the reference's class
 * and method names (Grid3D, EMVS::MapperEMVS, ...) on top of this ABI.
 *
 * Conventions
 *  - plain pointers and sizes only; every function returns a dsi_status_t
 *    (0 = ok) and never aborts or throws (the reference glog-CHECK-aborts);
 *    dsi_last_error() gives a thread-local message for the last failure.
 *  - volumes are fp32, layout volume[x + Nx*(y + Ny*z)] (cartesian3dgrid.h:34-35).
 *  - "host" pointers are ordinary CPU memory; "_dev" pointers are HIP device
 *    memory on the context's GPU.
 *  - all work of one context is issued on that context's HIP stream, in call
 *    order; functions that return host data synchronise, the others are
 *    asynchronous.  Distinct contexts may be used from distinct threads.
 *  - there is NO CPU fallback: creating a context fails when no gfx950 device
 *    (or no HIP device at all) is present.
 */
#ifndef DSI_ENGINE_H
#define DSI_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define DSI_API __attribute__((visibility("default")))
#else
#define DSI_API
#endif

#define DSI_ENGINE_ABI_VERSION 10
#define DSI_PACKET_SIZE 1024 /* mapper_emvs_stereo.hpp:152 packet_size_ */

typedef enum {
    DSI_OK = 0,
    DSI_ERR_INVALID = 1,       /* bad argument (null pointer, non-positive size, ...) */
    DSI_ERR_TOO_FEW_EVENTS = 2,/* evaluateDSI returns false: events.size() < 1024 (mapper_emvs_stereo.cpp:71-75) */
    DSI_ERR_HIP = 3,           /* a HIP runtime call failed */
    DSI_ERR_SHAPE = 4,         /* grids of different dimensions (reference: std::out_of_range from .at()) */
    DSI_ERR_BAD_OP = 5,        /* "Improper fusion method selected" (process1.cpp:155-157) */
    DSI_ERR_NO_DEVICE = 6,     /* no usable gfx950 GPU */
    DSI_ERR_CONTEXT = 7,       /* objects belong to different devices (or a batch to another context) */
    DSI_ERR_COMM = 8           /* RCCL could not be loaded or a collective call failed */
} dsi_status_t;

/* camera-fusion op codes = --stereo_fusion values (main.cpp:89, process1.cpp:136-158) */
typedef enum {
    DSI_FUSE_MIN = 1,  /* Grid3D::minTwoGrids            cartesian3dgrid.h:111-117 */
    DSI_FUSE_HM = 2,   /* Grid3D::harmonicMeanTwoGrids   cartesian3dgrid.h:119-127 */
    DSI_FUSE_GM = 3,   /* Grid3D::geometricMeanTwoGrids  cartesian3dgrid.h:150-156 */
    DSI_FUSE_AM = 4,   /* Grid3D::arithmeticMeanTwoGrids cartesian3dgrid.h:158-164 */
    DSI_FUSE_RMS = 5,  /* Grid3D::rmsTwoGrids            cartesian3dgrid.h:141-148 */
    DSI_FUSE_MAX = 6   /* Grid3D::maxTwoGrids            cartesian3dgrid.h:184-190 */
} dsi_fuse_op_t;

/* accumulate / finalize modes.  0 and 1 are the reference's temporal-fusion accumulators.
 * 2..5 are the n-ary forms of the 2-ary camera-fusion ops, which the reference does not have:
 * it applies GM / AM / RMS to two cameras only and silently ignores a third one
 * (process1.cpp:169-191).  They follow SURVEY.md 8(d) cfg 5 / 8(e): each is an accumulation
 * that is a plain sum, min or max over the maps -- i.e. directly an RCCL ncclSum / ncclMin /
 * ncclMax all-reduce across GPUs (dsi_acc_reduce_op) -- followed by a local finalize:
 *   LOG_SUM  acc += log v            finalize exp(acc / n)    geometric mean; 0 if any v is 0
 *   SQ_SUM   acc += v*v              finalize sqrt(acc / n)   root mean square
 *   MIN/MAX  acc = min/max(acc, v)   finalize: nothing
 * For n = 2: MIN / MAX / SUM(AM) equal the reference's 2-ary ops bit for bit; LOG_SUM equals
 * sqrt(a*g) (geometricMeanTwoGrids) within 16 ulp for values in [2^-20, 2^20] (1e-5 relative
 * over the whole float range: the accumulator is an fp32 sum of fp32 logarithms) and SQ_SUM
 * equals rmsTwoGrids within 2 ulp; tests/test_gpu_parity.py states and checks both.  log and exp
 * are spelled out in IEEE double operations, so GPU and CPU oracle agree bit for bit. */
typedef enum {
    DSI_ACC_SUM = 0,     /* addTwoGrids / computeAMfromSum            cartesian3dgrid.h:64-70, 87-93 */
    DSI_ACC_INV_SUM = 1, /* addInverseOfTwoGrids / computeHMfromSumOfInv cartesian3dgrid.h:72-86 */
    DSI_ACC_LOG_SUM = 2, /* n-ary geometricMeanTwoGrids               cartesian3dgrid.h:150-156 */
    DSI_ACC_SQ_SUM = 3,  /* n-ary rmsTwoGrids                         cartesian3dgrid.h:141-148 */
    DSI_ACC_MIN = 4,     /* n-ary minTwoGrids                         cartesian3dgrid.h:111-117 */
    DSI_ACC_MAX = 5,     /* n-ary maxTwoGrids                         cartesian3dgrid.h:184-190 */
    /* Only for the one-pass forms dsi_grid_fuse_n / dsi_mapper_depth_map_of_fusion_n with n = 2, 4 or 8 (not an
     * accumulation: dsi_grid_accumulate* / _finalize reject it): the geometric mean as the balanced TREE of the
     * reference's own 2-ary op, sqrt(sqrt(a b) sqrt(c d)) for n = 4 -- what calling geometricMeanTwoGrids
     * (cartesian3dgrid.h:150-156) on pairs of grids and then on the results gives, bit for bit (n = 2 IS that
     * member).  A plain HBM stream (~2 fp32 operations per source), where LOG_SUM costs ~100 fp64 operations. */
    DSI_ACC_GM_TREE = 6
} dsi_acc_mode_t;

/* how an accumulator of a given mode combines across GPUs */
typedef enum { DSI_REDUCE_SUM = 0, DSI_REDUCE_MIN = 1, DSI_REDUCE_MAX = 2 } dsi_reduce_op_t;

/* which voting kernel MapperEMVS::fillVoxelGrid is replaced by */
typedef enum {
    DSI_VOTE_AUTO = 0,
    DSI_VOTE_GLOBAL_ATOMIC = 1, /* thread = event, global_atomic_add_f32 into the DSI */
    DSI_VOTE_LDS_BANDS = 2,     /* plane x row-band privatised in LDS as Q33.31 fixed point (ds_add_u64), coalesced flush */
    DSI_VOTE_FUSED_ARGMAX = 3   /* reported only: the same bands, fused with the camera fusion and the arg-max
                                   (dsi_mapper_depth_map_of_events); not selectable with dsi_mapper_set_vote_algo */
} dsi_vote_algo_t;

typedef struct dsi_context dsi_context_t; /* one GPU + one HIP stream + scratch */
typedef struct dsi_grid dsi_grid_t;       /* device-resident Grid3D */
typedef struct dsi_mapper dsi_mapper_t;   /* device-resident MapperEMVS */
typedef struct dsi_batch dsi_batch_t;     /* device-resident, packetised events of one evaluateDSI call */

DSI_API const char *dsi_last_error(void);
DSI_API int dsi_abi_version(void);
/* 0: the production library (libdsi_engine.so).  1: the EXPERIMENTS flavour (libdsi_engine_experiments.so, built with
 * -DDSI_TIMING_EXPERIMENTS by `python -m dvs_mcemvs_amd.build --experiments`), which also reads the environment knobs
 * of the timing experiments quoted in NOTEBOOK.md and exports dsi_test_* hooks -- some of which make the DSIs WRONG on
 * purpose; never ship or benchmark it.  The production library reads no environment and exports no dsi_test_*. */
DSI_API int dsi_build_flavour(void);
/* number of HIP devices visible, or 0 (never fails) */
DSI_API int dsi_device_count(void);

/* ------------------------------------------------------------------ context */
DSI_API int dsi_context_create(int device_id, dsi_context_t **out);
/* fails with DSI_ERR_CONTEXT (and destroys nothing) while grids, mappers or batches created from the context are
 * alive: each holds a pointer to it.  Destroy them first. */
DSI_API int dsi_context_destroy(dsi_context_t *ctx);
DSI_API int dsi_context_synchronize(dsi_context_t *ctx);
/* the hipStream_t all work of this context is issued on */
DSI_API void *dsi_context_stream(dsi_context_t *ctx);
DSI_API int dsi_context_device(dsi_context_t *ctx);
/* device-side ordering between two contexts (streams) of one GPU: work queued on ctx after this call
 * starts only when everything queued on `other` before this call has finished.  The host does not
 * wait.  (The reference is single-threaded; this lets each MapperEMVS own a stream, e.g. the two
 * cameras of process_1, and meet at the fusion.) */
DSI_API int dsi_context_wait_for(dsi_context_t *ctx, dsi_context_t *other);
/* HIP-event stopwatch on the context's stream (bench.py: the replacement of the
 * std::chrono timers at process1.cpp:72-85,132-166). stop synchronises. */
DSI_API int dsi_context_timer_start(dsi_context_t *ctx);
DSI_API int dsi_context_timer_stop(dsi_context_t *ctx, float *elapsed_ms);
/* a timeline of HIP events on the context's stream: _mark records one (asynchronous); _read waits for the last
 * one, returns the n-1 intervals between consecutive marks in milliseconds (the first `capacity` of them) and
 * clears the timeline.  bench.py marks every step to report the spread of the per-step times. */
DSI_API int dsi_context_timeline_mark(dsi_context_t *ctx);
DSI_API int dsi_context_timeline_read(dsi_context_t *ctx, float *intervals_ms, size_t capacity, size_t *n_intervals);

/* ------------------------------------------------------------------- Grid3D */
/* Grid3D::Grid3D(dimX,dimY,dimZ) + allocate + resetGrid (cartesian3dgrid.cpp:30-46) */
DSI_API int dsi_grid_create(dsi_context_t *ctx, int nx, int ny, int nz, dsi_grid_t **out);
/* same, over caller-owned device memory of nx*ny*nz floats, 16-byte aligned (not zeroed, not freed) */
DSI_API int dsi_grid_wrap(dsi_context_t *ctx, int nx, int ny, int nz, void *data_dev, dsi_grid_t **out);
DSI_API int dsi_grid_destroy(dsi_grid_t *g);
/* Grid3D::getDimensions (cartesian3dgrid.h:230-235) */
DSI_API int dsi_grid_dims(const dsi_grid_t *g, int *nx, int *ny, int *nz);
/* Grid3D::resetGrid (cartesian3dgrid.cpp:67-70) */
DSI_API int dsi_grid_reset(dsi_grid_t *g);
/* Grid3D::getPointerToSlice(0) (cartesian3dgrid.h:237-240), as a device pointer */
DSI_API void *dsi_grid_device_ptr(dsi_grid_t *g);
DSI_API int dsi_grid_upload(dsi_grid_t *g, const float *host);
DSI_API int dsi_grid_download(dsi_grid_t *g, float *host);
/* dst = op(dst, src), in place, op in 1..6 (cartesian3dgrid.h:111-190) */
DSI_API int dsi_grid_fuse2(dsi_grid_t *dst, const dsi_grid_t *src, int op);
/* dst = op(a, b) in one pass: exactly what the reference's sequence
 * "dst.resetGrid(); dst.addTwoGrids(a); dst.<op>TwoGrids(b)" (process1.cpp:126-158,
 * process2.cpp:159-189) leaves in dst, without the two extra sweeps over the volume */
DSI_API int dsi_grid_fuse2_into(dsi_grid_t *dst, const dsi_grid_t *a, const dsi_grid_t *b, int op);
/* Grid3D::harmonicMeanTwoGrids(grid2, n) (cartesian3dgrid.h:130-139) */
DSI_API int dsi_grid_fuse_hm_n(dsi_grid_t *dst, const dsi_grid_t *src, int n);
/* dst = identity element of the mode: 0 for the four sums (what the reference's resetGrid()
 * before the temporal loop does, process2.cpp:203), +inf for MIN, -inf for MAX */
DSI_API int dsi_grid_accumulate_begin(dsi_grid_t *dst, int mode);
/* Grid3D::addTwoGrids / addInverseOfTwoGrids (cartesian3dgrid.h:64-78) and the n-ary modes,
 * mode = dsi_acc_mode_t */
DSI_API int dsi_grid_accumulate(dsi_grid_t *dst, const dsi_grid_t *src, int mode);
/* Grid3D::computeAMfromSum / computeHMfromSumOfInv (cartesian3dgrid.h:80-93) and the n-ary modes */
DSI_API int dsi_grid_finalize(dsi_grid_t *dst, int mode, int n);
/* dst = n-ary fusion of srcs[0..n) in ONE pass over the volumes: exactly dsi_grid_accumulate_begin(dst,
 * mode); dsi_grid_accumulate(dst, srcs[i], mode) for i = 0..n-1; dsi_grid_finalize(dst, mode, n) -- same
 * operations in the same order, hence the same bits -- with (n+1)*4 instead of (3n+2)*4 bytes of
 * traffic per voxel.  2 <= n <= 8; dst must not be one of the sources. */
DSI_API int dsi_grid_fuse_n(dsi_grid_t *dst, const dsi_grid_t *const *srcs, int n, int mode);
/* dsi_reduce_op_t of a mode (never fails for a valid mode; -1 otherwise) */
DSI_API int dsi_acc_reduce_op(int mode);
/* Grid3D::collapseMaxZSlice (cartesian3dgrid.cpp:115-137): conf[ny*nx] f32,
 * idx[ny*nx] u8, first maximum wins.  Host outputs; synchronises. */
DSI_API int dsi_grid_collapse_max_z(dsi_grid_t *g, float *conf_host, uint8_t *idx_host);
/* same, device outputs, asynchronous.  planes_dev (nz floats) + depth_dev may be
 * NULL; when given, depth_dev[p] = planes_dev[idx[p]]
 * (MapperEMVS::convertDepthIndicesToValues, mapper_emvs_stereo.cpp:302-313). */
DSI_API int dsi_grid_collapse_max_z_dev(dsi_grid_t *g, float *conf_dev, uint8_t *idx_dev,
                                const float *planes_dev, float *depth_dev);
/* Grid3D::computeMeanSquare (cartesian3dgrid.cpp:164-174), accumulated in double */
DSI_API int dsi_grid_mean_square(dsi_grid_t *g, double *out);

/* ---------------------------------------------------------- multi-GPU (RCCL) */
/* The temporal fusion of process_2 (process2.cpp:211-242: one accumulate per sub-interval, one
 * finalize) and of the sliding window (main.cpp:177) shards by time slice: every GPU accumulates
 * its slices locally, then ONE all-reduce of the accumulator over xGMI replaces the rest of the
 * loop.  The engine calls RCCL itself (librccl is loaded on first use; nothing
 * above this ABI needs torch or MPI).  Two ways to form a communicator:
 *   - one process, n GPUs (the reference is one process): dsi_comm_create_all over n contexts on n
 *     distinct devices; collectives through the *_all entry points (one group call);
 *   - one process per GPU: rank 0 calls dsi_comm_unique_id and hands the 128 bytes to the other
 *     ranks by any side channel (file, socket, MPI, a torch.distributed store), then every rank
 *     calls dsi_comm_create_rank.
 * A collective on a grid is issued on that grid's context stream, ordered like any other grid
 * operation; the host does not wait. */
typedef struct dsi_comm dsi_comm_t;
#define DSI_COMM_ID_BYTES 128
DSI_API int dsi_comm_unique_id(uint8_t id[DSI_COMM_ID_BYTES]);
DSI_API int dsi_comm_create_rank(dsi_context_t *ctx, const uint8_t id[DSI_COMM_ID_BYTES], int nranks, int rank,
                                 dsi_comm_t **out);
/* out receives n communicators, out[i] on contexts[i]'s device (n distinct devices) */
DSI_API int dsi_comm_create_all(dsi_context_t *const *contexts, int n, dsi_comm_t **out);
DSI_API int dsi_comm_destroy(dsi_comm_t *comm);
DSI_API int dsi_comm_rank(const dsi_comm_t *comm);
DSI_API int dsi_comm_size(const dsi_comm_t *comm);
/* what RCCL ITSELF reports for this communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), as opposed
 * to what it was created with: bench.py prints it per rank as the proof of the rank count.  Any pointer may be NULL. */
DSI_API int dsi_comm_query(const dsi_comm_t *comm, int *nranks, int *rank, int *device);
/* in-place all-reduce of a grid over the communicator, op = dsi_reduce_op_t (use
 * dsi_acc_reduce_op(mode) for an accumulator).  The grid must live on the communicator's device. */
DSI_API int dsi_grid_allreduce(dsi_comm_t *comm, dsi_grid_t *g, int op);
/* the same for the n ranks of one process (comms from dsi_comm_create_all, grids[i] on comms[i]) */
DSI_API int dsi_grid_allreduce_all(dsi_comm_t *const *comms, dsi_grid_t *const *grids, int n, int op);

/* --------------------------------------------------------------- MapperEMVS */
typedef struct {
    int sensor_width;   /* cam.fullResolution() (mapper_emvs_stereo.cpp:34-36) */
    int sensor_height;
    float K[4];         /* fx, fy, cx, cy of the projection matrix (mapper_emvs_stereo.cpp:46-48) */
    int dim_x;          /* ShapeDSI (mapper_emvs_stereo.hpp:40-65); 0 = sensor size (:216-217) */
    int dim_y;
    int dim_z;          /* <= 256 (main.cpp:156) */
    float min_depth;
    float max_depth;
    float fov_deg;      /* < 10: use the camera focal length (mapper_emvs_stereo.cpp:219-229) */
    int inverse_depth;  /* 0: LinearDepthVector (CMake default), 1: -DUSE_INVERSE_DEPTH */
    const float *lut;   /* host, 2*W*H floats, entry y*W+x = undistorted (u,v)
                           (precomputeRectifiedPoints, mapper_emvs_stereo.cpp:256-299); NULL = identity */
    /* Plane sharding (multi-GPU): this mapper owns planes
     * [plane_begin, plane_begin + plane_count) of the dim_z planes defined above; its DSI has
     * plane_count planes.  z0 stays plane 0 of the FULL depth vector (mapper_emvs_stereo.cpp:111,163)
     * and planes are independent in fillVoxelGrid (:168), so the shard's DSI equals those planes of
     * the unsharded DSI bit for bit.  plane_count == 0: all planes from plane_begin on. */
    int plane_begin;
    int plane_count;
} dsi_mapper_config_t;

/* MapperEMVS::MapperEMVS(cam, dsi_shape) (mapper_emvs_stereo.cpp:29-64, setupDSI :208-241) */
DSI_API int dsi_mapper_create(dsi_context_t *ctx, const dsi_mapper_config_t *cfg, dsi_mapper_t **out);
DSI_API int dsi_mapper_destroy(dsi_mapper_t *m);
/* public member MapperEMVS::dsi_ (mapper_emvs_stereo.hpp:116); owned by the mapper */
DSI_API dsi_grid_t *dsi_mapper_grid(dsi_mapper_t *m);
/* virtual camera {fx,fy,cx,cy} (:231-239), plane depths raw_depths_vec_ (:213-214;
 * raw_depths needs dim_z floats) and the resolved dimensions; any pointer may be NULL */
DSI_API int dsi_mapper_geometry(const dsi_mapper_t *m, float *Kv, float *raw_depths, int *nx, int *ny, int *nz);
/* first plane this mapper owns (0 unless plane-sharded); raw_depths / nz above describe the owned planes */
DSI_API int dsi_mapper_plane_begin(const dsi_mapper_t *m);
/* the WHOLE depth vector of the ShapeDSI (dim_z floats; global arg-max index -> depth of a plane-sharded
 * DSI); either pointer may be NULL */
DSI_API int dsi_mapper_full_depths(const dsi_mapper_t *m, float *raw_depths, int *dim_z);
DSI_API int dsi_mapper_set_vote_algo(dsi_mapper_t *m, int algo);
/* tuning knobs of DSI_VOTE_LDS_BANDS; 0 = automatic */
DSI_API int dsi_mapper_set_band_params(dsi_mapper_t *m, int band_rows, int chunks, int block_threads);
/* DSI_VOTE_LDS_BANDS has two lane mappings: one packet's run per wave pass (long runs) or the
 * runs of 64 packets packed back to back into the lanes (short runs: wide / tall grids).
 * mode -1 = automatic (by expected run length: 7 or 5), 0 = per packet, 1 = packed,
 * 2 = groups of consecutive packets sorted together (one long run per group),
 * 3 = packed with the compiled (not hand-scheduled) wave loop, for A/B tests,
 * 4 = groups with the hand-scheduled wave loop (long runs for wide grids),
 * 5 = packed with a vector (prefix-sum + tail-bit) slot -> record mapping instead of the scalar run
 *     bookkeeping (short runs: wide grids), hand-scheduled batches,
 * 6 = mapping 5 all compiled, for A/B tests,
 * 7 = mapping 1 with DEALT passes: the waves of a workgroup draw their passes from a counter instead of taking
 *     every 16th one (the automatic choice wherever mapping 1 used to be chosen),
 * 8 = OPT-IN, never chosen automatically: mapping 7 on PAIRED 32-bit cells -- one 64-bit LDS atomic updates the two cells
 *     (x, x + 1) of a row, a vote is two atomics instead of four (the voting kernels are bound by the LDS atomic unit).
 *     The price is the numerical contract: the weights of cartesian3dgrid.h:261-270 are added as ROUNDED Q.19 integers
 *     (error <= 2^-20 per vote, about that of the reference's own fp32 "+=") instead of the exact Q33.31 sums every other
 *     mapping keeps -- a record whose weight on a cell is below 2^-20 adds nothing, so bursts of events on one sub-pixel
 *     location can lose up to 2^-20 per record on the neighbouring cells --, the DSI is no longer bit-identical to
 *     theirs, and a 32-bit cell holds 8,192 full votes per work
 *     item: the voting kernel reports any cell that reached HALF of that (dsi_mapper_paired_overflow), in which case the
 *     caller repeats the call with another mapping.  1024-thread workgroups, the evaluate / fillVoxelGrid path only. */
DSI_API int dsi_mapper_set_packed_lanes(dsi_mapper_t *m, int mode);
/* lane mapping 8: *overflowed <- 1 if a paired cell of the last vote reached 2^31 (4,096 full votes: half its capacity;
 * beyond 8,192 a cell wraps, so a set flag means "the DSI may be wrong, vote again with an exact mapping").  0 for every
 * other mapping.  Synchronises. */
DSI_API int dsi_mapper_paired_overflow(dsi_mapper_t *m, int *overflowed);
/* Lane mappings 5 / 6 (wide grids) read, per (band, plane, packet), the run of the packet's records the band must look
 * at.  As a table that is bands x planes x packets words -- 6.1 GB per camera at 1024 x 1024 x 256 with 100 M events, 3.9 ms to
 * write -- so from min_packets packets per call on the voting kernel derives the runs itself, per pass, from the
 * packets' transposed row tables (the same runs: fillVoxelGrid's loop mapper_emvs_stereo.cpp:168-203 visits the same
 * events either way, and the DSI is bit-identical).  min_packets < 0 = the default (8192), 0 = always, a huge value = never. */
DSI_API int dsi_mapper_set_inline_cuts(dsi_mapper_t *m, long long min_packets);

/* MapperEMVS::fillVoxelGrid(event_locations_z0, camera_centers)
 * (mapper_emvs_stereo.cpp:151-205) -- the exact-parity test point.  xy_z0: 2 floats
 * per event (elements [0],[1] of each Vector4f), n_packets*1024 events; centers: 3
 * floats per packet.  Host inputs.  Accumulates into the mapper's grid WITHOUT
 * resetting it (the reset is evaluateDSI's, :145). */
DSI_API int dsi_mapper_fill_voxel_grid(dsi_mapper_t *m, const float *xy_z0, const float *centers, size_t n_packets);

/* Device-resident input of one evaluateDSI call: raw events (pixel x,y; polarity and
 * timestamps are not needed past packetisation) plus the packetisation done by
 * mapper_emvs_stereo.cpp:88-105: packet k covers events
 * [packet_first[k], packet_first[k]+1024) (packet_first == NULL: k*1024) and has
 * pose Rt[12k..12k+11] = R (row-major 3x3) then t of T_ev_rv, already cast to float.
 * An event whose pixel lies outside the sensor of the mapper that evaluates the batch has no
 * rectification-LUT entry (the reference would index past its matrix, :134); with a LUT it is
 * dropped -- it votes on no plane -- instead of being looked up out of bounds. */
DSI_API int dsi_batch_create(dsi_context_t *ctx, const uint16_t *x, const uint16_t *y, size_t n_events,
                     const uint32_t *packet_first, const float *Rt, size_t n_packets, dsi_batch_t **out);
/* Host-fed streams (a 50 ms window every 50 ms, main.cpp:177): page-locked host memory lets the
 * upload run as one DMA without the runtime's staging copies, and lets the host go on while it
 * runs.  dsi_host_alloc returns such memory (hipHostMalloc; any context of the process may use it).
 * dsi_batch_create_async is dsi_batch_create for arrays that live in it: it returns as soon as the
 * copies are queued on the copy stream.  The arrays must stay unchanged until the upload is done --
 * dsi_batch_uploaded(b) (non-blocking: 1 done, 0 in flight), or any later synchronisation of a
 * context that evaluated the batch. */
DSI_API int dsi_host_alloc(size_t bytes, void **out);
DSI_API int dsi_host_free(void *p);
DSI_API int dsi_batch_create_async(dsi_context_t *ctx, const uint16_t *x, const uint16_t *y, size_t n_events,
                           const uint32_t *packet_first, const float *Rt, size_t n_packets, dsi_batch_t **out);
DSI_API int dsi_batch_uploaded(const dsi_batch_t *b);
DSI_API int dsi_batch_destroy(dsi_batch_t *b);
DSI_API size_t dsi_batch_num_packets(const dsi_batch_t *b);

/* MapperEMVS::evaluateDSI past the pose lookup: per-packet camera centre and H_z0
 * (:108-126), per-event LUT + z0 warp (:129-142), resetGrid (:145), fillVoxelGrid
 * (:146).  Asynchronous on the context's stream. */
DSI_API int dsi_mapper_evaluate_batch(dsi_mapper_t *m, const dsi_batch_t *batch);

/* MapperEMVS::evaluateDSI(events, trajectory, T_rv_w) (mapper_emvs_stereo.cpp:67-148),
 * host inputs.  Events: pixel coordinates and timestamps in seconds, time-sorted.
 * Trajectory: n_poses control poses T_w_c as {tx,ty,tz,qw,qx,qy,qz} at ascending
 * times (LinearTrajectory, trajectory.hpp:81-128; SE(3) interpolation with
 * minkindr semantics).  T_rv_w: same 7-double layout.  Returns
 * DSI_ERR_TOO_FEW_EVENTS where the reference returns false.  n_voted (optional)
 * receives the number of events actually voted (packets * 1024). */
DSI_API int dsi_mapper_evaluate(dsi_mapper_t *m, const uint16_t *x, const uint16_t *y, const double *ts,
                        size_t n_events, const double *traj_times, const double *traj_poses,
                        size_t n_poses, const double *T_rv_w, size_t *n_voted);

/* host-side packetisation + pose pipeline of evaluateDSI, exposed so that callers can
 * build a dsi_batch_t once and keep it resident.  packet_first (capacity
 * n_events/1024+1) and Rt (12 floats per packet) are outputs; *n_packets the count.
 * Returns DSI_ERR_TOO_FEW_EVENTS for n_events < 1024. */
DSI_API int dsi_packetize(const double *ts, size_t n_events, const double *traj_times,
                  const double *traj_poses, size_t n_poses, const double *T_rv_w,
                  uint32_t *packet_first, float *Rt, size_t *n_packets);
/* the same with the timestamps read where they lie in an array of structs (the reference holds
 * std::vector<dvs_msgs::Event>, mapper_emvs_stereo.cpp:91: events[current_event_ + packet_size_/2].ts): event i's
 * timestamp is the double at ts_first + i * stride_bytes.  Only one timestamp per packet is read (mapper_emvs_stereo.cpp:88-99),
 * so a caller need not copy n_events doubles out of its structs first. */
DSI_API int dsi_packetize_strided(const void *ts_first, size_t stride_bytes, size_t n_events, const double *traj_times,
                          const double *traj_poses, size_t n_poses, const double *T_rv_w,
                          uint32_t *packet_first, float *Rt, size_t *n_packets);
/* LinearTrajectory::getPoseAt (trajectory.hpp:92-126); returns DSI_ERR_INVALID when
 * the reference returns false (no extrapolation). out = 7 doubles. */
DSI_API int dsi_pose_at(const double *traj_times, const double *traj_poses, size_t n_poses, double t, double *out);

/* arg-max + depth of the mapper's own grid: collapseMaxZSlice (cartesian3dgrid.cpp:115-137)
 * + convertDepthIndicesToValues (mapper_emvs_stereo.cpp:302-313) applied to the raw
 * (pre-filter) indices.  Host outputs (any may be NULL); synchronises. */
DSI_API int dsi_mapper_depth_map(dsi_mapper_t *m, float *depth_host, float *conf_host, uint8_t *idx_host);
/* the same for any grid of the mapper's shape (e.g. a fused DSI); asynchronous variant
 * keeps results in the mapper's device buffers until dsi_mapper_fetch_depth_map */
DSI_API int dsi_mapper_depth_map_of(dsi_mapper_t *m, dsi_grid_t *g);
/* depth map of op(a, b) (op = dsi_fuse_op_t) WITHOUT materialising the fused DSI: exactly what
 * "fused.resetGrid(); fused.addTwoGrids(a); fused.<op>TwoGrids(b)" (process1.cpp:126-158) followed by
 * dsi_mapper_depth_map_of(m, fused) gives, bit for bit, in one pass over a and b.  For streams of
 * windows that only keep the depth map (main.cpp:177-302). */
DSI_API int dsi_mapper_depth_map_of_fusion(dsi_mapper_t *m, const dsi_grid_t *a, const dsi_grid_t *b, int op);
/* the n-ary form: depth map of dsi_grid_fuse_n(srcs, n, mode) (mode = dsi_acc_mode_t; the n-camera
 * generalisation of process1.cpp:126-191, SURVEY 8d cfg 5) without materialising the fused DSI --
 * the same per-voxel operations in the same order, so the same bits as fuse-then-collapse. */
DSI_API int dsi_mapper_depth_map_of_fusion_n(dsi_mapper_t *m, const dsi_grid_t *const *srcs, int n, int mode);
/* The depth map of n = 1, 2 or 3 cameras' events WITHOUT building their DSIs: bit for bit what
 *     for c in 0..n-1: dsi_mapper_evaluate_batch(mappers[c], batches[c]);                 (process1.cpp:76-117)
 *     n == 2: dsi_mapper_depth_map_of_fusion(out, grid(mappers[0]), grid(mappers[1]), op);  (:126-166, :222 -> mapper_emvs_stereo.cpp:368)
 *     n == 1: dsi_mapper_depth_map_of(out, grid(mappers[0]))
 *     n == 3: the trinocular rig of process1.cpp:105-117, :169-191 -- fused = op(grid 0, grid 1), then
 *             op 1: min(fused, grid 2), op 2: harmonicMeanTwoGrids(grid 2, 3), op 6: max(fused, grid 2),
 *             ops 3, 4, 5: camera 2 is ignored like the reference does ("case 3: break;"), its events are
 *             not even voted -- followed by dsi_mapper_depth_map_of(out, fused)
 * leaves in out's depth-map buffers (dsi_mapper_fetch_depth_map*), computed by one kernel that votes a
 * (band, plane) of each camera into LDS, applies the 2-ary op per voxel and keeps the running arg-max
 * in registers -- no DSI is written or read (at 512x512x200 a window saves 840 MB of HBM traffic).  For
 * streams that keep only the depth maps (the --full_seq loop, main.cpp:177-302).  The mappers' grids
 * are NOT touched (they keep whatever an earlier evaluate left there); `out` may be one of the
 * mappers; all must share one context and one DSI shape; the cameras need distinct mappers.  op =
 * dsi_fuse_op_t (ignored for n == 1).  A batch without packets stands for an all-zero DSI (evaluateDSI
 * returned false, mapper_emvs_stereo.cpp:71-75).  Asynchronous on the context's stream. */
DSI_API int dsi_mapper_depth_map_of_events(dsi_mapper_t *out, dsi_mapper_t *const *mappers,
                                           const dsi_batch_t *const *batches, int n, int op);
/* The same for the n-camera rigs the reference's process_1 has no switch for (BASELINE configs[4]: four cameras,
 * geometric mean): bit for bit what
 *     for c in 0..n-1: dsi_mapper_evaluate_batch(mappers[c], batches[c]);
 *     dsi_mapper_depth_map_of_fusion_n(out, {grid(mappers[c])}, n, mode)
 * leaves in out's depth-map buffers, by the one kernel that votes, fuses and keeps the running arg-max -- no DSI written
 * (at 1024 x 1024 x 256: 4 GiB not written and not read back).  mode = DSI_ACC_GM_TREE, the balanced tree of
 * Grid3D::geometricMeanTwoGrids (cartesian3dgrid.h:150-156): sqrt(sqrt(c0 c1) sqrt(c2 c3)); n = 4, or 2 (where the tree IS
 * the reference's 2-ary op).  Everything else as dsi_mapper_depth_map_of_events. */
DSI_API int dsi_mapper_depth_map_of_events_n(dsi_mapper_t *out, dsi_mapper_t *const *mappers,
                                             const dsi_batch_t *const *batches, int n, int mode);
DSI_API int dsi_mapper_fetch_depth_map(dsi_mapper_t *m, float *depth_host, float *conf_host, uint8_t *idx_host);
/* the same without waiting: the copies are queued (on the context's copy stream, behind the arg-max
 * only -- not behind later work of the compute stream); the outputs (page-locked memory from
 * dsi_host_alloc) are valid after dsi_mapper_fetch_wait */
DSI_API int dsi_mapper_fetch_depth_map_async(dsi_mapper_t *m, float *depth_host, float *conf_host,
                                             uint8_t *idx_host);
/* the same copies queued on the context's COMPUTE stream, in order behind whatever it already holds -- for pipelines that
 * give every window in flight its own context (main.cpp:177 as a stream: include/dsi_process.hpp full_sequence_depth_maps)
 * and fetch right after the window's kernels: no second stream per context (the runtime maps streams onto four hardware
 * queues by default; contexts with two streams each made windows' uploads and copies queue behind other windows' kernels).
 * Valid after dsi_mapper_fetch_wait, like the _async form. */
DSI_API int dsi_mapper_fetch_depth_map_in_order(dsi_mapper_t *m, float *depth_host, float *conf_host, uint8_t *idx_host);
DSI_API int dsi_mapper_fetch_wait(dsi_mapper_t *m);

/* Plane sharding (one DSI too big or too slow for one GPU): every rank's mapper owns a
 * plane range (dsi_mapper_config_t.plane_begin / plane_count) and grid g holds that range of the
 * (fused) DSI.  collapseMaxZSlice of the whole DSI = local collapse, then ONE all-reduce(MAX) of
 * 64-bit keys (confidence bits << 8 | 255 - global plane index: larger confidence wins, on equal
 * confidence the smaller index -- the first-maximum rule of std::max_element,
 * cartesian3dgrid.cpp:115-137), then index -> depth over the FULL depth vector.  Results stay in
 * the mapper's device buffers (dsi_mapper_fetch_depth_map), identical on every rank. */
DSI_API int dsi_mapper_depth_map_sharded(dsi_mapper_t *m, dsi_grid_t *g, dsi_comm_t *comm);
DSI_API int dsi_mapper_depth_map_sharded_all(dsi_mapper_t *const *mappers, dsi_grid_t *const *grids,
                                             dsi_comm_t *const *comms, int n);

/* The last step of the temporal fusion across GPUs (process2.cpp:211-242 sharded by time slice) with half the
 * xGMI bytes of dsi_grid_allreduce + dsi_grid_finalize + dsi_mapper_depth_map_of: the accumulator `acc` (every
 * rank's local dsi_grid_accumulate results, mode = dsi_acc_mode_t) is REDUCE-SCATTERED by planes -- rank r of n
 * receives the reduced planes [r q, (r+1) q), q = dimZ / n; the dimZ mod n planes left over are all-reduced --,
 * each rank finalises (mode, n_maps) and arg-maxes the planes it owns, and one all-reduce(MAX) of the packed
 * (confidence, plane) keys (8 bytes per pixel) gives every rank the depth map of the fused DSI
 * (dsi_mapper_fetch_depth_map*).  `acc` is consumed: afterwards only the owned planes hold (finalised) values.
 * The mapper must own the whole depth vector.  (reduce-scatter + local finalize / arg-max +
 * gather of the small maps.) */
DSI_API int dsi_mapper_depth_map_reduce_scattered(dsi_mapper_t *m, dsi_grid_t *acc, dsi_comm_t *comm, int mode,
                                                  int n_maps);
DSI_API int dsi_mapper_depth_map_reduce_scattered_all(dsi_mapper_t *const *mappers, dsi_grid_t *const *accs,
                                                      dsi_comm_t *const *comms, int n, int mode, int n_maps);

/* The same three steps with the transport left to the caller, and the partition arithmetic they share (pure host
 * functions, no GPU needed): dsi_mapper_depth_map_reduce_scattered is
 *     reduce (sum / min / max over the ranks) of planes [own_begin, +own_count) to their owner and of planes
 *         [tail_begin, +tail_count) to everybody                                  <- dsi_scatter_plan, ncclReduceScatter + ncclAllReduce
 *     dsi_mapper_depth_map_scattered_local: finalize + arg-max of the owned planes -> packed keys
 *     MAX of the keys over the ranks                                              <- ncclAllReduce(MAX)
 *     dsi_mapper_depth_map_from_keys: keys -> confidence / index / depth
 * so that a caller with another transport (the 2-rank tests carry the two exchanges through host memory; MPI) runs
 * exactly the code the RCCL path runs around its collectives. */
typedef struct {
    int q;          /* planes per rank of the reduce-scatter part: dimZ / nranks */
    int own_begin;  /* this rank's reduce-scatter range: [rank q, (rank + 1) q) */
    int own_count;
    int tail_begin; /* the dimZ mod nranks planes left over: all-reduced, every rank owns them */
    int tail_count;
} dsi_scatter_plan_t;
DSI_API int dsi_scatter_plan(int dim_z, int nranks, int rank, dsi_scatter_plan_t *out);
/* plane sharding (dsi_mapper_config_t.plane_begin / plane_count): contiguous balanced ranges, the first
 * dimZ mod nranks ranks own one plane more */
DSI_API int dsi_plane_range(int dim_z, int nranks, int rank, int *begin, int *count);
/* the arg-max key of a pixel, confidence bits << 8 | 255 - (idx_local + plane_begin), on host arrays: what
 * k_pack_argmax / k_unpack_argmax compute on the device (MAX over shards = collapseMaxZSlice over all planes) */
DSI_API int dsi_argmax_keys_pack(const float *conf, const uint8_t *idx_local, size_t n, int plane_begin, uint64_t *keys);
DSI_API int dsi_argmax_keys_unpack(const uint64_t *keys, size_t n, float *conf, uint8_t *idx_global);
DSI_API int dsi_mapper_depth_map_scattered_local(dsi_mapper_t *m, dsi_grid_t *acc, int nranks, int rank, int mode,
                                                 int n_maps);
/* the mapper's packed arg-max keys (dimY * dimX words), host <-> device; synchronise */
DSI_API int dsi_mapper_argmax_keys_download(dsi_mapper_t *m, uint64_t *keys_host);
DSI_API int dsi_mapper_argmax_keys_upload(dsi_mapper_t *m, const uint64_t *keys_host);
DSI_API int dsi_mapper_depth_map_from_keys(dsi_mapper_t *m);

/* Exact tie resolver: makes the arg-max index map the reference's on every pixel -- VERIFIED EMPIRICALLY per call, not
 * guaranteed a priori (see premise_ok).
 * The engine sums a voxel's votes exactly (64-bit fixed point, rounded once); the reference adds them in fp32 in event
 * order, rounding after every vote (cartesian3dgrid.h:261-270 inside mapper_emvs_stereo.cpp:197-201).  The two sums
 * agree to ~1e-5, so the first-maximum plane (cartesian3dgrid.cpp:132-134) can differ only in columns whose best
 * planes are closer than that -- and there the depth is off by a whole plane.  This call finds the columns of the
 * fused DSI  op(grid(mappers[0]), grid(mappers[1]))  (n == 1: of grid(mappers[0])) that have two or more planes within
 * rel_gap of the column's maximum, re-sums exactly those voxels of every camera in the REFERENCE's order (per camera
 * one pass that asks, per packet and contending voxel, which events the plane transfer takes into the voxel's 2 x 2
 * neighbourhood and votes those with the reference's coordinates, accept test and weights; the recorded votes sorted
 * by (voxel, event index) and added one by one in fp32), re-applies the fusion op, re-picks the first maximum and
 * patches confidence / index / depth in `out`'s depth-map buffers -- all on the device; the host reads counters only.
 * Preconditions: grid(mappers[i]) holds dsi_mapper_evaluate_batch(mappers[i], batches[i]); `out` holds the depth map
 * of that fusion (dsi_mapper_depth_map_of / _of_fusion / _of_events).  n = 1 or 2; op = dsi_fuse_op_t.
 * rel_gap 0 = 2.5e-4: 15x the largest difference ever observed between the two summation orders (1.6e-5) and 2.5x the
 * tolerance every voxel of every parity test is held to.  The rigorous worst case of a voxel, (votes - 1) * 2^-24
 * (max_rel_bound), EXCEEDS that gap for voxels with more than ~4,200 votes, so the gap is a premise, and the call
 * checks it: max_order_diff is the largest |engine value - reference-order value| / max(1, value) over the voxels it
 * re-summed -- the column maxima of the near-tie columns, the most-voted voxels of the volume among them.  If it reaches
 * rel_gap / 8 the pass is repeated with a four times wider gap (gap_widenings, at most 3); premise_ok = 0 when even the
 * last pass did not hold it: the call still returns DSI_OK, the caller decides.  A WIDENED pass that asks for more votes
 * than the pass can record (a workgroup's segment table, the sort key) does not turn into an error either: the last
 * completed pass's patch and statistics stand, premise_ok = 0.  Optional and off the throughput path (elapsed_ms).
 * Synchronises. */
typedef struct {
    float rel_gap;          /* in (0 = 2.5e-4); out: the gap of the last pass (wider than asked for after gap_widenings) */
    int near_tie_pixels;    /* out: columns with >= 2 contending planes */
    int candidate_voxels;   /*      contending voxels re-summed (per camera) */
    int candidate_planes;   /*      distinct planes among them */
    long long votes;        /*      votes re-summed, all cameras */
    int changed_pixels;     /*      pixels whose plane index a pass changed, SUMMED over the passes (a widened pass counts against
                                    the already patched map: with gap_widenings > 0 a pixel can be counted more than once) */
    double max_rel_bound;   /*      see above */
    double max_order_diff;  /*      see above */
    float elapsed_ms;       /*      wall time of the call */
    int gap_widenings;      /*      passes repeated with a 4 x wider gap because max_order_diff reached rel_gap / 8 (0..3) */
    int premise_ok;         /*      1: 8 * max_order_diff < rel_gap held in the last pass.  0: it did not even at the widest gap
                                    tried -- the index map is then NOT known to equal the reference's */
    int columns_bounded;    /*      near-tie columns (of the last pass) whose gap also covers the WORST-CASE reordering error of
                                    their own most-voted contender, 2 ((votes) 2^-24 + 2^-22): for those no plane with at most
                                    that many votes can have been left out wrongly.  (Planes outside the gap with more votes than
                                    every contender are not covered -- a step keeps no per-voxel vote counts --, so this is a
                                    statistic beside premise_ok, not a proof; the proof, with counted votes, is
                                    dsi_mapper_prove_near_ties below.) */
} dsi_resolve_info_t;
DSI_API int dsi_mapper_resolve_near_ties(dsi_mapper_t *out, dsi_mapper_t *const *mappers,
                                         const dsi_batch_t *const *batches, int n, int op, dsi_resolve_info_t *info);

/* The resolver's premise as a PROOF, column by column (ABI 10) -- a verification pass, not part of a step.
 * dsi_mapper_resolve_near_ties re-sums the planes within rel_gap of a column's maximum in the reference's order and trusts
 * that no plane outside that gap can win under the reference's arithmetic; it checks this on the voxels it re-sums
 * (premise_ok), not on the others.  This call closes the gap: it COUNTS the votes of every voxel of every camera (the
 * transfer of mapper_emvs_stereo.cpp:177-195 with the IEEE divide and the accept test of cartesian3dgrid.h:255-259, one
 * global atomic per vote on a dimZ x dimY x dimX volume of 32-bit counters per camera), and then bounds, per voxel, the
 * value the reference holds there: its n non-negative weights added one by one in fp32 stay within
 * (n - 1) u / (1 - (n - 1) u), u = 2^-24, of their real sum (recursive summation), the engine's value is that sum with
 * every weight truncated to the 2^-31 grid and one rounding to fp32 (the mappers' DSIs must be such exact sums:
 * DSI_VOTE_LDS_BANDS, not the paired lane mapping 8 -- DSI_ERR_INVALID otherwise), and the camera fusion
 * (cartesian3dgrid.h:108-190) is monotone with at most five roundings.  A column is PROVEN when every plane below best - rel_gap * best has an upper bound
 * strictly below the lower bound of the maximum's plane: then the reference's first maximum (cartesian3dgrid.cpp:132-134)
 * is among the planes the resolver re-sums exactly, and the resolved index is the reference's.
 *   mappers / batches / n / op: as for dsi_mapper_resolve_near_ties (the cameras' DSIs voted from these batches);
 *   out: any mapper of their shape and context (its scratch holds the counters; its depth map is not touched).
 * gap_needed: the smallest rel_gap that would have taken every offending plane into the re-summed set (0 when all columns
 * are proven); resolving again with a gap a little above it and proving again then proves every column. */
typedef struct {
    float rel_gap;              /* in: the gap the resolver ran with (0: its default, 2.5e-4) */
    long long columns;          /* pixels examined: all of them */
    long long columns_proven;
    long long columns_unproven;
    double gap_needed;
    long long max_votes;        /* most votes in one voxel of one camera */
    float elapsed_ms;
    long long columns_resolved_fully; /* 0 from this call; the proven-mode helpers (process_1_exact_depth_map, process.py) count
                                         here the unproven columns they then re-summed on ALL planes (dsi_mapper_exact_voxels +
                                         dsi_reference_fuse2 + dsi_mapper_patch_depth_map): exact by construction */
} dsi_prove_info_t;
DSI_API int dsi_mapper_prove_near_ties(dsi_mapper_t *out, dsi_mapper_t *const *mappers,
                                       const dsi_batch_t *const *batches, int n, int op, dsi_prove_info_t *info);
/* The same proof for n <= 8 cameras fused by an n-ary mode (BASELINE configs[4]'s four-camera rig): fused = the grid that
 * holds dsi_grid_fuse_n(..., mode) of the mappers' DSIs -- the grid dsi_grid_near_tie_voxels took the near-tie columns from
 * (process.exact_depth_map_nary / the building blocks re-sum them); mode = DSI_ACC_GM_TREE (n = 2, 4, 8), DSI_ACC_MIN,
 * DSI_ACC_MAX or DSI_ACC_SUM (the arithmetic mean).  info->rel_gap: the gap those columns were taken with (0: 2.5e-4).
 * The oracle can only cover a strip of a 1024 x 1024 x 256 volume in reasonable time; this covers every column. */
DSI_API int dsi_mapper_prove_near_ties_n(dsi_mapper_t *out, dsi_grid_t *fused, dsi_mapper_t *const *mappers,
                                         const dsi_batch_t *const *batches, int n, int mode, dsi_prove_info_t *info);
/* The proof's pieces for ANY fusion topology (Alg. 2's camera-then-time fusion, process2.cpp:98-249; compositions of your own):
 * INTERVAL GRIDS.  lo / hi are two grids that enclose, voxel by voxel, the value the reference holds.
 *  - dsi_mapper_reference_interval: lo, hi <- the bounds of the DSI `m` built from `batch` (its votes counted as in
 *    dsi_mapper_prove_near_ties; scratch: any mapper of that shape, e.g. the output mapper);
 *  - carry them through the SAME grid ops the values went through (dsi_grid_fuse2, dsi_grid_accumulate, dsi_grid_finalize,
 *    dsi_grid_fuse_n ... on lo and on hi separately): the reference's voxel-wise ops are monotone non-decreasing in their
 *    operands -- except the 1 / (0.01 + g) inside the harmonic accumulation, which the final n / sum turns around, so for
 *    DSI_ACC_INV_SUM the lower bounds accumulate into the lower result as they are;
 *  - dsi_grid_widen_interval after every such step with the number of fp32 roundings the step performs per voxel (fuse2: 5;
 *    accumulate: 3; finalize: 2; an n-ary tree: 2 per level): lo <- lo (1 - k u) rounded down, hi <- hi (1 + k u) rounded up;
 *  - dsi_grid_prove_columns: `fused` holds the engine's values (the near-tie columns and the threshold come from them);
 *    a column is proven when every plane below best - rel_gap * best has hi strictly below lo of the maximum's plane.
 *    info as for dsi_mapper_prove_near_ties (max_votes: from the dsi_mapper_reference_interval calls on `scratch` since its
 *    last proof); dsi_mapper_proof_unproven lists the columns that are not.
 * process.exact_depth_map_process_2_proven is Alg. 2 done that way. */
DSI_API int dsi_mapper_reference_interval(dsi_mapper_t *scratch, dsi_mapper_t *m, const dsi_batch_t *batch, dsi_grid_t *lo,
                                          dsi_grid_t *hi);
DSI_API int dsi_grid_widen_interval(dsi_grid_t *lo, dsi_grid_t *hi, int roundings);
DSI_API int dsi_grid_prove_columns(dsi_mapper_t *scratch, dsi_grid_t *fused, dsi_grid_t *lo, dsi_grid_t *hi, dsi_prove_info_t *info);
/* votes[i] <- the number of votes voxel voxels[i] (z * dimY * dimX + y * dimX + x) of camera `camera` (0 .. 7) received
 * according to the counters the LAST dsi_mapper_prove_near_ties on `out` made (what its bounds were computed from; the
 * same number dsi_mapper_exact_voxels reports from the resolver's own event pass). */
DSI_API int dsi_mapper_proof_votes(dsi_mapper_t *out, int camera, const uint32_t *voxels, size_t n, uint32_t *votes);
/* The columns the LAST dsi_mapper_prove_near_ties on `out` could not prove: pixels[i] = y * dimX + x, gaps[i] (optional) = the
 * rel_gap that column alone would need (1 = down to its all-but-zero planes: a column whose maximum is a handful of tiny
 * weights, where the engine's 2^-31 weight grid is as coarse as the values).  *n may exceed capacity (nothing is written
 * beyond it).  Few and hard columns are cheaper to re-sum on ALL their planes (dsi_mapper_exact_voxels) than to widen the
 * gap of every column for. */
DSI_API int dsi_mapper_proof_unproven(dsi_mapper_t *out, uint32_t *pixels, float *gaps, size_t capacity, size_t *n);

/* The resolver's building blocks, for fusion topologies it does not cover itself (Alg. 2's camera-then-time fusion,
 * process2.cpp:98-249; n cameras): find the near-tie columns of ANY grid, get the reference-order value of ANY list of
 * voxels of the DSI a mapper builds from a batch, recombine them on the host with the reference's scalar ops, patch
 * the depth map.  dvs_mcemvs_amd/process.py::exact_depth_map_process_2 is Alg. 2 done that way.
 *  - dsi_grid_near_tie_voxels: the voxels (z * dimY * dimX + y * dimX + x) of g within rel_gap of their column's
 *    maximum, for the columns that have >= 2 of them; a column's run contiguous, planes ascending.  *n_voxels may
 *    exceed capacity (then nothing beyond capacity was written: call again with more room).  scratch: any mapper of
 *    g's context and shape (its resolver scratch is used).  dimZ <= 256 (main.cpp:156), else DSI_ERR_SHAPE.
 *  - dsi_mapper_exact_voxels: values[i] <- the value voxel voxels[i] of the DSI of (m, batch) has when its votes are
 *    added in fp32 in event order (resetGrid, then += per vote: mapper_emvs_stereo.cpp:145, :197-201,
 *    cartesian3dgrid.h:261-270); votes[i] (optional) <- their number.  m's grid is not touched.  voxels: any order,
 *    duplicates allowed.
 *  - dsi_reference_fuse2 / _accumulate / _finalize: the reference's voxel-wise ops on host arrays (op = dsi_fuse_op_t;
 *    mode = DSI_ACC_SUM or DSI_ACC_INV_SUM), bit for bit what the device kernels compute.
 *  - dsi_mapper_patch_depth_map: overwrite n pixels (y * dimX + x) of the raw depth map m holds: index, confidence,
 *    depth = plane of the index. */
DSI_API int dsi_grid_near_tie_voxels(dsi_mapper_t *scratch, dsi_grid_t *g, float rel_gap, uint32_t *voxels, size_t capacity,
                                     size_t *n_voxels, size_t *n_columns);
DSI_API int dsi_mapper_exact_voxels(dsi_mapper_t *m, const dsi_batch_t *batch, const uint32_t *voxels, size_t n,
                                    float *values, uint32_t *votes);
DSI_API int dsi_reference_fuse2(int op, const float *a, const float *g, size_t n, float *out);
DSI_API int dsi_reference_accumulate(int mode, float *acc, const float *g, size_t n);
DSI_API int dsi_reference_finalize(int mode, float *acc, size_t n, int n_maps);
DSI_API int dsi_mapper_patch_depth_map(dsi_mapper_t *m, const uint32_t *pixels, const uint8_t *idx, const float *conf,
                                       size_t n);

/* OptionsDepthMap (mapper_emvs_stereo.hpp:68-82), the fields the depth-map extraction reads */
typedef struct {
    int adaptive_threshold_kernel_size; /* --adaptive_threshold_kernel_size, default 5 (main.cpp:73) */
    double adaptive_threshold_c;        /* --adaptive_threshold_c, default 5 (main.cpp:74) */
    int median_filter_size;             /* --median_filter_size, default 5 (main.cpp:75) */
    double max_confidence;              /* --max_confidence (main.cpp:97); written to conf(0,0) */
} dsi_depthmap_options_t;

/* MapperEMVS::getDepthMapFromDSI(depth_map, confidence_map, mask, options)
 * (mapper_emvs_stereo.cpp:339-437) for grid g (NULL = the mapper's own DSI), entirely on the
 * device: collapseMaxZSlice (:368), conf(0,0) = max_confidence + cv::normalize to 8 bit
 * (:393-397), Gaussian adaptive threshold (:403-409), masked Huang median of the depth indices
 * (:420-423, median_filtering.cpp), removeMaskBoundary (:426-427) and
 * convertDepthIndicesToValues of the filtered indices (:435).  The Telea inpainting of
 * depth_map_dense (:430-436) is not reproduced.  Host outputs (any may be NULL): depth f32,
 * confidence f32 (element (0,0) overwritten like the reference does), mask u8 in {0,1},
 * filtered indices u8.  Synchronises. */
DSI_API int dsi_mapper_get_depth_map_from_dsi(dsi_mapper_t *m, dsi_grid_t *g, const dsi_depthmap_options_t *opts,
                                      float *depth_host, float *conf_host, uint8_t *mask_host,
                                      uint8_t *idx_filtered_host);

/* The part of getDepthMapFromDSI after collapseMaxZSlice (mapper_emvs_stereo.cpp:390-437), applied to
 * the raw depth map this mapper holds on the device -- the result of the last
 * dsi_mapper_depth_map_of / _of_fusion / _of_fusion_n / _of_events call.  This is how a window loop
 * that never materialises a DSI (dsi_mapper_depth_map_of_events, mode DSI_VOTE_FUSED_ARGMAX) obtains
 * the reference's filtered per-window outputs (main.cpp:177-302 calls getDepthMapFromDSI on the fused
 * DSI).  Same outputs and bits as dsi_mapper_get_depth_map_from_dsi on the same arg-max.  The raw map
 * is consumed (confidence normalised in place): DSI_ERR_INVALID if there is none, or on a second
 * call.  Synchronises. */
DSI_API int dsi_mapper_filter_depth_map(dsi_mapper_t *m, const dsi_depthmap_options_t *opts, float *depth_host,
                                        float *conf_host, uint8_t *mask_host, uint8_t *idx_filtered_host);

/* HIP-event stopwatch around the voting kernel (the replacement of fillVoxelGrid's hot
 * loop, mapper_emvs_stereo.cpp:168-203) on the context's stream: enable, run any number of
 * evaluate/fill calls, then read the summed kernel time and launch count (synchronises and
 * clears).  bench.py derives the roofline figure of the dominant kernel from this. */
DSI_API int dsi_mapper_set_kernel_timing(dsi_mapper_t *m, int enable);
DSI_API int dsi_mapper_vote_kernel_time(dsi_mapper_t *m, float *total_ms, int *launches);

/* Work the voting kernel does on one batch (bench.py's roofline): accepted_event_planes = events x planes that passed
 * the accept test of Grid3D::accumulateGridValueAt (cartesian3dgrid.h:255-259) = the sum of the DSI (the four
 * bilinear weights of a vote sum to 1); accepted_records = the same count after the packet sort has merged the
 * events of a packet that share a pixel into one record = a quarter of the LDS atomics actually issued.  Votes the
 * batch twice and leaves the mapper's DSI as dsi_mapper_evaluate_batch does.  Synchronises. */
DSI_API int dsi_mapper_vote_statistics(dsi_mapper_t *m, const dsi_batch_t *batch, double *accepted_event_planes,
                                       double *accepted_records);

/* diagnostics of the last evaluate/fill call: which kernel ran, bands, chunks */
typedef struct {
    int algo;          /* dsi_vote_algo_t actually used */
    int bands;         /* row bands per plane (LDS algo) */
    int band_rows;     /* owned rows per band */
    int chunks;        /* packet chunks (partial DSIs) */
    int block_threads;
    size_t lds_bytes;
    size_t n_packets;
    int packed;        /* lane mapping that ran (see dsi_mapper_set_packed_lanes) */
    int group_packets; /* packets sorted together by mapping 2 */
} dsi_vote_info_t;
DSI_API int dsi_mapper_last_vote_info(const dsi_mapper_t *m, dsi_vote_info_t *info);

#ifdef __cplusplus
}
#endif
#endif /* DSI_ENGINE_H */
