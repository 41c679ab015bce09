// dsi_kernels.h -- internal launch interface between the C-ABI host layer
// (dsi_engine.cpp) and the gfx950 kernels (dsi_kernels.hip).  Not installed.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace dsi {

constexpr int kPacket = 1024;  // mapper_emvs_stereo.hpp:152

// Per (packet, plane) transfer coefficients of mapper_emvs_stereo.cpp:177-182,
// plus what the banded kernel needs to divide quickly.  32 bytes so that one
// s_load_dwordx8 fetches it.
struct PlaneCoef {
    float a, bx, by, d;
    float r;         // RN(1/d) for the residual-corrected division
    uint32_t flags;  // kCoefSkip / kCoefSlow / kCoefInvertible; bits 16-31: the inline cuts' widening in 1/256 rows
    float d_a, by_a;  // the inverse of the row transfer, y0 = Y * d_a - by_a (d / a, by / a): a band's rows -> the packet's
                      // z0 rows -> its run of records.  k_plane_coef turns them into the cut table; with
                      // BandPlan::cuts_inline the vector-fill stream does, per pass, from the transposed row table
};
constexpr uint32_t kCoefSkip = 1u;  // no event of this (packet, plane) can be accepted
constexpr uint32_t kCoefSlow = 2u;  // use the IEEE divide and the whole packet
constexpr uint32_t kCoefInvertible = 4u;  // d_a / by_a are usable by the inline cuts (else the run is the whole packet)

// One entry of a grouped packet: an event's z0 location and how many events of the packet
// share exactly that location (identical raw pixel in one 1024-event packet: hot pixels,
// bursts).  Voting m identical events = one vote of m times the fixed-point weight, exactly.
struct EvRec {
    float x, y;
    uint32_t m;
};

struct Geom {
    int nx, ny, nz;
    float vfx, vfy, vcx, vcy;  // virtual camera (geometry_utils.hpp:30-47)
    float kfx, kfy, kcx, kcy;  // sensor projection intrinsics K_ (mapper_emvs_stereo.cpp:46-48)
    float z0;                  // raw_depths_vec_[0]
};

// lane mappings 5 / 6: 64-bit words of LDS scratch per wave (64 tail-bit words, reused as the run table
// of the hand-scheduled loop, whose entry 64 serves the slots beyond a pass)
constexpr int kVfillScratchWords = 65;

struct BandPlan {
    int bands;          // row bands per plane
    int band_rows;      // owned rows per band (last band may own fewer)
    int chunks;         // packet chunks; each writes its own partial DSI when > 1
    int block_threads;  // 256 / 512 / 1024
    int packed;         // lane mapping: 0 k_vote_bands, 1 k_vote_bands_packed (asm), 2 k_vote_groups,
                        // 3 k_vote_bands_packed (compiled loop), 4 k_vote_groups (asm),
                        // 5 k_vote_bands_packed with the vector fill (wide grids), 6 the same all compiled,
                        // 7 mapping 1 with dealt passes, 8 mapping 7 on PAIRED 32-bit Q.19 cells (opt-in: not the exact sums)
    int group_packets;  // mapping 2: packets sorted together (power of two <= 32)
    int row_pad;        // z0 rows binned over [-row_pad, ny + row_pad) by k_sort_packets
    int pass_lg;        // packed mappings: log2(packets a wave takes per pass); 0 = automatic
    size_t lds_bytes;   // (band_rows + 1) * nx * 8 (u64 fixed-point accumulators; +1 = the carry row)
                        // + mappings 5 / 6: kVfillScratchWords * 8 B per wave (tail-bit words / run table)
    int scratch_offset; // mapping 5: byte offset of that area behind the band
    int persistent;     // packed mappings: workgroups pull work items from per-XCD counters
    int raw_out;        // the voting kernel's `out` is 1: the chunks' partial volumes of raw 64-bit sums (unsigned long long
                        // [chunks][partial_stride]), summed exactly by k_reduce_partials; 0: the fp32 DSI itself (one chunk)
    int halo;           // 1: a band also takes the events of the row above its first owned row and keeps a halo row
                        // on either side in LDS ((band_rows + 2) rows; k_vote_fuse_argmax), 0: carry row ((band_rows + 1))
    int experiment;     // DSI_EXPERIMENT (timing experiments only, results are wrong): 1 no votes, 2 no flush
    int interleave;     // k_vote_fuse_argmax: 1 an XCD's workgroups take the pairs of its stretch in turn, 2 they DRAW them from
                        // the XCD's counter behind the keys (and from the other XCDs' once theirs is dry); see the kernel
    int cuts_inline;    // mappings 5 / 6 with many packets: NO cut table (bands x planes x packets words: 6.1 GB per camera at
                        // 1024 x 1024 x 256 with 100 M events, 3.9 ms to write); the voting kernel's `cuts` argument is then the
                        // TRANSPOSED row table u16 [ny + 2 row_pad + 3][rs_stride] (k_transpose_rowstart) and every pass
                        // derives its packets' runs from it and from PlaneCoef::d_a / by_a
    int rs_stride;      // cuts_inline: packets per row of the transposed table (a multiple of 64)
};

// lane mapping 8: 8-byte words per band row of paired 32-bit cells (dsi_vote_asm.h)
inline int paired_row_words_host(int nx) { return 2 * ((nx >> 1) + 1); }

// distance in voxels between the partial volumes of consecutive packet chunks: the volume size
// rounded up to 4 voxels, so that every partial volume starts 16-byte aligned
__host__ __device__ inline size_t partial_stride(size_t n_voxels) { return (n_voxels + 3) & ~(size_t)3; }

// ---- stage A ---------------------------------------------------------------
hipError_t launch_packet_geometry(hipStream_t s, const float* Rt, int np, const Geom& g,
                                  float* centers, float* H);
hipError_t launch_warp_z0(hipStream_t s, const uint16_t* ex, const uint16_t* ey,
                          const uint32_t* packet_first, int np, const float* H,
                          const float2* lut, int sensor_w, int sensor_h, float2* xy);
// ---- stage B, global-atomic form ------------------------------------------
hipError_t launch_vote_global(hipStream_t s, const float2* xy, const float* centers, int np,
                              const float* planes, const Geom& g, float* dsi);
// ---- stage B, LDS row-band form -------------------------------------------
// the same with stage A (per-packet geometry + z0 warp) fused in: raw events in, centers out
hipError_t launch_sort_packets_raw(hipStream_t s, const float* Rt, const uint16_t* ex, const uint16_t* ey,
                                   const uint32_t* packet_first, const float2* lut, int sensor_w, int sensor_h, const Geom& g,
                                   float* centers, int np, int pad, EvRec* sxy, uint32_t* nvalid,
                                   uint16_t* rowstart, int unit_multiplicity = 0);
hipError_t launch_sort_packets(hipStream_t s, const float2* xy, int np, int ny, int nz, int pad,
                               EvRec* sxy, uint32_t* nvalid, uint16_t* rowstart);
hipError_t launch_plane_coef(hipStream_t s, const float* centers, const float* planes,
                             const uint16_t* rowstart, uint32_t* nvalid, int np,
                             const Geom& g, const BandPlan& bp, PlaneCoef* coef, uint32_t* cuts);
hipError_t launch_vote_bands(hipStream_t s, const EvRec* sxy, const PlaneCoef* coef,
                             const uint32_t* cuts, const uint32_t* slow_any, int np, const Geom& g,
                             const BandPlan& bp, void* out, unsigned long long* seam);
// seam rows (first row of every band but the first, of every plane of every chunk volume in `out`) =
// fl((head + carry) * 2^-31) from the voting kernel's 64-bit sums seam[chunks][nz][bands][2][nx]
hipError_t launch_seam_rows(hipStream_t s, const unsigned long long* seam, int chunks, const Geom& g,
                            const BandPlan& bp, void* out);
hipError_t launch_sort_groups(hipStream_t s, const float2* xy, int np, int S, int ny, int nz, int pad,
                              EvRec* sxy, uint8_t* spk, uint32_t* nvalid, uint16_t* rowstart);
hipError_t launch_group_cuts(hipStream_t s, const uint32_t* prow, const uint16_t* rowstart, int np,
                             int S, const Geom& g, const BandPlan& bp, uint32_t* gcuts);
hipError_t launch_vote_groups(hipStream_t s, const EvRec* sxy, const uint8_t* spk,
                              const PlaneCoef* coef, const uint32_t* gcuts, const uint32_t* slow_any,
                              int np, int S, const Geom& g, const BandPlan& bp, void* out, unsigned long long* seam);
// dsi (+)= fl(sum of the chunks' raw partial volumes).  seam != nullptr (with g, bp; needs an even nx and < 2^32 voxels:
// reduce_can_fold_seams): the seam rows are read from the bands' head / carry sums instead, i.e. launch_seam_rows is
// folded in and must NOT be run on the partial volumes first
hipError_t launch_reduce_partials(hipStream_t s, const unsigned long long* partials, int chunks, size_t n,
                                  float* dsi, int accumulate, const unsigned long long* seam = nullptr,
                                  const Geom* g = nullptr, const BandPlan* bp = nullptr);
inline bool reduce_can_fold_seams(const Geom& g, size_t n) { return (g.nx & 1) == 0 && n <= 0xffffffffull; }
// ---- stage B fused with the camera fusion and the arg-max (no DSI leaves the CU) ----
// the per-camera tables of the banded vote (k_sort_packets / k_plane_coef outputs, built with bp.halo = 1)
struct FusedCamera {
    const EvRec* sxy;
    const PlaneCoef* coef;
    const uint32_t* cuts;
    const uint32_t* slow_any;
    int np;
};
constexpr int kFusedMaxCameras = 4;
struct FusedCameras {
    FusedCamera cam[kFusedMaxCameras];
    int n;  // 1 (vote -> arg-max), 2 (vote x 2 -> op -> arg-max), 3 (process1.cpp:169-191: the trinocular rig --
            // op 1 min, 2 harmonicMeanTwoGrids(g, 3), 6 max of the two-camera result and camera 2) or 4 (round 6: the
            // balanced tree of the reference's 2-ary geometric mean, cartesian3dgrid.h:150-156 -- DSI_ACC_GM_TREE)
};
// LAYOUT LOCK: k_vote_fuse_argmax reads cam[c] with scalar loads straight from the kernel-argument segment
// (__builtin_amdgcn_kernarg_segment_ptr), which holds because (i) `cams` is the kernel's FIRST parameter, so cam[0]
// sits at offset 0 of the segment, and (ii) a by-value struct parameter keeps its host layout there (code object v5:
// by-value aggregates are copied verbatim, aligned to their natural alignment).  Reordering the kernel's parameters or
// these fields breaks (i) / (ii); the asserts below and the one in the kernel catch the second kind at compile time.
static_assert(sizeof(FusedCamera) == 40 && alignof(FusedCamera) == 8, "FusedCamera: 4 pointers + int, 8-byte aligned");
static_assert(offsetof(FusedCamera, sxy) == 0 && offsetof(FusedCamera, coef) == 8 && offsetof(FusedCamera, cuts) == 16 &&
                  offsetof(FusedCamera, slow_any) == 24 && offsetof(FusedCamera, np) == 32,
              "FusedCamera field offsets are read from the kernarg segment");
static_assert(offsetof(FusedCameras, cam) == 0 && sizeof(FusedCameras) == kFusedMaxCameras * sizeof(FusedCamera) + 8,
              "FusedCameras: the camera table starts the struct (and so the kernel-argument segment)");
// The preparation of up to three cameras in two launches instead of two per camera (stage A + packet sort; coefficient /
// cut tables), optionally counting the records per (band, plane) pair for launch_fused_splits.
struct PrepCameraArgs {
    const float* Rt;
    const uint16_t *ex, *ey;
    const uint32_t* packet_first;
    const float2* lut;
    int sensor_w, sensor_h;
    float* centers;
    int np;
    EvRec* sxy;
    uint32_t* nvalid;
    uint16_t* rowstart;
    const float* planes;
    PlaneCoef* coef;
    uint32_t* cuts;
    uint32_t* pair_work;  // [bands * nz] or nullptr
    Geom g;               // this camera's own intrinsics, virtual camera and z0 (process1.cpp:73-110: one mapper
                          // per camera, each built from its own calibration); nx, ny, nz equal for all cameras
};
hipError_t launch_prepare_cameras(hipStream_t s, const PrepCameraArgs* cams, int n, const Geom& g, const BandPlan& bp);
// splits[fused_grid_blocks() + 1] <- cuts of the pair list into stretches of equal cost (records + fixed_per_pair
// each); prefix: n_pairs words of scratch.  n_pairs <= fused_max_pairs()
hipError_t launch_fused_splits(hipStream_t s, const uint32_t* work0, const uint32_t* work1, int n_pairs, uint32_t fixed_per_pair,
                               unsigned long long* prefix, uint32_t* splits);
int fused_max_pairs();
// keys[ny * nx] (zeroed by the caller) receive max over planes of conf_bits << 8 | 255 - plane: feed
// launch_unpack_argmax.  splits: optional balanced partition of the (band-major) pair list, one entry
// per workgroup + 1 (fused_grid_blocks() workgroups)
int fused_grid_blocks();
size_t fused_max_cells(int mapping, int n_cameras = 2);  // (band_rows + 2) * nx may not exceed this (four cameras: 16 cells per thread)
hipError_t launch_vote_fuse_argmax(hipStream_t s, const FusedCameras& cams, const Geom& g, const BandPlan& bp, int op,
                                   const uint32_t* splits, unsigned long long* keys, unsigned long long* trace = nullptr);
size_t fused_trace_words();  // trace: [workgroup][64 phases][16 waves][4 stamps] of 100 MHz ticks, or nullptr
// ---- Grid3D ops ------------------------------------------------------------
hipError_t launch_fuse2(hipStream_t s, float* a, const float* g, size_t n, int op);
hipError_t launch_fuse2_into(hipStream_t s, float* dst, const float* a, const float* g, size_t n,
                             int op);
hipError_t launch_fuse_hm_n(hipStream_t s, float* a, const float* g, size_t n, int n_maps);
hipError_t launch_accumulate(hipStream_t s, float* acc, const float* g, size_t n, int mode);
hipError_t launch_finalize(hipStream_t s, float* acc, size_t n, int mode, int n_maps);
hipError_t launch_fill(hipStream_t s, float* a, size_t n, float v);
// the depth map's arrays stored by a kernel into mapped page-locked host memory (device-side addresses; any may be null)
hipError_t launch_store_depth_map(hipStream_t s, const float* depth, const float* conf, const uint8_t* idx, size_t npix,
                                  float* depth_host_dev, float* conf_host_dev, uint8_t* idx_host_dev);
hipError_t launch_fuse_n(hipStream_t s, float* dst, const float* const* srcs, int n_src, size_t n, int mode);
hipError_t launch_pack_argmax(hipStream_t s, const float* conf, const uint8_t* idx, int n, int plane_begin,
                              unsigned long long* keys, int combine = 0);
// clear != 0: the keys are zeroed as they are read (the fused vote kernel needs them zero before it runs) -- and so are the
// kFusedKeyTail 64-bit words BEHIND the n keys, which such a buffer must have: the fused kernel's pair-dealing counters
constexpr int kFusedKeyTail = 8;
hipError_t launch_unpack_argmax(hipStream_t s, unsigned long long* keys, int n, const float* planes_full,
                                float* conf, uint8_t* idx, float* depth, int clear = 0);
hipError_t launch_collapse_max_z(hipStream_t s, const float* dsi, int nx, int ny, int nz,
                                 float* conf, uint8_t* idx, const float* planes, float* depth);
hipError_t launch_depth_map_filters(hipStream_t s, float* conf, const uint8_t* idx, int nx, int ny,
                                    int ksize, double C, int median_size, double max_confidence,
                                    const float* planes, uint32_t* minmax_scratch, uint8_t* conf8,
                                    uint8_t* mask, uint8_t* idx_filtered, float* depth);
hipError_t launch_collapse_max_z_fused(hipStream_t s, const float* a, const float* b, int nx, int ny, int nz, int op,
                                       float* conf, uint8_t* idx, const float* planes, float* depth);
hipError_t launch_collapse_max_z_fused_n(hipStream_t s, const float* const* srcs, int n_src, int mode, int nx, int ny,
                                         int nz, float* conf, uint8_t* idx, const float* planes, float* depth);
hipError_t launch_mean_square(hipStream_t s, const float* dsi, size_t n, double* accum);

// ---- exact tie resolver (dsi_mapper_resolve_near_ties): see the kernels' comment ----
// cand[<= cap] <- voxel indices (z * npix + p) of every plane whose value is within rel_gap of its column's maximum, for
// the columns that have >= 2 such planes, a column's run contiguous and ascending in z; counters[0] = voxels (may exceed
// cap: nothing beyond cap is written), counters[1] = columns.  b == nullptr: the values of `a`; else op(a, b)
// cols (<= cols_cap entries) <- per near-tie column (first entry of its run in cand, pixel, contenders, 0).  nz <= 256
hipError_t launch_tie_candidates(hipStream_t s, const float* a, const float* b, int op, int npix, int nz, float rel_gap,
                                 unsigned* counters, uint32_t* cand, uint32_t cap, uint4* cols, uint32_t cols_cap);
// desc[c] <- (x | y << 16, z) of voxel vox[c]; plane_bits (optional, 8 zeroed words): bit z <- plane z holds one of them
hipError_t launch_tie_desc(hipStream_t s, const uint32_t* vox, int n, int nx, int npix, uint2* desc, unsigned* plane_bits);
// The resolver's event pass, inverted (see k_tie_hits_binned): every vote of the batch that lands on one of the voxels
// `desc` -> keys[] = (rank_base + index of the voxel) << pos_bits | position of the vote in the reference's loop over events
// (packet * 1024 + slot), wts[] = the bilinear weight the reference adds.  Output in segments of tie_segment_records()
// records, *seg_counter of them handed out so far (it keeps counting beyond cap_segs: flags bit 1, nothing written there;
// flags bit 0: a block overflowed its segment table); unused tails hold sentinel keys sentinel_rank << pos_bits.  Several
// launches (cameras) may append to the same arrays.  *total_hits += the real votes
int tie_segment_records();
int tie_block_capacity_records();  // votes one workgroup of the pass can record
hipError_t launch_tie_hits_binned(hipStream_t s, const uint16_t* ex, const uint16_t* ey, const uint32_t* packet_first, const float* H,
                                  const float2* lut, int sensor_w, int sensor_h, const float* centers, const float* planes, const Geom& g,
                                  int np, const uint2* desc, int nsv, unsigned rank_base, unsigned pos_bits, unsigned sentinel_rank,
                                  unsigned* seg_counter, unsigned cap_segs, unsigned* flags, unsigned long long* total_hits,
                                  unsigned long long* keys, float* wts);
// per near-tie column: op(exact camera 0, exact camera 1) (op 0: one camera), first maximum, patch; stats[0] = max float bits
// of diff, [1] = max count, [2] += changed pixels, [3] += columns whose gap covers their own contenders' worst-case bound
hipError_t launch_tie_pick(hipStream_t s, int op, const uint4* cols, int n_cols, const uint32_t* vox, int nsv, int npix,
                           const float* exact, const uint32_t* count, const float* diff, const float* planes, float* conf,
                           uint8_t* idx, float* depth, unsigned* stats, float rel_gap);
// round 6, instead of a device-wide sort: the recorded votes partitioned by rank (camera * nsv + voxel) into runs[], each run
// ordered by event position in LDS and added one by one in fp32.  counts / cursor: n_cams * nsv words, starts: one more (scratch);
// runs: n_rec words; wts is consumed (it receives the weights in event order).  exact[cam * nsv + c], count[...] <- sequential fp32
// sum / number of the votes of voxel c of camera cam; diff[...] (optional; needs grid0 (and grid1 for two cameras)) <-
// |grid_cam[vox[c]] - exact| / max(1, |exact|).  No library call.
// the resolver's premise as a per-column proof (dsi_mapper_prove_near_ties): H (nz * ny * nx zeroed words) <- the reference's
// votes per INTEGER location (a voxel's votes: the four locations whose 2 x 2 footprint holds it); then, per column, every plane
// below best - rel_gap * best against the maximum's plane with rigorous bounds of both reference values.  stats5 (zeroed):
// columns proven, columns not, float bits of the rel_gap that would take the offending planes in, most votes in a voxel,
// entries of unproven[] (pixel, float bits of the gap that column needs)
hipError_t launch_count_votes(hipStream_t s, const uint16_t* ex, const uint16_t* ey, const uint32_t* packet_first, const float* H9,
                              const float2* lut, int sensor_w, int sensor_h, const float* centers, const float* planes, const Geom& g,
                              int np, uint32_t* H);
// the proof's pieces for any fusion topology: interval grids (lo / hi enclose the reference's value voxel by voxel)
hipError_t launch_interval_of_counts(hipStream_t s, const float* dsi, const uint32_t* H, int nx, int ny, int nz, float* lo, float* hi,
                                     unsigned* most /* atomicMax: most votes in a voxel */);
hipError_t launch_interval_widen(hipStream_t s, float* lo, float* hi, size_t n, int roundings);
hipError_t launch_prove_columns(hipStream_t s, const float* fused, const float* lo, const float* hi, int npix, int nz, float rel_gap,
                                unsigned* stats5, uint2* unproven);
// ... and for n <= 8 cameras fused by an n-ary mode (DSI_ACC_GM_TREE 6 with n = 2, 4, 8; DSI_ACC_MIN 4, DSI_ACC_MAX 5, DSI_ACC_SUM 0):
// `fused` = the engine's fused grid, whose values decide the near-tie columns and the threshold
hipError_t launch_tie_prove_n(hipStream_t s, const float* fused, const float* const* e, const uint32_t* const* h, int n, int mode, int nx,
                              int ny, int nz, float rel_gap, unsigned* stats5, uint2* unproven);
hipError_t launch_tie_votes_of(hipStream_t s, const uint32_t* H, const uint32_t* vox, int n, int nx, int ny, uint32_t* votes);
hipError_t launch_tie_prove(hipStream_t s, const float* a, const float* b, const uint32_t* Ha, const uint32_t* Hb, int op, int nx,
                            int ny, int nz, float rel_gap, unsigned* stats5, uint2* unproven /* nx * ny entries, or nullptr */);
// cursor: tie_partition_cursor_words(n_ranks) words (the per-stretch table of the LDS path, or one cursor per rank)
size_t tie_partition_cursor_words(size_t n_ranks);
hipError_t launch_tie_partition_sums(hipStream_t s, const unsigned long long* keys, const float* wts, unsigned long long n_rec,
                                     unsigned pos_bits, uint32_t* counts, uint32_t* starts, uint32_t* cursor, unsigned long long* runs,
                                     const uint32_t* vox, int nsv, int n_cams, const float* grid0, const float* grid1, float* exact,
                                     uint32_t* count, float* diff);
hipError_t launch_tie_patch(hipStream_t s, const uint32_t* pix, const uint8_t* new_idx, const float* new_conf, int n,
                            const float* planes, float* conf, uint8_t* idx, float* depth);

// test hook: q[i] = residual-corrected division, ref[i] = n[i] / d[i]
hipError_t launch_div_probe(hipStream_t s, const float* n, const float* d, size_t count,
                            float* q, float* ref);

size_t max_dynamic_lds();

}  // namespace dsi
