// dsi_engine.cpp -- C ABI (include/dsi_engine.h) over the gfx950 kernels.
// Host-side mirror of the reference objects on the hot path:
//   dsi_grid_t    <-> Grid3D        (cartesian3dgrid.h / cartesian3dgrid.cpp)
//   dsi_mapper_t  <-> MapperEMVS    (mapper_emvs_stereo.hpp / .cpp)
//   dsi_batch_t   <-> the (events, per-packet pose) pairs evaluateDSI iterates over
// There is no CPU implementation behind this ABI: without a gfx950 device every
// constructor fails with DSI_ERR_NO_DEVICE.
#include "../../include/dsi_engine.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: librccl is loaded at run time (load_rccl)

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "dsi_host.hpp"
#include "dsi_kernels.h"

#pragma clang fp contract(off)

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(DSI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                 \
    } while (0)

#define REQUIRE(cond, code, ...)                  \
    do {                                          \
        if (!(cond)) return fail(code, __VA_ARGS__); \
    } while (0)

// device buffer that only ever grows
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);  // implicit device sync
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return e;
        }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

template <typename T>
struct TmpDev {  // call-scoped device memory
    T* p = nullptr;
    hipError_t alloc(size_t n) { return hipMalloc(reinterpret_cast<void**>(&p), (n ? n : 1) * sizeof(T)); }
    ~TmpDev()
    {
        if (p) (void)hipFree(p);
    }
};

}  // namespace

struct dsi_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    std::vector<hipEvent_t> marks;  // dsi_context_timeline_mark
    size_t n_marks = 0;
    hipEvent_t sync_ev = nullptr;  // dsi_context_wait_for: "everything queued on this stream so far"
    double* ms_accum = nullptr;    // device scalar for mean-square
    // Event batches are uploaded on their own stream so that the upload of the next batch (camera,
    // window) overlaps the voting of the current one; "ready" / "freed" events order the two streams.
    // ONE upload stream per device, shared by its contexts (uploads share the PCIe link anyway): the runtime maps
    // streams onto four hardware queues by default, and a pipeline of windows -- one context per window in flight --
    // that gave every context two streams of its own had its uploads queue behind another window's kernel
    // (profiles/r04_cpp_window_stream.txt: 0.41 ms per window waiting for uploads).
    hipStream_t upload_stream = nullptr;
    // device -> host copies of depth maps that must not wait for later work of the compute stream
    // (dsi_mapper_fetch_depth_map[_async]); created on first use
    hipStream_t copy_stream = nullptr;
    // Released event-batch blocks, reused by the next dsi_batch_create on this context: a stream of
    // windows then costs no hipMalloc / hipFree (hipFree synchronises the device).
    struct PoolBlock {
        void* p;
        size_t bytes;
        hipEvent_t freed;  // recorded on `stream` when the batch was released: its last reader is done
    };
    std::vector<PoolBlock> batch_pool;
    // grow-only scratch of Grid3D::collapseMaxZSlice's host-output form (no hipMalloc / hipFree --
    // i.e. no device synchronisation -- per call)
    void* collapse_scratch = nullptr;
    size_t collapse_scratch_bytes = 0;
    // grids, mappers and batches created from this context and not yet destroyed: each holds a pointer to it, so the
    // context refuses to go while any is alive (destroying it first was a use-after-free in their destroy)
    std::atomic<int> children{0};
};

// the per-device upload stream (see dsi_context::upload_stream)
struct SharedUpload {
    hipStream_t stream = nullptr;
    int refs = 0;
};
std::mutex g_upload_mu;
std::map<int, SharedUpload> g_upload;

hipError_t upload_stream_acquire(int device, hipStream_t* out)
{
    std::lock_guard<std::mutex> lock(g_upload_mu);
    SharedUpload& u = g_upload[device];
    if (!u.stream)
        if (hipError_t e = hipStreamCreateWithFlags(&u.stream, hipStreamNonBlocking)) return e;
    ++u.refs;
    *out = u.stream;
    return hipSuccess;
}

void upload_stream_release(int device)
{
    std::lock_guard<std::mutex> lock(g_upload_mu);
    auto it = g_upload.find(device);
    if (it == g_upload.end()) return;
    if (--it->second.refs <= 0) {
        if (it->second.stream) {
            (void)hipStreamSynchronize(it->second.stream);
            (void)hipStreamDestroy(it->second.stream);
        }
        g_upload.erase(it);
    }
}

hipError_t copy_stream_of(dsi_context* ctx, hipStream_t* out)
{
    if (!ctx->copy_stream)
        if (hipError_t e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking)) return e;
    *out = ctx->copy_stream;
    return hipSuccess;
}

bool pool_take(dsi_context* ctx, size_t bytes, dsi_context::PoolBlock* out)
{
    size_t best = (size_t)-1;
    for (size_t i = 0; i < ctx->batch_pool.size(); ++i) {
        const size_t sz = ctx->batch_pool[i].bytes;
        if (sz >= bytes && sz <= 4 * bytes + 65536 && (best == (size_t)-1 || sz < ctx->batch_pool[best].bytes))
            best = i;
    }
    if (best != (size_t)-1) {
        *out = ctx->batch_pool[best];
        ctx->batch_pool.erase(ctx->batch_pool.begin() + (long)best);
        return true;
    }
    out->p = nullptr;
    out->bytes = bytes;
    out->freed = nullptr;
    if (hipMalloc(&out->p, bytes) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&out->freed, hipEventDisableTiming) != hipSuccess) {
        (void)hipFree(out->p);
        return false;
    }
    (void)hipEventRecord(out->freed, ctx->stream);
    return true;
}

void pool_give(dsi_context* ctx, const dsi_context::PoolBlock& blk)
{
    if (!blk.p) return;
    (void)hipEventRecord(blk.freed, ctx->stream);  // everything that read the block was queued before
    ctx->batch_pool.push_back(blk);
    if (ctx->batch_pool.size() > 8) {  // keep a handful; drop the oldest
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ctx->batch_pool.front().p);
        (void)hipEventDestroy(ctx->batch_pool.front().freed);
        ctx->batch_pool.erase(ctx->batch_pool.begin());
    }
}

struct dsi_grid {
    dsi_context* ctx = nullptr;
    int nx = 0, ny = 0, nz = 0;
    size_t n = 0;
    float* data = nullptr;
    bool owned = false;
};

struct dsi_batch {
    dsi_context* ctx = nullptr;
    dsi_context::PoolBlock block{nullptr, 0, nullptr};  // one device allocation holding the four arrays below
    hipEvent_t ready = nullptr;  // recorded on the copy stream after the upload
    uint16_t *x = nullptr, *y = nullptr;
    uint32_t* first = nullptr;
    float* Rt = nullptr;
    size_t n_events = 0, n_packets = 0;
};

// grow-only device scratch of the exact tie resolver (dsi_mapper_resolve_near_ties), owned by the output mapper
struct TieScratch {
    DevBuf<uint32_t> cand, count, rank_count, rank_start, rank_cursor;
    DevBuf<uint2> desc;
    DevBuf<uint4> cols;
    // 32-bit words: [0] contending voxels, [1] near-tie columns (k_tie_candidates); [2] output segments handed out,
    // [3] overflow flags (k_tie_hits_binned); [4] max order difference (float bits), [5] max votes of a voxel (k_tie_pick);
    // [6] changed pixels (k_tie_pick); [8..9] one 64-bit word: real votes recorded; [16..23] planes with a contender (k_tie_desc)
    DevBuf<unsigned long long> counters;
    DevBuf<unsigned long long> keys, keys2;
    DevBuf<float> w, exact, diff;
    DevBuf<char> tmp;  // dsi_mapper_patch_depth_map: the new indices
    DevBuf<uint32_t> votes[8];  // dsi_mapper_prove_near_ties(_n): the cameras' vote counters per integer location
    bool votes_valid[8] = {false, false, false, false, false, false, false, false};
    DevBuf<unsigned> interval_most;  // dsi_mapper_reference_interval: most votes in a voxel since the last dsi_grid_prove_columns
    bool interval_most_live = false;
    DevBuf<uint2> unproven;     // ... and the columns it could not prove (pixel, float bits of the gap the column needs)
    size_t n_unproven = 0;
    unsigned* host = nullptr;  // page-locked copy of the counters: the three reads of a call are plain DMAs
    hipError_t host_counters(unsigned** out)
    {
        if (!host)
            if (hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&host), 64 * sizeof(unsigned), hipHostMallocDefault)) return e;
        *out = host;
        return hipSuccess;
    }
    void release()
    {
        if (host) (void)hipHostFree(host);
        host = nullptr;
        cand.release(); count.release(); desc.release(); cols.release(); counters.release();
        for (auto& v : votes) v.release();
        unproven.release(); interval_most.release();
        keys.release(); keys2.release(); w.release(); exact.release(); diff.release(); tmp.release();
        rank_count.release(); rank_start.release(); rank_cursor.release();
    }
};

struct dsi_mapper {
    dsi_context* ctx = nullptr;
    TieScratch tie;
    int sensor_w = 0, sensor_h = 0;
    dsi::Geom geom{};
    std::vector<float> planes;  // raw_depths_vec_ (of the planes this mapper owns)
    std::vector<float> planes_full;  // the whole depth vector (plane-sharded arg-max: index -> depth)
    float* planes_full_dev = nullptr;
    DevBuf<unsigned long long> argmax_keys;
    DevBuf<unsigned long long> fused_keys;   // dsi_mapper_depth_map_of_events: one arg-max key per pixel, kept zero between calls
    DevBuf<unsigned long long> fused_trace;  // test hook: time stamps of the fused kernel's phases
    bool fused_trace_on = false;
    int unit_multiplicity = 0;  // set only inside dsi_mapper_vote_statistics
    // fused kernel: records per (band, plane) pair of this camera's tables; the balanced partition; the cost of
    // a phase's set-up, pass switches, barriers and read-back in record units (measured with tools/fused_trace.py at
    // 512 x 512 x 200: a phase without records takes 8.6 us, a full one 18.7 us for ~38 k records)
    DevBuf<uint32_t> pair_work, fused_splits;
    DevBuf<unsigned long long> fused_prefix;
    int fused_fixed_cost = -1;  // < 0: equal pair counts per workgroup (default: balancing by records did not pay at
                                // 512 x 512 x 200 -- 446 vs 447 us, the slow workgroups are slow per record, not by count)
    int want_pass_lg = 0;  // test hook: log2(packets per pass) of the packed / vector-fill streams (0 = automatic)
    int plane_begin = 0;        // first owned plane of the full depth vector (plane sharding)
    float* planes_dev = nullptr;
    float2* lut_dev = nullptr;
    dsi_grid* grid = nullptr;
    // Opt-in experiment (DSI_PREP_OVERLAP=1): stage A, the packet sort and the coefficient tables of
    // evaluateDSI on a second stream, overlapping whatever the context's stream is still doing
    // (typically the previous camera's voting kernel).  Measured: the voting kernel loses more
    // (1.21 -> 1.38 ms: wave slots, LDS and L2 taken by the preparation kernels) than the overlap
    // hides, 2.80 -> 2.92 ms per step, so it is off by default.
    hipStream_t prep_stream = nullptr;
    hipEvent_t ev_prep = nullptr, ev_vote_done = nullptr;
    bool vote_recorded = false;
    int prep_overlap = 0;
    int algo = DSI_VOTE_AUTO;
    int want_band_rows = 0, want_chunks = 0, want_block = 0;
    int want_packed = -1;  // -1 automatic, 0 per-packet waves, 1 packed lanes, 2 packet groups
    bool keep_z0 = false;  // tests: materialise event_locations_z0 (stage A as separate kernels)
    dsi_vote_info_t info{};
    // scratch
    DevBuf<float> centers, H, Rt_tmp, conf, depth;
    DevBuf<unsigned long long> partials;  // [chunks][partial_stride]: raw 64-bit sums per chunk (several chunks, or accumulation)
    DevBuf<float2> xy;
    DevBuf<dsi::EvRec> sxy;
    DevBuf<unsigned long long> seam;  // [chunks][nz][bands][2][nx]: 64-bit sums of each band's first row and of the row below it
    DevBuf<uint32_t> nvalid, cuts, gcuts;
    size_t cuts_inline_min_packets = 8192;   // vote_device: from this many packets on the vector fill derives its cuts itself
    bool info_cuts_inline = false;           // the last banded vote did
    DevBuf<uint8_t> spk;
    DevBuf<uint16_t> rowstart;
    DevBuf<dsi::PlaneCoef> coef;
    DevBuf<uint8_t> idx, conf8, mask, idx_filtered;
    DevBuf<uint32_t> minmax;
    bool depth_valid = false;
    bool fused_keys_dirty = false;  // the fused kernel may have written keys that no unpack cleared
    // The depth map leaves the device on the context's COPY stream, so that fetching window w's map
    // does not queue behind window w+1's kernels on the compute stream: "ready" (compute stream, after
    // the arg-max) gates the copies, "read" (copy stream, after them) gates the next arg-max into the
    // same buffers.
    hipEvent_t ev_depth_ready = nullptr, ev_depth_read = nullptr;
    bool depth_read_pending = false;
    // HIP-event stopwatch around the dominant (voting) kernel, for bench.py's roofline
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timing_pairs;  // recorded, not yet read
    std::vector<hipEvent_t> timing_pool;
};

namespace {

int set_device(const dsi_context* ctx)
{
    HIP_TRY(hipSetDevice(ctx->device));
    return DSI_OK;
}

// ShapeDSI + setupDSI (mapper_emvs_stereo.cpp:208-241), depth_vector.hpp:76-163
void make_planes(float min_depth, float max_depth, int nz, bool inverse, std::vector<float>* out)
{
    if (min_depth > max_depth) std::swap(min_depth, max_depth);  // depth_vector.hpp:33-36
    out->resize(nz);
    if (!inverse) {
        const float mult = (float)nz / (max_depth - min_depth);  // :88
        for (int i = 0; i < nz; ++i) (*out)[i] = min_depth + (float)i / mult;  // :93
    } else {
        const float inv_min = 1.f / min_depth, inv_max = 1.f / max_depth;  // :131-132
        const float mult = (float)nz / (inv_min - inv_max);
        for (int i = 0; i < nz; ++i) {
            const float rho = inv_max + (float)i / mult;  // :138
            (*out)[i] = 1.f / rho;                        // :145-148
        }
    }
}

float virtual_focal(float cam_fx, float fov_deg, int dim_x)
{
    if (fov_deg < 10.f) return cam_fx;  // mapper_emvs_stereo.cpp:220-224
    const float fov_rad = (float)((double)fov_deg * 3.1415926535897932384626433832795 / 180.0);
    return (float)(0.5 * (double)(float)dim_x / std::tan(0.5 * (double)fov_rad));  // :227-228
}

// Choose the LDS band decomposition for a grid and a packet count.
bool plan_bands(const dsi_mapper* m, size_t n_packets, dsi::BandPlan* bp)
{
    const dsi::Geom& g = m->geom;
    // Q33.31 LDS accumulators, one per cell -- or, lane mapping 8 (opt-in, never chosen automatically), rows of paired 32-bit
    // cells: 2 * ((nx >> 1) + 1) words of 8 bytes
    const size_t row_bytes = (size_t)(m->want_packed == 8 ? dsi::paired_row_words_host(g.nx) : g.nx) * sizeof(unsigned long long);
    const int block_threads = m->want_packed == 8 ? 1024 : (m->want_block > 0 ? m->want_block : 1024);
    // Lane mapping and workgroups per CU, by the records of a packet a band sees (its run):
    // rows + 1 of Ny rows see 1024 * (rows + 1) / Ny events of a packet.
    //  * TWO 1024-thread workgroups per CU (32 waves; half the LDS each) with the packed mapping and its
    //    hand-scheduled scalar run bookkeeping (1) when half the LDS still gives runs of >= ~56
    //    records (346x260x100: 106; measured against one workgroup per CU with either mapping, 10 M
    //    events: 480x360x100 (57) 1.33 vs 1.44 / 1.48 ms, 400x300x64 (85) 0.89 vs 0.91 / 0.96);
    //  * else ONE workgroup per CU; the scalar bookkeeping per run piece then dominates when the run
    //    is short and the vector fill (5) takes over below ~72 records (mapping 1 / mapping 5:
    //    1024x1024x256 (20) 7.6 / 4.25 ms, 800x600x128 (42) 2.43 / 1.84, 720x540x100 (53) 1.68 / 1.48,
    //    640x480x100 (68) 1.55 / 1.43, 512x512x200 (78) 2.81 / 2.63 at 10 M events but 0.216 / 0.222
    //    at a 0.5 M-event window and 0.379 / 0.380 at 1 M).
    // (The packed mapping is 1.1x (240x180) to 3x faster than the per-packet mapping 0.)
    int packed = m->want_packed;
    bool auto_two_per_cu = false;
    if (packed < 0) {
        const long rows_full = std::max<long>(1, (long)(dsi::max_dynamic_lds() / row_bytes) - 1);
        const long rows_half = (long)((dsi::max_dynamic_lds() / 2) / row_bytes) - 1;
        // (round 3: wherever the packed mapping is chosen it is lane mapping 7 -- the same hand-scheduled loop with
        //  DEALT passes: 10 M events, same box, mapping 1 / 7: 346x260x100 1.198 / 1.187 ms, 240x180x100 1.187 / 1.150,
        //  512x512x200 2.80 / 2.53; the vector fill keeps its range: 640x480x100 1.40 (5) vs 1.46 (7), 800x600x128
        //  1.78 vs 2.07, 1024x1024x256 4.32 vs 6.42)
        if (rows_half >= 4 && 1024L * (rows_half + 1) / g.ny >= 56) {
            packed = 7;
            auto_two_per_cu = true;
        } else {
            packed = 1024L * (rows_full + 1) / g.ny < 72 ? 5 : 7;
        }
    }
    // lane mapping 5 keeps 64 tail-bit words per wave behind the band
    const size_t scratch_bytes = (packed == 5 || packed == 6) ? (size_t)(block_threads / 64) * dsi::kVfillScratchWords * 8 : 0;
    const long max_rows_total = (long)((dsi::max_dynamic_lds() - scratch_bytes) / row_bytes);
    if (max_rows_total < 2 || g.nx < 2 || g.ny < 2 || g.ny > 16000) return false;
    long max_owned = max_rows_total - 1;  // + the carry row
    if (m->want_band_rows > 0) {
        max_owned = std::min<long>(max_owned, m->want_band_rows);
    } else {
        // two workgroups per CU (above; a mapping forced by the caller keeps the older, stricter rule)
        const long half_rows = (long)((dsi::max_dynamic_lds() / 2 - scratch_bytes) / row_bytes) - 1;
        if (half_rows >= 4 && (auto_two_per_cu || 1024L * (half_rows + 1) / g.ny >= 96)) max_owned = half_rows;
    }
    int bands = (int)((g.ny + max_owned - 1) / max_owned);
    int band_rows = (g.ny + bands - 1) / bands;  // balanced
    if (m->want_band_rows > 0) band_rows = (int)std::min<long>(m->want_band_rows, max_owned);
    bands = (g.ny + band_rows - 1) / band_rows;
    bp->bands = bands;
    bp->band_rows = band_rows;
    bp->scratch_offset = (int)((size_t)(band_rows + 1) * row_bytes);
    bp->lds_bytes = (size_t)(band_rows + 1) * row_bytes + scratch_bytes;
    bp->block_threads = block_threads;
    bp->row_pad = std::min(g.ny, 4096);  // z0 locations spill up to ~ny rows outside the grid
    // expected events of one packet in one band
    const long run = 1024L * (band_rows + 1) / g.ny;
    bp->packed = packed;
    // mapping 2 sorts S consecutive packets together so that a run holds >= ~512 events
    // (mapping 4, the hand-scheduled loop, packs the runs of consecutive groups into the lanes, so
    //  it only needs runs of a few batches)
    const long want_run = bp->packed == 4 ? 160 : 512;
    int S = 1;
    while (S < 32 && (long)S * std::max<long>(run, 1) < want_run) S <<= 1;
#ifdef DSI_TIMING_EXPERIMENTS  // (tuning knobs exist only in builds made with build.py --experiments)
    if (const char* e = std::getenv("DSI_GROUP_PACKETS")) {
        const int v = std::atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32) S = v;
    }
#endif
    bp->group_packets = S;
    // Persistent workgroups (the grid is what the chip holds at once; workgroups pull work items from
    // per-XCD counters) pay off when only ONE workgroup fits a CU -- nothing else hides the ~5 us it
    // takes to retire a 16-wave, 160 KB workgroup and launch the next (512x512x200, 500 k events:
    // 0.232 -> 0.222 ms).  With two per CU the hardware dispatcher already overlaps them and the item
    // loop only adds a barrier and an atomic (346x260x100: 1.175 ms plain, 1.204 ms persistent).
    const bool two_per_cu = bp->lds_bytes * 2 <= dsi::max_dynamic_lds() && bp->block_threads <= 1024;
    bp->persistent = ((bp->packed == 1 || bp->packed == 3 || bp->packed == 5 || bp->packed == 6 || bp->packed == 7 || bp->packed == 8) && !two_per_cu) ? 1 : 0;
    bp->experiment = 0;
    bp->pass_lg = (m->want_pass_lg >= 1 && m->want_pass_lg <= 6) ? m->want_pass_lg : 0;  // test hook dsi_test_pass_lg
#ifdef DSI_TIMING_EXPERIMENTS
    // Environment knobs of the timing experiments quoted in NOTEBOOK.md.  They are compiled in ONLY by
    // `python -m dvs_mcemvs_amd.build --experiments` (a stray variable in a job's environment must
    // not be able to change -- DSI_EXPERIMENT: corrupt -- the results of the production library).
    if (const char* e = std::getenv("DSI_PERSISTENT")) {  // A/B: 0 off, 1 on wherever the kernel supports it
        const int v = std::atoi(e);
        bp->persistent = (v != 0 && (bp->packed == 1 || bp->packed == 3 || bp->packed == 5 || bp->packed == 6 || bp->packed == 7)) ? 1 : 0;
    }
    if (const char* e = std::getenv("DSI_EXPERIMENT")) {  // 1 no votes, 2 no flush: WRONG results, timing only
        bp->experiment = std::atoi(e);
        static bool warned = false;
        if (!warned && bp->experiment) {
            std::fprintf(stderr, "dsi_engine: DSI_EXPERIMENT=%d -- timing experiment, the DSIs are WRONG\n", bp->experiment);
            warned = true;
        }
    }
    if (const char* e = std::getenv("DSI_PASS_LG")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 6) bp->pass_lg = v;
    }
#endif
    int chunks = m->want_chunks;
    if (chunks <= 0) {
        // Work items = chunks x bands x planes: ~8 per CU keep the tail of the last round short; every
        // chunk beyond the first costs a partial volume to write and to reduce.
        // (round 3: a chunk's partial volume holds raw 64-bit sums, 8 bytes per voxel to write and to read back, so
        //  fewer chunks than in round 2: 346x260x100, 10 M events: 1 / 2 / 4 chunks 2.695 / 2.668 / 2.717 ms per step)
        const long items_target = 4L * 256;
        chunks = (int)std::max<long>(1, (items_target + (long)bands * g.nz - 1) / ((long)bands * g.nz));
        // below ~8 M events one chunk is faster when the planes alone fill the chip 1.5 times: no
        // partial volumes to write and reduce, fewer workgroup set-ups (measured at 346x260x100:
        // 1.85x at 0.1 M events, 1.25x at 1 M, 1.03x at 5 M, 0.99x at 10 M)
        // (round 3, dealt passes + raw partial volumes: one chunk also at 10 M events -- 346x260x100, same box,
        //  1 / 2 / 3 / 4 chunks: 2.633 / 2.675 / 2.717 / 2.751 ms per step)
        if ((long)bands * g.nz >= 768) chunks = 1;
        // keep (chunks * bands) a multiple of 8 so that all XCDs get the same number of pairs
        int step = 8;
        for (int f = 2; f <= 8; f *= 2)
            if (bands % f == 0) step = 8 / f;
        if (chunks > 1) {  // whole groups of 8 pairs -- unless that multiplies the partial volumes (an odd
                           // number of bands would turn 2 chunks into 8: 640x480x100 with 17 bands paid
                           // 0.3 ms per step for it); the pairs beyond the last whole group are dealt plane
                           // by plane over all XCDs anyway
            const int rounded = ((chunks + step - 1) / step) * step;
            if (rounded <= chunks + chunks / 2) chunks = rounded;
        }
        chunks = (int)std::min<size_t>((size_t)chunks, std::max<size_t>(1, n_packets));
        const size_t vol_bytes = (size_t)g.nx * g.ny * g.nz * sizeof(float);
        const size_t budget = (size_t)16 << 30;  // partial DSIs (8 bytes per voxel) may use up to 16 GiB of HBM
        while (chunks > 1 && (size_t)chunks * (2 * vol_bytes + 32) > budget) --chunks;
    }
    bp->chunks = std::max(1, chunks);
    return true;
}

// Band decomposition of the fused vote -> camera fusion -> arg-max kernel (k_vote_fuse_argmax): one
// 1024-thread workgroup per CU, the band as tall as the LDS (and the per-thread register arrays) allow,
// a halo row on either side of the owned rows instead of a carry row, one chunk.
bool plan_fused(const dsi_mapper* m, size_t n_packets_max, dsi::BandPlan* bp, int n_cameras = 2)
{
    const dsi::Geom& g = m->geom;
    const size_t row_bytes = (size_t)g.nx * sizeof(unsigned long long);
    if (g.nx < 2 || g.ny < 2 || g.ny > 16000 || g.nz > 256) return false;
    int packed = m->want_packed;
    if (!(packed == 1 || packed == 3 || packed == 5 || packed == 6)) {
        // the rule of plan_bands for one workgroup per CU: vector fill below ~72 records per (packet, band)
        const long rows_full = (long)std::min(dsi::max_dynamic_lds() / row_bytes, dsi::fused_max_cells(1, n_cameras) / (size_t)g.nx);
        packed = 1024L * rows_full / g.ny < 72 ? 5 : 1;
    }
    // the hand-scheduled loops address records with 32-bit byte offsets (see vote_device)
    if ((n_packets_max + 1) * dsi::kPacket * sizeof(dsi::EvRec) > 0xffffffffull) packed = packed == 1 ? 3 : (packed == 5 ? 6 : packed);
    const size_t scratch_bytes = (packed == 5 || packed == 6) ? (size_t)(1024 / 64) * dsi::kVfillScratchWords * 8 : 0;
    const long max_rows_total =
        (long)std::min((dsi::max_dynamic_lds() - scratch_bytes) / row_bytes, dsi::fused_max_cells(packed, n_cameras) / (size_t)g.nx);
    if (max_rows_total < 3) return false;
    long max_owned = max_rows_total - 2;  // + the two halo rows
    if (m->want_band_rows > 0) max_owned = std::min<long>(max_owned, m->want_band_rows);
    int bands = (int)((g.ny + max_owned - 1) / max_owned);
    int band_rows = (g.ny + bands - 1) / bands;  // balanced
    if (m->want_band_rows > 0) band_rows = (int)max_owned;
    bands = (g.ny + band_rows - 1) / band_rows;
    *bp = dsi::BandPlan{};
    bp->bands = bands;
    bp->band_rows = band_rows;
    bp->chunks = 1;
    bp->block_threads = 1024;
    bp->packed = packed;
    bp->group_packets = 1;
    bp->row_pad = std::min(g.ny, 4096);
    bp->pass_lg = (m->want_pass_lg >= 1 && m->want_pass_lg <= 6) ? m->want_pass_lg : 0;
    bp->scratch_offset = (int)((size_t)(band_rows + 2) * row_bytes);
    bp->lds_bytes = (size_t)(band_rows + 2) * row_bytes + scratch_bytes;
    bp->persistent = 0;
    bp->halo = 1;
    bp->experiment = 0;
    bp->interleave = -1;  // decided by the caller from the cameras' packet counts (depth_map_of_events_impl)
#ifdef DSI_TIMING_EXPERIMENTS
    if (const char* e = std::getenv("DSI_FUSED_2CU"))  // 0: a small band still runs one workgroup per CU (A/B)
        if (std::atoi(e) == 0) bp->experiment = 300;
    if (const char* e = std::getenv("DSI_FUSED_DEFER"))  // 0: camera 1's fusion + arg-max inside its read-back, as before round 6 (A/B)
        if (std::atoi(e) == 0) bp->experiment = 301;
    if (const char* e = std::getenv("DSI_FUSED_INTERLEAVE"))  // A/B: 0 contiguous pieces, 1 pairs in turn, 2 pairs drawn
        bp->interleave = std::min(2, std::max(0, std::atoi(e)));
#endif
    return true;
}

int upload_async(dsi_context* ctx, void* dst, const void* src, size_t bytes)
{
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return DSI_OK;
}

hipEvent_t timing_event(dsi_mapper* m)
{
    if (!m->timing_pool.empty()) {
        hipEvent_t e = m->timing_pool.back();
        m->timing_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct VoteTimer {  // records an event pair around the voting kernel when timing is on
    dsi_mapper* m;
    hipEvent_t a = nullptr, b = nullptr;
    explicit VoteTimer(dsi_mapper* m_) : m(m_)
    {
        if (!m->timing) return;
        a = timing_event(m);
        b = timing_event(m);
        if (a && b) (void)hipEventRecord(a, m->ctx->stream);
    }
    void stop()
    {
        if (!a || !b) return;
        (void)hipEventRecord(b, m->ctx->stream);
        m->timing_pairs.emplace_back(a, b);
        a = b = nullptr;
    }
};

// The stream the preparation kernels of an evaluate call go to: the mapper's own second stream,
// ordered after this mapper's previous vote (which still reads the scratch tables), or the
// context's stream when overlapping is off.
int prep_begin(dsi_mapper* m, hipStream_t* ps)
{
    *ps = m->ctx->stream;
    if (!m->prep_overlap) return DSI_OK;
    if (!m->prep_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&m->prep_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_prep, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_vote_done, hipEventDisableTiming));
    }
    if (m->vote_recorded) HIP_TRY(hipStreamWaitEvent(m->prep_stream, m->ev_vote_done, 0));
    *ps = m->prep_stream;
    return DSI_OK;
}

// the context's stream continues only after the preparation kernels
int prep_end(dsi_mapper* m, hipStream_t ps)
{
    if (ps == m->ctx->stream) return DSI_OK;
    HIP_TRY(hipEventRecord(m->ev_prep, ps));
    HIP_TRY(hipStreamWaitEvent(m->ctx->stream, m->ev_prep, 0));
    return DSI_OK;
}

int vote_done(dsi_mapper* m)
{
    if (!m->prep_stream) return DSI_OK;
    HIP_TRY(hipEventRecord(m->ev_vote_done, m->ctx->stream));
    m->vote_recorded = true;
    return DSI_OK;
}

// before a kernel overwrites the mapper's depth-map buffers: the previous fetch must have read them
int depth_buffers_acquire(dsi_mapper* m)
{
    if (m->depth_read_pending) {
        HIP_TRY(hipStreamWaitEvent(m->ctx->stream, m->ev_depth_read, 0));
        m->depth_read_pending = false;
    }
    return DSI_OK;
}

// after the kernels that produce the depth map have been queued on the mapper's stream
int depth_buffers_ready(dsi_mapper* m)
{
    if (!m->ev_depth_ready) {
        HIP_TRY(hipEventCreateWithFlags(&m->ev_depth_ready, hipEventDisableTiming));
        // (the in-order fetch STORES the maps into mapped host memory from a kernel: the event that says "read" must
        //  release those stores to the system, or the host may see stale maps in non-coherent pinned memory -- ADVICE r05)
        HIP_TRY(hipEventCreateWithFlags(&m->ev_depth_read, hipEventDisableTiming | hipEventReleaseToSystem));
    }
    HIP_TRY(hipEventRecord(m->ev_depth_ready, m->ctx->stream));
    m->depth_valid = true;
    return DSI_OK;
}

// device -> host copies of the depth map: on the context's copy stream behind the arg-max only (see
// dsi_mapper::ev_depth_ready), or -- in_order -- on the compute stream behind whatever it holds
int depth_buffers_fetch(dsi_mapper* m, float* depth_host, float* conf_host, uint8_t* idx_host, bool wait, bool in_order = false)
{
    dsi_context* ctx = m->ctx;
    hipStream_t cs = ctx->stream;
    const size_t npix = (size_t)m->geom.nx * m->geom.ny;
    if (!in_order) {
        HIP_TRY(copy_stream_of(ctx, &cs));
        HIP_TRY(hipStreamWaitEvent(cs, m->ev_depth_ready, 0));
    } else {
        // destinations in mapped page-locked memory (dsi_host_alloc): the CUs store the maps there, no copy engine involved
        // (a device -> host copy queued behind this window's kernel would hold up the next window's uploads)
        auto mapped = [](void* host, void** dev) {
            *dev = nullptr;
            if (!host) return true;
            hipPointerAttribute_t attr{};
            if (hipPointerGetAttributes(&attr, host) != hipSuccess) {
                (void)hipGetLastError();  // (an ordinary pointer: not an error of ours)
                return false;
            }
            if (attr.type != hipMemoryTypeHost) return false;
            return hipHostGetDevicePointer(dev, host, 0) == hipSuccess && *dev;
        };
        void *dd = nullptr, *dc = nullptr, *di = nullptr;
        // (the kernel stores the index map as 32-bit words: an interior, misaligned idx_host takes the copy path below)
        if (mapped(depth_host, &dd) && mapped(conf_host, &dc) && mapped(idx_host, &di) && npix % 4 == 0 &&
            reinterpret_cast<uintptr_t>(di) % 4 == 0) {
            HIP_TRY(dsi::launch_store_depth_map(cs, m->depth.p, m->conf.p, m->idx.p, npix, static_cast<float*>(dd),
                                                static_cast<float*>(dc), static_cast<uint8_t*>(di)));
            HIP_TRY(hipEventRecord(m->ev_depth_read, cs));
            m->depth_read_pending = true;
            if (wait) HIP_TRY(hipEventSynchronize(m->ev_depth_read));
            return DSI_OK;
        }
    }
    if (depth_host) HIP_TRY(hipMemcpyAsync(depth_host, m->depth.p, npix * sizeof(float), hipMemcpyDeviceToHost, cs));
    if (conf_host) HIP_TRY(hipMemcpyAsync(conf_host, m->conf.p, npix * sizeof(float), hipMemcpyDeviceToHost, cs));
    if (idx_host) HIP_TRY(hipMemcpyAsync(idx_host, m->idx.p, npix, hipMemcpyDeviceToHost, cs));
    HIP_TRY(hipEventRecord(m->ev_depth_read, cs));
    m->depth_read_pending = true;
    if (wait) HIP_TRY(hipEventSynchronize(m->ev_depth_read));
    return DSI_OK;
}

// fillVoxelGrid on device data.  xy: np*1024 z0 locations (reference order);
// centers: np*3.  accumulate != 0: add to the grid's current contents.  ps: the stream the sort and
// coefficient kernels go to (see prep_begin); the caller has already put xy / centers on it.
int vote_device(dsi_mapper* m, const float2* xy, const float* centers, size_t np, bool accumulate,
                hipStream_t ps, const dsi_batch* raw = nullptr)
{
    // raw != nullptr: the events of `raw` have not been through stage A yet (xy / centers are the
    // mapper's buffers to fill).  The LDS-band path with per-packet sorting does stage A inside the
    // sort kernel; the other paths run the two stage-A kernels first.
    auto stage_a = [&]() -> int {
        if (!raw || np == 0) return DSI_OK;
        HIP_TRY(m->H.reserve(np * 9));
        HIP_TRY(m->xy.reserve(np * dsi::kPacket));
        HIP_TRY(dsi::launch_packet_geometry(ps, raw->Rt, (int)np, m->geom, m->centers.p, m->H.p));
        HIP_TRY(dsi::launch_warp_z0(ps, raw->x, raw->y, raw->first, (int)np, m->H.p, m->lut_dev, m->sensor_w, m->sensor_h,
                                    m->xy.p));
        xy = m->xy.p;
        return DSI_OK;
    };
    dsi_context* ctx = m->ctx;
    dsi_grid* g = m->grid;
    const dsi::Geom& geom = m->geom;
    dsi::BandPlan bp{};
    int algo = m->algo;
    const bool can_band = plan_bands(m, np, &bp);
    if (algo == DSI_VOTE_AUTO) algo = can_band ? DSI_VOTE_LDS_BANDS : DSI_VOTE_GLOBAL_ATOMIC;
    if (algo == DSI_VOTE_LDS_BANDS && !can_band)
        return fail(DSI_ERR_INVALID, "grid rows of %d floats do not fit the LDS band kernel", geom.nx);

    m->info = dsi_vote_info_t{};
    m->info.algo = algo;
    m->info.n_packets = np;
    m->depth_valid = false;

    if (algo == DSI_VOTE_GLOBAL_ATOMIC) {
        if (int rc = stage_a()) return rc;
        if (int rc = prep_end(m, ps)) return rc;
        if (!accumulate) HIP_TRY(hipMemsetAsync(g->data, 0, g->n * sizeof(float), ctx->stream));
        VoteTimer vt(m);
        HIP_TRY(dsi::launch_vote_global(ctx->stream, xy, centers, (int)np, m->planes_dev, geom, g->data));
        vt.stop();
        return vote_done(m);
    }

    // the hand-scheduled loops address records with 32-bit byte offsets (12 B per record):
    // beyond 2^32 / 12 records (349,525 packets = 358 M events in ONE call) use the compiled loops
    if ((np + 1) * dsi::kPacket * sizeof(dsi::EvRec) > 0xffffffffull) {
        if (bp.packed == 1 || bp.packed == 7) bp.packed = 3;
        if (bp.packed == 8) return fail(DSI_ERR_INVALID, "lane mapping 8 (paired cells) addresses records with 32 bits: %zu packets are too many", np);
        if (bp.packed == 4) bp.packed = 2;
        if (bp.packed == 5) bp.packed = 6;
    }

    m->info.bands = bp.bands;
    m->info.band_rows = bp.band_rows;
    m->info.chunks = bp.chunks;
    m->info.block_threads = bp.block_threads;
    m->info.lds_bytes = bp.lds_bytes;
    m->info.packed = bp.packed;
    m->info.group_packets = bp.group_packets;
    if (np == 0) {
        if (int rc = prep_end(m, ps)) return rc;
        if (!accumulate) HIP_TRY(hipMemsetAsync(g->data, 0, g->n * sizeof(float), ctx->stream));
        return vote_done(m);
    }
    HIP_TRY(m->sxy.reserve(np * dsi::kPacket + 1));  // + the multiplicity-0 dummy record
    HIP_TRY(m->nvalid.reserve(np + (size_t)geom.nz + 9));  // + one "needs IEEE divide" word per plane + 8 work counters + the paired cells' overflow word
    HIP_TRY(m->rowstart.reserve(np * (size_t)(geom.ny + 2 * bp.row_pad + 3) + 2));  // (+ slack: k_plane_coef copies 32-bit words)
    HIP_TRY(m->coef.reserve(np * geom.nz + 1));       // + the dummy record's "coefficients"
    // Wide grids with many packets (the vector fill): no cut table -- at 1024 x 1024 x 256 with 100 M events it is 6.1 GB per
    // camera and 3.9 ms to write -- but the packets' row tables transposed (u16 [ny + 2 pad + 3][stride], 0.6 GB there), from
    // which the voting kernel's passes derive their runs (BandPlan::cuts_inline).  Below ~8 k packets the table stays: a work
    // item is then short and the two dependent look-ups at its start would show.
    bp.rs_stride = (int)((np + 63) / 64 * 64);
    // (the voting kernel addresses the transposed table with 32-bit byte offsets)
    const bool rs_fits = (size_t)(geom.ny + 2 * bp.row_pad + 3) * (size_t)bp.rs_stride * 2 < ((size_t)1 << 32);
    bp.cuts_inline = ((bp.packed == 5 || bp.packed == 6) && np >= (size_t)m->cuts_inline_min_packets && rs_fits) ? 1 : 0;
    if (bp.cuts_inline)
        HIP_TRY(m->cuts.reserve(((size_t)(geom.ny + 2 * bp.row_pad + 3) * (size_t)bp.rs_stride + 1) / 2));  // (u16 entries in u32 words)
    else
        HIP_TRY(m->cuts.reserve(np * geom.nz * bp.bands));
    m->info_cuts_inline = bp.cuts_inline != 0;
    HIP_TRY(m->seam.reserve((size_t)bp.chunks * geom.nz * bp.bands * 2 * geom.nx));
    const bool direct = (bp.chunks == 1 && !accumulate);
    bp.raw_out = direct ? 0 : 1;
    if (!direct) HIP_TRY(m->partials.reserve((size_t)bp.chunks * dsi::partial_stride(g->n)));

    if (bp.packed == 2 || bp.packed == 4) {
        if (int rc = stage_a()) return rc;
        const int S = bp.group_packets;
        const size_t ngroups = (np + S - 1) / S;
        HIP_TRY(m->spk.reserve(np * dsi::kPacket));
        HIP_TRY(m->gcuts.reserve(ngroups * geom.nz * bp.bands));
        HIP_TRY(m->rowstart.reserve(ngroups * (size_t)(geom.ny + 2 * bp.row_pad + 3)));
        HIP_TRY(dsi::launch_sort_groups(ps, xy, (int)np, S, geom.ny, geom.nz, bp.row_pad, m->sxy.p, m->spk.p,
                                        m->nvalid.p, m->rowstart.p));
        // per-packet coefficients + row-bin ranges (cuts buffer), then the per-group runs
        HIP_TRY(dsi::launch_plane_coef(ps, centers, m->planes_dev, m->rowstart.p, m->nvalid.p, (int)np,
                                       geom, bp, m->coef.p, m->cuts.p));
        HIP_TRY(dsi::launch_group_cuts(ps, m->cuts.p, m->rowstart.p, (int)np, S, geom, bp, m->gcuts.p));
        if (int rc = prep_end(m, ps)) return rc;
        VoteTimer vt(m);
        HIP_TRY(dsi::launch_vote_groups(ctx->stream, m->sxy.p, m->spk.p, m->coef.p, m->gcuts.p, m->nvalid.p + np, (int)np, S, geom,
                                        bp, direct ? (void*)g->data : (void*)m->partials.p, m->seam.p));
        vt.stop();
    } else {
    if (raw && !m->keep_z0) {
        HIP_TRY(dsi::launch_sort_packets_raw(ps, raw->Rt, raw->x, raw->y, raw->first, m->lut_dev, m->sensor_w, m->sensor_h, geom,
                                             m->centers.p, (int)np, bp.row_pad, m->sxy.p, m->nvalid.p,
                                             m->rowstart.p, m->unit_multiplicity));
    } else {
        if (int rc = stage_a()) return rc;
        HIP_TRY(dsi::launch_sort_packets(ps, xy, (int)np, geom.ny, geom.nz, bp.row_pad, m->sxy.p, m->nvalid.p,
                                         m->rowstart.p));
    }
    HIP_TRY(dsi::launch_plane_coef(ps, centers, m->planes_dev, m->rowstart.p, m->nvalid.p, (int)np,
                                   geom, bp, m->coef.p, m->cuts.p));
    if (int rc = prep_end(m, ps)) return rc;
    VoteTimer vt(m);
    HIP_TRY(dsi::launch_vote_bands(ctx->stream, m->sxy.p, m->coef.p, m->cuts.p, m->nvalid.p + np, (int)np, geom, bp,
                                   direct ? (void*)g->data : (void*)m->partials.p, m->seam.p));
    vt.stop();
    }
    // seam rows from the exact 64-bit sums of the two bands that meet there; then (several chunks, or accumulation) the
    // sum over the chunks' raw partial volumes -- which takes the seam rows straight from the bands' sums when it can
    const bool fold = !direct && dsi::reduce_can_fold_seams(geom, g->n);
    if (!fold)
        HIP_TRY(dsi::launch_seam_rows(ctx->stream, m->seam.p, bp.chunks, geom, bp, direct ? (void*)g->data : (void*)m->partials.p));
    if (!direct)
        HIP_TRY(dsi::launch_reduce_partials(ctx->stream, m->partials.p, bp.chunks, g->n, g->data, accumulate ? 1 : 0,
                                            fold ? m->seam.p : nullptr, &geom, &bp));
    return vote_done(m);
}

bool same_shape(const dsi_grid* a, const dsi_grid* b)
{
    return a->nx == b->nx && a->ny == b->ny && a->nz == b->nz;
}

}  // namespace

extern "C" {

const char* dsi_last_error(void) { return g_last_error.c_str(); }
int dsi_abi_version(void) { return DSI_ENGINE_ABI_VERSION; }

int dsi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* ------------------------------------------------------------------ context */
int dsi_context_create(int device_id, dsi_context_t** out)
{
    REQUIRE(out, DSI_ERR_INVALID, "out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(DSI_ERR_NO_DEVICE, "no HIP device visible (this engine has no CPU fallback)");
    REQUIRE(device_id >= 0 && device_id < n, DSI_ERR_NO_DEVICE, "device %d out of range (%d devices)",
            device_id, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(DSI_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 only", device_id,
                    prop.gcnArchName);
    HIP_TRY(hipSetDevice(device_id));
    dsi_context* ctx = new (std::nothrow) dsi_context();
    REQUIRE(ctx, DSI_ERR_INVALID, "out of host memory");
    ctx->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = upload_stream_acquire(device_id, &ctx->upload_stream);
    if (e == hipSuccess) e = hipEventCreate(&ctx->t0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->t1);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ctx->ms_accum), sizeof(double));
    if (e != hipSuccess) {
        dsi_context_destroy(ctx);
        return fail(DSI_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return DSI_OK;
}

int dsi_context_destroy(dsi_context_t* ctx)
{
    if (!ctx) return DSI_OK;
    REQUIRE(ctx->children.load() == 0, DSI_ERR_CONTEXT,
            "%d object(s) created from this context (grids, mappers, batches) are still alive: destroy them first",
            ctx->children.load());
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->upload_stream) {
        (void)hipStreamSynchronize(ctx->upload_stream);
        upload_stream_release(ctx->device);
        ctx->upload_stream = nullptr;
    }
    for (auto& blk : ctx->batch_pool) {
        (void)hipFree(blk.p);
        (void)hipEventDestroy(blk.freed);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->ms_accum) (void)hipFree(ctx->ms_accum);
    if (ctx->collapse_scratch) (void)hipFree(ctx->collapse_scratch);
    for (hipEvent_t ev : ctx->marks) (void)hipEventDestroy(ev);
    ctx->marks.clear();
    if (ctx->t0) (void)hipEventDestroy(ctx->t0);
    if (ctx->t1) (void)hipEventDestroy(ctx->t1);
    if (ctx->sync_ev) (void)hipEventDestroy(ctx->sync_ev);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return DSI_OK;
}

int dsi_context_synchronize(dsi_context_t* ctx)
{
    REQUIRE(ctx, DSI_ERR_INVALID, "ctx is null");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DSI_OK;
}

int dsi_context_wait_for(dsi_context_t* ctx, dsi_context_t* other)
{
    REQUIRE(ctx && other, DSI_ERR_INVALID, "ctx is null");
    REQUIRE(ctx->device == other->device, DSI_ERR_INVALID, "contexts are on different devices");
    if (ctx == other) return DSI_OK;
    if (int rc = set_device(ctx)) return rc;
    if (!other->sync_ev) HIP_TRY(hipEventCreateWithFlags(&other->sync_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(other->sync_ev, other->stream));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, other->sync_ev, 0));
    return DSI_OK;
}

void* dsi_context_stream(dsi_context_t* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int dsi_context_device(dsi_context_t* ctx) { return ctx ? ctx->device : -1; }

int dsi_context_timer_start(dsi_context_t* ctx)
{
    REQUIRE(ctx, DSI_ERR_INVALID, "ctx is null");
    HIP_TRY(hipEventRecord(ctx->t0, ctx->stream));
    return DSI_OK;
}

int dsi_context_timer_stop(dsi_context_t* ctx, float* elapsed_ms)
{
    REQUIRE(ctx && elapsed_ms, DSI_ERR_INVALID, "null argument");
    HIP_TRY(hipEventRecord(ctx->t1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->t1));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, ctx->t0, ctx->t1));
    return DSI_OK;
}

int dsi_context_timeline_mark(dsi_context_t* ctx)
{
    REQUIRE(ctx, DSI_ERR_INVALID, "context is null");
    if (int rc = set_device(ctx)) return rc;
    if (ctx->n_marks == ctx->marks.size()) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        ctx->marks.push_back(e);
    }
    HIP_TRY(hipEventRecord(ctx->marks[ctx->n_marks], ctx->stream));
    ++ctx->n_marks;
    return DSI_OK;
}

int dsi_context_timeline_read(dsi_context_t* ctx, float* intervals_ms, size_t capacity, size_t* n_intervals)
{
    REQUIRE(ctx && n_intervals, DSI_ERR_INVALID, "null argument");
    if (int rc = set_device(ctx)) return rc;
    const size_t n = ctx->n_marks > 0 ? ctx->n_marks - 1 : 0;
    *n_intervals = n;
    if (ctx->n_marks) HIP_TRY(hipEventSynchronize(ctx->marks[ctx->n_marks - 1]));
    for (size_t i = 0; i < n && i < capacity && intervals_ms; ++i)
        HIP_TRY(hipEventElapsedTime(&intervals_ms[i], ctx->marks[i], ctx->marks[i + 1]));
    ctx->n_marks = 0;
    return DSI_OK;
}

/* ------------------------------------------------------------------- Grid3D */
static int grid_make(dsi_context_t* ctx, int nx, int ny, int nz, void* wrap, dsi_grid_t** out)
{
    REQUIRE(ctx && out, DSI_ERR_INVALID, "null argument");
    *out = nullptr;
    REQUIRE(nx > 0 && ny > 0 && nz > 0, DSI_ERR_INVALID, "grid dimensions must be positive (%d,%d,%d)", nx,
            ny, nz);
    if (int rc = set_device(ctx)) return rc;
    dsi_grid* g = new (std::nothrow) dsi_grid();
    REQUIRE(g, DSI_ERR_INVALID, "out of host memory");
    g->ctx = ctx;
    g->nx = nx;
    g->ny = ny;
    g->nz = nz;
    g->n = (size_t)nx * ny * nz;
    if (wrap) {
        g->data = static_cast<float*>(wrap);
        g->owned = false;
    } else {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&g->data), g->n * sizeof(float));
        if (e == hipSuccess) e = hipMemsetAsync(g->data, 0, g->n * sizeof(float), ctx->stream);
        if (e != hipSuccess) {
            if (g->data) (void)hipFree(g->data);
            delete g;
            return fail(DSI_ERR_HIP, "grid allocation of %zu bytes failed: %s", g->n * sizeof(float),
                        hipGetErrorString(e));
        }
        g->owned = true;
    }
    ++ctx->children;
    *out = g;
    return DSI_OK;
}

int dsi_grid_create(dsi_context_t* ctx, int nx, int ny, int nz, dsi_grid_t** out)
{
    return grid_make(ctx, nx, ny, nz, nullptr, out);
}

int dsi_grid_wrap(dsi_context_t* ctx, int nx, int ny, int nz, void* data_dev, dsi_grid_t** out)
{
    REQUIRE(data_dev, DSI_ERR_INVALID, "data_dev is null");
    // the voxel-wise kernels sweep volumes as float4
    REQUIRE((reinterpret_cast<uintptr_t>(data_dev) & 15u) == 0, DSI_ERR_INVALID,
            "wrapped device memory must be 16-byte aligned");
    return grid_make(ctx, nx, ny, nz, data_dev, out);
}

int dsi_grid_destroy(dsi_grid_t* g)
{
    if (!g) return DSI_OK;
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    if (g->owned && g->data) (void)hipFree(g->data);
    --g->ctx->children;
    delete g;
    return DSI_OK;
}

int dsi_grid_dims(const dsi_grid_t* g, int* nx, int* ny, int* nz)
{
    REQUIRE(g, DSI_ERR_INVALID, "grid is null");
    if (nx) *nx = g->nx;
    if (ny) *ny = g->ny;
    if (nz) *nz = g->nz;
    return DSI_OK;
}

int dsi_grid_reset(dsi_grid_t* g)
{
    REQUIRE(g, DSI_ERR_INVALID, "grid is null");
    if (int rc = set_device(g->ctx)) return rc;
    HIP_TRY(hipMemsetAsync(g->data, 0, g->n * sizeof(float), g->ctx->stream));
    return DSI_OK;
}

void* dsi_grid_device_ptr(dsi_grid_t* g) { return g ? g->data : nullptr; }

int dsi_grid_upload(dsi_grid_t* g, const float* host)
{
    REQUIRE(g && host, DSI_ERR_INVALID, "null argument");
    if (int rc = set_device(g->ctx)) return rc;
    HIP_TRY(hipMemcpyAsync(g->data, host, g->n * sizeof(float), hipMemcpyHostToDevice, g->ctx->stream));
    HIP_TRY(hipStreamSynchronize(g->ctx->stream));
    return DSI_OK;
}

int dsi_grid_download(dsi_grid_t* g, float* host)
{
    REQUIRE(g && host, DSI_ERR_INVALID, "null argument");
    if (int rc = set_device(g->ctx)) return rc;
    HIP_TRY(hipMemcpyAsync(host, g->data, g->n * sizeof(float), hipMemcpyDeviceToHost, g->ctx->stream));
    HIP_TRY(hipStreamSynchronize(g->ctx->stream));
    return DSI_OK;
}

static int check_pair(const dsi_grid_t* dst, const dsi_grid_t* src)
{
    REQUIRE(dst && src, DSI_ERR_INVALID, "null grid");
    // grids of two contexts (streams) of one device may be combined: the op runs on dst's stream and
    // the caller orders the streams (dsi_context_wait_for)
    REQUIRE(dst->ctx->device == src->ctx->device, DSI_ERR_CONTEXT, "grids live on different devices");
    REQUIRE(same_shape(dst, src), DSI_ERR_SHAPE, "grid shapes differ: (%d,%d,%d) vs (%d,%d,%d)", dst->nx,
            dst->ny, dst->nz, src->nx, src->ny, src->nz);
    return set_device(dst->ctx);
}

int dsi_grid_fuse2(dsi_grid_t* dst, const dsi_grid_t* src, int op)
{
    if (int rc = check_pair(dst, src)) return rc;
    REQUIRE(op >= 1 && op <= 6, DSI_ERR_BAD_OP, "improper fusion method %d (expected 1..6)", op);
    HIP_TRY(dsi::launch_fuse2(dst->ctx->stream, dst->data, src->data, dst->n, op));
    return DSI_OK;
}

int dsi_grid_fuse2_into(dsi_grid_t* dst, const dsi_grid_t* a, const dsi_grid_t* b, int op)
{
    if (int rc = check_pair(dst, a)) return rc;
    if (int rc = check_pair(dst, b)) return rc;
    REQUIRE(op >= 1 && op <= 6, DSI_ERR_BAD_OP, "improper fusion method %d (expected 1..6)", op);
    REQUIRE(dst != a && dst != b, DSI_ERR_INVALID, "dst must differ from the operands (use dsi_grid_fuse2 in place)");
    HIP_TRY(dsi::launch_fuse2_into(dst->ctx->stream, dst->data, a->data, b->data, dst->n, op));
    return DSI_OK;
}

int dsi_grid_fuse_hm_n(dsi_grid_t* dst, const dsi_grid_t* src, int n)
{
    if (int rc = check_pair(dst, src)) return rc;
    REQUIRE(n >= 2, DSI_ERR_INVALID, "n-ary harmonic mean needs n >= 2 (got %d)", n);
    HIP_TRY(dsi::launch_fuse_hm_n(dst->ctx->stream, dst->data, src->data, dst->n, n));
    return DSI_OK;
}

static bool valid_acc_mode(int mode) { return mode >= DSI_ACC_SUM && mode <= DSI_ACC_MAX; }
// the one-pass n-ary fusions also take DSI_ACC_GM_TREE (n = 2, 4, 8): not an accumulation, so not in the streaming API
static bool valid_fuse_n_mode(int mode, int n)
{
    return valid_acc_mode(mode) || (mode == DSI_ACC_GM_TREE && (n == 2 || n == 4 || n == 8));
}

int dsi_acc_reduce_op(int mode)
{
    if (!valid_acc_mode(mode)) return -1;
    if (mode == DSI_ACC_MIN) return DSI_REDUCE_MIN;
    if (mode == DSI_ACC_MAX) return DSI_REDUCE_MAX;
    return DSI_REDUCE_SUM;
}

int dsi_grid_accumulate_begin(dsi_grid_t* dst, int mode)
{
    REQUIRE(dst, DSI_ERR_INVALID, "grid is null");
    REQUIRE(valid_acc_mode(mode), DSI_ERR_BAD_OP, "bad accumulate mode %d", mode);
    if (int rc = set_device(dst->ctx)) return rc;
    if (mode == DSI_ACC_MIN || mode == DSI_ACC_MAX) {
        const float v = mode == DSI_ACC_MIN ? INFINITY : -INFINITY;
        HIP_TRY(dsi::launch_fill(dst->ctx->stream, dst->data, dst->n, v));
    } else {
        HIP_TRY(hipMemsetAsync(dst->data, 0, dst->n * sizeof(float), dst->ctx->stream));
    }
    return DSI_OK;
}

int dsi_grid_accumulate(dsi_grid_t* dst, const dsi_grid_t* src, int mode)
{
    if (int rc = check_pair(dst, src)) return rc;
    REQUIRE(valid_acc_mode(mode), DSI_ERR_BAD_OP, "bad accumulate mode %d", mode);
    HIP_TRY(dsi::launch_accumulate(dst->ctx->stream, dst->data, src->data, dst->n, mode));
    return DSI_OK;
}

int dsi_grid_fuse_n(dsi_grid_t* dst, const dsi_grid_t* const* srcs, int n, int mode)
{
    REQUIRE(dst && srcs, DSI_ERR_INVALID, "null argument");
    REQUIRE(valid_fuse_n_mode(mode, n), DSI_ERR_BAD_OP, "bad fusion mode %d for %d grids (DSI_ACC_GM_TREE needs 2, 4 or 8)", mode, n);
    REQUIRE(n >= 1 && n <= 8, DSI_ERR_INVALID, "1 <= n <= 8 sources (got %d)", n);
    const float* ptrs[8];
    for (int i = 0; i < n; ++i) {
        if (int rc = check_pair(dst, srcs[i])) return rc;
        REQUIRE(srcs[i] != dst, DSI_ERR_INVALID, "dst must not be one of the sources");
        ptrs[i] = srcs[i]->data;
    }
    HIP_TRY(dsi::launch_fuse_n(dst->ctx->stream, dst->data, ptrs, n, dst->n, mode));
    return DSI_OK;
}

int dsi_grid_finalize(dsi_grid_t* dst, int mode, int n)
{
    REQUIRE(dst, DSI_ERR_INVALID, "grid is null");
    REQUIRE(valid_acc_mode(mode), DSI_ERR_BAD_OP, "bad finalize mode %d", mode);
    REQUIRE(n >= 1, DSI_ERR_INVALID, "number of maps must be >= 1 (got %d)", n);
    if (int rc = set_device(dst->ctx)) return rc;
    HIP_TRY(dsi::launch_finalize(dst->ctx->stream, dst->data, dst->n, mode, n));
    return DSI_OK;
}

int dsi_grid_collapse_max_z_dev(dsi_grid_t* g, float* conf_dev, uint8_t* idx_dev, const float* planes_dev,
                                float* depth_dev)
{
    REQUIRE(g && conf_dev && idx_dev, DSI_ERR_INVALID, "null argument");
    REQUIRE(g->nz <= 256, DSI_ERR_INVALID, "arg-max indices are u8: dimZ must be <= 256 (got %d)", g->nz);
    REQUIRE(!depth_dev || planes_dev, DSI_ERR_INVALID, "depth output needs the plane depths");
    if (int rc = set_device(g->ctx)) return rc;
    HIP_TRY(dsi::launch_collapse_max_z(g->ctx->stream, g->data, g->nx, g->ny, g->nz, conf_dev, idx_dev,
                                       planes_dev, depth_dev));
    return DSI_OK;
}

int dsi_grid_collapse_max_z(dsi_grid_t* g, float* conf_host, uint8_t* idx_host)
{
    REQUIRE(g && conf_host && idx_host, DSI_ERR_INVALID, "null argument");
    dsi_context* ctx = g->ctx;
    if (int rc = set_device(ctx)) return rc;
    const size_t npix = (size_t)g->nx * g->ny;
    const size_t need = npix * sizeof(float) + npix;
    if (ctx->collapse_scratch_bytes < need) {
        if (ctx->collapse_scratch) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipFree(ctx->collapse_scratch));
            ctx->collapse_scratch = nullptr;
            ctx->collapse_scratch_bytes = 0;
        }
        HIP_TRY(hipMalloc(&ctx->collapse_scratch, need));
        ctx->collapse_scratch_bytes = need;
    }
    float* conf = static_cast<float*>(ctx->collapse_scratch);
    uint8_t* idx = reinterpret_cast<uint8_t*>(conf + npix);
    if (int rc = dsi_grid_collapse_max_z_dev(g, conf, idx, nullptr, nullptr)) return rc;
    HIP_TRY(hipMemcpyAsync(conf_host, conf, npix * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(idx_host, idx, npix, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return DSI_OK;
}

int dsi_grid_mean_square(dsi_grid_t* g, double* out)
{
    REQUIRE(g && out, DSI_ERR_INVALID, "null argument");
    dsi_context* ctx = g->ctx;
    if (int rc = set_device(ctx)) return rc;
    HIP_TRY(hipMemsetAsync(ctx->ms_accum, 0, sizeof(double), ctx->stream));
    HIP_TRY(dsi::launch_mean_square(ctx->stream, g->data, g->n, ctx->ms_accum));
    double sum = 0;
    HIP_TRY(hipMemcpyAsync(&sum, ctx->ms_accum, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = sum / (double)g->n;  // cartesian3dgrid.cpp:173
    return DSI_OK;
}

/* --------------------------------------------------------------- MapperEMVS */
int dsi_mapper_create(dsi_context_t* ctx, const dsi_mapper_config_t* cfg, dsi_mapper_t** out)
{
    REQUIRE(ctx && cfg && out, DSI_ERR_INVALID, "null argument");
    *out = nullptr;
    REQUIRE(cfg->sensor_width > 0 && cfg->sensor_height > 0, DSI_ERR_INVALID, "sensor size must be positive");
    REQUIRE(cfg->sensor_width <= 65536 && cfg->sensor_height <= 65536, DSI_ERR_INVALID,
            "event coordinates are u16");
    // glog CHECKs of setupDSI (mapper_emvs_stereo.cpp:210-211) and PinholeCamera
    // (geometry_utils.hpp:36-41) become error codes
    REQUIRE(cfg->min_depth > 0.f, DSI_ERR_INVALID, "min_depth must be > 0");
    REQUIRE(cfg->max_depth > cfg->min_depth, DSI_ERR_INVALID, "max_depth must be > min_depth");
    REQUIRE(cfg->dim_z >= 1 && cfg->dim_z <= 256, DSI_ERR_INVALID, "dimZ must be in 1..256 (main.cpp:156)");
    REQUIRE(cfg->dim_x >= 0 && cfg->dim_y >= 0, DSI_ERR_INVALID, "dimX/dimY must be >= 0");
    REQUIRE(cfg->plane_begin >= 0 && cfg->plane_count >= 0 && cfg->plane_begin < cfg->dim_z &&
                cfg->plane_begin + cfg->plane_count <= cfg->dim_z,
            DSI_ERR_INVALID, "plane range must lie inside [0, dimZ)");
    REQUIRE(cfg->K[0] > 0.f && cfg->K[1] > 0.f && cfg->K[2] > 0.f && cfg->K[3] > 0.f, DSI_ERR_INVALID,
            "camera fx, fy, cx, cy must be > 0");
    if (int rc = set_device(ctx)) return rc;

    dsi_mapper* m = new (std::nothrow) dsi_mapper();
    REQUIRE(m, DSI_ERR_INVALID, "out of host memory");
    m->ctx = ctx;
    ++ctx->children;
    m->sensor_w = cfg->sensor_width;
    m->sensor_h = cfg->sensor_height;
    dsi::Geom& g = m->geom;
    g.nx = cfg->dim_x > 0 ? cfg->dim_x : cfg->sensor_width;   // :216
    g.ny = cfg->dim_y > 0 ? cfg->dim_y : cfg->sensor_height;  // :217
    g.nz = cfg->dim_z;
    g.kfx = cfg->K[0];
    g.kfy = cfg->K[1];
    g.kcx = cfg->K[2];
    g.kcy = cfg->K[3];
    const float f = virtual_focal(cfg->K[0], cfg->fov_deg, g.nx);
    g.vfx = f;  // :236-239: PinholeCamera(dimX, dimY, f, f, cam.cx(), cam.cy())
    g.vfy = f;
    g.vcx = cfg->K[2];
    g.vcy = cfg->K[3];
    make_planes(cfg->min_depth, cfg->max_depth, cfg->dim_z, cfg->inverse_depth != 0, &m->planes);
    g.z0 = m->planes[0];  // :111, :163 -- of the full depth vector, also for a plane shard
    m->planes_full = m->planes;
    m->plane_begin = cfg->plane_begin;
#ifdef DSI_TIMING_EXPERIMENTS
    if (const char* e = std::getenv("DSI_PREP_OVERLAP")) m->prep_overlap = std::atoi(e) != 0;
#endif
    g.nz = cfg->plane_count > 0 ? cfg->plane_count : cfg->dim_z - cfg->plane_begin;
    m->planes = std::vector<float>(m->planes.begin() + m->plane_begin, m->planes.begin() + m->plane_begin + g.nz);

    int rc = DSI_OK;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->planes_dev), g.nz * sizeof(float));
    if (e == hipSuccess)
        e = hipMemcpy(m->planes_dev, m->planes.data(), g.nz * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && cfg->lut) {
        const size_t bytes = (size_t)cfg->sensor_width * cfg->sensor_height * sizeof(float2);
        e = hipMalloc(reinterpret_cast<void**>(&m->lut_dev), bytes);
        if (e == hipSuccess) e = hipMemcpy(m->lut_dev, cfg->lut, bytes, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) rc = fail(DSI_ERR_HIP, "mapper setup failed: %s", hipGetErrorString(e));
    if (rc == DSI_OK) rc = dsi_grid_create(ctx, g.nx, g.ny, g.nz, &m->grid);  // :240
    if (rc != DSI_OK) {
        std::string keep = g_last_error;
        dsi_mapper_destroy(m);
        g_last_error = keep;
        return rc;
    }
    *out = m;
    return DSI_OK;
}

int dsi_mapper_destroy(dsi_mapper_t* m)
{
    if (!m) return DSI_OK;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->prep_stream) {
        (void)hipStreamSynchronize(m->prep_stream);
        (void)hipStreamDestroy(m->prep_stream);
        (void)hipEventDestroy(m->ev_prep);
        (void)hipEventDestroy(m->ev_vote_done);
    }
    if (m->ev_depth_ready) {
        if (m->ctx->copy_stream) (void)hipStreamSynchronize(m->ctx->copy_stream);
        (void)hipEventDestroy(m->ev_depth_ready);
        (void)hipEventDestroy(m->ev_depth_read);
    }
    if (m->grid) dsi_grid_destroy(m->grid);
    if (m->planes_dev) (void)hipFree(m->planes_dev);
    if (m->planes_full_dev) (void)hipFree(m->planes_full_dev);
    m->argmax_keys.release();
    m->tie.release();
    if (m->lut_dev) (void)hipFree(m->lut_dev);
    m->centers.release();
    m->H.release();
    m->partials.release();
    m->fused_trace.release();
    m->fused_keys.release();
    m->pair_work.release();
    m->fused_splits.release();
    m->fused_prefix.release();
    m->Rt_tmp.release();
    m->conf.release();
    m->depth.release();
    m->xy.release();
    m->sxy.release();
    m->seam.release();
    m->nvalid.release();
    m->gcuts.release();
    m->spk.release();
    m->rowstart.release();
    m->cuts.release();
    m->coef.release();
    m->idx.release();
    m->conf8.release();
    m->mask.release();
    m->idx_filtered.release();
    m->minmax.release();
    for (auto& pr : m->timing_pairs) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (hipEvent_t e : m->timing_pool) (void)hipEventDestroy(e);
    --m->ctx->children;
    delete m;
    return DSI_OK;
}

dsi_grid_t* dsi_mapper_grid(dsi_mapper_t* m) { return m ? m->grid : nullptr; }

int dsi_mapper_plane_begin(const dsi_mapper_t* m) { return m ? m->plane_begin : 0; }

int dsi_mapper_full_depths(const dsi_mapper_t* m, float* raw_depths, int* dim_z)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    if (raw_depths) std::memcpy(raw_depths, m->planes_full.data(), m->planes_full.size() * sizeof(float));
    if (dim_z) *dim_z = (int)m->planes_full.size();
    return DSI_OK;
}

int dsi_mapper_geometry(const dsi_mapper_t* m, float* Kv, float* raw_depths, int* nx, int* ny, int* nz)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    if (Kv) {
        Kv[0] = m->geom.vfx;
        Kv[1] = m->geom.vfy;
        Kv[2] = m->geom.vcx;
        Kv[3] = m->geom.vcy;
    }
    if (raw_depths) std::memcpy(raw_depths, m->planes.data(), m->planes.size() * sizeof(float));
    if (nx) *nx = m->geom.nx;
    if (ny) *ny = m->geom.ny;
    if (nz) *nz = m->geom.nz;
    return DSI_OK;
}

int dsi_mapper_set_vote_algo(dsi_mapper_t* m, int algo)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    REQUIRE(algo >= DSI_VOTE_AUTO && algo <= DSI_VOTE_LDS_BANDS, DSI_ERR_INVALID, "unknown vote algorithm %d",
            algo);
    m->algo = algo;
    return DSI_OK;
}

int dsi_mapper_set_band_params(dsi_mapper_t* m, int band_rows, int chunks, int block_threads)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    REQUIRE(band_rows >= 0 && chunks >= 0, DSI_ERR_INVALID, "band_rows and chunks must be >= 0");
    REQUIRE(block_threads == 0 || block_threads == 256 || block_threads == 512 || block_threads == 1024,
            DSI_ERR_INVALID, "block_threads must be 0, 256, 512 or 1024");
    m->want_band_rows = band_rows;
    m->want_chunks = chunks;
    m->want_block = block_threads;
    return DSI_OK;
}

int dsi_mapper_set_packed_lanes(dsi_mapper_t* m, int mode)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    REQUIRE(mode >= -1 && mode <= 8, DSI_ERR_INVALID, "mode must be -1 (auto) or 0..8");
    m->want_packed = mode;
    return DSI_OK;
}

int dsi_mapper_paired_overflow(dsi_mapper_t* m, int* overflowed)
{
    REQUIRE(m && overflowed, DSI_ERR_INVALID, "null argument");
    *overflowed = 0;
    if (m->info.algo != DSI_VOTE_LDS_BANDS || m->info.packed != 8 || m->info.n_packets == 0) return DSI_OK;
    if (int rc = set_device(m->ctx)) return rc;
    uint32_t word = 0;
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    HIP_TRY(hipMemcpy(&word, m->nvalid.p + m->info.n_packets + (size_t)m->geom.nz + 8, sizeof word, hipMemcpyDeviceToHost));
    *overflowed = word != 0u ? 1 : 0;
    return DSI_OK;
}

int dsi_mapper_set_inline_cuts(dsi_mapper_t* m, long long min_packets)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    m->cuts_inline_min_packets = min_packets < 0 ? (size_t)8192 : (size_t)min_packets;
    return DSI_OK;
}

int dsi_mapper_fill_voxel_grid(dsi_mapper_t* m, const float* xy_z0, const float* centers, size_t n_packets)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    REQUIRE(n_packets == 0 || (xy_z0 && centers), DSI_ERR_INVALID, "null input");
    REQUIRE(n_packets < ((size_t)1 << 21), DSI_ERR_INVALID, "too many packets in one call");
    dsi_context* ctx = m->ctx;
    if (int rc = set_device(ctx)) return rc;
    if (n_packets == 0) return DSI_OK;
    HIP_TRY(m->xy.reserve(n_packets * dsi::kPacket));
    HIP_TRY(m->centers.reserve(n_packets * 3));
    if (int rc = upload_async(ctx, m->xy.p, xy_z0, n_packets * dsi::kPacket * sizeof(float2))) return rc;
    if (int rc = upload_async(ctx, m->centers.p, centers, n_packets * 3 * sizeof(float))) return rc;
    // (host inputs were uploaded on the context's stream: no second stream here)
    int rc = vote_device(m, m->xy.p, m->centers.p, n_packets, /*accumulate=*/true, ctx->stream);
    // the host buffers are pageable: do not return before the copies have read them
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return rc;
}

static int batch_create(dsi_context_t* ctx, const uint16_t* x, const uint16_t* y, size_t n_events,
                        const uint32_t* packet_first, const float* Rt, size_t n_packets, dsi_batch_t** out,
                        bool wait)
{
    REQUIRE(ctx && out, DSI_ERR_INVALID, "null argument");
    *out = nullptr;
    REQUIRE(n_events == 0 || (x && y), DSI_ERR_INVALID, "null event arrays");
    REQUIRE(n_packets == 0 || Rt, DSI_ERR_INVALID, "null pose array");
    REQUIRE(n_events < ((size_t)1 << 32), DSI_ERR_INVALID, "at most 2^32-1 events per batch");
    REQUIRE(n_packets < ((size_t)1 << 21), DSI_ERR_INVALID, "too many packets in one batch");
    bool regular = true;  // packet k starts at event k * 1024 (no pose look-up failed, mapper_emvs_stereo.cpp:95-99)
    for (size_t k = 0; k < n_packets; ++k) {
        const size_t first = packet_first ? packet_first[k] : k * dsi::kPacket;
        REQUIRE(first + dsi::kPacket <= n_events, DSI_ERR_INVALID,
                "packet %zu [%zu, %zu) exceeds the %zu events given", k, first, first + dsi::kPacket, n_events);
        regular = regular && first == k * dsi::kPacket;
    }
    // ... then the table is implied and does not travel: the runtime moves a copy of a few KB with a shader, which waits for
    // a free CU behind a persistent voting kernel and held up a window's other uploads by 0.2 ms (profiles/r05_cpp_window_stream_timeline.txt)
    if (regular) packet_first = nullptr;
    if (int rc = set_device(ctx)) return rc;
    dsi_batch* b = new (std::nothrow) dsi_batch();
    REQUIRE(b, DSI_ERR_INVALID, "out of host memory");
    b->ctx = ctx;
    ++ctx->children;
    b->n_events = n_events;
    b->n_packets = n_packets;
    // one block: Rt | first | x | y (256-byte aligned parts), uploaded on the context's stream
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t rt_bytes = up(std::max<size_t>(n_packets, 1) * 12 * sizeof(float));
    const size_t first_bytes = packet_first ? up(std::max<size_t>(n_packets, 1) * sizeof(uint32_t)) : 0;
    const size_t ev_bytes = up(std::max<size_t>(n_events, 1) * sizeof(uint16_t));
    if (!pool_take(ctx, rt_bytes + first_bytes + 2 * ev_bytes, &b->block)) {
        --ctx->children;
        delete b;
        return fail(DSI_ERR_HIP, "batch allocation of %zu bytes failed", rt_bytes + first_bytes + 2 * ev_bytes);
    }
    char* base = static_cast<char*>(b->block.p);
    b->Rt = reinterpret_cast<float*>(base);
    b->first = packet_first ? reinterpret_cast<uint32_t*>(base + rt_bytes) : nullptr;
    b->x = reinterpret_cast<uint16_t*>(base + rt_bytes + first_bytes);
    b->y = reinterpret_cast<uint16_t*>(base + rt_bytes + first_bytes + ev_bytes);
    hipStream_t cs = ctx->upload_stream;
    hipError_t e = hipEventCreateWithFlags(&b->ready, hipEventDisableTiming);
    // the block's previous reader (a kernel on the compute stream) must be done before we overwrite it
    if (e == hipSuccess) e = hipStreamWaitEvent(cs, b->block.freed, 0);
    if (e == hipSuccess && n_events) e = hipMemcpyAsync(b->x, x, n_events * sizeof(uint16_t), hipMemcpyHostToDevice, cs);
    if (e == hipSuccess && n_events)
        e = hipMemcpyAsync(b->y, y, n_events * sizeof(uint16_t), hipMemcpyHostToDevice, cs);
    if (e == hipSuccess && n_packets)
        e = hipMemcpyAsync(b->Rt, Rt, n_packets * 12 * sizeof(float), hipMemcpyHostToDevice, cs);
    if (e == hipSuccess && n_packets && packet_first)
        e = hipMemcpyAsync(b->first, packet_first, n_packets * sizeof(uint32_t), hipMemcpyHostToDevice, cs);
    if (e == hipSuccess) e = hipEventRecord(b->ready, cs);
    // the host arrays are the caller's: they must be consumed before this call returns (this waits
    // for the copies only, not for whatever the compute stream is doing) -- unless the caller
    // vouches for them (dsi_batch_create_async: page-locked memory, left alone until uploaded)
    if (e == hipSuccess && wait) e = hipStreamSynchronize(cs);
    if (e != hipSuccess) {
        dsi_batch_destroy(b);
        return fail(DSI_ERR_HIP, "batch upload failed: %s", hipGetErrorString(e));
    }
    *out = b;
    return DSI_OK;
}

int dsi_batch_create(dsi_context_t* ctx, const uint16_t* x, const uint16_t* y, size_t n_events,
                     const uint32_t* packet_first, const float* Rt, size_t n_packets, dsi_batch_t** out)
{
    return batch_create(ctx, x, y, n_events, packet_first, Rt, n_packets, out, /*wait=*/true);
}

int dsi_batch_create_async(dsi_context_t* ctx, const uint16_t* x, const uint16_t* y, size_t n_events,
                           const uint32_t* packet_first, const float* Rt, size_t n_packets, dsi_batch_t** out)
{
    return batch_create(ctx, x, y, n_events, packet_first, Rt, n_packets, out, /*wait=*/false);
}

int dsi_batch_uploaded(const dsi_batch_t* b)
{
    if (!b || !b->ready) return 1;
    // (anything but "not ready" ends a caller's polling loop: a failed copy surfaces at the next synchronisation)
    return hipEventQuery(b->ready) == hipErrorNotReady ? 0 : 1;
}

int dsi_host_alloc(size_t bytes, void** out)
{
    REQUIRE(out, DSI_ERR_INVALID, "out is null");
    *out = nullptr;
    REQUIRE(bytes > 0, DSI_ERR_INVALID, "bytes must be > 0");
    // coherent + mapped whatever HIP_HOST_COHERENT says: kernels store depth maps into these blocks
    // (dsi_mapper_fetch_depth_map_in_order) and the host reads them after an event
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
    return DSI_OK;
}

int dsi_host_free(void* p)
{
    if (!p) return DSI_OK;
    HIP_TRY(hipHostFree(p));
    return DSI_OK;
}

int dsi_batch_destroy(dsi_batch_t* b)
{
    if (!b) return DSI_OK;
    (void)hipSetDevice(b->ctx->device);
    // an upload that nobody waited for yet (async batch destroyed unused): order the block's reuse after it
    if (b->ready) (void)hipStreamWaitEvent(b->ctx->stream, b->ready, 0);
    pool_give(b->ctx, b->block);  // no wait: the block's "freed" event orders its reuse
    if (b->ready) (void)hipEventDestroy(b->ready);
    --b->ctx->children;
    delete b;
    return DSI_OK;
}

size_t dsi_batch_num_packets(const dsi_batch_t* b) { return b ? b->n_packets : 0; }

int dsi_mapper_evaluate_batch(dsi_mapper_t* m, const dsi_batch_t* batch)
{
    REQUIRE(m && batch, DSI_ERR_INVALID, "null argument");
    REQUIRE(m->ctx == batch->ctx, DSI_ERR_CONTEXT, "mapper and batch belong to different contexts");
    dsi_context* ctx = m->ctx;
    if (int rc = set_device(ctx)) return rc;
    const size_t np = batch->n_packets;
    hipStream_t ps = ctx->stream;
    if (np) {
        HIP_TRY(m->centers.reserve(np * 3));
        if (int rc = prep_begin(m, &ps)) return rc;
        if (batch->ready) HIP_TRY(hipStreamWaitEvent(ps, batch->ready, 0));  // uploaded on the copy stream
    }
    // stage A (:108-142) happens inside vote_device (fused into the packet sort on the default path);
    // resetGrid (:145) is folded into the vote: accumulate = false
    return vote_device(m, m->xy.p, m->centers.p, np, /*accumulate=*/false, ps, batch);
}

int dsi_pose_at(const double* traj_times, const double* traj_poses, size_t n_poses, double t, double* out)
{
    REQUIRE(traj_times && traj_poses && out, DSI_ERR_INVALID, "null argument");
    dsi::host::Pose T;
    if (!dsi::host::pose_at(traj_times, traj_poses, n_poses, t, &T))
        return fail(DSI_ERR_INVALID, "cannot extrapolate the trajectory to t=%.9f", t);
    T.to7(out);
    return DSI_OK;
}

int dsi_packetize(const double* ts, size_t n_events, const double* traj_times, const double* traj_poses,
                  size_t n_poses, const double* T_rv_w, uint32_t* packet_first, float* Rt, size_t* n_packets)
{
    REQUIRE(n_packets && T_rv_w && traj_times && traj_poses, DSI_ERR_INVALID, "null argument");
    REQUIRE(n_events == 0 || ts, DSI_ERR_INVALID, "null timestamps");
    REQUIRE(n_events < ((size_t)1 << 32), DSI_ERR_INVALID, "at most 2^32-1 events");
    *n_packets = 0;
    std::vector<uint32_t> first;
    std::vector<float> rt;
    if (!dsi::host::packetize(ts, n_events, traj_times, traj_poses, n_poses, dsi::host::Pose::from7(T_rv_w),
                              &first, &rt))
        return fail(DSI_ERR_TOO_FEW_EVENTS, "number of events (%zu) < packet size (%d)", n_events,
                    DSI_PACKET_SIZE);
    *n_packets = first.size();
    if (packet_first && !first.empty()) std::memcpy(packet_first, first.data(), first.size() * sizeof(uint32_t));
    if (Rt && !rt.empty()) std::memcpy(Rt, rt.data(), rt.size() * sizeof(float));
    return DSI_OK;
}

int dsi_packetize_strided(const void* ts_first, size_t stride_bytes, size_t n_events, const double* traj_times,
                          const double* traj_poses, size_t n_poses, const double* T_rv_w, uint32_t* packet_first, float* Rt,
                          size_t* n_packets)
{
    REQUIRE(n_packets && T_rv_w && traj_times && traj_poses, DSI_ERR_INVALID, "null argument");
    REQUIRE(n_events == 0 || ts_first, DSI_ERR_INVALID, "null timestamps");
    REQUIRE(stride_bytes >= sizeof(double), DSI_ERR_INVALID, "stride of %zu bytes is smaller than a timestamp", stride_bytes);
    REQUIRE(n_events < ((size_t)1 << 32), DSI_ERR_INVALID, "at most 2^32-1 events");
    *n_packets = 0;
    std::vector<uint32_t> first;
    std::vector<float> rt;
    const char* base = static_cast<const char*>(ts_first);
    auto ts_of = [base, stride_bytes](size_t i) {
        double t;
        std::memcpy(&t, base + i * stride_bytes, sizeof t);  // (no alignment assumed)
        return t;
    };
    if (!dsi::host::packetize_with(ts_of, n_events, traj_times, traj_poses, n_poses, dsi::host::Pose::from7(T_rv_w), &first, &rt))
        return fail(DSI_ERR_TOO_FEW_EVENTS, "number of events (%zu) < packet size (%d)", n_events, DSI_PACKET_SIZE);
    *n_packets = first.size();
    if (packet_first && !first.empty()) std::memcpy(packet_first, first.data(), first.size() * sizeof(uint32_t));
    if (Rt && !rt.empty()) std::memcpy(Rt, rt.data(), rt.size() * sizeof(float));
    return DSI_OK;
}

int dsi_mapper_evaluate(dsi_mapper_t* m, const uint16_t* x, const uint16_t* y, const double* ts,
                        size_t n_events, const double* traj_times, const double* traj_poses, size_t n_poses,
                        const double* T_rv_w, size_t* n_voted)
{
    REQUIRE(m && traj_times && traj_poses && T_rv_w, DSI_ERR_INVALID, "null argument");
    REQUIRE(n_events == 0 || (x && y && ts), DSI_ERR_INVALID, "null event arrays");
    REQUIRE(n_events < ((size_t)1 << 32), DSI_ERR_INVALID, "at most 2^32-1 events");
    if (n_voted) *n_voted = 0;
    std::vector<uint32_t> first;
    std::vector<float> rt;
    if (!dsi::host::packetize(ts, n_events, traj_times, traj_poses, n_poses, dsi::host::Pose::from7(T_rv_w),
                              &first, &rt))
        return fail(DSI_ERR_TOO_FEW_EVENTS, "number of events (%zu) < packet size (%d)", n_events,
                    DSI_PACKET_SIZE);
    dsi_batch_t* b = nullptr;
    int rc = dsi_batch_create(m->ctx, x, y, n_events, first.data(), rt.data(), first.size(), &b);
    if (rc != DSI_OK) return rc;
    rc = dsi_mapper_evaluate_batch(m, b);
    std::string keep = g_last_error;
    dsi_batch_destroy(b);  // synchronises the stream
    g_last_error = keep;
    if (rc == DSI_OK && n_voted) *n_voted = first.size() * (size_t)DSI_PACKET_SIZE;
    return rc;
}

// A kernel on `reader`'s stream has just been queued that READS a grid of context `owner`: whatever `owner`
// queues from now on (the next vote or fusion overwriting that grid) must start after it (the
// one-way wait_for before the arg-max left a write-after-read hazard for grids of another context).  Device-
// side ordering only; nothing to do within one context (one in-order stream).
static int release_to(dsi_context* owner, dsi_context* reader)
{
    if (owner == reader) return DSI_OK;
    return dsi_context_wait_for(owner, reader);
}

int dsi_mapper_depth_map_of(dsi_mapper_t* m, dsi_grid_t* g)
{
    REQUIRE(m && g, DSI_ERR_INVALID, "null argument");
    REQUIRE(m->ctx->device == g->ctx->device, DSI_ERR_CONTEXT, "mapper and grid live on different devices");
    REQUIRE(same_shape(m->grid, g), DSI_ERR_SHAPE, "grid shape differs from the mapper's DSI");
    if (int rc = set_device(m->ctx)) return rc;
    const size_t npix = (size_t)g->nx * g->ny;
    HIP_TRY(m->conf.reserve(npix));
    HIP_TRY(m->depth.reserve(npix));
    HIP_TRY(m->idx.reserve(npix));
    // The arg-max writes the MAPPER's buffers and the filters / copies that follow run on the mapper's
    // stream, so the kernel goes to the mapper's stream; a grid that lives in another context (stream)
    // of the same device is first waited for: everything queued on its stream so far.
    REQUIRE(g->nz <= 256, DSI_ERR_INVALID, "arg-max indices are u8: dimZ must be <= 256 (got %d)", g->nz);
    if (int rc = dsi_context_wait_for(m->ctx, g->ctx)) return rc;
    if (int rc = depth_buffers_acquire(m)) return rc;
    HIP_TRY(dsi::launch_collapse_max_z(m->ctx->stream, g->data, g->nx, g->ny, g->nz, m->conf.p, m->idx.p,
                                       m->planes_dev, m->depth.p));
    if (int rc = release_to(g->ctx, m->ctx)) return rc;
    return depth_buffers_ready(m);
}

int dsi_mapper_depth_map_of_fusion(dsi_mapper_t* m, const dsi_grid_t* a, const dsi_grid_t* b, int op)
{
    REQUIRE(m && a && b, DSI_ERR_INVALID, "null argument");
    REQUIRE(m->ctx->device == a->ctx->device && m->ctx->device == b->ctx->device, DSI_ERR_CONTEXT,
            "mapper and grids live on different devices");
    REQUIRE(same_shape(m->grid, a) && same_shape(m->grid, b), DSI_ERR_SHAPE, "grid shape differs from the mapper's DSI");
    REQUIRE(op >= 1 && op <= 6, DSI_ERR_BAD_OP, "improper fusion method %d (expected 1..6)", op);
    REQUIRE(a->nz <= 256, DSI_ERR_INVALID, "arg-max indices are u8: dimZ must be <= 256 (got %d)", a->nz);
    if (int rc = set_device(m->ctx)) return rc;
    const size_t npix = (size_t)a->nx * a->ny;
    HIP_TRY(m->conf.reserve(npix));
    HIP_TRY(m->depth.reserve(npix));
    HIP_TRY(m->idx.reserve(npix));
    if (int rc = dsi_context_wait_for(m->ctx, a->ctx)) return rc;
    if (int rc = dsi_context_wait_for(m->ctx, b->ctx)) return rc;
    if (int rc = depth_buffers_acquire(m)) return rc;
    HIP_TRY(dsi::launch_collapse_max_z_fused(m->ctx->stream, a->data, b->data, a->nx, a->ny, a->nz, op, m->conf.p,
                                             m->idx.p, m->planes_dev, m->depth.p));
    if (int rc = release_to(a->ctx, m->ctx)) return rc;
    if (int rc = release_to(b->ctx, m->ctx)) return rc;
    return depth_buffers_ready(m);
}

int dsi_mapper_depth_map_of_fusion_n(dsi_mapper_t* m, const dsi_grid_t* const* srcs, int n, int mode)
{
    REQUIRE(m && srcs, DSI_ERR_INVALID, "null argument");
    REQUIRE(valid_fuse_n_mode(mode, n), DSI_ERR_BAD_OP, "bad fusion mode %d for %d grids (DSI_ACC_GM_TREE needs 2, 4 or 8)", mode, n);
    REQUIRE(n >= 1 && n <= 8, DSI_ERR_INVALID, "1 <= n <= 8 sources (got %d)", n);
    const float* ptrs[8];
    for (int i = 0; i < n; ++i) {
        REQUIRE(srcs[i], DSI_ERR_INVALID, "source %d is null", i);
        REQUIRE(m->ctx->device == srcs[i]->ctx->device, DSI_ERR_CONTEXT, "mapper and grids live on different devices");
        REQUIRE(same_shape(m->grid, srcs[i]), DSI_ERR_SHAPE, "grid shape differs from the mapper's DSI");
        ptrs[i] = srcs[i]->data;
    }
    REQUIRE(m->grid->nz <= 256, DSI_ERR_INVALID, "arg-max indices are u8: dimZ must be <= 256 (got %d)", m->grid->nz);
    if (int rc = set_device(m->ctx)) return rc;
    const size_t npix = (size_t)m->grid->nx * m->grid->ny;
    HIP_TRY(m->conf.reserve(npix));
    HIP_TRY(m->depth.reserve(npix));
    HIP_TRY(m->idx.reserve(npix));
    for (int i = 0; i < n; ++i)
        if (int rc = dsi_context_wait_for(m->ctx, srcs[i]->ctx)) return rc;
    if (int rc = depth_buffers_acquire(m)) return rc;
    HIP_TRY(dsi::launch_collapse_max_z_fused_n(m->ctx->stream, ptrs, n, mode, m->grid->nx, m->grid->ny, m->grid->nz,
                                               m->conf.p, m->idx.p, m->planes_dev, m->depth.p));
    for (int i = 0; i < n; ++i)
        if (int rc = release_to(srcs[i]->ctx, m->ctx)) return rc;
    return depth_buffers_ready(m);
}

static int depth_map_of_events_impl(dsi_mapper_t* out, dsi_mapper_t* const* mappers, const dsi_batch_t* const* batches, int n,
                                    int op, bool gm_tree4);

int dsi_mapper_depth_map_of_events(dsi_mapper_t* out, dsi_mapper_t* const* mappers, const dsi_batch_t* const* batches, int n,
                                   int op)
{
    REQUIRE(out && mappers && batches, DSI_ERR_INVALID, "null argument");
    REQUIRE(n >= 1 && n <= 3, DSI_ERR_INVALID, "1, 2 or 3 cameras (got %d)", n);
    REQUIRE(n == 1 || (op >= 1 && op <= 6), DSI_ERR_BAD_OP, "improper fusion method %d (expected 1..6)", op);
    return depth_map_of_events_impl(out, mappers, batches, n, op, false);
}

int dsi_mapper_depth_map_of_events_n(dsi_mapper_t* out, dsi_mapper_t* const* mappers, const dsi_batch_t* const* batches, int n,
                                     int mode)
{
    REQUIRE(out && mappers && batches, DSI_ERR_INVALID, "null argument");
    REQUIRE(mode == DSI_ACC_GM_TREE, DSI_ERR_BAD_OP, "the DSI-less n-camera path fuses by the geometric-mean tree (DSI_ACC_GM_TREE); got mode %d", mode);
    REQUIRE(n == 2 || n == 4, DSI_ERR_INVALID, "the geometric-mean tree takes 2 or 4 cameras here (got %d)", n);
    // (two cameras: the tree IS the reference's 2-ary op, cartesian3dgrid.h:150-156)
    return depth_map_of_events_impl(out, mappers, batches, n, DSI_FUSE_GM, n == 4);
}

static int depth_map_of_events_impl(dsi_mapper_t* out, dsi_mapper_t* const* mappers, const dsi_batch_t* const* batches, int n,
                                    int op, bool gm_tree4)
{
    dsi_context* ctx = out->ctx;
    size_t np_max = 0;
    for (int i = 0; i < n; ++i) {
        REQUIRE(mappers[i] && batches[i], DSI_ERR_INVALID, "camera %d: null mapper or batch", i);
        REQUIRE(mappers[i]->ctx == ctx && batches[i]->ctx == ctx, DSI_ERR_CONTEXT,
                "mappers, batches and the output mapper must share one context");
        REQUIRE(same_shape(out->grid, mappers[i]->grid), DSI_ERR_SHAPE, "camera %d: DSI shape differs from the output mapper's", i);
        for (int k = 0; k < i; ++k)
            REQUIRE(mappers[i] != mappers[k], DSI_ERR_INVALID, "the cameras need distinct mappers (their tables are per mapper)");
    }
    // process1.cpp:169-191: the third camera enters only through min (1), harmonicMeanTwoGrids(g, 3) (2) and max (6);
    // "case 3: break; case 4: break; case 5: break;" -- its DSI is built and then ignored, so it is not built here
    if (n == 3 && !gm_tree4 && (op == 3 || op == 4 || op == 5)) n = 2;
    for (int i = 0; i < n; ++i) np_max = std::max(np_max, batches[i]->n_packets);
    REQUIRE(out->geom.nz <= 256, DSI_ERR_INVALID, "arg-max indices are u8: dimZ must be <= 256 (got %d)", out->geom.nz);
    if (int rc = set_device(ctx)) return rc;
    dsi::BandPlan bp{};
    // every knob of this path (lane mapping, band height, the kernel timer, the experiments flavour's pass size, partition
    // cost and tracing) is read from ONE object, the output mapper; the vote info is recorded on it and on the cameras
    REQUIRE(plan_fused(out, np_max, &bp, n), DSI_ERR_INVALID, "grid rows of %d floats do not fit the fused kernel", out->geom.nx);
    if (bp.interleave < 0) {
        // Which pairs a workgroup takes: a contiguous piece of its XCD's stretch (0), or -- all 32 workgroups of an XCD on
        // consecutive planes of ONE band -- every 32nd pair (1) / the next pair drawn from the XCD's counter (2).  With
        // contiguous pieces an XCD works on bands / 8 + 1 bands at once; when their records (all cameras') exceed its 4 MB of
        // L2 every phase re-reads its band from the Infinity Cache -- four cameras x 2 M events at 1024 x 1024 x 256: 13 MB,
        // 4.6 TB/s; in turn: 6.10 -> 5.76 ms per step (kernel 5.89 -> 5.55), drawn: 5.50 (a workgroup's 74 pairs differ in
        // cost; the draw is one atomic per pair).  A 50 ms stereo window (2.5 MB) measures the same all three ways -- its
        // kernel lasts ceil(2800 pairs / 256 workgroups) = 11 pairs of equal cost however they are dealt -- and keeps the
        // contiguous pieces (one band change per workgroup).
        double records_bytes = 0.0;
        for (int i = 0; i < n; ++i) records_bytes += (double)batches[i]->n_packets * dsi::kPacket * sizeof(dsi::EvRec);
        const double band_bytes = records_bytes * (double)(bp.band_rows + 2) / (double)std::max(1, out->geom.ny);
        const int at_once = std::min(32, bp.bands / 8 + 1);
        bp.interleave = band_bytes * at_once > 3.0e6 ? 2 : 0;
    }
    const dsi::Geom& geom = mappers[0]->geom;
    hipStream_t st = ctx->stream;
    dsi::FusedCameras cams{};
    cams.n = n;
    dsi::PrepCameraArgs prep[dsi::kFusedMaxCameras] = {};
    const int n_pairs = bp.bands * geom.nz;
    const bool balance = n <= 2 && n_pairs <= dsi::fused_max_pairs() && out->fused_fixed_cost >= 0;
    for (int i = 0; i < n; ++i) {
        dsi_mapper* m = mappers[i];
        const dsi_batch* b = batches[i];
        const size_t np = b->n_packets;
        // the per-camera tables of the banded vote, built for bands with halo rows
        HIP_TRY(m->centers.reserve(std::max<size_t>(np, 1) * 3));
        HIP_TRY(m->sxy.reserve(np * dsi::kPacket + 1));
        HIP_TRY(m->nvalid.reserve(np + (size_t)geom.nz + 9));
        HIP_TRY(m->rowstart.reserve(np * (size_t)(geom.ny + 2 * bp.row_pad + 3) + 2));
        HIP_TRY(m->coef.reserve(np * geom.nz + 1));
        HIP_TRY(m->cuts.reserve(std::max<size_t>(np * geom.nz * bp.bands, 64)));
        HIP_TRY(m->pair_work.reserve((size_t)n_pairs));
        if (b->ready) HIP_TRY(hipStreamWaitEvent(st, b->ready, 0));  // uploaded on the copy stream
        if (np == 0) {
            // evaluateDSI returns false (mapper_emvs_stereo.cpp:71-75): this camera's DSI is all zero
            HIP_TRY(hipMemsetAsync(m->nvalid.p, 0, ((size_t)geom.nz + 8) * sizeof(uint32_t), st));
            HIP_TRY(hipMemsetAsync(m->pair_work.p, 0, (size_t)n_pairs * sizeof(uint32_t), st));
        }
        prep[i] = dsi::PrepCameraArgs{b->Rt, b->x, b->y, b->first, m->lut_dev, m->sensor_w, m->sensor_h, m->centers.p, (int)np,
                                      m->sxy.p, m->nvalid.p, m->rowstart.p, m->planes_dev, m->coef.p, m->cuts.p,
                                      balance ? m->pair_work.p : nullptr, m->geom};
        cams.cam[i] = dsi::FusedCamera{m->sxy.p, m->coef.p, m->cuts.p, m->nvalid.p + np, (int)np};
        m->info = dsi_vote_info_t{};
        m->info.algo = DSI_VOTE_FUSED_ARGMAX;
        m->info.n_packets = np;
        m->info.bands = bp.bands;
        m->info.band_rows = bp.band_rows;
        m->info.chunks = 1;
        m->info.block_threads = bp.block_threads;
        m->info.lds_bytes = bp.lds_bytes;
        m->info.packed = bp.packed;
        m->info.group_packets = 1;
    }
    out->info = mappers[0]->info;
    out->info.n_packets = np_max;
    // stage A + packet sort of all cameras in one launch, their coefficient / cut tables in another
    // (a window's ~490 packets per camera do not fill the chip: two launches each would only add latency)
    HIP_TRY(dsi::launch_prepare_cameras(st, prep, n, geom, bp));
    // workgroup -> stretch of (band, plane) pairs, balanced by the records the tables hold per pair
    const uint32_t* splits = nullptr;
    if (balance) {
        HIP_TRY(out->fused_splits.reserve((size_t)dsi::fused_grid_blocks() + 1));
        HIP_TRY(out->fused_prefix.reserve((size_t)n_pairs));
        HIP_TRY(dsi::launch_fused_splits(st, mappers[0]->pair_work.p, n > 1 ? mappers[1]->pair_work.p : nullptr, n_pairs,
                                         (uint32_t)(out->fused_fixed_cost * n), out->fused_prefix.p, out->fused_splits.p));
        splits = out->fused_splits.p;
    }
    const size_t npix = (size_t)geom.nx * geom.ny;
    HIP_TRY(out->conf.reserve(npix));
    HIP_TRY(out->depth.reserve(npix));
    HIP_TRY(out->idx.reserve(npix));
    if (int rc = depth_buffers_acquire(out)) return rc;
    // one key per pixel, zero before the kernel: zeroed once when (re)allocated, then by every unpack
    // (or after a call that failed between the voting kernel and the unpack)
    if (out->fused_keys.cap < npix + dsi::kFusedKeyTail || out->fused_keys_dirty) {
        HIP_TRY(out->fused_keys.reserve(npix + dsi::kFusedKeyTail));  // (+ the pair-dealing counters)
        HIP_TRY(hipMemsetAsync(out->fused_keys.p, 0, out->fused_keys.cap * sizeof(unsigned long long), st));
    }
    out->fused_keys_dirty = true;
    {
        VoteTimer vt(out);
        HIP_TRY(dsi::launch_vote_fuse_argmax(st, cams, geom, bp, op, splits, out->fused_keys.p,
                                             out->fused_trace_on ? out->fused_trace.p : nullptr));
        vt.stop();
    }
    // keys -> confidence, index, depth over the planes the cameras voted (mapper_emvs_stereo.cpp:302-313)
    HIP_TRY(dsi::launch_unpack_argmax(st, out->fused_keys.p, (int)npix, out->planes_dev, out->conf.p, out->idx.p,
                                      out->depth.p, /*clear=*/1));
    out->fused_keys_dirty = false;
    return depth_buffers_ready(out);
}

/* ---- exact tie resolver (include/dsi_engine.h) ---- */
namespace {
constexpr int kTieCounterWords = 24;  // 32-bit words of TieScratch::counters ([16..23]: one bit per plane with a contender)
constexpr int kTieCounterPassWords = 16;  // the words an event pass starts from zero

unsigned bits_for(unsigned long long values)  // bits that hold 0 .. values - 1
{
    unsigned b = 1;
    while (b < 64 && ((unsigned long long)1 << b) < values) ++b;
    return b;
}
}  // namespace

// The values the REFERENCE's summation order gives the voxels ts.cand[0 .. nsv) (linear indices z * npix + p, any order, no
// duplicates) of the DSIs mappers[c] build from batches[c], c < n (1 or 2): ts.exact[c * nsv + i], ts.count[c * nsv + i] on
// the device.  Per camera one inverted event pass (k_tie_hits_binned: the packet's z0 locations binned by tile in LDS,
// every voxel asks which of them the plane transfer can take into its 2 x 2 neighbourhood, those are voted with the
// reference's coordinates, accept test and weights); the recorded votes of all cameras sorted on the device by (camera,
// voxel, event index) and added one by one in fp32 by one thread per (camera, voxel).  grid_stats: also the maxima of
// |grid value - reference-order value| and of the votes per voxel into ts.counters.  Host round trips: ONE read of two
// counters (how many records to sort).
// camera centre + H_z0 of every packet (what the event pass needs besides the raw events).  Depends on no counter: the
// resolver queues it BEFORE it waits for the number of contending voxels, so that the launches behind that wait are not
// held up by it (the dispatch timeline showed ~40 us of idle stream between the wait and the first event pass)
static int tie_packet_geometry(hipStream_t st, dsi_mapper* const* ms, const dsi_batch* const* bs, int n)
{
    for (int c = 0; c < n; ++c) {
        dsi_mapper* m = ms[c];
        const size_t np = bs[c]->n_packets;
        if (!np) continue;
        HIP_TRY(m->H.reserve(np * 9));
        HIP_TRY(m->centers.reserve(np * 3));
        if (bs[c]->ready) HIP_TRY(hipStreamWaitEvent(st, bs[c]->ready, 0));
        // (the events' z0 locations are computed by the event pass itself, packet by packet: no z0 array)
        HIP_TRY(dsi::launch_packet_geometry(st, bs[c]->Rt, (int)np, m->geom, m->centers.p, m->H.p));
    }
    return DSI_OK;
}

static int tie_exact_values_dev(TieScratch& ts, hipStream_t st, dsi_mapper* const* ms, const dsi_batch* const* bs, int n, int nsv,
                                bool grid_stats, long long* votes, bool* table_overflow = nullptr, bool geometry_ready = false)
{
    *votes = 0;
    if (table_overflow) *table_overflow = false;
    if (nsv <= 0) return DSI_OK;
    const dsi::Geom& g0 = ms[0]->geom;
    const int npix = g0.nx * g0.ny;
    size_t np_max = 1;
    for (int c = 0; c < n; ++c) np_max = std::max(np_max, bs[c]->n_packets);
    const unsigned pos_bits = bits_for((unsigned long long)np_max * dsi::kPacket);
    const unsigned rank_bits = bits_for((unsigned long long)n * nsv + 1);  // + 1: the sentinel rank of unused record slots
    REQUIRE(pos_bits + rank_bits <= 64, DSI_ERR_INVALID, "too many voxels x events for a 64-bit sort key");
    const unsigned sentinel = (unsigned)n * (unsigned)nsv;
    const size_t seg = (size_t)dsi::tie_segment_records();
    HIP_TRY(ts.counters.reserve(kTieCounterWords / 2));
    HIP_TRY(ts.desc.reserve((size_t)nsv));
    HIP_TRY(ts.exact.reserve((size_t)n * nsv));
    HIP_TRY(ts.count.reserve((size_t)n * nsv));
    unsigned* cnt = reinterpret_cast<unsigned*>(ts.counters.p);
    HIP_TRY(hipMemsetAsync(cnt + kTieCounterPassWords, 0, (kTieCounterWords - kTieCounterPassWords) * sizeof(unsigned), st));
    HIP_TRY(dsi::launch_tie_desc(st, ts.cand.p, nsv, g0.nx, npix, ts.desc.p, cnt + kTieCounterPassWords));
    if (!geometry_ready)
        if (int rc = tie_packet_geometry(st, ms, bs, n)) return rc;
    // ONE pass per camera into the scratch already held (4 M records to begin with); a pass that runs out of segments only
    // counts what it would have needed, and everything is repeated once with that much room
    size_t cap = std::max<size_t>(ts.keys.cap, (size_t)1 << 22);
    unsigned host_cnt[4] = {0, 0, 0, 0};  // segments, flags, votes (64 bits)
    for (int attempt = 0;; ++attempt) {
        cap = (cap + seg - 1) / seg * seg;
        HIP_TRY(ts.keys.reserve(cap));
        HIP_TRY(ts.keys2.reserve(cap));
        HIP_TRY(ts.w.reserve(cap));
        const unsigned cap_segs = (unsigned)std::min<size_t>(ts.keys.cap / seg, 0x7fffffffu);
        HIP_TRY(hipMemsetAsync(cnt + 2, 0, (kTieCounterPassWords - 2) * sizeof(unsigned), st));
        for (int c = 0; c < n; ++c) {
            dsi_mapper* m = ms[c];
            if (!bs[c]->n_packets) continue;
            HIP_TRY(dsi::launch_tie_hits_binned(st, bs[c]->x, bs[c]->y, bs[c]->first, m->H.p, m->lut_dev, m->sensor_w, m->sensor_h,
                                                m->centers.p, m->planes_dev, m->geom, (int)bs[c]->n_packets, ts.desc.p, nsv,
                                                (unsigned)c * (unsigned)nsv, pos_bits, sentinel, cnt + 2, cap_segs, cnt + 3,
                                                ts.counters.p + 4, ts.keys.p, ts.w.p));
        }
        unsigned* pinned = nullptr;
        HIP_TRY(ts.host_counters(&pinned));
        HIP_TRY(hipMemcpyAsync(pinned, cnt, kTieCounterPassWords * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        host_cnt[0] = pinned[2];
        host_cnt[1] = pinned[3];
        host_cnt[2] = pinned[8];
        host_cnt[3] = pinned[9];
        if ((host_cnt[1] & 1u) && table_overflow) {  // (a WIDENED pass of the resolver: the caller keeps the last good pass)
            *table_overflow = true;
            return DSI_OK;
        }
        REQUIRE(!(host_cnt[1] & 1u), DSI_ERR_INVALID, "a workgroup recorded more than %d k votes: too many voxels asked for",
                dsi::tie_block_capacity_records() >> 10);
        if (!(host_cnt[1] & 2u)) break;
        REQUIRE(attempt == 0, DSI_ERR_INVALID, "the vote count changed between two passes");
        cap = (size_t)host_cnt[0] * seg;
        REQUIRE(cap < ((size_t)1 << 33), DSI_ERR_INVALID, "%zu votes to re-sum: too many voxels asked for", cap);
    }
    unsigned long long real_votes = 0;
    std::memcpy(&real_votes, host_cnt + 2, sizeof real_votes);
    *votes = (long long)real_votes;
    const size_t n_rec = (size_t)host_cnt[0] * seg;  // with the sentinel tails of the blocks' last segments
    // the votes partitioned by (camera, voxel), each voxel's run put in event order in LDS and added one by one (round 6: no
    // device-wide sort, no library call)
    REQUIRE(pos_bits <= 32 && n_rec < ((size_t)1 << 32), DSI_ERR_INVALID, "%zu votes to re-sum: too many voxels asked for", n_rec);
    if (grid_stats) HIP_TRY(ts.diff.reserve((size_t)n * nsv));
    HIP_TRY(ts.rank_count.reserve((size_t)n * nsv));
    HIP_TRY(ts.rank_start.reserve((size_t)n * nsv + 1));
    HIP_TRY(ts.rank_cursor.reserve(dsi::tie_partition_cursor_words((size_t)n * nsv)));
    HIP_TRY(dsi::launch_tie_partition_sums(st, ts.keys.p, ts.w.p, n_rec, pos_bits, ts.rank_count.p, ts.rank_start.p, ts.rank_cursor.p,
                                           ts.keys2.p, ts.cand.p, nsv, n, grid_stats ? ms[0]->grid->data : nullptr,
                                           grid_stats && n > 1 ? ms[1]->grid->data : nullptr, ts.exact.p, ts.count.p,
                                           grid_stats ? ts.diff.p : nullptr));
    return DSI_OK;
}

// columns of op(a, b) (b == nullptr: of a) with >= 2 planes within rel_gap of the column's maximum: their voxels into
// ts.cand (a column's run contiguous and ascending in z), the columns into ts.cols (k_tie_candidates); counts to the host
static int tie_candidates_dev(TieScratch& ts, hipStream_t st, const float* a, const float* b, int op, int npix, int nz, float rel_gap,
                              unsigned* n_cand, unsigned* n_columns)
{
    const size_t nvox = (size_t)npix * nz;
    unsigned counters[2] = {0, 0};
    HIP_TRY(ts.counters.reserve(kTieCounterWords / 2));
    size_t cap = std::max<size_t>(ts.cand.cap, std::min<size_t>(nvox, (size_t)1 << 20));
    size_t cols_cap = std::max<size_t>(ts.cols.cap, std::min<size_t>((size_t)npix, (size_t)1 << 18));
    for (;;) {
        HIP_TRY(ts.cand.reserve(cap));
        HIP_TRY(ts.cols.reserve(cols_cap));
        HIP_TRY(hipMemsetAsync(ts.counters.p, 0, kTieCounterWords * sizeof(unsigned), st));
        unsigned* d_cnt = reinterpret_cast<unsigned*>(ts.counters.p);
        HIP_TRY(dsi::launch_tie_candidates(st, a, b, op, npix, nz, rel_gap, d_cnt, ts.cand.p, (uint32_t)std::min<size_t>(cap, 0xffffffffu),
                                           ts.cols.p, (uint32_t)std::min<size_t>(cols_cap, 0xffffffffu)));
        unsigned* pinned = nullptr;
        HIP_TRY(ts.host_counters(&pinned));
        HIP_TRY(hipMemcpyAsync(pinned, d_cnt, sizeof counters, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        counters[0] = pinned[0];
        counters[1] = pinned[1];
        if (counters[0] <= cap && counters[1] <= cols_cap) break;
        REQUIRE(cap < nvox || cols_cap < (size_t)npix, DSI_ERR_INVALID, "more contending voxels than voxels");
        cap = std::min<size_t>(nvox, std::max<size_t>(cap, counters[0]));
        cols_cap = std::min<size_t>((size_t)npix, std::max<size_t>(cols_cap, counters[1]));
    }
    *n_cand = counters[0];
    *n_columns = counters[1];
    return DSI_OK;
}

int dsi_mapper_resolve_near_ties(dsi_mapper_t* out, dsi_mapper_t* const* mappers, const dsi_batch_t* const* batches, int n,
                                 int op, dsi_resolve_info_t* info)
{
    REQUIRE(out && mappers && batches && info, DSI_ERR_INVALID, "null argument");
    REQUIRE(n == 1 || n == 2, DSI_ERR_INVALID, "1 or 2 cameras (got %d)", n);
    REQUIRE(n == 1 || (op >= 1 && op <= 6), DSI_ERR_BAD_OP, "improper fusion method %d (expected 1..6)", op);
    float rel_gap = info->rel_gap > 0.f ? info->rel_gap : 2.5e-4f;
    REQUIRE(rel_gap < 0.5f, DSI_ERR_INVALID, "rel_gap %g is not a rounding-sized gap", (double)rel_gap);
    dsi_context* ctx = out->ctx;
    for (int i = 0; i < n; ++i) {
        REQUIRE(mappers[i] && batches[i], DSI_ERR_INVALID, "camera %d: null mapper or batch", i);
        REQUIRE(mappers[i]->ctx == ctx && batches[i]->ctx == ctx, DSI_ERR_CONTEXT,
                "mappers, batches and the output mapper must share one context");
        REQUIRE(same_shape(out->grid, mappers[i]->grid), DSI_ERR_SHAPE, "camera %d: DSI shape differs from the output mapper's", i);
    }
    REQUIRE(n == 1 || mappers[0] != mappers[1], DSI_ERR_INVALID, "the cameras need distinct mappers");
    REQUIRE(out->depth_valid, DSI_ERR_INVALID, "the output mapper holds no raw depth map to resolve");
    const dsi::Geom& g0 = out->geom;
    const int npix = g0.nx * g0.ny, nz = g0.nz;
    const size_t nvox = (size_t)npix * nz;
    REQUIRE(nvox < ((size_t)1 << 32), DSI_ERR_INVALID, "the resolver addresses voxels with 32 bits");
    if (int rc = set_device(ctx)) return rc;
    hipStream_t st = ctx->stream;
    const auto t_begin = std::chrono::steady_clock::now();
    *info = dsi_resolve_info_t{};
    TieScratch& ts = out->tie;  // grow-only scratch of the output mapper: a stream of calls allocates nothing
    auto finish = [&]() {
        info->elapsed_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        return DSI_OK;
    };
    // The premise -- the two summation orders differ by far less than the gap whose columns are re-summed -- is CHECKED,
    // not assumed: max_order_diff is measured on the re-summed voxels (the column maxima of the near-tie columns, among
    // them the most-voted voxels of the volume), and a pass that finds it above rel_gap / 8 is repeated with a gap four
    // times as wide (at most three times); premise_ok tells whether the last pass held it.
    // A WIDENED pass can ask for orders of magnitude more voxels than the first; if it overflows a workgroup's segment
    // table (or the 32-bit bookkeeping) the call does not fail: it keeps the last completed pass's patch and statistics
    // and returns DSI_OK with premise_ok = 0 -- "the caller decides", as the header says.
    {
        dsi_mapper* ms0[2] = {mappers[0], n == 2 ? mappers[1] : nullptr};
        const dsi_batch* bs0[2] = {batches[0], n == 2 ? batches[1] : nullptr};
        if (int rc = tie_packet_geometry(st, ms0, bs0, n)) return rc;  // (queued ahead of the first wait below)
    }
    for (int widenings = 0;; ++widenings) {
        const dsi_resolve_info_t last_good = *info;
        info->rel_gap = rel_gap;
        info->gap_widenings = widenings;
        // 1. the contending voxels (device lists)
        unsigned n_cand = 0, n_columns = 0;
        if (int rc = tie_candidates_dev(ts, st, mappers[0]->grid->data, n == 2 ? mappers[1]->grid->data : nullptr, op, npix, nz, rel_gap,
                                        &n_cand, &n_columns))
            return rc;
        info->near_tie_pixels = (int)n_columns;
        info->candidate_voxels = (int)n_cand;
        info->premise_ok = 1;
        if (!n_cand) return finish();
        // 2 + 3. per camera: the contending voxels' values in the reference's summation order (device)
        long long votes = 0;
        dsi_mapper* ms[2] = {mappers[0], n == 2 ? mappers[1] : nullptr};
        const dsi_batch* bs[2] = {batches[0], n == 2 ? batches[1] : nullptr};
        bool overflow = false;
        if (int rc = tie_exact_values_dev(ts, st, ms, bs, n, (int)n_cand, /*grid_stats=*/true, &votes, widenings > 0 ? &overflow : nullptr,
                                          /*geometry_ready=*/true)) {
            if (widenings == 0) return rc;
            overflow = true;  // (too many voxels x events for the sort key, > 2^33 votes: the same verdict)
            g_last_error.clear();
        }
        if (overflow) {
            *info = last_good;  // (premise_ok = 0 there: that is why this pass was tried)
            return finish();
        }
        info->votes = votes;
        // 4. fuse, first maximum per column, patch (device)
        if (int rc = depth_buffers_acquire(out)) return rc;
        unsigned* cnt = reinterpret_cast<unsigned*>(ts.counters.p);
        HIP_TRY(dsi::launch_tie_pick(st, n == 2 ? op : 0, ts.cols.p, (int)n_columns, ts.cand.p, (int)n_cand, npix, ts.exact.p, ts.count.p,
                                     ts.diff.p, out->planes_dev, out->conf.p, out->idx.p, out->depth.p, cnt + 4, rel_gap));
        if (int rc = depth_buffers_ready(out)) return rc;
        unsigned stats[4] = {0, 0, 0, 0};
        unsigned* pinned = nullptr;
        HIP_TRY(ts.host_counters(&pinned));
        HIP_TRY(hipMemcpyAsync(pinned, cnt, kTieCounterWords * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::memcpy(stats, pinned + 4, sizeof stats);
        info->candidate_planes = 0;
        for (int wd = 0; wd < kTieCounterWords - kTieCounterPassWords; ++wd) info->candidate_planes += __builtin_popcount(pinned[kTieCounterPassWords + wd]);
        float diff = 0.f;
        std::memcpy(&diff, &stats[0], sizeof diff);
        info->max_order_diff = (double)diff;  // (of the last pass: what its premise is checked with)
        info->max_rel_bound = stats[1] > 1 ? (double)(stats[1] - 1) * 5.9604644775390625e-8 : 0.0;
        info->changed_pixels += (int)stats[2];
        info->columns_bounded = (int)stats[3];
        info->premise_ok = (8.0 * (double)diff < (double)rel_gap) ? 1 : 0;  // (false for a NaN difference, too)
        if (info->premise_ok || widenings == 3 || rel_gap * 4.f >= 0.5f) return finish();
        rel_gap *= 4.f;
    }
}

int dsi_mapper_prove_near_ties(dsi_mapper_t* out, dsi_mapper_t* const* mappers, const dsi_batch_t* const* batches, int n, int op,
                               dsi_prove_info_t* info)
{
    REQUIRE(out && mappers && batches && info, DSI_ERR_INVALID, "null argument");
    REQUIRE(n == 1 || n == 2, DSI_ERR_INVALID, "1 or 2 cameras (got %d)", n);
    REQUIRE(n == 1 || (op >= 1 && op <= 6), DSI_ERR_BAD_OP, "improper fusion method %d (expected 1..6)", op);
    const float rel_gap = info->rel_gap > 0.f ? info->rel_gap : 2.5e-4f;
    REQUIRE(rel_gap < 0.5f, DSI_ERR_INVALID, "rel_gap %g is not a rounding-sized gap", (double)rel_gap);
    dsi_context* ctx = out->ctx;
    for (int i = 0; i < n; ++i) {
        REQUIRE(mappers[i] && batches[i], DSI_ERR_INVALID, "camera %d: null mapper or batch", i);
        REQUIRE(mappers[i]->ctx == ctx && batches[i]->ctx == ctx, DSI_ERR_CONTEXT,
                "mappers, batches and the output mapper must share one context");
        REQUIRE(same_shape(out->grid, mappers[i]->grid), DSI_ERR_SHAPE, "camera %d: DSI shape differs from the output mapper's", i);
        // the bounds describe the EXACT sums of the LDS-band mappings (weights truncated to 2^-31, one rounding): a DSI voted
        // with fp32 global atomics or with the paired 32-bit cells (rounded Q.19 weights) is not one
        REQUIRE(batches[i]->n_packets == 0 || (mappers[i]->info.algo == DSI_VOTE_LDS_BANDS && mappers[i]->info.packed != 8),
                DSI_ERR_INVALID, "camera %d: the proof needs a DSI of exact sums (DSI_VOTE_LDS_BANDS, not lane mapping 8)", i);
    }
    REQUIRE(n == 1 || mappers[0] != mappers[1], DSI_ERR_INVALID, "the cameras need distinct mappers");
    const dsi::Geom& g0 = out->geom;
    const size_t nvox = (size_t)g0.nx * g0.ny * g0.nz;
    if (int rc = set_device(ctx)) return rc;
    hipStream_t st = ctx->stream;
    const auto t_begin = std::chrono::steady_clock::now();
    *info = dsi_prove_info_t{};
    info->rel_gap = rel_gap;
    TieScratch& ts = out->tie;
    dsi_mapper* ms[2] = {mappers[0], n == 2 ? mappers[1] : nullptr};
    const dsi_batch* bs[2] = {batches[0], n == 2 ? batches[1] : nullptr};
    if (int rc = tie_packet_geometry(st, ms, bs, n)) return rc;
    for (bool& v : ts.votes_valid) v = false;
    for (int c = 0; c < n; ++c) {
        HIP_TRY(ts.votes[c].reserve(nvox));
        HIP_TRY(hipMemsetAsync(ts.votes[c].p, 0, nvox * sizeof(uint32_t), st));
        dsi_mapper* m = ms[c];
        if (!bs[c]->n_packets) continue;
        HIP_TRY(dsi::launch_count_votes(st, bs[c]->x, bs[c]->y, bs[c]->first, m->H.p, m->lut_dev, m->sensor_w, m->sensor_h, m->centers.p,
                                        m->planes_dev, m->geom, (int)bs[c]->n_packets, ts.votes[c].p));
    }
    HIP_TRY(ts.counters.reserve(kTieCounterWords / 2));
    unsigned* cnt = reinterpret_cast<unsigned*>(ts.counters.p);
    HIP_TRY(hipMemsetAsync(cnt, 0, 5 * sizeof(unsigned), st));
    HIP_TRY(ts.unproven.reserve((size_t)g0.nx * g0.ny));
    ts.n_unproven = 0;
    HIP_TRY(dsi::launch_tie_prove(st, ms[0]->grid->data, n == 2 ? ms[1]->grid->data : nullptr, ts.votes[0].p, n == 2 ? ts.votes[1].p : nullptr,
                                  op, g0.nx, g0.ny, g0.nz, rel_gap, cnt, ts.unproven.p));
    unsigned* pinned = nullptr;
    HIP_TRY(ts.host_counters(&pinned));
    HIP_TRY(hipMemcpyAsync(pinned, cnt, 5 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int c = 0; c < n; ++c) ts.votes_valid[c] = true;
    ts.n_unproven = pinned[4];
    info->columns = (long long)g0.nx * g0.ny;
    info->columns_proven = pinned[0];
    info->columns_unproven = pinned[1];
    float need = 0.f;
    std::memcpy(&need, &pinned[2], sizeof need);
    info->gap_needed = pinned[1] ? (double)need : 0.0;
    info->max_votes = pinned[3];
    info->elapsed_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return DSI_OK;
}

int dsi_mapper_prove_near_ties_n(dsi_mapper_t* out, dsi_grid_t* fused, dsi_mapper_t* const* mappers, const dsi_batch_t* const* batches,
                                 int n, int mode, dsi_prove_info_t* info)
{
    REQUIRE(out && fused && mappers && batches && info, DSI_ERR_INVALID, "null argument");
    REQUIRE(n >= 1 && n <= 8, DSI_ERR_INVALID, "1 to 8 cameras (got %d)", n);
    REQUIRE(mode == DSI_ACC_GM_TREE || mode == DSI_ACC_MIN || mode == DSI_ACC_MAX || mode == DSI_ACC_SUM, DSI_ERR_BAD_OP,
            "mode %d: the proof covers DSI_ACC_GM_TREE, DSI_ACC_MIN, DSI_ACC_MAX and DSI_ACC_SUM", mode);
    REQUIRE(mode != DSI_ACC_GM_TREE || n == 2 || n == 4 || n == 8, DSI_ERR_BAD_OP, "DSI_ACC_GM_TREE needs 2, 4 or 8 cameras (got %d)", n);
    const float rel_gap = info->rel_gap > 0.f ? info->rel_gap : 2.5e-4f;
    REQUIRE(rel_gap < 0.5f, DSI_ERR_INVALID, "rel_gap %g is not a rounding-sized gap", (double)rel_gap);
    dsi_context* ctx = out->ctx;
    REQUIRE(fused->ctx == ctx && same_shape(out->grid, fused), DSI_ERR_SHAPE, "the fused grid must have the output mapper's shape and context");
    for (int i = 0; i < n; ++i) {
        REQUIRE(mappers[i] && batches[i], DSI_ERR_INVALID, "camera %d: null mapper or batch", i);
        REQUIRE(mappers[i]->ctx == ctx && batches[i]->ctx == ctx, DSI_ERR_CONTEXT,
                "mappers, batches and the output mapper must share one context");
        REQUIRE(same_shape(out->grid, mappers[i]->grid), DSI_ERR_SHAPE, "camera %d: DSI shape differs from the output mapper's", i);
        REQUIRE(batches[i]->n_packets == 0 || (mappers[i]->info.algo == DSI_VOTE_LDS_BANDS && mappers[i]->info.packed != 8),
                DSI_ERR_INVALID, "camera %d: the proof needs a DSI of exact sums (DSI_VOTE_LDS_BANDS, not lane mapping 8)", i);
        for (int k = 0; k < i; ++k) REQUIRE(mappers[i] != mappers[k], DSI_ERR_INVALID, "the cameras need distinct mappers");
    }
    const dsi::Geom& g0 = out->geom;
    const size_t nvox = (size_t)g0.nx * g0.ny * g0.nz;
    if (int rc = set_device(ctx)) return rc;
    hipStream_t st = ctx->stream;
    const auto t_begin = std::chrono::steady_clock::now();
    *info = dsi_prove_info_t{};
    info->rel_gap = rel_gap;
    TieScratch& ts = out->tie;
    for (bool& v : ts.votes_valid) v = false;
    const float* e[8] = {};
    const uint32_t* h[8] = {};
    for (int c = 0; c < n; ++c) {
        dsi_mapper* m = mappers[c];
        const dsi_batch* b = batches[c];
        dsi_mapper* one_m[1] = {m};
        const dsi_batch* one_b[1] = {b};
        if (int rc = tie_packet_geometry(st, one_m, one_b, 1)) return rc;
        HIP_TRY(ts.votes[c].reserve(nvox));
        HIP_TRY(hipMemsetAsync(ts.votes[c].p, 0, nvox * sizeof(uint32_t), st));
        if (b->n_packets)
            HIP_TRY(dsi::launch_count_votes(st, b->x, b->y, b->first, m->H.p, m->lut_dev, m->sensor_w, m->sensor_h, m->centers.p,
                                            m->planes_dev, m->geom, (int)b->n_packets, ts.votes[c].p));
        e[c] = m->grid->data;
        h[c] = ts.votes[c].p;
    }
    HIP_TRY(ts.counters.reserve(kTieCounterWords / 2));
    unsigned* cnt = reinterpret_cast<unsigned*>(ts.counters.p);
    HIP_TRY(hipMemsetAsync(cnt, 0, 5 * sizeof(unsigned), st));
    HIP_TRY(ts.unproven.reserve((size_t)g0.nx * g0.ny));
    ts.n_unproven = 0;
    HIP_TRY(dsi::launch_tie_prove_n(st, fused->data, e, h, n, mode, g0.nx, g0.ny, g0.nz, rel_gap, cnt, ts.unproven.p));
    unsigned* pinned = nullptr;
    HIP_TRY(ts.host_counters(&pinned));
    HIP_TRY(hipMemcpyAsync(pinned, cnt, 5 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int c = 0; c < n; ++c) ts.votes_valid[c] = true;
    ts.n_unproven = pinned[4];
    info->columns = (long long)g0.nx * g0.ny;
    info->columns_proven = pinned[0];
    info->columns_unproven = pinned[1];
    float need = 0.f;
    std::memcpy(&need, &pinned[2], sizeof need);
    info->gap_needed = pinned[1] ? (double)need : 0.0;
    info->max_votes = pinned[3];
    info->elapsed_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return DSI_OK;
}

int dsi_mapper_reference_interval(dsi_mapper_t* scratch, dsi_mapper_t* m, const dsi_batch_t* batch, dsi_grid_t* lo, dsi_grid_t* hi)
{
    REQUIRE(scratch && m && batch && lo && hi, DSI_ERR_INVALID, "null argument");
    dsi_context* ctx = scratch->ctx;
    REQUIRE(m->ctx == ctx && batch->ctx == ctx && lo->ctx == ctx && hi->ctx == ctx, DSI_ERR_CONTEXT, "everything must share one context");
    REQUIRE(same_shape(scratch->grid, m->grid) && same_shape(m->grid, lo) && same_shape(m->grid, hi), DSI_ERR_SHAPE, "shapes differ");
    REQUIRE(lo != hi && lo != m->grid && hi != m->grid, DSI_ERR_INVALID, "lo, hi and the DSI must be three grids");
    REQUIRE(batch->n_packets == 0 || (m->info.algo == DSI_VOTE_LDS_BANDS && m->info.packed != 8), DSI_ERR_INVALID,
            "the bounds need a DSI of exact sums (DSI_VOTE_LDS_BANDS, not lane mapping 8)");
    const dsi::Geom& g0 = m->geom;
    const size_t nvox = (size_t)g0.nx * g0.ny * g0.nz;
    if (int rc = set_device(ctx)) return rc;
    hipStream_t st = ctx->stream;
    TieScratch& ts = scratch->tie;
    dsi_mapper* one_m[1] = {m};
    const dsi_batch* one_b[1] = {batch};
    if (int rc = tie_packet_geometry(st, one_m, one_b, 1)) return rc;
    HIP_TRY(ts.votes[7].reserve(nvox));  // (the last counter volume: the cameras' own of a proof stay valid)
    ts.votes_valid[7] = false;
    HIP_TRY(hipMemsetAsync(ts.votes[7].p, 0, nvox * sizeof(uint32_t), st));
    if (batch->n_packets)
        HIP_TRY(dsi::launch_count_votes(st, batch->x, batch->y, batch->first, m->H.p, m->lut_dev, m->sensor_w, m->sensor_h, m->centers.p,
                                        m->planes_dev, m->geom, (int)batch->n_packets, ts.votes[7].p));
    HIP_TRY(ts.interval_most.reserve(1));
    if (!ts.interval_most_live) {
        HIP_TRY(hipMemsetAsync(ts.interval_most.p, 0, sizeof(unsigned), st));
        ts.interval_most_live = true;
    }
    HIP_TRY(dsi::launch_interval_of_counts(st, m->grid->data, ts.votes[7].p, g0.nx, g0.ny, g0.nz, lo->data, hi->data, ts.interval_most.p));
    return DSI_OK;
}

int dsi_grid_widen_interval(dsi_grid_t* lo, dsi_grid_t* hi, int roundings)
{
    if (int rc = check_pair(lo, hi)) return rc;
    REQUIRE(lo != hi, DSI_ERR_INVALID, "lo and hi must be two grids");
    REQUIRE(roundings >= 1 && roundings <= 4096, DSI_ERR_INVALID, "roundings %d (1 .. 4096)", roundings);
    HIP_TRY(dsi::launch_interval_widen(lo->ctx->stream, lo->data, hi->data, lo->n, roundings));
    return DSI_OK;
}

int dsi_grid_prove_columns(dsi_mapper_t* scratch, dsi_grid_t* fused, dsi_grid_t* lo, dsi_grid_t* hi, dsi_prove_info_t* info)
{
    REQUIRE(scratch && fused && lo && hi && info, DSI_ERR_INVALID, "null argument");
    dsi_context* ctx = scratch->ctx;
    REQUIRE(fused->ctx == ctx && lo->ctx == ctx && hi->ctx == ctx, DSI_ERR_CONTEXT, "everything must share one context");
    REQUIRE(same_shape(scratch->grid, fused) && same_shape(fused, lo) && same_shape(fused, hi), DSI_ERR_SHAPE, "shapes differ");
    const float rel_gap = info->rel_gap > 0.f ? info->rel_gap : 2.5e-4f;
    REQUIRE(rel_gap < 0.5f, DSI_ERR_INVALID, "rel_gap %g is not a rounding-sized gap", (double)rel_gap);
    if (int rc = set_device(ctx)) return rc;
    hipStream_t st = ctx->stream;
    const auto t_begin = std::chrono::steady_clock::now();
    *info = dsi_prove_info_t{};
    info->rel_gap = rel_gap;
    TieScratch& ts = scratch->tie;
    HIP_TRY(ts.counters.reserve(kTieCounterWords / 2));
    unsigned* cnt = reinterpret_cast<unsigned*>(ts.counters.p);
    HIP_TRY(hipMemsetAsync(cnt, 0, 5 * sizeof(unsigned), st));
    const int npix = fused->nx * fused->ny;
    HIP_TRY(ts.unproven.reserve((size_t)npix));
    ts.n_unproven = 0;
    HIP_TRY(dsi::launch_prove_columns(st, fused->data, lo->data, hi->data, npix, fused->nz, rel_gap, cnt, ts.unproven.p));
    unsigned* pinned = nullptr;
    HIP_TRY(ts.host_counters(&pinned));
    HIP_TRY(hipMemcpyAsync(pinned, cnt, 5 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    if (ts.interval_most_live) HIP_TRY(hipMemcpyAsync(pinned + 8, ts.interval_most.p, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    ts.votes_valid[0] = true;  // (dsi_mapper_proof_unproven asks whether a proof has run)
    ts.n_unproven = pinned[4];
    info->columns = npix;
    info->columns_proven = pinned[0];
    info->columns_unproven = pinned[1];
    float need = 0.f;
    std::memcpy(&need, &pinned[2], sizeof need);
    info->gap_needed = pinned[1] ? (double)need : 0.0;
    info->max_votes = ts.interval_most_live ? pinned[8] : 0;
    ts.interval_most_live = false;  // (the next composition starts its own maximum)
    info->elapsed_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return DSI_OK;
}

int dsi_mapper_proof_unproven(dsi_mapper_t* out, uint32_t* pixels, float* gaps, size_t capacity, size_t* n)
{
    REQUIRE(out && n && (capacity == 0 || pixels), DSI_ERR_INVALID, "null argument");
    TieScratch& ts = out->tie;
    REQUIRE(ts.votes_valid[0], DSI_ERR_INVALID, "no proof has run on this mapper");
    *n = ts.n_unproven;
    const size_t take = std::min(capacity, ts.n_unproven);
    if (!take) return DSI_OK;
    if (int rc = set_device(out->ctx)) return rc;
    std::vector<uint2> host(take);
    HIP_TRY(hipMemcpyAsync(host.data(), ts.unproven.p, take * sizeof(uint2), hipMemcpyDeviceToHost, out->ctx->stream));
    HIP_TRY(hipStreamSynchronize(out->ctx->stream));
    for (size_t i = 0; i < take; ++i) {
        pixels[i] = host[i].x;
        if (gaps) std::memcpy(&gaps[i], &host[i].y, sizeof(float));
    }
    return DSI_OK;
}

int dsi_mapper_proof_votes(dsi_mapper_t* out, int camera, const uint32_t* voxels, size_t n, uint32_t* votes)
{
    REQUIRE(out && (n == 0 || (voxels && votes)), DSI_ERR_INVALID, "null argument");
    REQUIRE(camera >= 0 && camera < 8, DSI_ERR_INVALID, "camera %d (0 .. 7)", camera);
    const dsi::Geom& g0 = out->geom;
    const size_t nvox = (size_t)g0.nx * g0.ny * g0.nz;
    TieScratch& ts = out->tie;
    REQUIRE(ts.votes[camera].cap >= nvox && ts.votes_valid[camera], DSI_ERR_INVALID, "no proof has counted camera %d's votes on this mapper", camera);
    REQUIRE(n < ((size_t)1 << 31), DSI_ERR_INVALID, "too many voxels");
    for (size_t i = 0; i < n; ++i) REQUIRE(voxels[i] < nvox, DSI_ERR_INVALID, "voxel %zu (%u) is outside the grid", i, voxels[i]);
    if (n == 0) return DSI_OK;
    if (int rc = set_device(out->ctx)) return rc;
    hipStream_t st = out->ctx->stream;
    HIP_TRY(ts.cand.reserve(n));
    HIP_TRY(ts.count.reserve(n));
    HIP_TRY(hipMemcpyAsync(ts.cand.p, voxels, n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(dsi::launch_tie_votes_of(st, ts.votes[camera].p, ts.cand.p, (int)n, g0.nx, g0.ny, ts.count.p));
    HIP_TRY(hipMemcpyAsync(votes, ts.count.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));  // (pageable host arrays)
    return DSI_OK;
}

int dsi_grid_near_tie_voxels(dsi_mapper_t* scratch, dsi_grid_t* g, float rel_gap, uint32_t* voxels, size_t capacity,
                             size_t* n_voxels, size_t* n_columns)
{
    REQUIRE(scratch && g && n_voxels, DSI_ERR_INVALID, "null argument");
    REQUIRE(capacity == 0 || voxels, DSI_ERR_INVALID, "null output");
    REQUIRE(scratch->ctx == g->ctx, DSI_ERR_CONTEXT, "the scratch mapper must live in the grid's context");
    REQUIRE(same_shape(scratch->grid, g), DSI_ERR_SHAPE, "the scratch mapper's DSI shape differs from the grid's");
    if (!(rel_gap > 0.f)) rel_gap = 2.5e-4f;
    REQUIRE(rel_gap < 0.5f, DSI_ERR_INVALID, "rel_gap %g is not a rounding-sized gap", (double)rel_gap);
    REQUIRE(g->n < ((size_t)1 << 32), DSI_ERR_INVALID, "voxels are addressed with 32 bits");
    REQUIRE(g->nz <= 256, DSI_ERR_SHAPE, "the near-tie search holds a column's planes in four 64-bit masks: dimZ %d > 256", g->nz);
    if (int rc = set_device(g->ctx)) return rc;
    unsigned n_cand = 0, cols = 0;
    if (int rc = tie_candidates_dev(scratch->tie, g->ctx->stream, g->data, nullptr, 0, g->nx * g->ny, g->nz, rel_gap, &n_cand, &cols)) return rc;
    *n_voxels = n_cand;
    if (n_columns) *n_columns = cols;
    if (n_cand <= capacity && n_cand) HIP_TRY(hipMemcpy(voxels, scratch->tie.cand.p, (size_t)n_cand * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return DSI_OK;
}

int dsi_mapper_exact_voxels(dsi_mapper_t* m, const dsi_batch_t* batch, const uint32_t* voxels, size_t n, float* values,
                            uint32_t* votes)
{
    REQUIRE(m && batch, DSI_ERR_INVALID, "null argument");
    REQUIRE(n == 0 || (voxels && values), DSI_ERR_INVALID, "null array");
    REQUIRE(m->ctx == batch->ctx, DSI_ERR_CONTEXT, "mapper and batch belong to different contexts");
    const size_t nvox = (size_t)m->geom.nx * m->geom.ny * m->geom.nz;
    REQUIRE(nvox < ((size_t)1 << 32), DSI_ERR_INVALID, "voxels are addressed with 32 bits");
    for (size_t i = 0; i < n; ++i) REQUIRE(voxels[i] < nvox, DSI_ERR_INVALID, "voxel %zu (%u) outside the DSI", i, voxels[i]);
    if (n == 0) return DSI_OK;
    if (int rc = set_device(m->ctx)) return rc;
    hipStream_t st = m->ctx->stream;
    std::vector<uint32_t> sv(voxels, voxels + n);
    std::sort(sv.begin(), sv.end());
    sv.erase(std::unique(sv.begin(), sv.end()), sv.end());
    REQUIRE(sv.size() < ((size_t)1 << 31), DSI_ERR_INVALID, "too many voxels");
    TieScratch& ts = m->tie;
    HIP_TRY(ts.cand.reserve(sv.size()));
    HIP_TRY(hipMemcpyAsync(ts.cand.p, sv.data(), sv.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    long long total = 0;
    dsi_mapper* ms[1] = {m};
    const dsi_batch* bs[1] = {batch};
    if (int rc = tie_exact_values_dev(ts, st, ms, bs, 1, (int)sv.size(), /*grid_stats=*/false, &total)) return rc;
    std::vector<float> exact(sv.size());
    std::vector<uint32_t> count(sv.size());
    HIP_TRY(hipMemcpyAsync(exact.data(), ts.exact.p, sv.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(count.data(), ts.count.p, sv.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));  // (sv is pageable: its upload has completed by now, too)
    for (size_t i = 0; i < n; ++i) {
        const size_t c = (size_t)(std::lower_bound(sv.begin(), sv.end(), voxels[i]) - sv.begin());
        values[i] = exact[c];
        if (votes) votes[i] = count[c];
    }
    return DSI_OK;
}

int dsi_reference_fuse2(int op, const float* a, const float* g, size_t n, float* out)
{
    REQUIRE(n == 0 || (a && g && out), DSI_ERR_INVALID, "null array");
    REQUIRE(op >= 1 && op <= 6, DSI_ERR_BAD_OP, "improper fusion method %d (expected 1..6)", op);
    for (size_t i = 0; i < n; ++i) out[i] = dsi::host::fuse2(op, a[i], g[i]);
    return DSI_OK;
}

int dsi_reference_accumulate(int mode, float* acc, const float* g, size_t n)
{
    REQUIRE(n == 0 || (acc && g), DSI_ERR_INVALID, "null array");
    REQUIRE(mode == DSI_ACC_SUM || mode == DSI_ACC_INV_SUM, DSI_ERR_BAD_OP, "the reference accumulates sums (0) or inverse sums (1)");
    for (size_t i = 0; i < n; ++i) acc[i] = dsi::host::accumulate1(mode, acc[i], g[i]);
    return DSI_OK;
}

int dsi_reference_finalize(int mode, float* acc, size_t n, int n_maps)
{
    REQUIRE(n == 0 || acc, DSI_ERR_INVALID, "null array");
    REQUIRE(mode == DSI_ACC_SUM || mode == DSI_ACC_INV_SUM, DSI_ERR_BAD_OP, "the reference accumulates sums (0) or inverse sums (1)");
    REQUIRE(n_maps >= 1, DSI_ERR_INVALID, "number of maps must be >= 1 (got %d)", n_maps);
    for (size_t i = 0; i < n; ++i) acc[i] = dsi::host::finalize1(mode, acc[i], n_maps);
    return DSI_OK;
}

static int patch_depth_map(dsi_mapper_t* m, TieScratch& ts, const uint32_t* pix, const uint8_t* idx, const float* conf, size_t n)
{
    if (n == 0) return DSI_OK;
    hipStream_t st = m->ctx->stream;
    HIP_TRY(ts.count.reserve(n));
    HIP_TRY(ts.exact.reserve(n));
    HIP_TRY(ts.tmp.reserve(n));
    // (the patch lists reuse the resolver's scratch: pix <- count, conf <- exact, idx <- tmp)
    HIP_TRY(hipMemcpyAsync(ts.count.p, pix, n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ts.exact.p, conf, n * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ts.tmp.p, idx, n, hipMemcpyHostToDevice, st));
    if (int rc = depth_buffers_acquire(m)) return rc;
    HIP_TRY(dsi::launch_tie_patch(st, ts.count.p, reinterpret_cast<const uint8_t*>(ts.tmp.p), ts.exact.p, (int)n, m->planes_dev,
                                  m->conf.p, m->idx.p, m->depth.p));
    if (int rc = depth_buffers_ready(m)) return rc;
    HIP_TRY(hipStreamSynchronize(st));  // the host arrays are pageable
    return DSI_OK;
}

int dsi_mapper_patch_depth_map(dsi_mapper_t* m, const uint32_t* pixels, const uint8_t* idx, const float* conf, size_t n)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    REQUIRE(n == 0 || (pixels && idx && conf), DSI_ERR_INVALID, "null array");
    REQUIRE(m->depth_valid, DSI_ERR_INVALID, "the mapper holds no raw depth map to patch");
    const size_t npix = (size_t)m->geom.nx * m->geom.ny;
    for (size_t i = 0; i < n; ++i) {
        REQUIRE(pixels[i] < npix, DSI_ERR_INVALID, "pixel %zu (%u) outside the image", i, pixels[i]);
        REQUIRE((int)idx[i] < m->geom.nz, DSI_ERR_INVALID, "plane index %d outside the depth vector", (int)idx[i]);
    }
    if (int rc = set_device(m->ctx)) return rc;
    return patch_depth_map(m, m->tie, pixels, idx, conf, n);
}

int dsi_mapper_fetch_depth_map(dsi_mapper_t* m, float* depth_host, float* conf_host, uint8_t* idx_host)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    REQUIRE(m->depth_valid, DSI_ERR_INVALID, "no depth map has been computed since the last vote");
    if (int rc = set_device(m->ctx)) return rc;
    return depth_buffers_fetch(m, depth_host, conf_host, idx_host, /*wait=*/true);
}

int dsi_mapper_fetch_depth_map_async(dsi_mapper_t* m, float* depth_host, float* conf_host, uint8_t* idx_host)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    REQUIRE(m->depth_valid, DSI_ERR_INVALID, "no depth map has been computed since the last vote");
    if (int rc = set_device(m->ctx)) return rc;
    return depth_buffers_fetch(m, depth_host, conf_host, idx_host, /*wait=*/false);
}

int dsi_mapper_fetch_depth_map_in_order(dsi_mapper_t* m, float* depth_host, float* conf_host, uint8_t* idx_host)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    REQUIRE(m->depth_valid, DSI_ERR_INVALID, "no depth map has been computed since the last vote");
    if (int rc = set_device(m->ctx)) return rc;
    return depth_buffers_fetch(m, depth_host, conf_host, idx_host, /*wait=*/false, /*in_order=*/true);
}

int dsi_mapper_fetch_wait(dsi_mapper_t* m)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    if (!m->depth_read_pending) return DSI_OK;
    if (int rc = set_device(m->ctx)) return rc;
    HIP_TRY(hipEventSynchronize(m->ev_depth_read));
    return DSI_OK;
}

int dsi_mapper_depth_map(dsi_mapper_t* m, float* depth_host, float* conf_host, uint8_t* idx_host)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    if (int rc = dsi_mapper_depth_map_of(m, m->grid)) return rc;
    return dsi_mapper_fetch_depth_map(m, depth_host, conf_host, idx_host);
}

int dsi_mapper_set_kernel_timing(dsi_mapper_t* m, int enable)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    m->timing = enable != 0;
    return DSI_OK;
}

int dsi_mapper_vote_kernel_time(dsi_mapper_t* m, float* total_ms, int* launches)
{
    REQUIRE(m && total_ms && launches, DSI_ERR_INVALID, "null argument");
    if (int rc = set_device(m->ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    float total = 0.f;
    int n = 0;
    for (auto& pr : m->timing_pairs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            total += ms;
            ++n;
        }
        m->timing_pool.push_back(pr.first);
        m->timing_pool.push_back(pr.second);
    }
    m->timing_pairs.clear();
    *total_ms = total;
    *launches = n;
    return DSI_OK;
}

static int check_depthmap_options(const dsi_depthmap_options_t* opts)
{
    REQUIRE(opts->adaptive_threshold_kernel_size >= 1 && (opts->adaptive_threshold_kernel_size & 1) &&
                opts->adaptive_threshold_kernel_size <= 63,
            DSI_ERR_INVALID, "adaptive_threshold_kernel_size must be odd and in 1..63");
    REQUIRE(opts->median_filter_size >= 1 && (opts->median_filter_size & 1) && opts->median_filter_size <= 31,
            DSI_ERR_INVALID, "median_filter_size must be odd and in 1..31 (median_filtering.cpp:44)");
    return DSI_OK;
}

int dsi_mapper_filter_depth_map(dsi_mapper_t* m, const dsi_depthmap_options_t* opts, float* depth_host,
                                float* conf_host, uint8_t* mask_host, uint8_t* idx_filtered_host)
{
    REQUIRE(m && opts, DSI_ERR_INVALID, "null argument");
    if (int rc = check_depthmap_options(opts)) return rc;
    REQUIRE(m->depth_valid, DSI_ERR_INVALID,
            "no raw depth map on this mapper: call one of dsi_mapper_depth_map_of* first (a filtered map "
            "cannot be filtered again: the confidence image was normalised in place)");
    dsi_context* ctx = m->ctx;
    const int nx = m->geom.nx, ny = m->geom.ny;
    const size_t npix = (size_t)nx * ny;
    if (int rc = set_device(ctx)) return rc;
    if (int rc = depth_buffers_acquire(m)) return rc;  // an asynchronous fetch of the raw map may be reading them
    HIP_TRY(m->conf8.reserve(npix));
    HIP_TRY(m->mask.reserve(npix));
    HIP_TRY(m->idx_filtered.reserve(npix));
    HIP_TRY(m->minmax.reserve(2));
    HIP_TRY(dsi::launch_depth_map_filters(ctx->stream, m->conf.p, m->idx.p, nx, ny,
                                          opts->adaptive_threshold_kernel_size, opts->adaptive_threshold_c,
                                          opts->median_filter_size, opts->max_confidence, m->planes_dev,
                                          m->minmax.p, m->conf8.p, m->mask.p, m->idx_filtered.p, m->depth.p));
    if (depth_host)
        HIP_TRY(hipMemcpyAsync(depth_host, m->depth.p, npix * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (conf_host)
        HIP_TRY(hipMemcpyAsync(conf_host, m->conf.p, npix * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (mask_host) HIP_TRY(hipMemcpyAsync(mask_host, m->mask.p, npix, hipMemcpyDeviceToHost, ctx->stream));
    if (idx_filtered_host)
        HIP_TRY(hipMemcpyAsync(idx_filtered_host, m->idx_filtered.p, npix, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    m->depth_valid = false;  // m->depth now holds the filtered map, m->conf has (0,0) overwritten
    return DSI_OK;
}

int dsi_mapper_get_depth_map_from_dsi(dsi_mapper_t* m, dsi_grid_t* g, const dsi_depthmap_options_t* opts,
                                      float* depth_host, float* conf_host, uint8_t* mask_host,
                                      uint8_t* idx_filtered_host)
{
    REQUIRE(m && opts, DSI_ERR_INVALID, "null argument");
    if (int rc = check_depthmap_options(opts)) return rc;
    if (!g) g = m->grid;
    if (int rc = dsi_mapper_depth_map_of(m, g)) return rc;  // collapseMaxZSlice, :368 (acquires the buffers)
    return dsi_mapper_filter_depth_map(m, opts, depth_host, conf_host, mask_host, idx_filtered_host);
}

int dsi_mapper_last_vote_info(const dsi_mapper_t* m, dsi_vote_info_t* info)
{
    REQUIRE(m && info, DSI_ERR_INVALID, "null argument");
    *info = m->info;
    return DSI_OK;
}

/* Diagnostics of the voting kernel's work on one batch (bench.py: roofline.achieved / frac_issued).  Votes the batch
 * twice: once with every merged record counted once -- the sum of that DSI is the number of RECORDS the voting kernel
 * accepted, a quarter of the LDS atomics it issues -- and once as dsi_mapper_evaluate_batch does, whose DSI sums to the
 * accepted event-planes (the four bilinear weights of a vote sum to 1) and is what the mapper's grid holds afterwards.
 * The switch that makes the first pass is internal: no caller can leave it on.  Synchronises. */
int dsi_mapper_vote_statistics(dsi_mapper_t* m, const dsi_batch_t* batch, double* accepted_event_planes, double* accepted_records)
{
    REQUIRE(m && batch && accepted_event_planes && accepted_records, DSI_ERR_INVALID, "null argument");
    std::vector<float> host(m->grid->n);
    double sums[2] = {0.0, 0.0};
    for (int pass = 0; pass < 2; ++pass) {
        m->unit_multiplicity = pass == 0 ? 1 : 0;
        const int rc = dsi_mapper_evaluate_batch(m, batch);
        m->unit_multiplicity = 0;
        if (rc != DSI_OK) return rc;
        if (int rc2 = dsi_grid_download(m->grid, host.data())) return rc2;
        double t = 0.0;
        for (float v : host) t += (double)v;
        sums[pass] = t;
    }
    *accepted_records = sums[0];
    *accepted_event_planes = sums[1];
    return DSI_OK;
}

int dsi_build_flavour(void)
{
#ifdef DSI_TIMING_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

#ifdef DSI_TIMING_EXPERIMENTS
/* Everything from here to the #endif exists only in the EXPERIMENTS flavour of the library
 * (libdsi_engine_experiments.so, `python -m dvs_mcemvs_amd.build --experiments`): hooks of the timing experiments and
 * development tools quoted in NOTEBOOK.md.  The production library exports no dsi_test_* symbol (tests/test_abi.py). */
/* test hook (not in the public header): packets per pass of the voting streams, as a power of two (0 = automatic) */
DSI_API int dsi_test_pass_lg(dsi_mapper_t* m, int lg)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    m->want_pass_lg = lg;
    return DSI_OK;
}

/* test hook (not in the public header): the fixed cost of one phase of the fused kernel in record units (the
 * balanced partition's only tunable); < 0 switches the balancing off (equal pair counts per workgroup) */
DSI_API int dsi_test_fused_fixed_cost(dsi_mapper_t* m, int records)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    m->fused_fixed_cost = records;
    return DSI_OK;
}

/* test hooks (not in the public header): time stamps of the phases of the next fused vote kernels that
 * mapper m leads (dsi_mapper_depth_map_of_events with mappers[0] == m): [workgroup][64 phases][16 waves][4]
 * ticks of the 100 MHz clock; *_read synchronises, copies and switches tracing off */
DSI_API int dsi_test_fused_trace_enable(dsi_mapper_t* m, size_t* n_words)
{
    REQUIRE(m && n_words, DSI_ERR_INVALID, "null argument");
    if (int rc = set_device(m->ctx)) return rc;
    *n_words = dsi::fused_trace_words();
    HIP_TRY(m->fused_trace.reserve(*n_words));
    HIP_TRY(hipMemsetAsync(m->fused_trace.p, 0, *n_words * sizeof(unsigned long long), m->ctx->stream));
    m->fused_trace_on = true;
    return DSI_OK;
}

DSI_API int dsi_test_fused_trace_read(dsi_mapper_t* m, unsigned long long* host, size_t n_words)
{
    REQUIRE(m && host, DSI_ERR_INVALID, "null argument");
    REQUIRE(m->fused_trace_on && n_words <= m->fused_trace.cap, DSI_ERR_INVALID, "tracing is not enabled");
    if (int rc = set_device(m->ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    HIP_TRY(hipMemcpy(host, m->fused_trace.p, n_words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    m->fused_trace_on = false;
    return DSI_OK;
}

/* test hook (not in the public header): total length of all runs [lo,hi) of the last banded
 * vote, i.e. the number of event-planes the voting kernel looked at (>= the number accepted) */
DSI_API int dsi_test_run_length_total(dsi_mapper_t* m, unsigned long long* total, unsigned long long* entries)
{
    REQUIRE(m && total && entries, DSI_ERR_INVALID, "null argument");
    if (int rc = set_device(m->ctx)) return rc;
    size_t units = m->info.n_packets;
    REQUIRE(!m->info_cuts_inline, DSI_ERR_INVALID, "the last vote derived its runs in the kernel: there is no cut table to add up");
    const uint32_t* src = m->cuts.p;
    if (m->info.packed == 2 || m->info.packed == 4) {  // grouped mapping: one run per group of packets
        units = (m->info.n_packets + m->info.group_packets - 1) / m->info.group_packets;
        src = m->gcuts.p;
    }
    const size_t n = units * (size_t)m->geom.nz * (size_t)m->info.bands;
    std::vector<uint32_t> h(n);
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    if (n) HIP_TRY(hipMemcpy(h.data(), src, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    unsigned long long t = 0;
    for (uint32_t c : h) t += (c >> 16) - (c & 0xffffu);
    *total = t;
    *entries = n;
    return DSI_OK;
}

/* test hook (not in the public header): residual-corrected division vs IEEE divide */
DSI_API int dsi_test_div_probe(dsi_context_t* ctx, const float* n, const float* d, size_t count, float* q, float* ref)
{
    REQUIRE(ctx && n && d && q && ref, DSI_ERR_INVALID, "null argument");
    if (int rc = set_device(ctx)) return rc;
    float *dn = nullptr, *dd = nullptr, *dq = nullptr, *dr = nullptr;
    const size_t bytes = count * sizeof(float);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dn), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dd), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dq), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dr), bytes);
    if (e == hipSuccess) e = hipMemcpy(dn, n, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dd, d, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = dsi::launch_div_probe(ctx->stream, dn, dd, count, dq, dr);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(q, dq, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(ref, dr, bytes, hipMemcpyDeviceToHost);
    if (dn) (void)hipFree(dn);
    if (dd) (void)hipFree(dd);
    if (dq) (void)hipFree(dq);
    if (dr) (void)hipFree(dr);
    if (e != hipSuccess) return fail(DSI_ERR_HIP, "div probe failed: %s", hipGetErrorString(e));
    return DSI_OK;
}
#endif  // DSI_TIMING_EXPERIMENTS

}  // extern "C"

/* ---------------------------------------------------------- multi-GPU (RCCL) */
namespace {

// librccl entry points, resolved once (see load_rccl for which copy).
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                                  hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;     // what RCCL itself says about a communicator
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;  // (bench.py prints it: proof of the rank count)
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl* load_rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // The RCCL that belongs to the HIP runtime this process runs on: the librccl next to the loaded
        // libamdhip64 (/opt/rocm/lib, or torch's lib directory when torch brought its own runtime in
        // first).  Opening by soname instead could hand back an RCCL built for another runtime that
        // some other package loaded (seen: torch's librccl on /opt/rocm's libamdhip64 fails in
        // ncclCommInitAll).
        std::vector<std::string> names;
        Dl_info di;
        if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash);
                names.push_back(dir + "/librccl.so.1");
                names.push_back(dir + "/librccl.so");
            }
        }
        names.push_back("/opt/rocm/lib/librccl.so.1");
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        for (const std::string& n : names) {
            r.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.error = std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "?");
            return;
        }
        auto sym = [&](const char* name) -> void* {
            void* p = dlsym(r.lib, name);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.ReduceScatter = reinterpret_cast<decltype(r.ReduceScatter)>(sym("ncclReduceScatter"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
        r.CommCuDevice = reinterpret_cast<decltype(r.CommCuDevice)>(sym("ncclCommCuDevice"));
    });
    return &r;
}

#define RCCL_TRY(R, expr)                                                                          \
    do {                                                                                           \
        ncclResult_t e_ = (expr);                                                                  \
        if (e_ != ncclSuccess)                                                                     \
            return fail(DSI_ERR_COMM, "%s failed: %s (%s:%d)", #expr, (R)->GetErrorString(e_), __FILE__, \
                        __LINE__);                                                                 \
    } while (0)

int rccl_ready(Rccl** out)
{
    Rccl* r = load_rccl();
    if (!r->error.empty()) return fail(DSI_ERR_COMM, "%s", r->error.c_str());
    *out = r;
    return DSI_OK;
}

bool to_nccl_op(int op, ncclRedOp_t* out)
{
    switch (op) {
    case DSI_REDUCE_SUM: *out = ncclSum; return true;
    case DSI_REDUCE_MIN: *out = ncclMin; return true;
    case DSI_REDUCE_MAX: *out = ncclMax; return true;
    default: return false;
    }
}

}  // namespace

struct dsi_comm {
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, size = 1;
};

extern "C" {

int dsi_comm_unique_id(uint8_t id[DSI_COMM_ID_BYTES])
{
    REQUIRE(id, DSI_ERR_INVALID, "id is null");
    static_assert(sizeof(ncclUniqueId) == DSI_COMM_ID_BYTES, "RCCL unique id size");
    Rccl* r = nullptr;
    if (int rc = rccl_ready(&r)) return rc;
    ncclUniqueId u;
    RCCL_TRY(r, r->GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return DSI_OK;
}

int dsi_comm_create_rank(dsi_context_t* ctx, const uint8_t id[DSI_COMM_ID_BYTES], int nranks, int rank,
                         dsi_comm_t** out)
{
    REQUIRE(ctx && id && out, DSI_ERR_INVALID, "null argument");
    *out = nullptr;
    REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, DSI_ERR_INVALID, "rank %d of %d", rank, nranks);
    Rccl* r = nullptr;
    if (int rc = rccl_ready(&r)) return rc;
    if (int rc = set_device(ctx)) return rc;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    dsi_comm* c = new (std::nothrow) dsi_comm();
    REQUIRE(c, DSI_ERR_INVALID, "out of host memory");
    c->device = ctx->device;
    c->rank = rank;
    c->size = nranks;
    const ncclResult_t e = r->CommInitRank(&c->comm, nranks, u, rank);
    if (e != ncclSuccess) {
        delete c;
        return fail(DSI_ERR_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, r->GetErrorString(e));
    }
    *out = c;
    return DSI_OK;
}

int dsi_comm_create_all(dsi_context_t* const* contexts, int n, dsi_comm_t** out)
{
    REQUIRE(contexts && out && n >= 1, DSI_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) {
        REQUIRE(contexts[i], DSI_ERR_INVALID, "context %d is null", i);
        devs[i] = contexts[i]->device;
        for (int j = 0; j < i; ++j)
            REQUIRE(devs[j] != devs[i], DSI_ERR_CONTEXT, "contexts %d and %d share device %d (RCCL: one rank per GPU)",
                    j, i, devs[i]);
    }
    Rccl* r = nullptr;
    if (int rc = rccl_ready(&r)) return rc;
    std::vector<ncclComm_t> comms(n, nullptr);
    RCCL_TRY(r, r->CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        dsi_comm* c = new (std::nothrow) dsi_comm();
        if (!c) {
            for (int j = 0; j < n; ++j) {
                if (comms[j]) (void)r->CommDestroy(comms[j]);
                delete out[j];
                out[j] = nullptr;
            }
            return fail(DSI_ERR_INVALID, "out of host memory");
        }
        c->comm = comms[i];
        c->device = devs[i];
        c->rank = i;
        c->size = n;
        out[i] = c;
    }
    return DSI_OK;
}

int dsi_comm_destroy(dsi_comm_t* c)
{
    if (!c) return DSI_OK;
    Rccl* r = load_rccl();
    if (c->comm && r->CommDestroy) {
        (void)hipSetDevice(c->device);
        (void)r->CommDestroy(c->comm);
    }
    delete c;
    return DSI_OK;
}

int dsi_comm_rank(const dsi_comm_t* c) { return c ? c->rank : -1; }
int dsi_comm_size(const dsi_comm_t* c) { return c ? c->size : 0; }

static int allreduce_check(const dsi_comm_t* c, const dsi_grid_t* g)
{
    REQUIRE(c && g, DSI_ERR_INVALID, "null argument");
    REQUIRE(c->device == g->ctx->device, DSI_ERR_CONTEXT, "grid on device %d, communicator rank on device %d",
            g->ctx->device, c->device);
    return DSI_OK;
}

int dsi_grid_allreduce(dsi_comm_t* c, dsi_grid_t* g, int op)
{
    if (int rc = allreduce_check(c, g)) return rc;
    ncclRedOp_t nop;
    REQUIRE(to_nccl_op(op, &nop), DSI_ERR_BAD_OP, "bad reduce op %d", op);
    Rccl* r = nullptr;
    if (int rc = rccl_ready(&r)) return rc;
    if (int rc = set_device(g->ctx)) return rc;
    RCCL_TRY(r, r->AllReduce(g->data, g->data, g->n, ncclFloat32, nop, c->comm, g->ctx->stream));
    return DSI_OK;
}

int dsi_grid_allreduce_all(dsi_comm_t* const* comms, dsi_grid_t* const* grids, int n, int op)
{
    REQUIRE(comms && grids && n >= 1, DSI_ERR_INVALID, "bad argument");
    ncclRedOp_t nop;
    REQUIRE(to_nccl_op(op, &nop), DSI_ERR_BAD_OP, "bad reduce op %d", op);
    for (int i = 0; i < n; ++i) {
        if (int rc = allreduce_check(comms[i], grids[i])) return rc;
        REQUIRE(same_shape(grids[0], grids[i]), DSI_ERR_SHAPE, "grid %d has another shape", i);
    }
    Rccl* r = nullptr;
    if (int rc = rccl_ready(&r)) return rc;
    RCCL_TRY(r, r->GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(grids[i]->ctx->device);
        const ncclResult_t e = r->AllReduce(grids[i]->data, grids[i]->data, grids[i]->n, ncclFloat32, nop,
                                            comms[i]->comm, grids[i]->ctx->stream);
        if (e != ncclSuccess && bad == ncclSuccess) bad = e;
    }
    const ncclResult_t e2 = r->GroupEnd();
    if (bad != ncclSuccess) return fail(DSI_ERR_COMM, "ncclAllReduce failed: %s", r->GetErrorString(bad));
    if (e2 != ncclSuccess) return fail(DSI_ERR_COMM, "ncclGroupEnd failed: %s", r->GetErrorString(e2));
    return DSI_OK;
}

static int ensure_planes_full_dev(dsi_mapper_t* m)
{
    if (m->planes_full_dev) return DSI_OK;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->planes_full_dev), m->planes_full.size() * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(m->planes_full_dev, m->planes_full.data(), m->planes_full.size() * sizeof(float),
                           hipMemcpyHostToDevice, m->ctx->stream));
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));  // the source is host memory of this object: simplest
    return DSI_OK;
}

// local collapse + key packing of one shard (everything on the mapper's stream)
static int sharded_prepare(dsi_mapper_t* m, dsi_grid_t* g)
{
    REQUIRE(m && g, DSI_ERR_INVALID, "null argument");
    REQUIRE(m->ctx->device == g->ctx->device, DSI_ERR_CONTEXT, "mapper and grid live on different devices");
    REQUIRE(same_shape(m->grid, g), DSI_ERR_SHAPE, "grid shape differs from the mapper's DSI");
    REQUIRE(m->planes_full.size() <= 256, DSI_ERR_INVALID, "arg-max indices are u8");
    if (int rc = set_device(m->ctx)) return rc;
    const size_t npix = (size_t)g->nx * g->ny;
    HIP_TRY(m->conf.reserve(npix));
    HIP_TRY(m->depth.reserve(npix));
    HIP_TRY(m->idx.reserve(npix));
    HIP_TRY(m->argmax_keys.reserve(npix));
    if (int rc = ensure_planes_full_dev(m)) return rc;
    if (int rc = dsi_context_wait_for(m->ctx, g->ctx)) return rc;
    if (int rc = depth_buffers_acquire(m)) return rc;
    HIP_TRY(dsi::launch_collapse_max_z(m->ctx->stream, g->data, g->nx, g->ny, g->nz, m->conf.p, m->idx.p, nullptr,
                                       nullptr));
    HIP_TRY(dsi::launch_pack_argmax(m->ctx->stream, m->conf.p, m->idx.p, (int)npix, m->plane_begin,
                                    m->argmax_keys.p, 0));
    return release_to(g->ctx, m->ctx);  // the collapse READ g on the mapper's stream
}

static int sharded_finish(dsi_mapper_t* m)
{
    const size_t npix = (size_t)m->geom.nx * m->geom.ny;
    HIP_TRY(hipSetDevice(m->ctx->device));
    HIP_TRY(dsi::launch_unpack_argmax(m->ctx->stream, m->argmax_keys.p, (int)npix, m->planes_full_dev, m->conf.p,
                                      m->idx.p, m->depth.p));
    return depth_buffers_ready(m);
}

int dsi_mapper_depth_map_sharded(dsi_mapper_t* m, dsi_grid_t* g, dsi_comm_t* c)
{
    REQUIRE(c, DSI_ERR_INVALID, "communicator is null");
    if (int rc = sharded_prepare(m, g)) return rc;
    REQUIRE(c->device == m->ctx->device, DSI_ERR_CONTEXT, "communicator rank on another device");
    Rccl* r = nullptr;
    if (int rc = rccl_ready(&r)) return rc;
    const size_t npix = (size_t)g->nx * g->ny;
    RCCL_TRY(r, r->AllReduce(m->argmax_keys.p, m->argmax_keys.p, npix, ncclUint64, ncclMax, c->comm,
                             m->ctx->stream));
    return sharded_finish(m);
}

int dsi_mapper_depth_map_sharded_all(dsi_mapper_t* const* ms, dsi_grid_t* const* gs, dsi_comm_t* const* cs, int n)
{
    REQUIRE(ms && gs && cs && n >= 1, DSI_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; ++i) {
        REQUIRE(cs[i], DSI_ERR_INVALID, "communicator %d is null", i);
        if (int rc = sharded_prepare(ms[i], gs[i])) return rc;
        REQUIRE(cs[i]->device == ms[i]->ctx->device, DSI_ERR_CONTEXT, "communicator %d on another device", i);
        REQUIRE(ms[i]->geom.nx == ms[0]->geom.nx && ms[i]->geom.ny == ms[0]->geom.ny, DSI_ERR_SHAPE,
                "mapper %d has another image size", i);
    }
    Rccl* r = nullptr;
    if (int rc = rccl_ready(&r)) return rc;
    const size_t npix = (size_t)ms[0]->geom.nx * ms[0]->geom.ny;
    RCCL_TRY(r, r->GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(ms[i]->ctx->device);
        const ncclResult_t e = r->AllReduce(ms[i]->argmax_keys.p, ms[i]->argmax_keys.p, npix, ncclUint64, ncclMax,
                                            cs[i]->comm, ms[i]->ctx->stream);
        if (e != ncclSuccess && bad == ncclSuccess) bad = e;
    }
    const ncclResult_t e2 = r->GroupEnd();
    if (bad != ncclSuccess) return fail(DSI_ERR_COMM, "ncclAllReduce failed: %s", r->GetErrorString(bad));
    if (e2 != ncclSuccess) return fail(DSI_ERR_COMM, "ncclGroupEnd failed: %s", r->GetErrorString(e2));
    for (int i = 0; i < n; ++i)
        if (int rc = sharded_finish(ms[i])) return rc;
    return DSI_OK;
}

// ---- temporal fusion's last step with a reduce-scatter: rank r of n reduces only planes
// [r q, (r+1) q), q = nz / n, of the accumulator (the nz mod n planes left over are all-reduced: every rank then
// has them), finalises and arg-maxes what it owns, and ONE all-reduce(MAX) of the packed keys gives every rank
// the depth map.  Per rank (n-1)/n x S bytes cross xGMI instead of the all-reduce's 2 (n-1)/n x S.
static int scattered_check(dsi_mapper_t* m, dsi_grid_t* acc, dsi_comm_t* c, int mode)
{
    REQUIRE(m && acc && c, DSI_ERR_INVALID, "null argument");
    REQUIRE(valid_acc_mode(mode), DSI_ERR_BAD_OP, "bad accumulate mode %d", mode);
    REQUIRE(m->ctx->device == acc->ctx->device && c->device == acc->ctx->device, DSI_ERR_CONTEXT,
            "mapper, accumulator and communicator rank must live on one device");
    REQUIRE(same_shape(m->grid, acc), DSI_ERR_SHAPE, "accumulator shape differs from the mapper's DSI");
    REQUIRE(m->plane_begin == 0 && (int)m->planes_full.size() == acc->nz, DSI_ERR_INVALID,
            "the mapper must own the whole depth vector (time slices shard by time, not by plane)");
    REQUIRE(acc->nz <= 256, DSI_ERR_INVALID, "arg-max indices are u8: dimZ must be <= 256 (got %d)", acc->nz);
    return DSI_OK;
}

// the two collectives on the accumulator (callers bracket several ranks with ncclGroupStart / End); which planes go
// where is dsi::host::scatter_plan's business (shared with the host-staged tests)
static ncclResult_t scattered_reduce(Rccl* r, dsi_grid_t* acc, dsi_comm_t* c, ncclRedOp_t nop)
{
    const size_t plane = (size_t)acc->nx * acc->ny;
    dsi::host::ScatterPlan sp;
    if (!dsi::host::scatter_plan(acc->nz, c->size, c->rank, &sp)) return ncclInvalidArgument;
    ncclResult_t e = ncclSuccess;
    if (sp.q > 0)  // in place: the receive buffer is this rank's stretch of the send buffer
        e = r->ReduceScatter(acc->data, acc->data + (size_t)sp.own_begin * plane, (size_t)sp.q * plane, ncclFloat32, nop, c->comm,
                             acc->ctx->stream);
    if (e == ncclSuccess && sp.tail_count > 0) {
        float* tail = acc->data + (size_t)sp.tail_begin * plane;
        e = r->AllReduce(tail, tail, (size_t)sp.tail_count * plane, ncclFloat32, nop, c->comm, acc->ctx->stream);
    }
    return e;
}

// finalize + arg-max of the planes rank `rank` of `nranks` owns -> packed keys (on the mapper's stream)
static int scattered_local(dsi_mapper_t* m, dsi_grid_t* acc, int nranks, int rank, int mode, int n_maps)
{
    if (int rc = set_device(m->ctx)) return rc;
    const size_t plane = (size_t)acc->nx * acc->ny;
    dsi::host::ScatterPlan sp;
    REQUIRE(dsi::host::scatter_plan(acc->nz, nranks, rank, &sp), DSI_ERR_INVALID, "rank %d of %d", rank, nranks);
    HIP_TRY(m->conf.reserve(plane));
    HIP_TRY(m->depth.reserve(plane));
    HIP_TRY(m->idx.reserve(plane));
    HIP_TRY(m->argmax_keys.reserve(plane));
    if (int rc = ensure_planes_full_dev(m)) return rc;
    if (int rc = dsi_context_wait_for(m->ctx, acc->ctx)) return rc;
    if (int rc = depth_buffers_acquire(m)) return rc;
    hipStream_t st = m->ctx->stream;
    const int begin[2] = {sp.own_begin, sp.tail_begin}, count[2] = {sp.own_count, sp.tail_count};
    bool first = true;
    for (int k = 0; k < 2; ++k) {
        if (count[k] <= 0) continue;
        float* p = acc->data + (size_t)begin[k] * plane;
        HIP_TRY(dsi::launch_finalize(st, p, (size_t)count[k] * plane, mode, n_maps));
        HIP_TRY(dsi::launch_collapse_max_z(st, p, acc->nx, acc->ny, count[k], m->conf.p, m->idx.p, nullptr, nullptr));
        HIP_TRY(dsi::launch_pack_argmax(st, m->conf.p, m->idx.p, (int)plane, begin[k], m->argmax_keys.p, first ? 0 : 1));
        first = false;
    }
    if (first) HIP_TRY(hipMemsetAsync(m->argmax_keys.p, 0, plane * sizeof(unsigned long long), st));  // owns no plane
    // finalize WROTE the accumulator on the mapper's stream: whatever the accumulator's context queues next (the
    // next round's accumulateBegin) must come after it
    return release_to(acc->ctx, m->ctx);
}

int dsi_mapper_depth_map_reduce_scattered(dsi_mapper_t* m, dsi_grid_t* acc, dsi_comm_t* c, int mode, int n_maps)
{
    if (int rc = scattered_check(m, acc, c, mode)) return rc;
    Rccl* r = nullptr;
    if (int rc = rccl_ready(&r)) return rc;
    ncclRedOp_t nop;
    REQUIRE(to_nccl_op(dsi_acc_reduce_op(mode), &nop), DSI_ERR_BAD_OP, "mode %d has no reduce op", mode);
    if (int rc = set_device(acc->ctx)) return rc;
    RCCL_TRY(r, r->GroupStart());
    const ncclResult_t e = scattered_reduce(r, acc, c, nop);
    const ncclResult_t e2 = r->GroupEnd();
    if (e != ncclSuccess) return fail(DSI_ERR_COMM, "reduce-scatter failed: %s", r->GetErrorString(e));
    if (e2 != ncclSuccess) return fail(DSI_ERR_COMM, "ncclGroupEnd failed: %s", r->GetErrorString(e2));
    if (int rc = scattered_local(m, acc, c->size, c->rank, mode, n_maps)) return rc;
    const size_t npix = (size_t)acc->nx * acc->ny;
    RCCL_TRY(r, r->AllReduce(m->argmax_keys.p, m->argmax_keys.p, npix, ncclUint64, ncclMax, c->comm, m->ctx->stream));
    return sharded_finish(m);
}

int dsi_mapper_depth_map_reduce_scattered_all(dsi_mapper_t* const* ms, dsi_grid_t* const* accs, dsi_comm_t* const* cs, int n,
                                              int mode, int n_maps)
{
    REQUIRE(ms && accs && cs && n >= 1, DSI_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; ++i) {
        if (int rc = scattered_check(ms[i], accs[i], cs[i], mode)) return rc;
        REQUIRE(same_shape(accs[0], accs[i]), DSI_ERR_SHAPE, "accumulator %d has another shape", i);
    }
    Rccl* r = nullptr;
    if (int rc = rccl_ready(&r)) return rc;
    ncclRedOp_t nop;
    REQUIRE(to_nccl_op(dsi_acc_reduce_op(mode), &nop), DSI_ERR_BAD_OP, "mode %d has no reduce op", mode);
    RCCL_TRY(r, r->GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(accs[i]->ctx->device);
        const ncclResult_t e = scattered_reduce(r, accs[i], cs[i], nop);
        if (e != ncclSuccess && bad == ncclSuccess) bad = e;
    }
    ncclResult_t e2 = r->GroupEnd();
    if (bad != ncclSuccess) return fail(DSI_ERR_COMM, "reduce-scatter failed: %s", r->GetErrorString(bad));
    if (e2 != ncclSuccess) return fail(DSI_ERR_COMM, "ncclGroupEnd failed: %s", r->GetErrorString(e2));
    for (int i = 0; i < n; ++i)
        if (int rc = scattered_local(ms[i], accs[i], cs[i]->size, cs[i]->rank, mode, n_maps)) return rc;
    const size_t npix = (size_t)accs[0]->nx * accs[0]->ny;
    RCCL_TRY(r, r->GroupStart());
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(ms[i]->ctx->device);
        const ncclResult_t e = r->AllReduce(ms[i]->argmax_keys.p, ms[i]->argmax_keys.p, npix, ncclUint64, ncclMax, cs[i]->comm,
                                            ms[i]->ctx->stream);
        if (e != ncclSuccess && bad == ncclSuccess) bad = e;
    }
    e2 = r->GroupEnd();
    if (bad != ncclSuccess) return fail(DSI_ERR_COMM, "ncclAllReduce failed: %s", r->GetErrorString(bad));
    if (e2 != ncclSuccess) return fail(DSI_ERR_COMM, "ncclGroupEnd failed: %s", r->GetErrorString(e2));
    for (int i = 0; i < n; ++i)
        if (int rc = sharded_finish(ms[i])) return rc;
    return DSI_OK;
}

/* ---- the same three steps with the transport left to the caller (host-staged tests, other transports) ---- */
int dsi_scatter_plan(int nz, int nranks, int rank, dsi_scatter_plan_t* out)
{
    REQUIRE(out, DSI_ERR_INVALID, "out is null");
    dsi::host::ScatterPlan sp;
    REQUIRE(dsi::host::scatter_plan(nz, nranks, rank, &sp), DSI_ERR_INVALID, "bad partition: %d planes, rank %d of %d", nz, rank,
            nranks);
    out->q = sp.q;
    out->own_begin = sp.own_begin;
    out->own_count = sp.own_count;
    out->tail_begin = sp.tail_begin;
    out->tail_count = sp.tail_count;
    return DSI_OK;
}

int dsi_plane_range(int nz, int nranks, int rank, int* begin, int* count)
{
    REQUIRE(begin && count, DSI_ERR_INVALID, "null argument");
    REQUIRE(dsi::host::plane_range(nz, nranks, rank, begin, count), DSI_ERR_INVALID, "bad partition: %d planes, rank %d of %d",
            nz, rank, nranks);
    return DSI_OK;
}

int dsi_argmax_keys_pack(const float* conf, const uint8_t* idx_local, size_t n, int plane_begin, uint64_t* keys)
{
    REQUIRE(n == 0 || (conf && idx_local && keys), DSI_ERR_INVALID, "null argument");
    REQUIRE(plane_begin >= 0 && plane_begin <= 255, DSI_ERR_INVALID, "plane_begin %d outside 0..255", plane_begin);
    for (size_t i = 0; i < n; ++i) {
        const int gp = (int)idx_local[i] + plane_begin;
        REQUIRE(gp <= 255, DSI_ERR_INVALID, "global plane index %d does not fit the key's 8 bits", gp);
        keys[i] = dsi::host::argmax_key(conf[i], gp);
    }
    return DSI_OK;
}

int dsi_argmax_keys_unpack(const uint64_t* keys, size_t n, float* conf, uint8_t* idx)
{
    REQUIRE(n == 0 || (keys && conf && idx), DSI_ERR_INVALID, "null argument");
    for (size_t i = 0; i < n; ++i) {
        int gp = 0;
        dsi::host::argmax_unkey(keys[i], &conf[i], &gp);
        idx[i] = (uint8_t)gp;
    }
    return DSI_OK;
}

int dsi_mapper_depth_map_scattered_local(dsi_mapper_t* m, dsi_grid_t* acc, int nranks, int rank, int mode, int n_maps)
{
    REQUIRE(m && acc, DSI_ERR_INVALID, "null argument");
    REQUIRE(valid_acc_mode(mode), DSI_ERR_BAD_OP, "bad accumulate mode %d", mode);
    REQUIRE(m->ctx->device == acc->ctx->device, DSI_ERR_CONTEXT, "mapper and accumulator must live on one device");
    REQUIRE(same_shape(m->grid, acc), DSI_ERR_SHAPE, "accumulator shape differs from the mapper's DSI");
    REQUIRE(m->plane_begin == 0 && (int)m->planes_full.size() == acc->nz, DSI_ERR_INVALID,
            "the mapper must own the whole depth vector (time slices shard by time, not by plane)");
    REQUIRE(acc->nz <= 256, DSI_ERR_INVALID, "arg-max indices are u8: dimZ must be <= 256 (got %d)", acc->nz);
    REQUIRE(n_maps >= 1, DSI_ERR_INVALID, "number of maps must be >= 1 (got %d)", n_maps);
    return scattered_local(m, acc, nranks, rank, mode, n_maps);
}

int dsi_mapper_argmax_keys_download(dsi_mapper_t* m, uint64_t* host)
{
    REQUIRE(m && host, DSI_ERR_INVALID, "null argument");
    const size_t npix = (size_t)m->geom.nx * m->geom.ny;
    REQUIRE(m->argmax_keys.cap >= npix, DSI_ERR_INVALID, "this mapper holds no arg-max keys");
    if (int rc = set_device(m->ctx)) return rc;
    HIP_TRY(hipMemcpyAsync(host, m->argmax_keys.p, npix * sizeof(uint64_t), hipMemcpyDeviceToHost, m->ctx->stream));
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    return DSI_OK;
}

int dsi_mapper_argmax_keys_upload(dsi_mapper_t* m, const uint64_t* host)
{
    REQUIRE(m && host, DSI_ERR_INVALID, "null argument");
    const size_t npix = (size_t)m->geom.nx * m->geom.ny;
    if (int rc = set_device(m->ctx)) return rc;
    HIP_TRY(m->argmax_keys.reserve(npix));
    HIP_TRY(hipMemcpyAsync(m->argmax_keys.p, host, npix * sizeof(uint64_t), hipMemcpyHostToDevice, m->ctx->stream));
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));  // the source is pageable host memory
    return DSI_OK;
}

int dsi_mapper_depth_map_from_keys(dsi_mapper_t* m)
{
    REQUIRE(m, DSI_ERR_INVALID, "mapper is null");
    const size_t npix = (size_t)m->geom.nx * m->geom.ny;
    REQUIRE(m->argmax_keys.cap >= npix, DSI_ERR_INVALID, "this mapper holds no arg-max keys");
    if (int rc = set_device(m->ctx)) return rc;
    HIP_TRY(m->conf.reserve(npix));
    HIP_TRY(m->depth.reserve(npix));
    HIP_TRY(m->idx.reserve(npix));
    if (int rc = ensure_planes_full_dev(m)) return rc;
    if (int rc = depth_buffers_acquire(m)) return rc;
    return sharded_finish(m);
}

int dsi_comm_query(const dsi_comm_t* c, int* nranks, int* rank, int* device)
{
    REQUIRE(c && c->comm, DSI_ERR_INVALID, "communicator is null");
    Rccl* r = nullptr;
    if (int rc = rccl_ready(&r)) return rc;
    REQUIRE(r->CommCount && r->CommUserRank && r->CommCuDevice, DSI_ERR_COMM, "librccl lacks the communicator queries");
    int v = 0;
    if (nranks) {
        RCCL_TRY(r, r->CommCount(c->comm, &v));
        *nranks = v;
    }
    if (rank) {
        RCCL_TRY(r, r->CommUserRank(c->comm, &v));
        *rank = v;
    }
    if (device) {
        RCCL_TRY(r, r->CommCuDevice(c->comm, &v));
        *device = v;
    }
    return DSI_OK;
}

}  // extern "C"
