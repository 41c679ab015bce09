// dsi_process.hpp -- host orchestration of the reference's Alg. 1 / Alg. 2 over the engine, in C++,
// with the reference's function names and argument order (minus ROS / OpenCV types and the file
// output):
//
//   process_1   mapper_emvs_stereo/src/process1.cpp:28-224   one DSI per camera, camera fusion
//   process_2_exact_depth_map                                Alg. 2's fused DSI + an arg-max whose index map equals the CPU reference's on every pixel
//   process_1_exact_depth_map                                process_1 + arg-max whose index map equals the CPU reference's on every pixel
//   process_1_depth_map                                      the same + the arg-max of getDepthMapFromDSI, without
//                                                            writing any DSI (one fused kernel)
//   dsi::full_sequence_depth_maps   main.cpp:177-302          the --full_seq loop (process_method 1) as a stream of
//                                                            windows, `depth` of them in flight, one fused kernel each
//   process_2   mapper_emvs_stereo/src/process2.cpp:28-302   sub-intervals: camera fusion then
//                                                            temporal fusion, and the converse order
//   process_5   mapper_emvs_stereo/src/process5.cpp:28-260   process_2 with the right camera's
//                                                            sub-intervals circularly shifted
//   process_2_multi_gpu                                      process_2's temporal fusion over the GPUs of
//                                                            one node: sub-interval -> device, ONE RCCL
//                                                            all-reduce per accumulator
//
// Pure call sequencing: which mapper gets which events, where the reference view sits, the fusion
// order and op codes (1 min, 2 HM, 3 GM, 4 AM, 5 RMS, 6 max), the temporal accumulators
// (2 harmonic, 4 arithmetic; other codes do nothing, like the reference).  Every voxel operation is
// an engine kernel behind include/dsi_engine.h.
#ifndef DSI_PROCESS_HPP
#define DSI_PROCESS_HPP

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "dsi_engine.hpp"

namespace dsi {

// (T_a * T_b and inverse(T) live in dsi_engine.hpp, next to dsi::Transformation)

inline void fuseTwoGrids(Grid3D& dst, const Grid3D& g, int method, const char* what)
{
    switch (method) {  // process1.cpp:136-158
    case 1: dst.minTwoGrids(g); break;
    case 2: dst.harmonicMeanTwoGrids(g); break;
    case 3: dst.geometricMeanTwoGrids(g); break;
    case 4: dst.arithmeticMeanTwoGrids(g); break;
    case 5: dst.rmsTwoGrids(g); break;
    case 6: dst.maxTwoGrids(g); break;
    default: throw Error(DSI_ERR_BAD_OP, what);
    }
}

}  // namespace dsi

// Alg. 1.  A camera with no events (events2.empty(), the stereo case) is skipped like
// process1.cpp:105.  rv_pos: position of the reference view along the baseline (process1.cpp:58-66;
// a flag in the reference).  Returns T_rv_w; the fused DSI is mapper_fused.dsi_.
inline dsi::Transformation process_1(const LinearTrajectory& trajectory0, const LinearTrajectory& trajectory1,
                                     const LinearTrajectory& trajectory2, const std::vector<dsi::Event>& events0,
                                     const std::vector<dsi::Event>& events1, const std::vector<dsi::Event>& events2,
                                     EMVS::MapperEMVS& mapper_fused, EMVS::MapperEMVS& mapper0,
                                     EMVS::MapperEMVS& mapper1, EMVS::MapperEMVS& mapper2, double ts,
                                     int fusion_method, double rv_pos = 0.0)
{
    dsi::Transformation T_w_l;
    if (!trajectory0.getPoseAt(ts, T_w_l)) throw dsi::Error(DSI_ERR_INVALID, "no pose at the reference timestamp");
    dsi::Transformation baseline;
    baseline.t[0] = rv_pos;
    const dsi::Transformation T_rv_w = dsi::inverse(T_w_l * baseline);  // :56-68
    mapper0.evaluateDSI(events0, trajectory0, T_rv_w);                  // :76
    mapper1.evaluateDSI(events1, trajectory1, T_rv_w);                  // :94
    if (!events2.empty()) mapper2.evaluateDSI(events2, trajectory2, T_rv_w);  // :105-117
    mapper_fused.dsi_.resetGrid();                                      // :126
    mapper_fused.dsi_.addTwoGrids(mapper0.dsi_);                        // :127
    dsi::fuseTwoGrids(mapper_fused.dsi_, mapper1.dsi_, fusion_method, "Improper fusion method selected");
    if (!events2.empty()) {                                             // :169-191
        if (fusion_method == 1) mapper_fused.dsi_.minTwoGrids(mapper2.dsi_);
        else if (fusion_method == 2) mapper_fused.dsi_.harmonicMeanTwoGrids(mapper2.dsi_, 3);
        else if (fusion_method == 6) mapper_fused.dsi_.maxTwoGrids(mapper2.dsi_);
        // 3, 4, 5: the reference silently ignores the third camera
    }
    return T_rv_w;
}

// Alg. 1 for callers that keep only the depth map (the --full_seq loop, main.cpp:177-302: process_1 followed by
// getDepthMapFromDSI's arg-max): the same result as process_1(...) + mapper_fused.getDepthMapFromDSI(depth_map,
// confidence_map, depth_cell_indices), bit for bit, but through ONE kernel that votes the cameras band by band in LDS,
// fuses them per voxel and keeps the running arg-max on the CU (dsi_mapper_depth_map_of_events): no DSI is written.
// The mappers supply the cameras' geometry and scratch; their dsi_ members are NOT updated.  n cameras (1..3); n = 4 (round
// 6): the synthetic four-camera rig of BASELINE configs[4], for which process_1 has no switch -- fusion_method must be 3
// (geometric mean) and the cameras are fused by the balanced tree of Grid3D::geometricMeanTwoGrids
// (cartesian3dgrid.h:150-156; dsi_mapper_depth_map_of_events_n, DSI_ACC_GM_TREE).
inline dsi::Transformation process_1_depth_map_n(const LinearTrajectory* const* trs, const std::vector<dsi::Event>* const* evs,
                                                 EMVS::MapperEMVS* const* mappers, int n, EMVS::MapperEMVS& mapper_out, double ts,
                                                 int fusion_method, dsi::Image<float>& depth_map,
                                                 dsi::Image<float>& confidence_map, dsi::Image<uint8_t>& depth_cell_indices,
                                                 double rv_pos = 0.0)
{
    if (n < 1 || n > 4) throw dsi::Error(DSI_ERR_INVALID, "process_1_depth_map: 1 to 4 cameras");
    if (n == 4 && fusion_method != DSI_FUSE_GM)
        throw dsi::Error(DSI_ERR_BAD_OP, "process_1_depth_map: four cameras are fused by the geometric-mean tree (fusion_method 3)");
    dsi::Transformation T_w_l;
    if (!trs[0]->getPoseAt(ts, T_w_l)) throw dsi::Error(DSI_ERR_INVALID, "no pose at the reference timestamp");
    dsi::Transformation baseline;
    baseline.t[0] = rv_pos;
    const dsi::Transformation T_rv_w = dsi::inverse(T_w_l * baseline);  // process1.cpp:56-68
    double T7[7];
    T_rv_w.to7(T7);
    dsi_mapper_t* ms[4] = {nullptr, nullptr, nullptr, nullptr};
    dsi_batch_t* bs[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<uint16_t> xs, ys;
    std::vector<uint32_t> first;
    std::vector<float> Rt;
    dsi_context_t* ctx = mapper_out.context();
    try {
        for (int c = 0; c < n; ++c) {
            ms[c] = mappers[c]->handle();
            const size_t ne = evs[c]->size();
            xs.resize(ne);
            ys.resize(ne);
            for (size_t i = 0; i < ne; ++i) {
                xs[i] = (*evs[c])[i].x;
                ys[i] = (*evs[c])[i].y;
            }
            first.assign(ne / DSI_PACKET_SIZE + 1, 0u);
            Rt.assign(12 * first.size(), 0.f);
            size_t np = 0;
            const int rc = dsi_packetize_strided(ne ? &(*evs[c])[0].ts : nullptr, sizeof(dsi::Event), ne, trs[c]->times().data(),
                                                 trs[c]->poses7().data(), trs[c]->times().size(), T7, first.data(), Rt.data(), &np);
            if (rc == DSI_ERR_TOO_FEW_EVENTS) np = 0;  // evaluateDSI returns false: an all-zero DSI (mapper_emvs_stereo.cpp:71-75)
            else dsi::check(rc);
            dsi::check(dsi_batch_create(ctx, xs.data(), ys.data(), ne, first.data(), Rt.data(), np, &bs[c]));
        }
        if (n == 4)
            dsi::check(dsi_mapper_depth_map_of_events_n(mapper_out.handle(), ms, bs, n, DSI_ACC_GM_TREE));
        else
            dsi::check(dsi_mapper_depth_map_of_events(mapper_out.handle(), ms, bs, n, fusion_method));
        int nx, ny, nz;
        mapper_out.dsi_.getDimensions(&nx, &ny, &nz);
        depth_map = dsi::Image<float>(ny, nx);
        confidence_map = dsi::Image<float>(ny, nx);
        depth_cell_indices = dsi::Image<uint8_t>(ny, nx);
        dsi::check(dsi_mapper_fetch_depth_map(mapper_out.handle(), depth_map.data.data(), confidence_map.data.data(),
                                              depth_cell_indices.data.data()));
    } catch (...) {
        for (dsi_batch_t* b : bs) dsi_batch_destroy(b);
        throw;
    }
    for (dsi_batch_t* b : bs) dsi_batch_destroy(b);
    return T_rv_w;
}

// ALL planes of the listed columns (pixels y * dimX + x) re-summed per camera in the reference's order
// (dsi_mapper_exact_voxels), fused with the reference's scalar op (dsi_reference_fuse2; one camera: as they are), the first
// maximum taken (cartesian3dgrid.cpp:132-134) and out's depth map patched: those columns are then the reference's by
// construction.  For the FEW columns dsi_mapper_prove_near_ties cannot settle (planes x columns voxels per camera).
inline void resolve_columns_fully(dsi_mapper_t* out, dsi_mapper_t* const* ms, dsi_batch_t* const* bs, int n, int fusion_method,
                                  const std::vector<uint32_t>& pixels, int nx, int ny, int nz)
{
    if (pixels.empty()) return;
    const size_t npix = (size_t)nx * ny, ncol = pixels.size();
    std::vector<uint32_t> vox(ncol * (size_t)nz);
    for (size_t i = 0; i < ncol; ++i)
        for (int z = 0; z < nz; ++z) vox[i * (size_t)nz + (size_t)z] = (uint32_t)((size_t)z * npix + pixels[i]);
    std::vector<float> val[2], fused(vox.size());
    for (int c = 0; c < n; ++c) {
        val[c].resize(vox.size());
        dsi::check(dsi_mapper_exact_voxels(ms[c], bs[c], vox.data(), vox.size(), val[c].data(), nullptr));
    }
    if (n == 2) dsi::check(dsi_reference_fuse2(fusion_method, val[0].data(), val[1].data(), vox.size(), fused.data()));
    else fused = val[0];
    std::vector<uint8_t> idx(ncol);
    std::vector<float> conf(ncol);
    for (size_t i = 0; i < ncol; ++i) {
        size_t best = 0;
        for (size_t z = 1; z < (size_t)nz; ++z)
            if (fused[i * (size_t)nz + best] < fused[i * (size_t)nz + z]) best = z;  // the first maximum
        idx[i] = (uint8_t)best;
        conf[i] = fused[i * (size_t)nz + best];
    }
    dsi::check(dsi_mapper_patch_depth_map(out, pixels.data(), idx.data(), conf.data(), ncol));
}

// Alg. 1 + the arg-max of getDepthMapFromDSI with the EXACT TIE RESOLVER (dsi_mapper_resolve_near_ties): what
// process_1(...) + mapper_fused.getDepthMapFromDSI(depth_map, confidence_map, depth_cell_indices) gives, except that the
// plane index map equals the CPU reference's on EVERY pixel: the engine sums a voxel's votes exactly and rounds once, the
// reference adds them in fp32 in event order (cartesian3dgrid.h:261-270), so the first-maximum plane
// (cartesian3dgrid.cpp:132-134) can differ where a column's best planes are closer than that rounding; those columns'
// contending voxels are re-summed in the reference's order.  Two cameras; the camera DSIs and the fused DSI are written
// (mapper0.dsi_, mapper1.dsi_, mapper_fused.dsi_) like process_1 does.  info (optional): the resolver's statistics.
// proof (optional, ABI 10): PROVEN mode -- after the resolver, dsi_mapper_prove_near_ties counts every voxel's votes and checks
// with rigorous bounds of the reference's fp32 event-order sums that no plane outside the re-summed gap can be the reference's
// first maximum; where columns' bounds ask for a moderately wider gap the resolver runs again with it (and the proof again);
// the columns that remain (maxima made of a handful of tiny weights) are re-summed on ALL their planes (resolve_columns_fully).
// *proof holds the last pass: columns_unproven - columns_resolved_fully == 0 means the index map is the reference's on every
// pixel BY PROOF.  A verification pass (two global-atomic vote counts): tens of milliseconds at 10 M events per camera.
inline dsi::Transformation process_1_exact_depth_map(const LinearTrajectory& trajectory0, const LinearTrajectory& trajectory1,
                                                     const std::vector<dsi::Event>& events0, const std::vector<dsi::Event>& events1,
                                                     EMVS::MapperEMVS& mapper_fused, EMVS::MapperEMVS& mapper0,
                                                     EMVS::MapperEMVS& mapper1, double ts, int fusion_method,
                                                     dsi::Image<float>& depth_map, dsi::Image<float>& confidence_map,
                                                     dsi::Image<uint8_t>& depth_cell_indices, dsi_resolve_info_t* info = nullptr,
                                                     double rv_pos = 0.0, dsi_prove_info_t* proof = nullptr)
{
    dsi::Transformation T_w_l;
    if (!trajectory0.getPoseAt(ts, T_w_l)) throw dsi::Error(DSI_ERR_INVALID, "no pose at the reference timestamp");
    dsi::Transformation baseline;
    baseline.t[0] = rv_pos;
    const dsi::Transformation T_rv_w = dsi::inverse(T_w_l * baseline);  // process1.cpp:56-68
    double T7[7];
    T_rv_w.to7(T7);
    const LinearTrajectory* trs[2] = {&trajectory0, &trajectory1};
    const std::vector<dsi::Event>* evs[2] = {&events0, &events1};
    dsi_mapper_t* ms[2] = {mapper0.handle(), mapper1.handle()};
    dsi_batch_t* bs[2] = {nullptr, nullptr};
    dsi_context_t* ctx = mapper_fused.context();
    auto release = [&]() {
        for (dsi_batch_t* b : bs) dsi_batch_destroy(b);
    };
    try {
        std::vector<uint16_t> xs, ys;
        std::vector<uint32_t> first;
        std::vector<float> Rt;
        for (int c = 0; c < 2; ++c) {
            const size_t ne = evs[c]->size();
            xs.resize(ne);
            ys.resize(ne);
            for (size_t i = 0; i < ne; ++i) {
                xs[i] = (*evs[c])[i].x;
                ys[i] = (*evs[c])[i].y;
            }
            first.assign(ne / DSI_PACKET_SIZE + 1, 0u);
            Rt.assign(12 * first.size(), 0.f);
            size_t np = 0;
            const int rc = dsi_packetize_strided(ne ? &(*evs[c])[0].ts : nullptr, sizeof(dsi::Event), ne, trs[c]->times().data(),
                                                 trs[c]->poses7().data(), trs[c]->times().size(), T7, first.data(), Rt.data(), &np);
            if (rc == DSI_ERR_TOO_FEW_EVENTS) np = 0;  // evaluateDSI returns false (mapper_emvs_stereo.cpp:71-75)
            else dsi::check(rc);
            dsi::check(dsi_batch_create(ctx, xs.data(), ys.data(), ne, first.data(), Rt.data(), np, &bs[c]));
            dsi::check(dsi_mapper_evaluate_batch(ms[c], bs[c]));  // process1.cpp:76, :94
        }
        mapper_fused.dsi_.resetGrid();                 // :126
        mapper_fused.dsi_.addTwoGrids(mapper0.dsi_);   // :127
        dsi::fuseTwoGrids(mapper_fused.dsi_, mapper1.dsi_, fusion_method, "Improper fusion method selected");
        dsi::check(dsi_mapper_depth_map_of(mapper_fused.handle(), mapper_fused.dsi_.handle()));  // :222 -> collapseMaxZSlice
        dsi_resolve_info_t local{};
        dsi_resolve_info_t* ri = info ? info : &local;
        dsi::check(dsi_mapper_resolve_near_ties(mapper_fused.handle(), ms, bs, 2, fusion_method, ri));
        if (proof) {
            auto prove = [&]() {
                *proof = dsi_prove_info_t{};
                proof->rel_gap = ri->rel_gap;  // (the gap the resolver ended with: it may have widened its own)
                dsi::check(dsi_mapper_prove_near_ties(mapper_fused.handle(), ms, bs, 2, fusion_method, proof));
            };
            auto unproven = [&](std::vector<uint32_t>& pix, std::vector<float>& gaps) {
                size_t m = 0;
                dsi::check(dsi_mapper_proof_unproven(mapper_fused.handle(), nullptr, nullptr, 0, &m));
                pix.resize(m);
                gaps.resize(m);
                if (m) dsi::check(dsi_mapper_proof_unproven(mapper_fused.handle(), pix.data(), gaps.data(), m, &m));
            };
            prove();
            if (proof->columns_unproven) {
                // columns whose bounds reach beyond the gap: one more resolver pass if a moderate gap (<= 16 x the default) covers
                // some of them; what remains -- maxima made of a handful of tiny weights -- is re-summed on ALL its planes
                std::vector<uint32_t> pix;
                std::vector<float> gaps;
                unproven(pix, gaps);
                const double max_gap = 4e-3;
                double moderate = 0.0;
                for (float gp : gaps)
                    if ((double)gp * 1.05 <= max_gap) moderate = std::max(moderate, (double)gp * 1.05);
                if (moderate > 0.0) {
                    *ri = dsi_resolve_info_t{};
                    ri->rel_gap = (float)moderate;
                    dsi::check(dsi_mapper_resolve_near_ties(mapper_fused.handle(), ms, bs, 2, fusion_method, ri));
                    prove();
                    if (proof->columns_unproven) unproven(pix, gaps);
                    else pix.clear();
                }
                if (!pix.empty() && pix.size() <= 4096) {
                    int gx, gy, gz;
                    mapper_fused.dsi_.getDimensions(&gx, &gy, &gz);
                    resolve_columns_fully(mapper_fused.handle(), ms, bs, 2, fusion_method, pix, gx, gy, gz);
                    proof->columns_resolved_fully = (long long)pix.size();
                }
            }
        }
        int nx, ny, nz;
        mapper_fused.dsi_.getDimensions(&nx, &ny, &nz);
        depth_map = dsi::Image<float>(ny, nx);
        confidence_map = dsi::Image<float>(ny, nx);
        depth_cell_indices = dsi::Image<uint8_t>(ny, nx);
        dsi::check(dsi_mapper_fetch_depth_map(mapper_fused.handle(), depth_map.data.data(), confidence_map.data.data(),
                                              depth_cell_indices.data.data()));
    } catch (...) {
        release();
        throw;
    }
    release();
    return T_rv_w;
}

// two cameras (process1.cpp:76-166)
inline dsi::Transformation process_1_depth_map(const LinearTrajectory& trajectory0, const LinearTrajectory& trajectory1,
                                               const std::vector<dsi::Event>& events0, const std::vector<dsi::Event>& events1,
                                               EMVS::MapperEMVS& mapper_out, EMVS::MapperEMVS& mapper0,
                                               EMVS::MapperEMVS& mapper1, double ts, int fusion_method,
                                               dsi::Image<float>& depth_map, dsi::Image<float>& confidence_map,
                                               dsi::Image<uint8_t>& depth_cell_indices, double rv_pos = 0.0)
{
    const LinearTrajectory* trs[2] = {&trajectory0, &trajectory1};
    const std::vector<dsi::Event>* evs[2] = {&events0, &events1};
    EMVS::MapperEMVS* ms[2] = {&mapper0, &mapper1};
    return process_1_depth_map_n(trs, evs, ms, 2, mapper_out, ts, fusion_method, depth_map, confidence_map, depth_cell_indices,
                                 rv_pos);
}

// process_1's own argument order with the third camera (process1.cpp:28-41; EVIMO2): events2 empty = two cameras
// (:105, :169), otherwise the third camera enters through min / harmonicMeanTwoGrids(g, 3) / max (:169-191) and is
// ignored by fusion methods 3, 4, 5 like in the reference
inline dsi::Transformation process_1_depth_map(const LinearTrajectory& trajectory0, const LinearTrajectory& trajectory1,
                                               const LinearTrajectory& trajectory2, const std::vector<dsi::Event>& events0,
                                               const std::vector<dsi::Event>& events1, const std::vector<dsi::Event>& events2,
                                               EMVS::MapperEMVS& mapper_out, EMVS::MapperEMVS& mapper0,
                                               EMVS::MapperEMVS& mapper1, EMVS::MapperEMVS& mapper2, double ts,
                                               int fusion_method, dsi::Image<float>& depth_map,
                                               dsi::Image<float>& confidence_map, dsi::Image<uint8_t>& depth_cell_indices,
                                               double rv_pos = 0.0)
{
    const LinearTrajectory* trs[3] = {&trajectory0, &trajectory1, &trajectory2};
    const std::vector<dsi::Event>* evs[3] = {&events0, &events1, &events2};
    EMVS::MapperEMVS* ms[3] = {&mapper0, &mapper1, &mapper2};
    return process_1_depth_map_n(trs, evs, ms, events2.empty() ? 2 : 3, mapper_out, ts, fusion_method, depth_map,
                                 confidence_map, depth_cell_indices, rv_pos);
}

// ---------------------------------------------------------------------------------------------------------------------
// The --full_seq loop of main.cpp:177-302 with process_method 1, for callers that keep the windows' depth maps, as a
// STREAM: for (t = start; t + duration <= stop; t += out_skip) { events of [t, t + duration] of both cameras; reference
// view at t + duration (forward_looking) or at the middle; process_1; arg-max of getDepthMapFromDSI }.  The reference
// builds fresh mappers per window (main.cpp:262-275) and runs the windows one after the other; here `depth` windows are
// in flight, each in its own context (HIP streams) with its own mappers and page-locked result buffers: window w+1's
// uploads, packet sort and tables and the first workgroups of its voting kernel run while window w's kernel drains and
// its depth map travels to the host.  Per window ONE kernel votes both cameras band by band in LDS, fuses them and keeps
// the running arg-max (dsi_mapper_depth_map_of_events): the depth maps are those of process_1(...) +
// mapper_fused.getDepthMapFromDSI(...) before the filters, bit for bit (process_1_depth_map above is the one-window form).
//
// on_window(const dsi::WindowDepthMap&) is called once per window, in window order, from the calling thread; with
// options_depth_map it also carries the filtered outputs of main.cpp:281.
namespace dsi {

struct WindowDepthMap {
    int index = 0;                    // 0, 1, ... in the order of main.cpp:177
    double t_start = 0, t_stop = 0;   // the interval
    double ts = 0;                    // the reference view's timestamp (main.cpp:184-188)
    Transformation T_rv_w;            // process1.cpp:56-68
    size_t n_events[2] = {0, 0};      // events of the interval per camera (a camera with < 1024 votes nothing: :71-75)
    Image<float> depth_map, confidence_map;  // the raw arg-max (mapper_emvs_stereo.cpp:302-313), before the filters
    Image<uint8_t> depth_cell_indices;
    // with options_depth_map: what main.cpp:281's getDepthMapFromDSI(depth_map, confidence_map, mask, options) returns for
    // the window (adaptive threshold, masked median, border removal: mapper_emvs_stereo.cpp:390-437)
    Image<float> filtered_depth_map, filtered_confidence_map;
    Image<uint8_t> semidense_mask;
};

// where the calling thread of full_sequence_depth_maps spent its time (milliseconds, summed over the windows)
struct WindowStreamStats {
    double wait_prepare_ms = 0;  // waiting for the two preparation threads
    double wait_gpu_ms = 0;      // waiting for a slot's window to complete (dsi_mapper_fetch_wait)
    double wait_upload_ms = 0;   // waiting for a slot's previous uploads before its staging is overwritten
    double deliver_ms = 0;       // copying a window's maps out of the page-locked buffers, the filters, on_window
    double submit_ms = 0;        // queueing uploads, kernels and the asynchronous fetch
    double total_ms = 0;
    size_t windows = 0;
};

// events of a time-sorted vector with t_start <= ts <= t_stop: [begin, end)
// (a rosbag is cut at message granularity instead, data_loading.cpp:272-285: a few hundred events more)
inline void window_event_range(const std::vector<Event>& ev, double t_start, double t_stop, size_t* begin, size_t* end)
{
    size_t lo = 0, hi = ev.size();
    while (lo < hi) {  // first event with ts >= t_start
        const size_t mid = (lo + hi) / 2;
        if (ev[mid].ts < t_start) lo = mid + 1; else hi = mid;
    }
    *begin = lo;
    hi = ev.size();
    while (lo < hi) {  // first event with ts > t_stop
        const size_t mid = (lo + hi) / 2;
        if (ev[mid].ts <= t_stop) lo = mid + 1; else hi = mid;
    }
    *end = lo;
}

// host threads that turn a window's array of structs into the engine's arrays and look up the packets' poses while the
// calling thread hands a finished window to the caller and queues the next one
class WindowPrepWorker {
public:
    WindowPrepWorker() : th_([this] { loop(); }) {}
    ~WindowPrepWorker()
    {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    WindowPrepWorker(const WindowPrepWorker&) = delete;
    WindowPrepWorker& operator=(const WindowPrepWorker&) = delete;
    void start(std::function<void()> job)
    {
        {
            std::lock_guard<std::mutex> l(m_);
            job_ = std::move(job);
            has_ = true;
            done_ = false;
        }
        cv_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return done_; });
        if (err_) {
            std::exception_ptr e = err_;
            err_ = nullptr;
            std::rethrow_exception(e);
        }
    }

private:
    void loop()
    {
        for (;;) {
            std::unique_lock<std::mutex> l(m_);
            cv_.wait(l, [&] { return has_ || stop_; });
            if (stop_) return;
            std::function<void()> job = std::move(job_);
            has_ = false;
            l.unlock();
            try {
                job();
            } catch (...) {
                err_ = std::current_exception();
            }
            l.lock();
            done_ = true;
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool has_ = false, done_ = true, stop_ = false;
    std::exception_ptr err_ = nullptr;
    std::thread th_;  // (last: the thread starts when every other member exists)
};

// devices: slot k (window k, k + depth, ...) lives on devices[k % devices.size()] -- windows are independent (fresh
// mappers per window in the reference, main.cpp:262-275), so several GPUs of a node take them in turn, no collective
// ("replicas"); depth >= devices.size() to use them all.
template <typename OnWindow>
inline size_t full_sequence_depth_maps(const std::vector<int>& devices, const PinholeCameraModel& cam0, const PinholeCameraModel& cam1,
                                       const EMVS::ShapeDSI& dsi_shape, const LinearTrajectory& trajectory0,
                                       const LinearTrajectory& trajectory1, const std::vector<Event>& events0,
                                       const std::vector<Event>& events1, double start_time_s, double stop_time_s,
                                       double duration, double out_skip, bool forward_looking, int fusion_method,
                                       OnWindow&& on_window, int depth = 2, double rv_pos = 0.0,
                                       const EMVS::OptionsDepthMap* options_depth_map = nullptr,
                                       WindowStreamStats* stats = nullptr)
{
    using clock = std::chrono::steady_clock;
    const clock::time_point t_call = clock::now();
    WindowStreamStats st;
    auto since = [](clock::time_point t0) { return std::chrono::duration<double, std::milli>(clock::now() - t0).count(); };
    if (!(duration > 0) || !(out_skip > 0)) throw Error(DSI_ERR_INVALID, "full_sequence_depth_maps: duration and out_skip must be > 0");
    if (depth < 1) depth = 1;
    if (devices.empty()) throw Error(DSI_ERR_INVALID, "full_sequence_depth_maps: no device");
    struct Slot {
        Context ctx;
        EMVS::MapperEMVS m0, m1, out;
        dsi_batch_t* batch[2] = {nullptr, nullptr};
        void* host[3] = {nullptr, nullptr, nullptr};  // page-locked depth, confidence, indices
        // page-locked staging of the window's inputs (asynchronous uploads read them until the window is done):
        // x, y, packet_first, Rt per camera
        void* in[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
        size_t in_events[2] = {0, 0}, in_packets[2] = {0, 0};
        WindowDepthMap w;     // the window in flight (or just completed) in this slot
        WindowDepthMap next;  // the window being prepared for it
        size_t np[2] = {0, 0};
        int prc[2] = {DSI_OK, DSI_OK};
        bool busy = false;
        Slot(int dev, const PinholeCameraModel& c0, const PinholeCameraModel& c1, const EMVS::ShapeDSI& sh)
            : ctx(dev), m0(ctx, c0, sh), m1(ctx, c1, sh), out(ctx, c0, sh) {}
        ~Slot()
        {
            if (busy) dsi_mapper_fetch_wait(out.handle());
            for (dsi_batch_t* b : batch) dsi_batch_destroy(b);
            for (void* p : host) dsi_host_free(p);
            for (auto& cam : in)
                for (void* p : cam) dsi_host_free(p);
        }
        void reserve_inputs(int c, size_t ne, size_t np)
        {
            if (ne > in_events[c] || !in[c][0]) {
                const size_t cap = ne + ne / 4 + 1024;
                dsi_host_free(in[c][0]);
                dsi_host_free(in[c][1]);
                in[c][0] = in[c][1] = nullptr;
                check(dsi_host_alloc(cap * sizeof(uint16_t), &in[c][0]));
                check(dsi_host_alloc(cap * sizeof(uint16_t), &in[c][1]));
                in_events[c] = cap;
            }
            if (np > in_packets[c] || !in[c][2]) {
                const size_t cap = np + np / 4 + 16;
                dsi_host_free(in[c][2]);
                dsi_host_free(in[c][3]);
                in[c][2] = in[c][3] = nullptr;
                check(dsi_host_alloc(cap * sizeof(uint32_t), &in[c][2]));
                check(dsi_host_alloc(cap * 12 * sizeof(float), &in[c][3]));
                in_packets[c] = cap;
            }
        }
    };
    std::vector<std::unique_ptr<Slot>> slots;
    for (int k = 0; k < depth; ++k) slots.emplace_back(new Slot(devices[(size_t)k % devices.size()], cam0, cam1, dsi_shape));
    int nx = 0, ny = 0, nz = 0;
    slots[0]->out.dsi_.getDimensions(&nx, &ny, &nz);
    const size_t npix = (size_t)nx * ny;
    for (auto& s : slots) {
        check(dsi_host_alloc(npix * sizeof(float), &s->host[0]));
        check(dsi_host_alloc(npix * sizeof(float), &s->host[1]));
        check(dsi_host_alloc(npix, &s->host[2]));
    }
    // the slot's window is complete on the device and in s.host: wait for it, release its inputs
    auto complete = [&](Slot& s) {
        const clock::time_point t0 = clock::now();
        check(dsi_mapper_fetch_wait(s.out.handle()));
        st.wait_gpu_ms += since(t0);
        for (dsi_batch_t*& b : s.batch) {
            dsi_batch_destroy(b);
            b = nullptr;
        }
        s.busy = false;
    };
    // ... and goes to the caller (s.w still describes it)
    auto deliver = [&](Slot& s) {
        const clock::time_point t0 = clock::now();
        // (the slot's images keep their storage from window to window: one copy out of the page-locked buffers, no
        //  allocation, no zero-fill)
        auto fill = [&](auto& img, const void* src) {
            using T = typename std::remove_reference<decltype(img.data)>::type::value_type;
            img.rows = ny;
            img.cols = nx;
            img.data.assign(static_cast<const T*>(src), static_cast<const T*>(src) + npix);
        };
        fill(s.w.depth_map, s.host[0]);
        fill(s.w.confidence_map, s.host[1]);
        fill(s.w.depth_cell_indices, s.host[2]);
        if (options_depth_map)  // the filters run on the arg-max the slot's mapper still holds on the device
            s.out.filterDepthMap(s.w.filtered_depth_map, s.w.filtered_confidence_map, s.w.semidense_mask, *options_depth_map);
        on_window(static_cast<const WindowDepthMap&>(s.w));
        st.deliver_ms += since(t0);
    };
    const LinearTrajectory* trs[2] = {&trajectory0, &trajectory1};
    const std::vector<Event>* evs[2] = {&events0, &events1};
    std::vector<double> starts;  // main.cpp:177 (the loop variable is accumulated, like there)
    for (double interval_start = start_time_s; interval_start + duration <= stop_time_s; interval_start += out_skip)
        starts.push_back(interval_start);
    const size_t n_windows = starts.size();
    constexpr int kPrepSplit = 4;  // host threads per camera (a window's 500 k 24-byte events take one core 0.4-0.7 ms)
    WindowPrepWorker workers[2 * kPrepSplit];
    // prepare(i): window i's interval, reference view and event ranges, then the preparation threads fill the slot's
    // INPUT staging.  That staging was last read by the uploads of the window the slot holds (window i - depth), which
    // are long over -- they are waited for, not the window itself: its results are collected later, before submit(i).
    auto prepare = [&](size_t i) {
        Slot& s = *slots[i % slots.size()];
        {
            const clock::time_point t0 = clock::now();
            for (dsi_batch_t* b : s.batch)
                while (b && !dsi_batch_uploaded(b)) std::this_thread::yield();
            st.wait_upload_ms += since(t0);
        }
        const double interval_start = starts[i], interval_stop = interval_start + duration;
        const double ts = forward_looking ? interval_stop : (interval_start + interval_stop) / 2;  // main.cpp:184-188
        Transformation T_w_l;
        if (!trajectory0.getPoseAt(ts, T_w_l)) throw Error(DSI_ERR_INVALID, "no pose at the reference timestamp of a window");
        Transformation baseline;
        baseline.t[0] = rv_pos;
        s.next = WindowDepthMap{};
        s.next.index = (int)i;
        s.next.t_start = interval_start;
        s.next.t_stop = interval_stop;
        s.next.ts = ts;
        s.next.T_rv_w = inverse(T_w_l * baseline);  // process1.cpp:56-68
        double T7[7];
        s.next.T_rv_w.to7(T7);
        for (int c = 0; c < 2; ++c) {
            size_t a = 0, b = 0;
            window_event_range(*evs[c], interval_start, interval_stop, &a, &b);
            const size_t ne = b - a;
            s.next.n_events[c] = ne;
            s.reserve_inputs(c, ne, ne / DSI_PACKET_SIZE + 1);
            const Event* src = evs[c]->data() + a;
            uint16_t* xs = static_cast<uint16_t*>(s.in[c][0]);
            uint16_t* ys = static_cast<uint16_t*>(s.in[c][1]);
            uint32_t* first = static_cast<uint32_t*>(s.in[c][2]);
            float* Rt = static_cast<float*>(s.in[c][3]);
            const LinearTrajectory* tr = trs[c];
            size_t* np_c = &s.np[c];
            int* rc_c = &s.prc[c];
            for (int part = 0; part < kPrepSplit; ++part) {
                const size_t k0 = ne * (size_t)part / kPrepSplit, k1 = ne * (size_t)(part + 1) / kPrepSplit;
                workers[c * kPrepSplit + part].start([=] {
                    for (size_t k = k0; k < k1; ++k) {
                        xs[k] = src[k].x;
                        ys[k] = src[k].y;
                    }
                    // (one timestamp per packet is read, in place: mapper_emvs_stereo.cpp:88-99)
                    if (part == 0)
                        *rc_c = dsi_packetize_strided(ne ? &src[0].ts : nullptr, sizeof(Event), ne, tr->times().data(),
                                                      tr->poses7().data(), tr->times().size(), T7, first, Rt, np_c);
                });
            }
        }
    };
    auto submit = [&](Slot& s) {
        const clock::time_point t0 = clock::now();
        s.w.index = s.next.index;  // (the window's description; the images of s.w keep their storage)
        s.w.t_start = s.next.t_start;
        s.w.t_stop = s.next.t_stop;
        s.w.ts = s.next.ts;
        s.w.T_rv_w = s.next.T_rv_w;
        s.w.n_events[0] = s.next.n_events[0];
        s.w.n_events[1] = s.next.n_events[1];
        dsi_mapper_t* ms[2] = {s.m0.handle(), s.m1.handle()};
        for (int c = 0; c < 2; ++c) {
            if (s.prc[c] == DSI_ERR_TOO_FEW_EVENTS) s.np[c] = 0;  // evaluateDSI returns false: an all-zero DSI (mapper_emvs_stereo.cpp:71-75)
            else if (s.prc[c] != DSI_OK) throw Error(s.prc[c], "dsi_packetize failed for a window (preparation thread)");
            check(dsi_batch_create_async(s.ctx.handle(), static_cast<uint16_t*>(s.in[c][0]), static_cast<uint16_t*>(s.in[c][1]),
                                         s.w.n_events[c], static_cast<uint32_t*>(s.in[c][2]), static_cast<float*>(s.in[c][3]), s.np[c],
                                         &s.batch[c]));
        }
        check(dsi_mapper_depth_map_of_events(s.out.handle(), ms, s.batch, 2, fusion_method));
        // (behind the window's kernels on its context's stream: ONE stream per window in flight + the device's upload stream)
        check(dsi_mapper_fetch_depth_map_in_order(s.out.handle(), static_cast<float*>(s.host[0]), static_cast<float*>(s.host[1]),
                                                  static_cast<uint8_t*>(s.host[2])));
        s.busy = true;
        st.submit_ms += since(t0);
    };
    auto wait_workers = [&]() {
        const clock::time_point t0 = clock::now();
        for (WindowPrepWorker& w : workers) w.wait();
        st.wait_prepare_ms += since(t0);
    };
    // Three things run side by side: the preparation threads turn window i+1's events into the engine's arrays, the GPU
    // works on windows i-1, i-2, ..., and this thread collects window i-depth from slot i's buffers, hands it to the caller
    // and queues window i.  (With ONE slot the next window's staging is the current window's: everything is serial.)
    const bool ahead = slots.size() >= 2;
    if (n_windows) prepare(0);
    for (size_t i = 0; i < n_windows; ++i) {
        Slot& s = *slots[i % slots.size()];
        wait_workers();
        if (ahead && i + 1 < n_windows) prepare(i + 1);
        if (s.busy) {  // the window this slot holds (s.w, s.host), before submit() replaces it
            complete(s);
            deliver(s);
        }
        submit(s);
        if (!ahead && i + 1 < n_windows) prepare(i + 1);
    }
    wait_workers();
    // the windows still in flight, oldest first
    for (size_t k = 0; k < slots.size(); ++k) {
        Slot& s = *slots[(n_windows + k) % slots.size()];
        if (s.busy) {
            complete(s);
            deliver(s);
        }
    }
    if (stats) {
        st.total_ms = since(t_call);
        st.windows = n_windows;
        *stats = st;
    }
    return n_windows;
}

// one GPU
template <typename OnWindow>
inline size_t full_sequence_depth_maps(int device, const PinholeCameraModel& cam0, const PinholeCameraModel& cam1,
                                       const EMVS::ShapeDSI& dsi_shape, const LinearTrajectory& trajectory0,
                                       const LinearTrajectory& trajectory1, const std::vector<Event>& events0,
                                       const std::vector<Event>& events1, double start_time_s, double stop_time_s,
                                       double duration, double out_skip, bool forward_looking, int fusion_method,
                                       OnWindow&& on_window, int depth = 2, double rv_pos = 0.0,
                                       const EMVS::OptionsDepthMap* options_depth_map = nullptr,
                                       WindowStreamStats* stats = nullptr)
{
    return full_sequence_depth_maps(std::vector<int>{device}, cam0, cam1, dsi_shape, trajectory0, trajectory1, events0, events1,
                                    start_time_s, stop_time_s, duration, out_skip, forward_looking, fusion_method,
                                    std::forward<OnWindow>(on_window), depth, rv_pos, options_depth_map, stats);
}

}  // namespace dsi

struct Process2Result {
    dsi::Transformation T_rv_w;
    Grid3D left, right;  // temporal fusion of the left / right camera's sub-interval DSIs
};

// Alg. 2 (shuffle_right = false) and process_5 (true).
inline Process2Result process_2(dsi::Context& ctx, const dsi::PinholeCameraModel& cam0,
                                const dsi::PinholeCameraModel& cam1, const LinearTrajectory& trajectory0,
                                const LinearTrajectory& trajectory1, const std::vector<dsi::Event>& events0,
                                const std::vector<dsi::Event>& events1, const EMVS::ShapeDSI& dsi_shape,
                                const int num_subintervals, EMVS::MapperEMVS& mapper_fused,
                                EMVS::MapperEMVS& mapper_fused_camera_time, double ts, int stereo_fusion,
                                int temporal_fusion, bool shuffle_right = false)
{
    EMVS::MapperEMVS mapper0(ctx, cam0, dsi_shape), mapper1(ctx, cam1, dsi_shape);
    int nx, ny, nz;
    mapper0.dsi_.getDimensions(&nx, &ny, &nz);
    Grid3D sub(ctx, nx, ny, nz);  // mapper_fused_subinterval.dsi_
    Process2Result out;
    out.left.allocate(ctx, nx, ny, nz);
    out.right.allocate(ctx, nx, ny, nz);
    dsi::Transformation T_w_l;
    if (!trajectory0.getPoseAt(ts, T_w_l)) throw dsi::Error(DSI_ERR_INVALID, "no pose at the reference timestamp");
    out.T_rv_w = dsi::inverse(T_w_l);  // process2.cpp:79-81
    const size_t per0 = events0.size() / (size_t)num_subintervals;  // :46-47, integer division
    const size_t per1 = events1.size() / (size_t)num_subintervals;
    mapper_fused.dsi_.resetGrid();  // :90
    size_t idx1 = shuffle_right ? (size_t)(num_subintervals / 2) * per1 : 0;  // process5.cpp:89-93
    for (int k = 0; k < num_subintervals; ++k) {
        const std::vector<dsi::Event> ev0(events0.begin() + (long)(k * per0), events0.begin() + (long)((k + 1) * per0));
        std::vector<dsi::Event> ev1;
        if (!shuffle_right) {
            ev1.assign(events1.begin() + (long)(k * per1), events1.begin() + (long)((k + 1) * per1));  // :132-134
        } else if (idx1 + per1 >= events1.size()) {  // process5.cpp:136-150: tail, then the head
            ev1.assign(events1.begin() + (long)idx1, events1.end());
            const size_t rest = idx1 + per1 - events1.size();
            ev1.insert(ev1.end(), events1.begin(), events1.begin() + (long)rest);
            idx1 = rest;
        } else {
            ev1.assign(events1.begin() + (long)idx1, events1.begin() + (long)(idx1 + per1));
            idx1 += per1;
        }
        mapper0.dsi_.resetGrid();
        mapper0.evaluateDSI(ev0, trajectory0, out.T_rv_w);  // :119
        mapper1.dsi_.resetGrid();
        mapper1.evaluateDSI(ev1, trajectory1, out.T_rv_w);  // :146
        sub.resetGrid();                                    // :159
        sub.addTwoGrids(mapper0.dsi_);                      // :160
        dsi::fuseTwoGrids(sub, mapper1.dsi_, stereo_fusion, "Improper stereo fusion method selected");  // :168-189
        if (temporal_fusion == 2) {  // :216-226
            out.left.addInverseOfTwoGrids(mapper0.dsi_);
            out.right.addInverseOfTwoGrids(mapper1.dsi_);
            mapper_fused.dsi_.addInverseOfTwoGrids(sub);
            if (k == num_subintervals - 1) {
                out.left.computeHMfromSumOfInv(num_subintervals);
                out.right.computeHMfromSumOfInv(num_subintervals);
                mapper_fused.dsi_.computeHMfromSumOfInv(num_subintervals);
            }
        } else if (temporal_fusion == 4) {  // :229-239
            out.left.addTwoGrids(mapper0.dsi_);
            out.right.addTwoGrids(mapper1.dsi_);
            mapper_fused.dsi_.addTwoGrids(sub);
            if (k == num_subintervals - 1) {
                out.left.computeAMfromSum(num_subintervals);
                out.right.computeAMfromSum(num_subintervals);
                mapper_fused.dsi_.computeAMfromSum(num_subintervals);
            }
        }  // 1, 3, 5, 6: nothing happens (:213, :227, :240-243)
    }
    // converse order: time first, cameras second (:266-289).  The reference swaps cases 3 and 4
    // here (3 -> arithmetic, 4 -> geometric, process2.cpp:274-279); kept as is.
    mapper_fused_camera_time.dsi_.addTwoGrids(out.left);  // :266
    static const int converse[7] = {0, 1, 2, 4, 3, 5, 6};
    if (stereo_fusion < 1 || stereo_fusion > 6) throw dsi::Error(DSI_ERR_BAD_OP, "Improper stereo fusion method selected");
    dsi::fuseTwoGrids(mapper_fused_camera_time.dsi_, out.right, converse[stereo_fusion],
                      "Improper stereo fusion method selected");
    return out;
}

// Alg. 2 on the GPUs of one node (BASELINE configs[3]): sub-interval k goes to device
// k mod n, every device fuses the cameras of its sub-intervals and accumulates them locally
// (process2.cpp:98-242), then ONE RCCL all-reduce over xGMI of each accumulator (sum; harmonic:
// of the inverse sums, cartesian3dgrid.h:72-86) replaces the rest of the temporal loop, and every
// device finalizes its copy.  One host thread drives all devices, like the reference's single
// process; work on different devices overlaps because every call is asynchronous on its
// context's stream.  Events never cross devices.
//   ctxs / comms: one context per device and the communicator ranks from dsi::Comm::createAll(ctxs)
//   (comms may be empty when there is one device).
// The result is on EVERY device (fused[i], left[i], right[i]); the converse order
// (process2.cpp:266-289) is computed on device 0 into camera_time.
struct Process2MultiResult {
    dsi::Transformation T_rv_w;
    std::vector<Grid3D> fused, left, right;  // per device: time-fused camera fusion, left / right temporal DSIs
    Grid3D camera_time;                      // device 0
};

inline Process2MultiResult process_2_multi_gpu(const std::vector<dsi::Context*>& ctxs, std::vector<dsi::Comm>& comms,
                                               const dsi::PinholeCameraModel& cam0,
                                               const dsi::PinholeCameraModel& cam1,
                                               const LinearTrajectory& trajectory0,
                                               const LinearTrajectory& trajectory1,
                                               const std::vector<dsi::Event>& events0,
                                               const std::vector<dsi::Event>& events1,
                                               const EMVS::ShapeDSI& dsi_shape, const int num_subintervals, double ts,
                                               int stereo_fusion, int temporal_fusion)
{
    const int ndev = (int)ctxs.size();
    if (ndev < 1) throw dsi::Error(DSI_ERR_INVALID, "no device");
    if (ndev > 1 && (int)comms.size() != ndev) throw dsi::Error(DSI_ERR_INVALID, "one communicator rank per device");
    if (temporal_fusion != 2 && temporal_fusion != 4)
        throw dsi::Error(DSI_ERR_BAD_OP, "temporal fusion 2 (harmonic) or 4 (arithmetic) -- the others do nothing (process2.cpp:213-243)");
    const int mode = temporal_fusion == 2 ? DSI_ACC_INV_SUM : DSI_ACC_SUM;
    Process2MultiResult out;
    dsi::Transformation T_w_l;
    if (!trajectory0.getPoseAt(ts, T_w_l)) throw dsi::Error(DSI_ERR_INVALID, "no pose at the reference timestamp");
    out.T_rv_w = dsi::inverse(T_w_l);  // process2.cpp:79-81
    std::vector<std::unique_ptr<EMVS::MapperEMVS>> m0, m1;
    std::vector<Grid3D> sub(ndev);
    out.fused.resize(ndev);
    out.left.resize(ndev);
    out.right.resize(ndev);
    int nx = 0, ny = 0, nz = 0;
    for (int i = 0; i < ndev; ++i) {
        m0.emplace_back(new EMVS::MapperEMVS(*ctxs[i], cam0, dsi_shape));
        m1.emplace_back(new EMVS::MapperEMVS(*ctxs[i], cam1, dsi_shape));
        m0[i]->dsi_.getDimensions(&nx, &ny, &nz);
        sub[i].allocate(*ctxs[i], nx, ny, nz);
        for (Grid3D* g : {&out.fused[i], &out.left[i], &out.right[i]}) {
            g->allocate(*ctxs[i], nx, ny, nz);
            g->accumulateBegin(mode);  // resetGrid (process2.cpp:90)
        }
    }
    const size_t per0 = events0.size() / (size_t)num_subintervals;  // :46-47
    const size_t per1 = events1.size() / (size_t)num_subintervals;
    for (int k = 0; k < num_subintervals; ++k) {
        const int i = k % ndev;
        const std::vector<dsi::Event> ev0(events0.begin() + (long)(k * per0), events0.begin() + (long)((k + 1) * per0));
        const std::vector<dsi::Event> ev1(events1.begin() + (long)(k * per1), events1.begin() + (long)((k + 1) * per1));
        m0[i]->dsi_.resetGrid();                           // :100-101 (a sub-interval with < 1024 events votes nothing)
        m1[i]->dsi_.resetGrid();
        m0[i]->evaluateDSI(ev0, trajectory0, out.T_rv_w);  // :119
        m1[i]->evaluateDSI(ev1, trajectory1, out.T_rv_w);  // :146
        sub[i].resetGrid();                                // :159-160
        sub[i].addTwoGrids(m0[i]->dsi_);
        dsi::fuseTwoGrids(sub[i], m1[i]->dsi_, stereo_fusion, "Improper stereo fusion method selected");  // :168-189
        out.left[i].accumulate(m0[i]->dsi_, mode);   // :218-220 / :231-233
        out.right[i].accumulate(m1[i]->dsi_, mode);
        out.fused[i].accumulate(sub[i], mode);
    }
    if (ndev > 1) {  // the three accumulators, one group call each
        std::vector<dsi_comm_t*> cs;
        for (dsi::Comm& c : comms) cs.push_back(c.handle());
        for (std::vector<Grid3D>* acc : {&out.fused, &out.left, &out.right}) {
            std::vector<dsi_grid_t*> gs;
            for (Grid3D& g : *acc) gs.push_back(g.handle());
            dsi::check(dsi_grid_allreduce_all(cs.data(), gs.data(), ndev, dsi_acc_reduce_op(mode)));
        }
    }
    for (int i = 0; i < ndev; ++i)
        for (Grid3D* g : {&out.fused[i], &out.left[i], &out.right[i]}) g->finalize(mode, num_subintervals);  // :221-225
    // converse order on device 0 (:266-289; cases 3 and 4 swapped there, kept)
    out.camera_time.allocate(*ctxs[0], nx, ny, nz);
    out.camera_time.resetGrid();
    out.camera_time.addTwoGrids(out.left[0]);
    static const int converse[7] = {0, 1, 2, 4, 3, 5, 6};
    if (stereo_fusion < 1 || stereo_fusion > 6) throw dsi::Error(DSI_ERR_BAD_OP, "Improper stereo fusion method selected");
    dsi::fuseTwoGrids(out.camera_time, out.right[0], converse[stereo_fusion], "Improper stereo fusion method selected");
    for (int i = 0; i < ndev; ++i) ctxs[i]->synchronize();  // the mappers go out of scope
    return out;
}

// ---- Alg. 2 with a plane index map equal to the CPU reference's on EVERY pixel (BASELINE configs[3]) ----
// The resolver (dsi_mapper_resolve_near_ties) covers Alg. 1's topology; Alg. 2 fuses cameras per sub-interval and then
// over time (process2.cpp:98-249), so it is resolved from the same building blocks: the near-tie columns of the FINAL
// fused DSI (dsi_grid_near_tie_voxels), those voxels of every (sub-interval, camera) DSI re-summed in the reference's
// order (dsi_mapper_exact_voxels: fp32, event order, cartesian3dgrid.h:261-270), the reference's scalar ops on the host
// in process_2's order (dsi_reference_fuse2 / _accumulate / _finalize), the first maximum per column
// (cartesian3dgrid.cpp:132-134), and the few pixels patched (dsi_mapper_patch_depth_map).
namespace dsi {

struct ExactDepthMapInfo {
    size_t near_tie_pixels = 0, candidate_voxels = 0;
    long long votes = 0;
    int changed_pixels = 0;
};

// events [begin, end) of one camera as a resident batch for the reference view T7 (packetisation + pose pipeline of
// mapper_emvs_stereo.cpp:67-105); a batch without packets when evaluateDSI would return false
inline dsi_batch_t* make_batch(dsi_context_t* ctx, const std::vector<Event>& ev, size_t begin, size_t end,
                               const LinearTrajectory& tr, const double* T7)
{
    const size_t ne = end - begin;
    std::vector<uint16_t> xs(ne), ys(ne);
    for (size_t i = 0; i < ne; ++i) {
        xs[i] = ev[begin + i].x;
        ys[i] = ev[begin + i].y;
    }
    std::vector<uint32_t> first(ne / DSI_PACKET_SIZE + 1, 0u);
    std::vector<float> Rt(12 * first.size(), 0.f);
    size_t np = 0;
    const int rc = dsi_packetize_strided(ne ? &ev[begin].ts : nullptr, sizeof(Event), ne, tr.times().data(), tr.poses7().data(),
                                         tr.times().size(), T7, first.data(), Rt.data(), &np);
    if (rc == DSI_ERR_TOO_FEW_EVENTS) np = 0;
    else check(rc);
    dsi_batch_t* b = nullptr;
    check(dsi_batch_create(ctx, xs.data(), ys.data(), ne, first.data(), Rt.data(), np, &b));
    return b;
}

}  // namespace dsi

inline dsi::ExactDepthMapInfo process_2_exact_depth_map(dsi::Context& ctx, const dsi::PinholeCameraModel& cam0,
                                                        const dsi::PinholeCameraModel& cam1,
                                                        const LinearTrajectory& trajectory0, const LinearTrajectory& trajectory1,
                                                        const std::vector<dsi::Event>& events0,
                                                        const std::vector<dsi::Event>& events1,
                                                        const EMVS::ShapeDSI& dsi_shape, const int num_subintervals,
                                                        EMVS::MapperEMVS& mapper_fused, double ts, int stereo_fusion,
                                                        int temporal_fusion, dsi::Image<float>& depth_map,
                                                        dsi::Image<float>& confidence_map,
                                                        dsi::Image<uint8_t>& depth_cell_indices, float rel_gap = 0.f)
{
    EMVS::MapperEMVS mapper0(ctx, cam0, dsi_shape), mapper1(ctx, cam1, dsi_shape);
    EMVS::MapperEMVS* mappers[2] = {&mapper0, &mapper1};
    const std::vector<dsi::Event>* evs[2] = {&events0, &events1};
    const LinearTrajectory* trs[2] = {&trajectory0, &trajectory1};
    int nx, ny, nz;
    mapper0.dsi_.getDimensions(&nx, &ny, &nz);
    Grid3D sub(ctx, nx, ny, nz);
    dsi::Transformation T_w_l;
    if (!trajectory0.getPoseAt(ts, T_w_l)) throw dsi::Error(DSI_ERR_INVALID, "no pose at the reference timestamp");
    double T7[7];
    dsi::inverse(T_w_l).to7(T7);  // process2.cpp:79-81
    const int mode = temporal_fusion == 2 ? DSI_ACC_INV_SUM : (temporal_fusion == 4 ? DSI_ACC_SUM : -1);
    std::vector<dsi_batch_t*> batches;  // [sub-interval][camera]
    dsi::ExactDepthMapInfo info;
    auto release = [&]() {
        for (dsi_batch_t* b : batches) dsi_batch_destroy(b);
        batches.clear();
    };
    try {
        mapper_fused.dsi_.resetGrid();  // :90
        for (int k = 0; k < num_subintervals; ++k) {
            for (int c = 0; c < 2; ++c) {
                const size_t per = evs[c]->size() / (size_t)num_subintervals;  // :46-47
                dsi_batch_t* b = dsi::make_batch(ctx.handle(), *evs[c], (size_t)k * per, (size_t)(k + 1) * per, *trs[c], T7);
                batches.push_back(b);
                // :100-101 resetGrid + :119 / :146 evaluateDSI (a batch without packets -- evaluateDSI returned false -- leaves
                // the reset DSI: the vote of zero packets writes zeros)
                dsi::check(dsi_mapper_evaluate_batch(mappers[c]->handle(), b));
            }
            sub.resetGrid();  // :159-160
            sub.addTwoGrids(mapper0.dsi_);
            dsi::fuseTwoGrids(sub, mapper1.dsi_, stereo_fusion, "Improper stereo fusion method selected");  // :168-189
            if (mode >= 0) {
                mapper_fused.dsi_.accumulate(sub, mode);                                            // :218-220 / :231-233
                if (k == num_subintervals - 1) mapper_fused.dsi_.finalize(mode, num_subintervals);  // :221-225
            }
        }
        dsi::check(dsi_mapper_depth_map_of(mapper_fused.handle(), mapper_fused.dsi_.handle()));
        const size_t npix = (size_t)nx * ny;
        depth_map = dsi::Image<float>(ny, nx);
        confidence_map = dsi::Image<float>(ny, nx);
        depth_cell_indices = dsi::Image<uint8_t>(ny, nx);
        if (mode >= 0) {
            std::vector<uint32_t> vox(1 << 16);
            size_t n_vox = 0, n_cols = 0;
            for (;;) {
                dsi::check(dsi_grid_near_tie_voxels(mapper_fused.handle(), mapper_fused.dsi_.handle(), rel_gap, vox.data(), vox.size(),
                                                    &n_vox, &n_cols));
                if (n_vox <= vox.size()) break;
                vox.resize(n_vox);
            }
            vox.resize(n_vox);
            info.near_tie_pixels = n_cols;
            info.candidate_voxels = n_vox;
            if (n_vox) {
                std::vector<float> acc(n_vox, 0.f), a(n_vox), g(n_vox), f(n_vox);
                std::vector<uint32_t> votes(n_vox);
                for (int k = 0; k < num_subintervals; ++k) {
                    dsi::check(dsi_mapper_exact_voxels(mapper0.handle(), batches[2 * k], vox.data(), n_vox, a.data(), votes.data()));
                    for (uint32_t v : votes) info.votes += v;
                    dsi::check(dsi_mapper_exact_voxels(mapper1.handle(), batches[2 * k + 1], vox.data(), n_vox, g.data(), votes.data()));
                    for (uint32_t v : votes) info.votes += v;
                    dsi::check(dsi_reference_fuse2(stereo_fusion, a.data(), g.data(), n_vox, f.data()));
                    dsi::check(dsi_reference_accumulate(mode, acc.data(), f.data(), n_vox));
                }
                dsi::check(dsi_reference_finalize(mode, acc.data(), n_vox, num_subintervals));
                dsi::check(dsi_mapper_fetch_depth_map(mapper_fused.handle(), nullptr, nullptr, depth_cell_indices.data.data()));
                std::vector<uint32_t> pix;
                std::vector<uint8_t> idx;
                std::vector<float> conf;
                for (size_t i = 0; i < n_vox;) {  // a column's run is contiguous, planes ascending
                    const uint32_t p = vox[i] % (uint32_t)npix;
                    size_t best = i;
                    for (size_t j = i; j < n_vox && vox[j] % (uint32_t)npix == p; ++j, ++i)
                        if (acc[best] < acc[j]) best = j;  // the first maximum wins
                    pix.push_back(p);
                    idx.push_back((uint8_t)(vox[best] / (uint32_t)npix));
                    conf.push_back(acc[best]);
                    if (depth_cell_indices.data[p] != idx.back()) ++info.changed_pixels;
                }
                dsi::check(dsi_mapper_patch_depth_map(mapper_fused.handle(), pix.data(), idx.data(), conf.data(), pix.size()));
            }
        }
        dsi::check(dsi_mapper_fetch_depth_map(mapper_fused.handle(), depth_map.data.data(), confidence_map.data.data(),
                                              depth_cell_indices.data.data()));
    } catch (...) {
        release();
        throw;
    }
    release();
    return info;
}

inline Process2Result process_5(dsi::Context& ctx, const dsi::PinholeCameraModel& cam0,
                                const dsi::PinholeCameraModel& cam1, const LinearTrajectory& trajectory0,
                                const LinearTrajectory& trajectory1, const std::vector<dsi::Event>& events0,
                                const std::vector<dsi::Event>& events1, const EMVS::ShapeDSI& dsi_shape,
                                const int num_subintervals, EMVS::MapperEMVS& mapper_fused,
                                EMVS::MapperEMVS& mapper_fused_camera_time, double ts, int stereo_fusion,
                                int temporal_fusion)
{
    return process_2(ctx, cam0, cam1, trajectory0, trajectory1, events0, events1, dsi_shape, num_subintervals,
                     mapper_fused, mapper_fused_camera_time, ts, stereo_fusion, temporal_fusion, true);
}

#endif
