// C++ drop-in test: the reference's process_1 flow (process1.cpp:54-222) and, through
// include/dsi_process.hpp, process_1 / process_2 / process_5 as functions, written against
// include/dsi_engine.hpp, i.e. with the reference's own class and method names, checked
// against the CPU oracle (oracle/dsi_oracle.h).  Built and run by tests/test_cpp_adapter.py.
//
//   exit 0: parity ok      exit 3: no GPU (adapter threw DSI_ERR_NO_DEVICE)      else: failure
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "dsi_engine.hpp"
#include "dsi_oracle.h"
#include "dsi_process.hpp"

namespace {

struct Lcg {
    uint64_t s;
    explicit Lcg(uint64_t seed) : s(seed) {}
    double uni()
    {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        return (double)(s >> 11) / 9007199254740992.0;
    }
};

// a small rig: camera translating along x, looking down +z; events = projections of points
void make_events(int n, double x_off, const dsi::PinholeCameraModel& cam, uint64_t seed,
                 std::vector<dsi::Event>* ev, LinearTrajectory::PoseMap* poses)
{
    Lcg rng(seed);
    const int npts = 200;
    std::vector<double> P(3 * npts);
    for (int i = 0; i < npts; ++i) {
        const double z = 1.5 + 3.0 * rng.uni();
        P[3 * i] = (rng.uni() - 0.5) * 1.2 * z;
        P[3 * i + 1] = (rng.uni() - 0.5) * 0.9 * z;
        P[3 * i + 2] = z;
    }
    for (int k = 0; k <= 12; ++k) {  // control poses every 0.1 s: x = 0.4 t
        dsi::Transformation T;
        T.t[0] = 0.4 * (0.1 * k - 0.1) + x_off;
        (*poses)[0.1 * k - 0.1] = T;
    }
    ev->clear();
    while ((int)ev->size() < n) {
        const double t = 1.0 * ev->size() / n;  // increasing timestamps in [0,1)
        const int i = (int)(rng.uni() * npts) % npts;
        const double cx = 0.4 * t + x_off;
        const double u = cam.fx * (P[3 * i] - cx) / P[3 * i + 2] + cam.cx;
        const double v = cam.fy * P[3 * i + 1] / P[3 * i + 2] + cam.cy;
        dsi::Event e;
        e.ts = t;
        if (u < 0 || v < 0 || u >= cam.width - 1 || v >= cam.height - 1) {
            e.x = (uint16_t)(rng.uni() * cam.width);  // noise event
            e.y = (uint16_t)(rng.uni() * cam.height);
        } else {
            e.x = (uint16_t)std::lround(u);
            e.y = (uint16_t)std::lround(v);
        }
        ev->push_back(e);
    }
}

// oracle evaluateDSI for the same inputs
std::vector<float> oracle_dsi(const std::vector<dsi::Event>& ev, const LinearTrajectory& traj,
                              const dsi::Transformation& T_rv_w, const dsi::PinholeCameraModel& cam,
                              const EMVS::ShapeDSI& shape, std::vector<float>* planes_out)
{
    const int nx = cam.width, ny = cam.height, nz = (int)shape.dimZ_;
    std::vector<float> planes(nz);
    orc_depth_planes(shape.min_depth_, shape.max_depth_, nz, 0, planes.data());
    *planes_out = planes;
    const float K[4] = {cam.fx, cam.fy, cam.cx, cam.cy};
    const float f = orc_virtual_focal(cam.fx, shape.fov_, nx);
    const float Kv[4] = {f, f, cam.cx, cam.cy};
    std::vector<uint16_t> x, y;
    std::vector<float> centers, H;
    double T7[7];
    T_rv_w.to7(T7);
    size_t cur = 0, np = 0;
    while (cur + 1024 < ev.size()) {
        double Tw[7];
        if (!orc_pose_at(traj.times().data(), traj.poses7().data(), traj.times().size(), ev[cur + 512].ts, Tw)) {
            ++cur;
            continue;
        }
        float Rt[12];
        orc_event_pose_Rt(T7, Tw, Rt);
        centers.resize(3 * (np + 1));
        H.resize(9 * (np + 1));
        orc_packet_geometry(Rt, K, Kv, planes[0], &centers[3 * np], &H[9 * np]);
        for (int i = 0; i < 1024; ++i, ++cur) {
            x.push_back(ev[cur].x);
            y.push_back(ev[cur].y);
        }
        ++np;
    }
    std::vector<float> xy(2 * x.size());
    orc_warp_z0(x.data(), y.data(), x.size(), H.data(), nullptr, cam.width, xy.data());
    std::vector<float> dsi((size_t)nx * ny * nz, 0.f);
    orc_fill_voxel_grid(xy.data(), centers.data(), np, planes.data(), nz, Kv, nx, ny, dsi.data());
    return dsi;
}

double max_rel_err(const std::vector<float>& a, const std::vector<float>& b)
{
    double m = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        const double e = std::fabs((double)a[i] - b[i]) / std::fmax(1.0, std::fabs((double)b[i]));
        if (e > m) m = e;
    }
    return m;
}

}  // namespace

int main()
{
    try {
        dsi::Context ctx(0);
        dsi::PinholeCameraModel cam;
        cam.width = 80;
        cam.height = 60;
        cam.fx = cam.fy = 70.f;
        cam.cx = 40.f;
        cam.cy = 30.f;
        const EMVS::ShapeDSI dsi_shape(0, 0, 24, 1.0f, 5.0f, 0.f);

        std::vector<dsi::Event> events0, events1;
        LinearTrajectory::PoseMap p0, p1;
        make_events(9000, 0.0, cam, 11, &events0, &p0);
        make_events(9000, 0.2, cam, 12, &events1, &p1);
        const LinearTrajectory trajectory0(p0), trajectory1(p1);

        // process1.cpp:56-68: reference view = left camera pose at t_mid (rv_pos = 0)
        dsi::Transformation T_w_rv;
        if (!trajectory0.getPoseAt(0.5, T_w_rv)) return 10;
        dsi::Transformation T_rv_w;  // inverse of a pure translation
        for (int i = 0; i < 3; ++i) T_rv_w.t[i] = -T_w_rv.t[i];

        EMVS::MapperEMVS mapper0(ctx, cam, dsi_shape), mapper1(ctx, cam, dsi_shape), mapper_fused(ctx, cam, dsi_shape);
        if (!mapper0.evaluateDSI(events0, trajectory0, T_rv_w)) return 11;  // process1.cpp:76
        if (!mapper1.evaluateDSI(events1, trajectory1, T_rv_w)) return 12;  // process1.cpp:94
        if (mapper0.evaluateDSI(std::vector<dsi::Event>(events0.begin(), events0.begin() + 1023), trajectory0, T_rv_w))
            return 13;                                                       // < 1024 events -> false
        if (!mapper0.evaluateDSI(events0, trajectory0, T_rv_w)) return 14;

        std::vector<float> planes;
        const std::vector<float> ref0 = oracle_dsi(events0, trajectory0, T_rv_w, cam, dsi_shape, &planes);
        const std::vector<float> ref1 = oracle_dsi(events1, trajectory1, T_rv_w, cam, dsi_shape, &planes);
        const double e0 = max_rel_err(mapper0.dsi_.download(), ref0);
        const double e1 = max_rel_err(mapper1.dsi_.download(), ref1);
        std::printf("dsi0 max rel err %.3g, dsi1 max rel err %.3g, mean square %.6f\n", e0, e1,
                    mapper0.dsi_.computeMeanSquare());
        if (e0 > 1e-4 || e1 > 1e-4) return 20;
        if (mapper0.depthPlanes() != planes) return 21;

        for (int fusion_method = 1; fusion_method <= 6; ++fusion_method) {
            // process1.cpp:126-158
            mapper_fused.dsi_.resetGrid();
            mapper_fused.dsi_.addTwoGrids(mapper0.dsi_);
            switch (fusion_method) {
            case 1: mapper_fused.dsi_.minTwoGrids(mapper1.dsi_); break;
            case 2: mapper_fused.dsi_.harmonicMeanTwoGrids(mapper1.dsi_); break;
            case 3: mapper_fused.dsi_.geometricMeanTwoGrids(mapper1.dsi_); break;
            case 4: mapper_fused.dsi_.arithmeticMeanTwoGrids(mapper1.dsi_); break;
            case 5: mapper_fused.dsi_.rmsTwoGrids(mapper1.dsi_); break;
            case 6: mapper_fused.dsi_.maxTwoGrids(mapper1.dsi_); break;
            }
            std::vector<float> a = mapper0.dsi_.download();
            const std::vector<float> g = mapper1.dsi_.download();
            orc_fuse2(a.data(), g.data(), a.size(), fusion_method);
            if (mapper_fused.dsi_.download() != a) {  // bit exact on identical inputs
                std::printf("fusion %d differs\n", fusion_method);
                return 30 + fusion_method;
            }
        }
        // process1.cpp:222 -> mapper_emvs_stereo.cpp:368 (+ :302-313)
        mapper_fused.dsi_.resetGrid();
        mapper_fused.dsi_.addTwoGrids(mapper0.dsi_);
        mapper_fused.dsi_.harmonicMeanTwoGrids(mapper1.dsi_);
        dsi::Image<float> depth, conf;
        dsi::Image<uint8_t> idx;
        mapper_fused.getDepthMapFromDSI(depth, conf, idx);
        const std::vector<float> vol = mapper_fused.dsi_.download();
        std::vector<float> rconf(conf.data.size()), rdepth(conf.data.size());
        std::vector<uint8_t> ridx(conf.data.size());
        orc_collapse_max_z(vol.data(), cam.width, cam.height, (int)dsi_shape.dimZ_, rconf.data(), ridx.data());
        orc_indices_to_depth(ridx.data(), ridx.size(), planes.data(), rdepth.data());
        if (conf.data != rconf || idx.data != ridx || depth.data != rdepth) return 40;
        dsi::Image<float> c2;
        dsi::Image<uint8_t> i2;
        mapper_fused.dsi_.collapseMaxZSlice(&c2, &i2);
        if (c2.data != rconf || i2.data != ridx) return 41;
        // the full extraction with the reference's signature (adaptive threshold, median, border)
        {
            EMVS::OptionsDepthMap opts;
            opts.adaptive_threshold_c_ = 4.;
            opts.max_confidence = 50.;
            dsi::Image<float> dm, cm;
            dsi::Image<uint8_t> mk;
            mapper_fused.getDepthMapFromDSI(dm, cm, mk, opts);
            std::vector<float> oc = rconf, od(rconf.size());
            std::vector<uint8_t> c8(rconf.size()), om(rconf.size()), of(rconf.size());
            orc_depth_map_filters(oc.data(), ridx.data(), cam.width, cam.height, 5, 4., 5, 50., planes.data(),
                                  c8.data(), om.data(), of.data(), od.data());
            if (dm.data != od || cm.data != oc || mk.data != om) return 42;
        }
        // mismatched grids: the reference throws std::out_of_range from .at()
        Grid3D other(ctx, 8, 8, 8);
        try {
            mapper_fused.dsi_.addTwoGrids(other);
            return 50;
        } catch (const dsi::Error& e) {
            if (e.code != DSI_ERR_SHAPE) return 51;
        }
        // ---- process_1 / process_2 / process_5 as functions (include/dsi_process.hpp) against the
        //      same orchestration done with the oracle
        {
            EMVS::MapperEMVS mapper2(ctx, cam, dsi_shape);
            const std::vector<dsi::Event> none;
            const dsi::Transformation T1 = process_1(trajectory0, trajectory1, trajectory1, events0, events1, none,
                                                     mapper_fused, mapper0, mapper1, mapper2, 0.5, 3);
            if (std::fabs(T1.t[0] - T_rv_w.t[0]) > 1e-12) return 60;
            std::vector<float> a = ref0;
            orc_fuse2(a.data(), ref1.data(), a.size(), 3);
            if (max_rel_err(mapper_fused.dsi_.download(), a) > 3e-4) return 61;
            // the depth-map-only form (one fused kernel, no DSI written) gives the depth map of process_1 +
            // getDepthMapFromDSI bit for bit, for every fusion method
            for (int method = 1; method <= 6; ++method) {
                process_1(trajectory0, trajectory1, trajectory1, events0, events1, none, mapper_fused, mapper0, mapper1, mapper1,
                          0.5, method);
                dsi::Image<float> d1, c1, d2, c2;
                dsi::Image<uint8_t> i1, i2;
                mapper_fused.getDepthMapFromDSI(d1, c1, i1);
                EMVS::MapperEMVS cam_a(ctx, cam, dsi_shape), cam_b(ctx, cam, dsi_shape);
                process_1_depth_map(trajectory0, trajectory1, events0, events1, mapper_fused, cam_a, cam_b, 0.5, method, d2, c2, i2);
                if (d1.data != d2.data || c1.data != c2.data || i1.data != i2.data) {
                    std::printf("process_1_depth_map differs for fusion method %d\n", method);
                    return 64;
                }
                // main.cpp:281: the window ends in getDepthMapFromDSI(depth, confidence, mask, options); after
                // process_1_depth_map there is no DSI, filterDepthMap runs the same filters on the device-held arg-max
                EMVS::OptionsDepthMap opts;
                dsi::Image<float> fd1, fc1, fd2, fc2;
                dsi::Image<uint8_t> m1, m2;
                mapper_fused.getDepthMapFromDSI(fd1, fc1, m1, opts);   // process_1 left the fused DSI in mapper_fused
                process_1_depth_map(trajectory0, trajectory1, events0, events1, cam_a, mapper0, mapper1, 0.5, method, d2, c2, i2);
                cam_a.filterDepthMap(fd2, fc2, m2, opts);
                if (std::memcmp(fd1.data.data(), fd2.data.data(), fd1.data.size() * sizeof(float)) != 0 ||
                    fc1.data != fc2.data || m1.data != m2.data) {
                    std::printf("filterDepthMap differs for fusion method %d\n", method);
                    return 65;
                }
            }
            // the exact tie resolver: the index map of process_1_exact_depth_map IS the oracle's arg-max of the oracle's
            // fused volume (first maximum, cartesian3dgrid.cpp:132-134), on every pixel, for every fusion method
            for (int method = 1; method <= 6; ++method) {
                dsi::Image<float> d3, c3;
                dsi::Image<uint8_t> i3;
                dsi_resolve_info_t rinfo{};
                process_1_exact_depth_map(trajectory0, trajectory1, events0, events1, mapper_fused, mapper0, mapper1, 0.5, method, d3,
                                          c3, i3, &rinfo);
                std::vector<float> of = ref0;
                for (float& v : of) v = 0.f + v;
                orc_fuse2(of.data(), ref1.data(), of.size(), method);
                const size_t npix = i3.data.size();
                const size_t planes_n = of.size() / npix;
                size_t differ = 0;
                for (size_t p = 0; p < npix; ++p) {
                    size_t best = 0;
                    for (size_t z = 1; z < planes_n; ++z)
                        if (of[best * npix + p] < of[z * npix + p]) best = z;
                    differ += (size_t)i3.data[p] != best;
                }
                if (differ != 0 || rinfo.near_tie_pixels <= 0) {
                    std::printf("process_1_exact_depth_map: %zu pixels differ from the oracle for fusion method %d (%d near-tie columns)\n",
                                differ, method, rinfo.near_tie_pixels);
                    return 67;
                }
                // PROVEN mode (ABI 10): the same map, and every column's premise proven from the counted votes
                dsi::Image<float> d4, c4;
                dsi::Image<uint8_t> i4;
                dsi_resolve_info_t rinfo4{};
                dsi_prove_info_t proof{};
                process_1_exact_depth_map(trajectory0, trajectory1, events0, events1, mapper_fused, mapper0, mapper1, 0.5, method, d4,
                                          c4, i4, &rinfo4, 0.0, &proof);
                if (i4.data != i3.data || proof.columns != (long long)npix || proof.columns_unproven != proof.columns_resolved_fully ||
                    proof.columns_proven + proof.columns_unproven != (long long)npix || proof.max_votes <= 0) {
                    std::printf("proven mode, fusion method %d: %lld of %lld columns proven, %lld re-summed fully (gap %g, needed %g), "
                                "index maps %s\n", method, proof.columns_proven, proof.columns, proof.columns_resolved_fully,
                                (double)proof.rel_gap, proof.gap_needed, i4.data == i3.data ? "equal" : "DIFFER");
                    return 71;
                }
            }
            // three cameras (process1.cpp:105-117, :169-191): the right camera's second half of events plays camera 2
            {
                const std::vector<dsi::Event> events2(events1.begin() + (long)(events1.size() / 2), events1.end());
                EMVS::MapperEMVS out3(ctx, cam, dsi_shape);
                for (int method = 1; method <= 6; ++method) {
                    process_1(trajectory0, trajectory1, trajectory1, events0, events1, events2, mapper_fused, mapper0, mapper1,
                              mapper2, 0.5, method);
                    dsi::Image<float> d1, c1, d2, c2;
                    dsi::Image<uint8_t> i1, i2;
                    mapper_fused.getDepthMapFromDSI(d1, c1, i1);
                    process_1_depth_map(trajectory0, trajectory1, trajectory1, events0, events1, events2, out3, mapper0, mapper1,
                                        mapper2, 0.5, method, d2, c2, i2);
                    if (d1.data != d2.data || c1.data != c2.data || i1.data != i2.data) {
                        std::printf("three-camera process_1_depth_map differs for fusion method %d\n", method);
                        return 66;
                    }
                }
            }
            // four cameras (BASELINE configs[4]'s rig; process_1 has no switch for it): the DSI-less kernel with the geometric-mean
            // tree == evaluateDSI x 4 + the tree inside the arg-max kernel (dsi_mapper_depth_map_of_fusion_n), bit for bit
            {
                const std::vector<dsi::Event> ev2(events1.begin() + (long)(events1.size() / 2), events1.end());
                const std::vector<dsi::Event> ev3(events0.begin() + (long)(events0.size() / 3), events0.end());
                EMVS::MapperEMVS mapper3(ctx, cam, dsi_shape), out4(ctx, cam, dsi_shape);
                const LinearTrajectory* trs[4] = {&trajectory0, &trajectory1, &trajectory1, &trajectory0};
                const std::vector<dsi::Event>* evs[4] = {&events0, &events1, &ev2, &ev3};
                EMVS::MapperEMVS* ms4[4] = {&mapper0, &mapper1, &mapper2, &mapper3};
                dsi::Image<float> d4, c4;
                dsi::Image<uint8_t> i4;
                const dsi::Transformation T4 = process_1_depth_map_n(trs, evs, ms4, 4, out4, 0.5, DSI_FUSE_GM, d4, c4, i4);
                const dsi_grid_t* gs[4];
                for (int c = 0; c < 4; ++c) {
                    ms4[c]->evaluateDSI(*evs[c], *trs[c], T4);
                    gs[c] = ms4[c]->dsi_.handle();
                }
                dsi::check(dsi_mapper_depth_map_of_fusion_n(mapper_fused.handle(), gs, 4, DSI_ACC_GM_TREE));
                dsi::Image<float> d5(d4.rows, d4.cols), c5(d4.rows, d4.cols);
                dsi::Image<uint8_t> i5(d4.rows, d4.cols);
                dsi::check(dsi_mapper_fetch_depth_map(mapper_fused.handle(), d5.data.data(), c5.data.data(), i5.data.data()));
                bool any = false;
                for (float v : c4.data) any = any || v > 0.f;
                if (d4.data != d5.data || c4.data != c5.data || i4.data != i5.data || !any) {
                    std::printf("four-camera process_1_depth_map_n differs from evaluateDSI x 4 + the GM tree\n");
                    return 68;
                }
                try {
                    process_1_depth_map_n(trs, evs, ms4, 4, out4, 0.5, DSI_FUSE_HM, d4, c4, i4);
                    return 69;
                } catch (const dsi::Error& e) {
                    if (e.code != DSI_ERR_BAD_OP) return 69;
                }
            }
            try {
                process_1(trajectory0, trajectory1, trajectory1, events0, events1, none, mapper_fused, mapper0,
                          mapper1, mapper2, 0.5, 9);
                return 62;
            } catch (const dsi::Error& e) {
                if (e.code != DSI_ERR_BAD_OP) return 63;
            }
        }
        for (int variant = 0; variant < 2; ++variant) {
            const bool shuffle = variant == 1;
            const int n_sub = 3, stereo = shuffle ? 3 : 2, temporal = shuffle ? 4 : 2;
            EMVS::MapperEMVS fused_ct(ctx, cam, dsi_shape);
            fused_ct.dsi_.resetGrid();
            Process2Result r = shuffle ? process_5(ctx, cam, cam, trajectory0, trajectory1, events0, events1, dsi_shape,
                                                   n_sub, mapper_fused, fused_ct, 0.5, stereo, temporal)
                                       : process_2(ctx, cam, cam, trajectory0, trajectory1, events0, events1, dsi_shape,
                                                   n_sub, mapper_fused, fused_ct, 0.5, stereo, temporal);
            // oracle orchestration (process2.cpp:98-289, process5.cpp:89-150)
            const size_t n = ref0.size();
            std::vector<float> oleft(n, 0.f), oright(n, 0.f), ofused(n, 0.f);
            const size_t per0 = events0.size() / n_sub, per1 = events1.size() / n_sub;
            size_t idx1 = shuffle ? (size_t)(n_sub / 2) * per1 : 0;
            const int mode = temporal == 2 ? 1 : 0;
            for (int k = 0; k < n_sub; ++k) {
                const std::vector<dsi::Event> e0(events0.begin() + k * per0, events0.begin() + (k + 1) * per0);
                std::vector<dsi::Event> e1;
                if (!shuffle) {
                    e1.assign(events1.begin() + k * per1, events1.begin() + (k + 1) * per1);
                } else if (idx1 + per1 >= events1.size()) {
                    e1.assign(events1.begin() + idx1, events1.end());
                    const size_t rest = idx1 + per1 - events1.size();
                    e1.insert(e1.end(), events1.begin(), events1.begin() + rest);
                    idx1 = rest;
                } else {
                    e1.assign(events1.begin() + idx1, events1.begin() + idx1 + per1);
                    idx1 += per1;
                }
                const std::vector<float> d0 = oracle_dsi(e0, trajectory0, r.T_rv_w, cam, dsi_shape, &planes);
                const std::vector<float> d1 = oracle_dsi(e1, trajectory1, r.T_rv_w, cam, dsi_shape, &planes);
                std::vector<float> sub = d0;
                orc_fuse2(sub.data(), d1.data(), n, stereo);
                orc_accumulate(oleft.data(), d0.data(), n, mode);
                orc_accumulate(oright.data(), d1.data(), n, mode);
                orc_accumulate(ofused.data(), sub.data(), n, mode);
            }
            orc_finalize(oleft.data(), n, mode, n_sub);
            orc_finalize(oright.data(), n, mode, n_sub);
            orc_finalize(ofused.data(), n, mode, n_sub);
            std::vector<float> oct = oleft;
            static const int converse[7] = {0, 1, 2, 4, 3, 5, 6};
            orc_fuse2(oct.data(), oright.data(), n, converse[stereo]);
            const double el = max_rel_err(r.left.download(), oleft), er = max_rel_err(r.right.download(), oright);
            const double ef = max_rel_err(mapper_fused.dsi_.download(), ofused);
            const double ec = max_rel_err(fused_ct.dsi_.download(), oct);
            std::printf("%s: left %.3g right %.3g fused %.3g camera-time %.3g\n", shuffle ? "process_5" : "process_2",
                        el, er, ef, ec);
            if (el > 3e-4 || er > 3e-4 || ef > 1e-3 || ec > 1e-3) return 70 + variant;
            if (!shuffle) {
                // Alg. 2 through the resolver's building blocks: the plane index map IS the arg-max (first maximum) of the
                // oracle's time-fused volume, on every pixel
                EMVS::MapperEMVS exact(ctx, cam, dsi_shape);
                dsi::Image<float> de, ce;
                dsi::Image<uint8_t> ie;
                const dsi::ExactDepthMapInfo xi = process_2_exact_depth_map(ctx, cam, cam, trajectory0, trajectory1, events0, events1,
                                                                            dsi_shape, n_sub, exact, 0.5, stereo, temporal, de, ce, ie);
                const size_t npix = ie.data.size(), nplanes = n / npix;
                size_t differ = 0;
                for (size_t p = 0; p < npix; ++p) {
                    size_t best = 0;
                    for (size_t z = 1; z < nplanes; ++z)
                        if (ofused[best * npix + p] < ofused[z * npix + p]) best = z;
                    differ += (size_t)ie.data[p] != best;
                    if ((size_t)ie.data[p] == best && de.data[p] != planes[best]) return 75;
                }
                std::printf("process_2_exact_depth_map: %zu near-tie columns, %zu voxels, %lld votes re-summed, %d pixels changed, %zu differ\n",
                            xi.near_tie_pixels, xi.candidate_voxels, xi.votes, xi.changed_pixels, differ);
                if (differ != 0 || xi.near_tie_pixels == 0) return 74;
                if (exact.dsi_.download() != mapper_fused.dsi_.download()) return 76;  // the same fused DSI as process_2's
            }
        }
        // ---- process_2 over every GPU of the node (one here on the test box, eight under the driver's
        //      multi-GPU run): sub-interval -> device, ONE RCCL all-reduce per accumulator issued by the
        //      engine; must equal the single-device process_2 (to summation order when n > 1)
        {
            int ndev = dsi_device_count();
            if (const char* e = std::getenv("DSI_TEST_DEVICES")) ndev = std::min(ndev, std::atoi(e));
            std::vector<std::unique_ptr<dsi::Context>> owned;
            std::vector<dsi::Context*> ctxs;
            for (int i = 0; i < ndev; ++i) {
                owned.emplace_back(new dsi::Context(i));
                ctxs.push_back(owned.back().get());
            }
            std::vector<dsi::Comm> comms = dsi::Comm::createAll(ctxs);   // also with one device: RCCL is exercised
            if ((int)comms.size() != ndev || comms[0].size() != ndev) return 90;
            const int n_sub = 8;
            EMVS::MapperEMVS fused_ct(ctx, cam, dsi_shape);
            fused_ct.dsi_.resetGrid();
            Process2Result single = process_2(ctx, cam, cam, trajectory0, trajectory1, events0, events1, dsi_shape, n_sub,
                                              mapper_fused, fused_ct, 0.5, 2, 2);
            Process2MultiResult multi = process_2_multi_gpu(ctxs, comms, cam, cam, trajectory0, trajectory1, events0,
                                                            events1, dsi_shape, n_sub, 0.5, 2, 2);
            const std::vector<float> want = mapper_fused.dsi_.download();
            for (int i = 0; i < ndev; ++i) {
                const double e1 = max_rel_err(multi.fused[i].download(), want);
                const double e2 = max_rel_err(multi.left[i].download(), single.left.download());
                std::printf("process_2_multi_gpu device %d of %d: fused %.3g left %.3g\n", i, ndev, e1, e2);
                if (e1 > (ndev > 1 ? 1e-5 : 0.0) || e2 > (ndev > 1 ? 1e-5 : 0.0)) return 91;
            }
            if (max_rel_err(multi.camera_time.download(), fused_ct.dsi_.download()) > (ndev > 1 ? 1e-5 : 0.0)) return 92;
            // a plain grid all-reduce: sum over n copies of the same volume = n * volume
            std::vector<Grid3D> gs(ndev);
            std::vector<dsi_grid_t*> gh;
            std::vector<dsi_comm_t*> ch;
            int gx, gy, gz;
            mapper_fused.dsi_.getDimensions(&gx, &gy, &gz);
            for (int i = 0; i < ndev; ++i) {
                gs[i].allocate(*ctxs[i], gx, gy, gz);
                gs[i].upload(want);
                gh.push_back(gs[i].handle());
                ch.push_back(comms[i].handle());
            }
            dsi::check(dsi_grid_allreduce_all(ch.data(), gh.data(), ndev, DSI_REDUCE_SUM));
            std::vector<float> scaled = want;
            for (float& v : scaled) v *= (float)ndev;
            for (int i = 0; i < ndev; ++i)
                if (max_rel_err(gs[i].download(), scaled) > 1e-6) return 93;
        }
        // ---- the --full_seq loop as a stream (main.cpp:177-302, process_method 1): every window's depth map equals
        //      process_1_depth_map on the window's events, bit for bit, whatever the number of windows in flight;
        //      windows arrive in order; a window with < 1024 events of one camera still gives a (defined) map
        {
            struct Got {
                int index;
                double ts;
                size_t n0, n1;
                std::vector<float> depth, conf;
                std::vector<uint8_t> idx;
            };
            const double start = 0.0, stop = 0.97, duration = 0.4, skip = 0.14;  // windows [0, .4], [.14, .54], ..., [.56, .96]
            std::vector<Got> runs[3];
            const int depths[3] = {1, 2, 3};
            for (int r = 0; r < 3; ++r) {
                const size_t n = dsi::full_sequence_depth_maps(
                    0, cam, cam, dsi_shape, trajectory0, trajectory1, events0, events1, start, stop, duration, skip,
                    /*forward_looking=*/false, /*fusion_method=*/2,
                    [&](const dsi::WindowDepthMap& w) {
                        runs[r].push_back(Got{w.index, w.ts, w.n_events[0], w.n_events[1], w.depth_map.data, w.confidence_map.data,
                                              w.depth_cell_indices.data});
                    },
                    depths[r]);
                if (n != 5 || runs[r].size() != 5) {
                    std::printf("full_sequence_depth_maps: %zu windows, %zu delivered (expected 5)\n", n, runs[r].size());
                    return 100;
                }
                for (int i = 0; i < 5; ++i)
                    if (runs[r][i].index != i) return 101;  // in order
            }
            for (int r = 1; r < 3; ++r)
                for (int i = 0; i < 5; ++i)
                    if (runs[r][i].depth != runs[0][i].depth || runs[r][i].conf != runs[0][i].conf || runs[r][i].idx != runs[0][i].idx)
                        return 102;  // the number of windows in flight does not change a bit
            EMVS::MapperEMVS cam_a(ctx, cam, dsi_shape), cam_b(ctx, cam, dsi_shape), out_w(ctx, cam, dsi_shape);
            double t = start;
            for (int i = 0; i < 5; ++i, t += skip) {
                size_t a0, b0, a1, b1;
                dsi::window_event_range(events0, t, t + duration, &a0, &b0);
                dsi::window_event_range(events1, t, t + duration, &a1, &b1);
                if (b0 - a0 != runs[0][i].n0 || b1 - a1 != runs[0][i].n1 || b0 - a0 < 3000) return 103;
                const std::vector<dsi::Event> w0(events0.begin() + (long)a0, events0.begin() + (long)b0);
                const std::vector<dsi::Event> w1(events1.begin() + (long)a1, events1.begin() + (long)b1);
                dsi::Image<float> d, c;
                dsi::Image<uint8_t> ix;
                const double ts = (t + (t + duration)) / 2;
                if (ts != runs[0][i].ts) return 104;
                process_1_depth_map(trajectory0, trajectory1, w0, w1, out_w, cam_a, cam_b, ts, 2, d, c, ix);
                if (d.data != runs[0][i].depth || c.data != runs[0][i].conf || ix.data != runs[0][i].idx) {
                    std::printf("full_sequence_depth_maps: window %d differs from process_1_depth_map\n", i);
                    return 105;
                }
            }
            // the device-list form (slot k on devices[k % n]; here every slot on device 0) gives the same windows
            {
                std::vector<Got> again;
                dsi::full_sequence_depth_maps(std::vector<int>{0, 0}, cam, cam, dsi_shape, trajectory0, trajectory1, events0, events1,
                                              start, stop, duration, skip, false, 2,
                                              [&](const dsi::WindowDepthMap& w) {
                                                  again.push_back(Got{w.index, w.ts, w.n_events[0], w.n_events[1], w.depth_map.data,
                                                                      w.confidence_map.data, w.depth_cell_indices.data});
                                              },
                                              4);
                if (again.size() != 5) return 108;
                for (int i = 0; i < 5; ++i)
                    if (again[i].index != i || again[i].depth != runs[0][i].depth || again[i].idx != runs[0][i].idx) return 109;
            }
            // forward-looking reference view (main.cpp:184-185) and a right camera that has too few events in the windows
            const std::vector<dsi::Event> few(events1.begin(), events1.begin() + 500);
            int seen = 0;
            dsi::full_sequence_depth_maps(0, cam, cam, dsi_shape, trajectory0, trajectory1, events0, few, 0.1, 0.95, 0.4, 0.4, true, 6,
                                          [&](const dsi::WindowDepthMap& w) {
                                              ++seen;
                                              if (w.ts != w.t_stop) seen = -100;
                                              dsi::Image<float> d, c;
                                              dsi::Image<uint8_t> ix;
                                              size_t a0, b0, a1, b1;
                                              dsi::window_event_range(events0, w.t_start, w.t_stop, &a0, &b0);
                                              dsi::window_event_range(few, w.t_start, w.t_stop, &a1, &b1);
                                              const std::vector<dsi::Event> w0(events0.begin() + (long)a0, events0.begin() + (long)b0);
                                              const std::vector<dsi::Event> w1(few.begin() + (long)a1, few.begin() + (long)b1);
                                              process_1_depth_map(trajectory0, trajectory1, w0, w1, out_w, cam_a, cam_b, w.ts, 6, d, c, ix);
                                              if (d.data != w.depth_map.data || ix.data != w.depth_cell_indices.data) seen = -100;
                                          });
            if (seen != 2) return 106;
            // main.cpp:281: the filtered outputs of a window = filterDepthMap after process_1_depth_map on its events
            {
                EMVS::OptionsDepthMap opts;
                opts.adaptive_threshold_c_ = 4.;
                int bad = 0, n_w = 0;
                dsi::full_sequence_depth_maps(
                    0, cam, cam, dsi_shape, trajectory0, trajectory1, events0, events1, 0.1, 0.95, 0.4, 0.4, false, 2,
                    [&](const dsi::WindowDepthMap& w) {
                        ++n_w;
                        size_t a0, b0, a1, b1;
                        dsi::window_event_range(events0, w.t_start, w.t_stop, &a0, &b0);
                        dsi::window_event_range(events1, w.t_start, w.t_stop, &a1, &b1);
                        const std::vector<dsi::Event> w0(events0.begin() + (long)a0, events0.begin() + (long)b0);
                        const std::vector<dsi::Event> w1(events1.begin() + (long)a1, events1.begin() + (long)b1);
                        dsi::Image<float> d, c, fd, fc;
                        dsi::Image<uint8_t> ix, mk;
                        process_1_depth_map(trajectory0, trajectory1, w0, w1, out_w, cam_a, cam_b, w.ts, 2, d, c, ix);
                        out_w.filterDepthMap(fd, fc, mk, opts);
                        if (std::memcmp(fd.data.data(), w.filtered_depth_map.data.data(), fd.data.size() * sizeof(float)) != 0 ||
                            fc.data != w.filtered_confidence_map.data || mk.data != w.semidense_mask.data)
                            ++bad;
                        size_t kept = 0;
                        for (uint8_t v : mk.data) kept += v != 0;
                        if (kept == 0) ++bad;  // a mask that keeps nothing would prove nothing
                    },
                    2, 0.0, &opts);
                if (bad || n_w != 2) return 107;
            }
        }
        if (const char* out = std::getenv("DSI_TEST_NPY")) {  // for tests/test_cpp_adapter.py
            if (mapper_fused.dsi_.writeGridNpy(out) != 0) return 80;
            double sum = 0;
            for (float v : mapper_fused.dsi_.download()) sum += v;
            std::printf("npy sum %.9e\n", sum);
        }
        std::printf("process_1 / process_2 / process_5 through the C++ adapter: OK\n");
        return 0;
    } catch (const dsi::Error& e) {
        std::printf("dsi::Error %d: %s\n", e.code, e.what());
        return e.code == DSI_ERR_NO_DEVICE ? 3 : 2;
    }
}
