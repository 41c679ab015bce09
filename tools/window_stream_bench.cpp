// The --full_seq loop (main.cpp:177-302, process_method 1) through the C++ adapter as a stream:
// dsi::full_sequence_depth_maps on a synthetic stereo rig at BASELINE configs[2]'s shape -- 50 ms windows of
// 2 x ~500 k events (10 Mevents/s per camera), sensor 640 x 480, DSI 512 x 512 x 200, harmonic-mean fusion --
// fed from std::vector<dsi::Event> in pageable host memory, like the reference holds them.  Prints ms per window
// for 1, 2 and 3 windows in flight.
// build (__graft_entry__.build() does it; the rpath is relative to the binary):
//   g++ -std=c++17 -O2 -pthread tools/window_stream_bench.cpp -Iinclude -Ldvs_mcemvs_amd -ldsi_engine
//       '-Wl,-rpath,$ORIGIN/../dvs_mcemvs_amd' -Wl,-rpath,/opt/rocm/lib -o tools/window_stream_bench
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

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

void make_rig(size_t n, double seconds, double x_off, const dsi::PinholeCameraModel& cam, uint64_t seed,
              std::vector<dsi::Event>* ev, LinearTrajectory::PoseMap* poses)
{
    Lcg rng(seed);
    const int npts = 6000;
    std::vector<double> P(3 * npts);
    for (int i = 0; i < npts; ++i) {
        const double z = 6.0 + 60.0 * rng.uni();
        P[3 * i] = (rng.uni() - 0.5) * 2.2 * z;
        P[3 * i + 1] = (rng.uni() - 0.5) * 1.6 * z;
        P[3 * i + 2] = z;
    }
    const int ctrl = (int)std::ceil(seconds / 0.01) + 20;  // control poses every 10 ms: x = 1.0 t
    for (int k = 0; k < ctrl; ++k) {
        dsi::Transformation T;
        T.t[0] = 1.0 * (0.01 * k - 0.1) + x_off;
        (*poses)[0.01 * k - 0.1] = T;
    }
    ev->clear();
    ev->reserve(n);
    for (size_t k = 0; k < n; ++k) {
        const double t = seconds * (double)k / (double)n;
        const int i = (int)(rng.uni() * npts) % npts;
        const double cx = 1.0 * t + x_off;
        const double u = cam.fx * (P[3 * i] - cx) / P[3 * i + 2] + cam.cx;
        const double v = cam.fy * P[3 * i + 1] / P[3 * i + 2] + cam.cy;
        dsi::Event e;
        e.ts = t;
        if (u < 0 || v < 0 || u >= cam.width - 1 || v >= cam.height - 1 || rng.uni() < 0.1) {
            e.x = (uint16_t)(rng.uni() * cam.width);  // noise event
            e.y = (uint16_t)(rng.uni() * cam.height);
        } else {
            e.x = (uint16_t)std::lround(u);
            e.y = (uint16_t)std::lround(v);
        }
        ev->push_back(e);
    }
}
}  // namespace

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? std::atof(argv[1]) : 1.0;
    const double rate = 10e6, duration = 0.05;
    try {
        dsi::PinholeCameraModel cam;
        cam.width = 640;
        cam.height = 480;
        cam.fx = cam.fy = 320.f;
        cam.cx = 320.f;
        cam.cy = 240.f;
        const EMVS::ShapeDSI shape(512, 512, 200, 4.0f, 200.0f, 0.f);
        std::vector<dsi::Event> events0, events1;
        LinearTrajectory::PoseMap p0, p1;
        const size_t n = (size_t)(rate * seconds);
        make_rig(n, seconds, 0.0, cam, 21, &events0, &p0);
        make_rig(n, seconds, 0.3, cam, 22, &events1, &p1);
        const LinearTrajectory trajectory0(p0), trajectory1(p1);
        std::printf("%zu events per camera over %.2f s; %zu-byte events in std::vector (pageable)\n", n, seconds, sizeof(dsi::Event));
        for (int depth = 1; depth <= 3; ++depth) {
            for (int rep = 0; rep < 2; ++rep) {  // the first repetition warms the device pools and the clocks
                double checksum = 0;
                size_t ev_sum = 0;
                std::vector<std::chrono::steady_clock::time_point> at;
                dsi::WindowStreamStats stats;
                const auto t0 = std::chrono::steady_clock::now();
                const size_t nw = dsi::full_sequence_depth_maps(
                    0, cam, cam, shape, trajectory0, trajectory1, events0, events1, 0.0, seconds - 1e-9, duration, duration,
                    /*forward_looking=*/false, /*fusion_method=*/2,
                    [&](const dsi::WindowDepthMap& w) {
                        checksum += w.confidence_map.data[w.confidence_map.data.size() / 2 + 77] + w.depth_cell_indices.data[1000];
                        ev_sum += w.n_events[0] + w.n_events[1];
                        at.push_back(std::chrono::steady_clock::now());
                    },
                    depth, 0.0, nullptr, &stats);
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                // steady state: between the deliveries of window 2 and of the last window (the call also creates the slots'
                // contexts and mappers -- 3 x 210 MB of DSI each -- which a long sequence amortises)
                const double steady = std::chrono::duration<double, std::milli>(at.back() - at[2]).count() / (double)(nw - 3);
                if (rep == 1)
                    std::printf("depth %d: %zu windows, %.3f ms per window in steady state (%.0f windows/s, %.1f x real time); whole call "
                                "%.1f ms; %.0f events per window, checksum %.6g\n", depth, nw, steady, 1e3 / steady,
                                duration * 1e3 / steady, ms, (double)ev_sum / (double)nw, checksum);
                if (rep == 1)  // one machine-readable line per depth (bench.py: host_fed.cpp_stream)
                    std::printf("JSON {\"depth\": %d, \"windows\": %zu, \"ms_per_window\": %.4f, \"windows_per_s\": %.1f, \"x_real_time\": %.2f, "
                                "\"wait_prepare_ms\": %.4f, \"wait_gpu_ms\": %.4f, \"wait_upload_ms\": %.4f, \"deliver_ms\": %.4f, \"submit_ms\": %.4f}\n",
                                depth, nw, steady, 1e3 / steady, duration * 1e3 / steady, stats.wait_prepare_ms / (double)nw,
                                stats.wait_gpu_ms / (double)nw, stats.wait_upload_ms / (double)nw, stats.deliver_ms / (double)nw,
                                stats.submit_ms / (double)nw);
                if (rep == 1)
                    std::printf("         calling thread per window: wait for the preparation threads %.3f, wait for the GPU %.3f + %.3f (uploads), deliver %.3f, "
                                "submit %.3f ms (whole call %.1f ms)\n", stats.wait_prepare_ms / (double)nw, stats.wait_gpu_ms / (double)nw,
                                stats.wait_upload_ms / (double)nw,
                                stats.deliver_ms / (double)nw, stats.submit_ms / (double)nw, stats.total_ms);
            }
        }
        // the host's share: turning a window's array of structs into the engine's arrays (x, y; timestamps for the pose look-up)
        {
            std::vector<uint16_t> xs(600000), ys(600000);
            std::vector<double> tss(600000);
            const size_t ne = (size_t)(rate * duration);
            const auto t0 = std::chrono::steady_clock::now();
            double sink = 0;
            for (int rep = 0; rep < 20; ++rep) {
                const size_t a = (size_t)rep * ne % (n - ne);
                for (size_t i = 0; i < ne; ++i) {
                    const dsi::Event& e = events0[a + i];
                    xs[i] = e.x;
                    ys[i] = e.y;
                    tss[i] = e.ts;
                }
                sink += xs[ne / 2] + tss[ne / 3];
            }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::printf("host: array-of-structs -> arrays, one camera's window (%zu events): %.3f ms (%g)\n", ne, ms / 20, sink);
        }
    } catch (const std::exception& e) {
        std::printf("failed: %s\n", e.what());
        return 1;
    }
    return 0;
}
