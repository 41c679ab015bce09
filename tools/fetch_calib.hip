// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE report for THIS engine's access widths?
// MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE is exactly half the bytes of a wide coalesced
// 16-B-per-lane stream, "other access widths ... are uncalibrated: calibrate on a known byte count in your own
// access pattern".  The voting kernel reads 12-byte records (global_load_dwordx3) in runs of ~20-100 consecutive
// records and 32-byte coefficient sets per lane; this tool streams buffers far larger than the 256 MiB Infinity
// Cache with exactly those accesses, so that the bytes that must cross the fabric are known:
//   k_stream16   16 B per lane, consecutive            (the guide's reference case)
//   k_stream12   12 B per lane, consecutive records    (a run of records)
//   k_runs12     12 B per lane, runs of RUN consecutive records at scattered places (every record read once)
//   k_coef32     16 + 4 B of a 32-byte set per lane, consecutive sets (the coefficient gathers)
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o calib -- tools/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

struct Rec { float x, y; uint32_t m; };

__global__ void k_stream16(const float4* __restrict__ p, size_t n, float* out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ void k_stream12(const Rec* __restrict__ p, size_t n, float* out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const Rec v = p[i];
        acc += v.x + v.y + (float)v.m;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// wave w of the grid reads run r = perm-like(w, pass): RUN consecutive records starting at a scattered multiple of RUN
template <int RUN>
__global__ void k_runs12(const Rec* __restrict__ p, size_t n_runs, float* out)
{
    float acc = 0.f;
    const size_t lane = threadIdx.x & 63, wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave; r < n_runs; r += waves) {
        const size_t run = (r * 2654435761ull) % n_runs;  // a permutation when n_runs is a power of two ... odd multiplier
        for (size_t k = lane; k < RUN; k += 64) {
            const Rec v = p[run * RUN + k];
            acc += v.x + v.y + (float)v.m;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ void k_coef32(const uint4* __restrict__ p, size_t n_sets, float* out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_sets; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 a = p[2 * i];
        const uint32_t r = *reinterpret_cast<const uint32_t*>(p + 2 * i + 1);
        acc += __uint_as_float(a.x) + __uint_as_float(a.w) + __uint_as_float(r);
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main()
{
    const size_t bytes = (size_t)3 << 30;  // 3 GiB: 12x the Infinity Cache
    void* buf = nullptr;
    float* out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    hipMemset(buf, 0, bytes);
    const dim3 grid(256 * 8), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_stream16, grid, block, 0, 0, (const float4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(k_stream12, grid, block, 0, 0, (const Rec*)buf, bytes / 12, out);
        hipLaunchKernelGGL(k_runs12<64>, grid, block, 0, 0, (const Rec*)buf, (size_t)1 << 22, out);   // 2^22 runs x 64 x 12 B = 3 GiB
        hipLaunchKernelGGL(k_runs12<128>, grid, block, 0, 0, (const Rec*)buf, (size_t)1 << 21, out);
        hipLaunchKernelGGL(k_coef32, grid, block, 0, 0, (const uint4*)buf, bytes / 32, out);
    }
    hipDeviceSynchronize();
    std::printf("bytes that must cross the fabric per launch: stream16 %zu, stream12 %zu, runs12<64> %zu, runs12<128> %zu, "
                "coef32 %zu (whole 32-byte sets: the 12 unread bytes share their lines)\n",
                bytes, (bytes / 12) * 12, ((size_t)1 << 22) * 64 * 12, ((size_t)1 << 21) * 128 * 12, bytes);
    hipFree(buf);
    hipFree(out);
    return 0;
}
