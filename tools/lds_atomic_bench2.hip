// LDS-bound microbenchmark: 8 precomputed random addresses per lane, re-used every
// iteration with a small rotating offset, so that the loop is a pure stream of DS ops.
// Patterns mimic the voting kernel: cells of a (28 x 346) band.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kRows = 28, kNx = 346, kCells = kRows * kNx;  // 9688 cells

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int OP, int PAT>
__global__ __launch_bounds__(1024) void k(int iters, float* out)
{
    extern __shared__ unsigned char raw[];
    uint32_t* u = reinterpret_cast<uint32_t*>(raw);
    unsigned long long* u64 = reinterpret_cast<unsigned long long*>(raw);
    float* f = reinterpret_cast<float*>(raw);
    for (int i = threadIdx.x; i < kCells * 2 + 64; i += blockDim.x) u[i] = 0;
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const int lane = threadIdx.x & 63;
    int a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (PAT == 0) a[j] = (j * 64 + lane) % kCells;                 // linear
        else if (PAT == 1) a[j] = lcg(s) % (kCells - 8);               // random cell in band
        else if (PAT == 2) a[j] = ((lcg(s) % 3) + (lane / 8)) * kNx + lcg(s) % (kNx - 8);  // sorted-ish rows, random x
        else if (PAT == 3) a[j] = (lcg(s) % kRows) * kNx + ((lane & 15) * 21 + (lcg(s) % 16) * 16) % (kNx - 8);  // distinct x mod 16 per 16-lane group
        else {
            // "bank-aware order": the 32 lanes of a half-wave sit in ONE row (same for the half-wave:
            // seeded by wave id, half and j) and have distinct x mod 32; PAT 5 adds the distortion of
            // a plane transfer X = 1.02 * x0 (up to 7 cells across the row), PAT 6 a random +-3
            uint32_t hs = (threadIdx.x >> 5) * 7919u + j * 104729u + blockIdx.x * 31u + 17u;
            const int row = lcg(hs) % kRows;
            int x = (lane & 31) + 32 * (int)(lcg(s) % 10);             // 0..319, residue = lane & 31
            if (PAT == 5) x = (int)(1.02f * (float)x);
            if (PAT == 6) x += (int)(lcg(s) % 7) - 3;
            if (PAT == 7) x = (int)(1.005f * (float)x);
            x = x < 0 ? 0 : (x > kNx - 10 ? kNx - 10 : x);
            a[j] = row * kNx + x;
        }
    }
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const int o = it & 7;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = a[j] + o;
            if (OP == 0) __hip_atomic_fetch_add(&u64[idx], 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 1) __hip_atomic_fetch_add(&u[idx], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 2) acc += __hip_atomic_fetch_add(&u[idx], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 3) __hip_atomic_fetch_add(&f[idx], 0.25f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 4) { u[idx] = it; }
            else if (OP == 5) { acc += u[idx]; }
            else if (OP == 6) { u64[idx] = it; }
        }
    }
    __syncthreads();
    if (acc == 0x12345u) out[0] = 1.f;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)u[5];
}

template <int OP, int PAT>
void run(const char* name, float* d_out, int threads)
{
    const int iters = 2000, blocks = 256 * (1024 / threads);
    const size_t lds = (kCells * 2 + 64) * 4 + 64;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<OP, PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<OP, PAT>), dim3(blocks), dim3(threads), lds, 0, 50, d_out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<OP, PAT>), dim3(blocks), dim3(threads), lds, 0, iters, d_out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    hipError_t e = hipGetLastError(); if (e != hipSuccess) printf("ERR %s\n", hipGetErrorString(e));
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 1e-3 * 2.3e9;            // ~clock under load
    const double winstr = iters * 8.0 * 16.0;          // wave instructions per CU (256 blocks of 1024 threads = 16 waves per CU)
    printf("%-40s %8.3f ms  %6.2f cyc/wave-instr/CU\n", name, ms, cyc / winstr);
}

int main()
{
    float* d_out; (void)hipMalloc(&d_out, 64);
#define RUNALL(OP, label) \
    run<OP, 0>(label " linear", d_out, 1024); run<OP, 1>(label " random-in-band", d_out, 1024); \
    run<OP, 2>(label " rows-sorted", d_out, 1024); run<OP, 3>(label " x-mod16-distinct", d_out, 1024); \
    run<OP, 4>(label " halfwave-row x-mod32-distinct", d_out, 1024); run<OP, 5>(label " ... scaled 1.02", d_out, 1024); \
    run<OP, 6>(label " ... jitter +-3", d_out, 1024); run<OP, 7>(label " ... scaled 1.005", d_out, 1024);
    RUNALL(0, "ds_add_u64")
    RUNALL(1, "ds_add_u32")
    RUNALL(2, "ds_add_rtn_u32")
    run<3, 1>("ds_add_f32 random-in-band", d_out, 1024);
    RUNALL(4, "ds_write_b32")
    RUNALL(6, "ds_write_b64")
    return 0;
}
