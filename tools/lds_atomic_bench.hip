// Microbenchmark: LDS atomic / read / write rates on gfx950 for the address patterns the
// DSI voting kernel produces.  Build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o /tmp/ldsb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int kWords = 20000;   // 160 KB as u64; ~58 rows of 346 floats

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int OP, int PAT>
__global__ __launch_bounds__(1024) void k(int iters, float* out)
{
    extern __shared__ unsigned char raw[];
    float* f = reinterpret_cast<float*>(raw);
    uint32_t* u = reinterpret_cast<uint32_t*>(raw);
    unsigned long long* u64 = reinterpret_cast<unsigned long long*>(raw);
    double* f64 = reinterpret_cast<double*>(raw);
    for (int i = threadIdx.x; i < kWords * 2; i += blockDim.x) u[i] = 0;
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        int a;
        if (PAT == 0) a = (it * 64 + lane) % kWords;            // consecutive words: conflict free
        else if (PAT == 1) a = lcg(s) % kWords;                 // uniform random
        else if (PAT == 2) a = (it * 7) % kWords;               // all lanes same address
        else a = ((lcg(s) % 64) * 346 + (lcg(s) % 346)) % kWords; // random within 64 rows
        if (OP == 0) __hip_atomic_fetch_add(&f[a], 0.25f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (OP == 1) __hip_atomic_fetch_add(&u[a], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (OP == 2) __hip_atomic_fetch_add(&u64[a], 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (OP == 3) __hip_atomic_fetch_add(&f64[a], 0.25, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (OP == 4) { f[a] = 0.25f + it; }                  // plain write
        else if (OP == 5) { acc += f[a]; }                        // plain read
        else if (OP == 6) { float v = f[a]; f[a] = v + 0.25f; }   // non-atomic RMW
        else if (OP == 7) { acc += __hip_atomic_fetch_add(&f[a], 0.25f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } // rtn
        else if (OP == 8) { __hip_atomic_fetch_max(&u[a], (uint32_t)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    }
    __syncthreads();
    if (acc == 123.456f) out[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = f[5];
}

template <int OP, int PAT>
void run(const char* name, float* d_out)
{
    const int iters = 4000, blocks = 256, threads = 1024;
    const size_t lds = kWords * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<OP, PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<OP, PAT>), dim3(blocks), dim3(threads), lds, 0, 100, d_out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP, PAT>), dim3(blocks), dim3(threads), lds, 0, iters, d_out);
    hipEventRecord(b); hipEventSynchronize(b);
    { hipError_t e = hipGetLastError(); if (e != hipSuccess) printf("ERR %s\n", hipGetErrorString(e)); }
    float ms; hipEventElapsedTime(&ms, a, b);
    // wave-instructions per CU = iters * 16 waves
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-34s %8.3f ms  %7.1f cyc/wave-instr/CU  %6.2f lanes/clk/CU\n", name, ms, cyc / (iters * 16.0), iters * 1024.0 / cyc);
}

int main()
{
    float* d_out; hipMalloc(&d_out, 64);
    const char* pats[4] = {"linear", "random", "same-addr", "rand64rows"};
#define RUNALL(OP, label) \
    run<OP, 0>(label " linear", d_out); run<OP, 1>(label " random", d_out); run<OP, 2>(label " same-addr", d_out); run<OP, 3>(label " rand64rows", d_out);
    RUNALL(0, "ds_add_f32")
    RUNALL(7, "ds_add_rtn_f32")
    RUNALL(1, "ds_add_u32")
    RUNALL(2, "ds_add_u64")
    RUNALL(3, "ds_add_f64")
    RUNALL(8, "ds_max_u32")
    RUNALL(4, "ds_write_b32")
    RUNALL(5, "ds_read_b32")
    RUNALL(6, "read+add+write")
    (void)pats;
    return 0;
}
