// How many SALU instructions per cycle does a gfx950 CU issue with 32 resident waves?
// Each wave runs a loop of NS scalar + NV vector independent-ish instructions.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NS, int NV>
__global__ __launch_bounds__(1024) void k(int iters, float* out, int seed)
{
    int s0 = __builtin_amdgcn_readfirstlane(seed), s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    float v0 = threadIdx.x, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NS / 4; ++j) {
            asm volatile("s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 3\n s_xor_b32 %2, %2, %0\n s_add_i32 %3, %3, %1"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        }
#pragma unroll
        for (int j = 0; j < NV / 4; ++j) {
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        }
    }
    if (v0 + v1 + v2 + v3 == 123.f || s0 + s1 + s2 + s3 == 12345) out[0] = 1.f;
}

template <int NS, int NV>
void run(float* d)
{
    const int iters = 20000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<NS, NV>), dim3(512), dim3(1024), 0, 0, 100, d, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<NS, NV>), dim3(512), dim3(1024), 0, 0, iters, d, 1);   // 2 blocks per CU = 32 waves
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    { hipError_t e = hipGetLastError(); if (e != hipSuccess) printf("ERR %s\n", hipGetErrorString(e)); }
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 1e-3 * 2.3e9 / iters;   // CU cycles per loop iteration of all 32 waves
    printf("NS=%3d NV=%3d : %.3f ms  %7.1f CU-cycles/iter  => per wave-iter %.2f cycles/CU; SALU/clk/CU %.2f  VALU/clk/CU %.2f\n",
           NS, NV, ms, cyc, cyc / 32.0, NS * 32.0 / cyc, NV * 32.0 / cyc);
}

int main()
{
    float* d; (void)hipMalloc(&d, 64);
    run<40, 0>(d); run<80, 0>(d); run<0, 40>(d); run<0, 80>(d);
    run<40, 40>(d); run<40, 48>(d); run<20, 48>(d); run<8, 48>(d); run<80, 48>(d);
    return 0;
}
