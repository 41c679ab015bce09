// Issue rate of individual VALU instructions on gfx950: wave-instructions per clock per CU with
// 32 resident waves, four independent dependency chains per wave.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate_bench.hip -o tools/valu_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>

#define BENCH(NAME, ASM4, CLOB)                                                                    \
    __global__ __launch_bounds__(1024) void NAME(int iters, float* out)                             \
    {                                                                                               \
        typedef float v2f __attribute__((ext_vector_type(2)));                                      \
        v2f a0 = {(float)threadIdx.x, 1.5f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;           \
        unsigned long long q0 = threadIdx.x, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3;                 \
        unsigned m = threadIdx.x | 1u;                                                              \
        float f0 = threadIdx.x, f1 = f0 + 1.f, f2 = f0 + 2.f, f3 = f0 + 3.f;                        \
        for (int it = 0; it < iters; ++it) {                                                        \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) { asm volatile(ASM4                      \
                : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3),   \
                  "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)                                            \
                : "v"(m) : CLOB); }                                                                 \
        }                                                                                           \
        if (f0 + f1 + f2 + f3 + a0.x + a1.x + a2.x + a3.x + a0.y + a1.y + a2.y + a3.y == 123.f ||                       \
            q0 + q1 + q2 + q3 == 12345)                                                             \
            out[0] = 1.f;                                                                           \
    }

BENCH(k_fma, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1", "memory")
BENCH(k_mul, "v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0", "memory")
BENCH(k_pk_mul, "v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %6\n v_pk_mul_f32 %6, %6, %7\n v_pk_mul_f32 %7, %7, %4", "memory")
BENCH(k_pk_fma, "v_pk_fma_f32 %4, %4, %5, %6\n v_pk_fma_f32 %5, %5, %6, %7\n v_pk_fma_f32 %6, %6, %7, %4\n v_pk_fma_f32 %7, %7, %4, %5", "memory")
BENCH(k_pk_add, "v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %6\n v_pk_add_f32 %6, %6, %7\n v_pk_add_f32 %7, %7, %4", "memory")
BENCH(k_mad64, "v_mad_u64_u32 %8, vcc, %0, %12, %8\n v_mad_u64_u32 %9, vcc, %1, %12, %9\n v_mad_u64_u32 %10, vcc, %2, %12, %10\n v_mad_u64_u32 %11, vcc, %3, %12, %11", "vcc")
BENCH(k_mullo, "v_mul_lo_u32 %0, %0, %12\n v_mul_lo_u32 %1, %1, %12\n v_mul_lo_u32 %2, %2, %12\n v_mul_lo_u32 %3, %3, %12", "memory")
BENCH(k_mul24, "v_mul_u32_u24 %0, %0, %12\n v_mul_u32_u24 %1, %1, %12\n v_mul_u32_u24 %2, %2, %12\n v_mul_u32_u24 %3, %3, %12", "memory")
BENCH(k_mulhi24, "v_mul_hi_u32_u24 %0, %0, %12\n v_mul_hi_u32_u24 %1, %1, %12\n v_mul_hi_u32_u24 %2, %2, %12\n v_mul_hi_u32_u24 %3, %3, %12", "memory")
BENCH(k_cvtu, "v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3", "memory")
BENCH(k_floor, "v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3", "memory")
BENCH(k_fract, "v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3", "memory")
BENCH(k_or3, "v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %1, %1, %2, %3\n v_or3_b32 %2, %2, %3, %0\n v_or3_b32 %3, %3, %0, %1", "memory")
BENCH(k_add3, "v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %0\n v_add3_u32 %3, %3, %0, %1", "memory")
BENCH(k_lshladd64, "v_lshl_add_u64 %8, %8, 3, %9\n v_lshl_add_u64 %9, %9, 3, %10\n v_lshl_add_u64 %10, %10, 3, %11\n v_lshl_add_u64 %11, %11, 3, %8", "memory")
BENCH(k_mov64, "v_mov_b64 %8, %9\n v_mov_b64 %9, %10\n v_mov_b64 %10, %11\n v_mov_b64 %11, %8", "memory")
BENCH(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc", "memory")
BENCH(k_cnd_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n v_cndmask_b32_e64 %1, %1, %2, s[10:11]\n v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n v_cndmask_b32_e64 %3, %3, %0, s[10:11]", "memory")
BENCH(k_cnd_indep, "v_cndmask_b32 %0, %0, %12, vcc\n v_cndmask_b32 %1, %1, %12, vcc\n v_cndmask_b32 %2, %2, %12, vcc\n v_cndmask_b32 %3, %3, %12, vcc", "memory")
BENCH(k_cmp, "v_cmp_gt_u32 vcc, %0, %1\n v_cmp_gt_u32 vcc, %1, %2\n v_cmp_gt_u32 vcc, %2, %3\n v_cmp_gt_u32 vcc, %3, %0", "vcc")
BENCH(k_cmp_cnd, "v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_gt_u32 vcc, %2, %3\n v_cndmask_b32 %3, %3, %0, vcc", "vcc")
BENCH(k_cmpx, "v_cmpx_gt_u32 exec, %0, %1\n s_mov_b64 exec, -1\n v_cmpx_gt_u32 exec, %2, %3\n s_mov_b64 exec, -1", "memory")
BENCH(k_and_ashr, "v_ashrrev_i32 %0, 31, %1\n v_and_b32 %1, %1, %2\n v_ashrrev_i32 %2, 31, %3\n v_and_b32 %3, %3, %0", "memory")
BENCH(k_bfi, "v_bfi_b32 %0, %0, %1, %2\n v_bfi_b32 %1, %1, %2, %3\n v_bfi_b32 %2, %2, %3, %0\n v_bfi_b32 %3, %3, %0, %1", "memory")
BENCH(k_cvti, "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3", "memory")
BENCH(k_addf, "v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0", "memory")
BENCH(k_subu, "v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_sub_u32 %3, %3, %0", "memory")
BENCH(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3", "memory")
BENCH(k_muli24, "v_mul_i32_i24 %0, %0, %1\n v_mul_i32_i24 %1, %1, %2\n v_mul_i32_i24 %2, %2, %3\n v_mul_i32_i24 %3, %3, %0", "memory")
BENCH(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1", "memory")

template <typename K>
void run(const char* name, K kern, float* d)
{
    const int iters = 8000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(512), dim3(1024), 0, 0, 10, d);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(512), dim3(1024), 0, 0, iters, d);  // 2 blocks per CU = 32 waves
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("ERR %s\n", hipGetErrorString(e));
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 1e-3 * 2.4e9 / iters;  // CU cycles per loop iteration of all 32 waves
    printf("%-12s %8.3f ms   %.2f wave-instr/clk/CU\n", name, ms, 64.0 * 32.0 / cyc);
}

int main()
{
    float* d;
    (void)hipMalloc(&d, 64);
#define RUN(K) run(#K, K, d)
    RUN(k_fma); RUN(k_mul); RUN(k_pk_mul); RUN(k_pk_fma); RUN(k_pk_add); RUN(k_mad64); RUN(k_mullo);
    RUN(k_mul24); RUN(k_mulhi24); RUN(k_cvtu); RUN(k_floor); RUN(k_fract); RUN(k_or3); RUN(k_add3);
    RUN(k_lshladd64); RUN(k_mov64); RUN(k_cndmask); RUN(k_cnd_sgpr); RUN(k_cnd_indep); RUN(k_cmp); RUN(k_cmp_cnd); RUN(k_cmpx); RUN(k_and_ashr); RUN(k_bfi); RUN(k_cvti); RUN(k_addf); RUN(k_subu); RUN(k_rcp); RUN(k_muli24); RUN(k_mad24);
    return 0;
}
