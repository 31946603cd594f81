// What do the vector instructions of the edge kernels cost on one SIMD?  (gfx950 dev probe; one wave per SIMD, 256 workgroups)
//   hipcc -O3 --offload-arch=gfx950 tools/dev/valu_cost.hip -o /tmp/valu_cost && /tmp/valu_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a = threadIdx.x * 0.001f + 1.0f, b = 1.0001f;
    f32x2 p0 = {a, a + 1}, p1 = {a + 2, a + 3}, p2 = {a + 4, a + 5}, p3 = {a + 6, a + 7}, pb = {b, b}, pa = {a, a};
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) {          // 8 x v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b), "v"(a));
            } else if (MODE == 1) {   // 8 x v_pk_fma_f32 (two results each)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pa));
            } else if (MODE == 2) {   // 8 x v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
            } else if (MODE == 3) {   // 8 x v_exp_f32
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            } else if (MODE == 4) {   // 8 x v_mov_b32
                asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            } else if (MODE == 5) {   // 8 x v_permlane32_swap
                asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                             "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            } else if (MODE == 6) {   // 8 x s_nop 3
                asm volatile("s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3\n s_nop 3");
            } else if (MODE == 7) {   // 8 x v_add_f32 with DPP row_shr (cross-lane add in one instruction)
                asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1\n v_add_f32_dpp %1, %1, %1 row_shr:2\n v_add_f32_dpp %2, %2, %2 row_shr:4\n v_add_f32_dpp %3, %3, %3 row_shr:8\n"
                             "v_add_f32_dpp %4, %4, %4 row_shr:1\n v_add_f32_dpp %5, %5, %5 row_shr:2\n v_add_f32_dpp %6, %6, %6 row_shr:4\n v_add_f32_dpp %7, %7, %7 row_shr:8"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            } else if (MODE == 8) {   // 8 x v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
            } else if (MODE == 9) {   // 8 x v_rcp_f32
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            }
        }
    }
    float r = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
    if (r == 12345.678f) out[threadIdx.x] = r;
}
template <int MODE>
float run(float* d, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<256, 256>>>(d, iters);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<MODE><<<256, 256>>>(d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}
int main() {
    float* d; (void)hipMalloc(&d, 4096);
    const int it = 20000;
    const double n = 32.0 * it;       // instructions per wave
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_exp_f32", "v_mov_b32", "v_permlane32/16_swap", "s_nop 3", "v_add_f32 dpp row_shr",
                           "v_pk_add_f32", "v_rcp_f32"};
    float us[10] = {run<0>(d, it), run<1>(d, it), run<2>(d, it), run<3>(d, it), run<4>(d, it), run<5>(d, it), run<6>(d, it), run<7>(d, it), run<8>(d, it), run<9>(d, it)};
    printf("one wave per SIMD, %d instructions per wave; cycles at 2.4 GHz nominal (the chip may clock lower)\n", (int)n);
    for (int m = 0; m < 10; ++m) printf("%-24s %8.1f us  %5.2f ns/instr  ~%4.1f cycles\n", names[m], us[m], us[m] * 1e3 / n, us[m] * 1e3 / n * 2.4);
    return 0;
}
