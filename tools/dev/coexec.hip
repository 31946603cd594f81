// Does fp32 MFMA overlap with VALU work of the partner wave on the same SIMD?  (gfx950 dev probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE_LO, int MODE_HI>   // role of waves 0..3 / 4..7: 0 idle, 1 fp32 mfma 16x16x4, 2 valu fma, 3 bf16 mfma 16x16x32, 4 fp32 mfma 32x32x2, 5 transcendental
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wv = threadIdx.x >> 6;
    const int mode = wv < 4 ? MODE_LO : MODE_HI;
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0 = {0}, d1 = {0};
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
    bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
    if (mode == 1) {
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
    } else if (mode == 2) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v0 = fmaf(v0, b, a); v1 = fmaf(v1, b, a); v2 = fmaf(v2, b, a); v3 = fmaf(v3, b, a);
                v4 = fmaf(v4, b, a); v5 = fmaf(v5, b, a); v6 = fmaf(v6, b, a); v7 = fmaf(v7, b, a);
            }
        }
    } else if (mode == 3) {
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, c3, 0, 0, 0);
        }
    } else if (mode == 4) {
        for (int i = 0; i < iters; ++i) {
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
        }
    } else if (mode == 5) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                v0 = __builtin_amdgcn_exp2f(v0); v1 = __builtin_amdgcn_exp2f(v1); v2 = __builtin_amdgcn_exp2f(v2); v3 = __builtin_amdgcn_exp2f(v3);
                v4 = __builtin_amdgcn_rcpf(v4); v5 = __builtin_amdgcn_rcpf(v5); v6 = __builtin_amdgcn_rcpf(v6); v7 = __builtin_amdgcn_rcpf(v7);
            }
        }
    }
    float r = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    if (r == 12345.678f) out[threadIdx.x] = r;
}
template <int LO, int HI>
float run(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<LO, HI><<<256, 512>>>(d, iters);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<LO, HI><<<256, 512>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}
int main() {
    float* d; hipMalloc(&d, 4096);
    const int it = 20000;
    printf("iters %d per wave; per iteration: 4 MFMA f32 16x16x4 (128 cyc) | 32 v_fma | 4 MFMA bf16 16x16x32 | 2 MFMA f32 32x32x2 (128 cyc) | 8 exp + 8 rcp\n", it);
    printf("f32 mfma16 alone (4 waves/CU)   %8.1f us\n", run<1, 0>(d, it));
    printf("valu fma alone                  %8.1f us\n", run<2, 0>(d, it));
    printf("f32 mfma16 + valu partner       %8.1f us\n", run<1, 2>(d, it));
    printf("f32 mfma16 + f32 mfma16 partner %8.1f us\n", run<1, 1>(d, it));
    printf("valu + valu partner             %8.1f us\n", run<2, 2>(d, it));
    printf("bf16 mfma alone                 %8.1f us\n", run<3, 0>(d, it));
    printf("bf16 mfma + valu partner        %8.1f us\n", run<3, 2>(d, it));
    printf("f32 mfma32 alone                %8.1f us\n", run<4, 0>(d, it));
    printf("f32 mfma32 + valu partner       %8.1f us\n", run<4, 2>(d, it));
    printf("trans alone                     %8.1f us\n", run<5, 0>(d, it));
    printf("f32 mfma16 + trans partner      %8.1f us\n", run<1, 5>(d, it));
    return 0;
}
