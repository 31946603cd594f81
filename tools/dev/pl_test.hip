#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, float* o16, float* o32, float* osum) {
    float v = in[threadIdx.x];
    auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    float s16 = __uint_as_float(q[0]) + __uint_as_float(q[1]);
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s16), __float_as_uint(s16), false, false);
    o16[threadIdx.x] = s16;
    o32[threadIdx.x] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    float w = v; w += __shfl_xor(w, 16, 64); w += __shfl_xor(w, 32, 64);
    osum[threadIdx.x] = w;
}
int main() {
    float h[64], a[64], b[64], c[64], *d, *da, *db, *dc;
    for (int i = 0; i < 64; ++i) h[i] = (float)(i * i % 37) + 0.125f * i;
    hipMalloc(&d, 256); hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dc, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, da, db, dc);
    hipMemcpy(a, da, 256, hipMemcpyDeviceToHost); hipMemcpy(b, db, 256, hipMemcpyDeviceToHost); hipMemcpy(c, dc, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        float e16 = h[i] + h[i ^ 16], e = h[i & 15] + h[(i & 15) + 16] + h[(i & 15) + 32] + h[(i & 15) + 48];
        if (a[i] != e16 || b[i] != c[i] || fabsf(b[i] - e) > 1e-3f) { ++bad; printf("lane %d: s16 %g (want %g) s %g shfl %g (want %g)\n", i, a[i], e16, b[i], c[i], e); }
    }
    printf("permlane swap test: %s\n", bad ? "FAIL" : "OK");
    return bad != 0;
}
