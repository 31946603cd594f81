// decide the "bf16 x 3 split" question with a probe, not a rewrite.
// One 32 x 32 mat-vec tile of the edge kernels (y = W v for 16 edges: today 4 ds_read_b128 + 16 v_mfma_f32_16x16x4_f32, csrc/egnn_mfma16.h::mv16)
// against the same product on the bf16 matrix pipe with 3-way split operands (x = hi + mid + lo, each bf16; the six products with
// i + j <= 2: 12 v_mfma_f32_16x16x32_bf16 with the weights PRE-SPLIT in the LDS image, the per-edge vector split in registers).
// Reports ns per tile per wave (dependent chain: every product feeds the next, as in the kernels) at one and two waves per SIMD,
// and the max error of both against fp64.   hipcc -O3 --offload-arch=gfx950 tools/dev/bf16_split.hip -o /tmp/bf16_split && /tmp/bf16_split
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int WS = 40;                    // fp32 image row stride (egnn_mfma16.h)

__device__ __forceinline__ f32x4 mfma_f32(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// the lane's 8 features of the L16 layout: lane (e = l & 15, g = l >> 4) owns features 16 b + 4 g + r;  K index j of the bf16 form <-> (b = j >> 2, r = j & 3)
__device__ __forceinline__ int feat_of(int g, int j) { return 16 * (j >> 2) + 4 * g + (j & 3); }

// 3-way split of 8 floats into three bf16x8 fragments (round to nearest even through the compiler's cvt)
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)x[j];
        const float r1 = x[j] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        hi[j] = h; mid[j] = m; lo[j] = (__bf16)r2;
    }
}

// MODE 0: fp32 MFMA tile chain; MODE 1: bf16 x 3 tile chain.  `iters` dependent tiles per wave.
template <int MODE>
__global__ __launch_bounds__(512) void chain_kernel(const float* __restrict__ W, const float* __restrict__ v0, float* __restrict__ out, int iters, float bias = 0.0f) {
    __shared__ float wf[32 * WS];
    __shared__ bf16x8 ws[3][2][64];           // [level][bo][lane]: the A fragment of lane (row i, g): W[16 bo + i][feat_of(g, j)], pre-split
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    for (int t = threadIdx.x; t < 1024; t += blockDim.x) wf[(t >> 5) * WS + (t & 31)] = W[t];
    if (threadIdx.x < 128) {
        const int bo = threadIdx.x >> 6, l = threadIdx.x & 63, ii = l & 15, gg = l >> 4;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = W[(16 * bo + ii) * 32 + feat_of(gg, j)];
        bf16x8 h, m, lo;
        split3(x, h, m, lo);
        ws[0][bo][l] = h; ws[1][bo][l] = m; ws[2][bo][l] = lo;
    }
    __syncthreads();
    // the tile's input vector: lane (edge e = i, g) holds features 16 b + 4 g + r as v[b][r]
    f32x4 v[2];
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[b][r] = v0[((size_t)(wave & 63) * 16 + i) * 32 + 16 * b + 4 * g + r];
    for (int it = 0; it < iters; ++it) {
        f32x4 acc[2] = {f32x4{bias, bias, bias, bias}, f32x4{bias, bias, bias, bias}};
        if (MODE == 0) {
            f32x4 a[2][2];
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bo = 0; bo < 2; ++bo) a[bi][bo] = *reinterpret_cast<const f32x4*>(wf + (16 * bo + i) * WS + 16 * bi + 4 * g);
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[0] = mfma_f32(a[bi][0][r], v[bi][r], acc[0]);
                    acc[1] = mfma_f32(a[bi][1][r], v[bi][r], acc[1]);
                }
        } else {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = v[j >> 2][j & 3];
            bf16x8 vh, vm, vl;
            split3(x, vh, vm, vl);
#pragma unroll
            for (int bo = 0; bo < 2; ++bo) {
                const bf16x8 ah = ws[0][bo][lane], am = ws[1][bo][lane], al = ws[2][bo][lane];
                // smallest terms first
                acc[bo] = mfma_bf16(al, vh, acc[bo]);
                acc[bo] = mfma_bf16(ah, vl, acc[bo]);
                acc[bo] = mfma_bf16(am, vm, acc[bo]);
                acc[bo] = mfma_bf16(am, vh, acc[bo]);
                acc[bo] = mfma_bf16(ah, vm, acc[bo]);
                acc[bo] = mfma_bf16(ah, vh, acc[bo]);
            }
        }
        // the next tile's input depends on this one (bounded: a contraction keeps the chain finite)
#pragma unroll
        for (int b = 0; b < 2; ++b) v[b] = acc[b] * 0.17f + v[b] * 0.5f;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[((size_t)wave * 16 + i) * 32 + 16 * b + 4 * g + r] = v[b][r];
}

int main() {
    const int n_w = 1024, n_v = 64 * 16 * 32;
    std::vector<float> W(n_w), V(n_v);
    srand(5);
    for (auto& w : W) w = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.35f;       // ~ xavier for 32 -> 32
    for (auto& x : V) x = ((rand() / (float)RAND_MAX) * 2 - 1) * 3.0f;
    float *dW, *dV, *dO;
    const int max_waves = 256 * 8;
    (void)hipMalloc(&dW, n_w * 4); (void)hipMalloc(&dV, n_v * 4); (void)hipMalloc(&dO, (size_t)max_waves * 16 * 32 * 4);
    (void)hipMemcpy(dW, W.data(), n_w * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dV, V.data(), n_v * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    // ---- accuracy: ONE tile (iters = 1) against fp64
    std::vector<float> o0((size_t)64 * 16 * 32), o1(o0.size());
    chain_kernel<0><<<16, 256>>>(dW, dV, dO, 1); (void)hipMemcpy(o0.data(), dO, o0.size() * 4, hipMemcpyDeviceToHost);
    chain_kernel<1><<<16, 256>>>(dW, dV, dO, 1); (void)hipMemcpy(o1.data(), dO, o1.size() * 4, hipMemcpyDeviceToHost);
    double e_f32 = 0, e_bf = 0, ymax = 0;
    for (int t = 0; t < 64 * 16; ++t)
        for (int o = 0; o < 32; ++o) {
            double y = 0;
            for (int k = 0; k < 32; ++k) y += (double)W[o * 32 + k] * (double)V[t * 32 + k];
            const double ref = y * 0.17f + (double)V[t * 32 + o] * 0.5;
            ymax = fmax(ymax, fabs(ref));
            e_f32 = fmax(e_f32, fabs(o0[t * 32 + o] - ref)); e_bf = fmax(e_bf, fabs(o1[t * 32 + o] - ref));
        }
    printf("one tile vs fp64 (|y| up to %.2f): fp32 MFMA max err %.3e   bf16 x 3 (6 products) max err %.3e\n", ymax, e_f32, e_bf);
    // ---- stress: wide dynamic range inside the vector (lognormal, sigma 3) and an accumulator that starts at a large bias
    for (int trial = 0; trial < 3; ++trial) {
        const float bias = trial == 0 ? 0.0f : (trial == 1 ? 37.5f : 4000.0f);
        std::vector<float> V2(n_v);
        for (auto& x : V2) { const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX; x = expf(3.0f * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2)) * ((rand() & 1) ? 1.f : -1.f); }
        (void)hipMemcpy(dV, V2.data(), n_v * 4, hipMemcpyHostToDevice);
        chain_kernel<0><<<16, 256>>>(dW, dV, dO, 1, bias); (void)hipMemcpy(o0.data(), dO, o0.size() * 4, hipMemcpyDeviceToHost);
        chain_kernel<1><<<16, 256>>>(dW, dV, dO, 1, bias); (void)hipMemcpy(o1.data(), dO, o1.size() * 4, hipMemcpyDeviceToHost);
        double ef = 0, eb = 0, sc = 0;
        for (int t = 0; t < 64 * 16; ++t)
            for (int o = 0; o < 32; ++o) {
                double y = bias, ab = fabs(bias);
                for (int k = 0; k < 32; ++k) { y += (double)W[o * 32 + k] * (double)V2[t * 32 + k]; ab += fabs((double)W[o * 32 + k] * (double)V2[t * 32 + k]); }
                const double ref = y * 0.17f + (double)V2[t * 32 + o] * 0.5;
                ef = fmax(ef, fabs(o0[t * 32 + o] - ref) / ab); eb = fmax(eb, fabs(o1[t * 32 + o] - ref) / ab);
            }
        printf("stress (lognormal vector, accumulator starts at %.1f): max error / sum|a b|: fp32 MFMA %.3e   bf16 x 3 %.3e\n", bias, ef, eb);
    }
    (void)hipMemcpy(dV, V.data(), n_v * 4, hipMemcpyHostToDevice);
    // ---- time: dependent chains, all CUs, one / two waves per SIMD
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps) {
        float ms[2];
        for (int mode = 0; mode < 2; ++mode) {
            auto launch = [&] { if (mode == 0) chain_kernel<0><<<256, 256 * wps>>>(dW, dV, dO, iters); else chain_kernel<1><<<256, 256 * wps>>>(dW, dV, dO, iters); };
            launch(); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms[mode], e0, e1);
        }
        printf("%d wave(s) per SIMD: fp32 16 x mfma_16x16x4_f32 %.1f ns per tile and SIMD   bf16 x 3 (split + 12 x mfma_16x16x32_bf16) %.1f ns   ratio %.2f x\n",
               wps, ms[0] * 1e6 / iters / wps, ms[1] * 1e6 / iters / wps, ms[0] / ms[1]);
    }
    return 0;
}
