// Batched weighted Kabsch pose head for gfx950.
//   moments  fp64 partial raw moments of the weighted clouds, several workgroups per pair
//   solve    H = sum w (s-cs)(t-ct)^T + 1e-6 I; one-sided Jacobi SVD of the 3x3 in fp64 on one lane;
//            R = V D U^T with D flipping the smallest singular direction when det(V U^T) < 0; t = ct - R cs.
// The reference's host-visible `if torch.det(R_b) < 0` (train :749) is a device-side select;
// nothing synchronises with the host.  Backward is the closed form through the polar factor.
//
// Reference: src/3dmatch_train_egnn_with_batch.py:726-758, src/eval_egnn_metrics.py:786-818.
#include "common.h"
#include "kabsch_device.h"

namespace eq {

template <int NV>
__device__ __forceinline__ void block_reduce_sum(float (&v)[NV], float* smem /* [4*NV] */, float* out /* LDS [NV] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float a = v[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        v[i] = a;
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) smem[wave * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < NV) out[threadIdx.x] = smem[threadIdx.x] + smem[NV + threadIdx.x] + smem[2 * NV + threadIdx.x] + smem[3 * NV + threadIdx.x];
    __syncthreads();
}

__global__ __launch_bounds__(256) void kabsch_moments_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                             const float* __restrict__ w, int n, float* __restrict__ aux) {
    kabsch_moments_block(src, tgt, w, n, aux, blockIdx.y, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(64) void kabsch_solve_kernel(int nb, float* __restrict__ Rout, float* __restrict__ tout,
                                                          float* __restrict__ aux) {
    kabsch_solve_wave(nb, Rout, tout, aux, blockIdx.x, threadIdx.x);
}

__global__ __launch_bounds__(256) void kabsch_bwd_kernel(const float* __restrict__ src, const float* __restrict__ tgt,
                                                         const float* __restrict__ w, const float* __restrict__ aux,
                                                         const float* __restrict__ gR, const float* __restrict__ gt, int n,
                                                         float* __restrict__ gsrc, float* __restrict__ gtgt,
                                                         float* __restrict__ gw) {
    __shared__ float sh[9 + 3 + 3];   // gH, g_cs, g_ct
    const int b = blockIdx.x;
    const float* ax = aux + (size_t)b * EQUIGSPR_KABSCH_AUX;
    if (threadIdx.x == 0) {
        double cs[3], U[9], st[3], R[9], G[9], g_cs[3], g_ct[3], gtv[3];
        for (int a = 0; a < 3; ++a) { cs[a] = ax[a]; st[a] = ax[15 + a]; gtv[a] = gt[b * 3 + a]; }
        for (int i = 0; i < 9; ++i) { U[i] = ax[6 + i]; R[i] = ax[18 + i]; }
        for (int a = 0; a < 3; ++a) {
            g_ct[a] = gtv[a];
            g_cs[a] = -(R[0 * 3 + a] * gtv[0] + R[1 * 3 + a] * gtv[1] + R[2 * 3 + a] * gtv[2]);
            for (int c = 0; c < 3; ++c) G[a * 3 + c] = (double)gR[b * 9 + a * 3 + c] - gtv[a] * cs[c];
        }
        // M = U^T R^T G U
        double T1[9], T2[9], M[9];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {                 // T1 = R^T G
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += R[k * 3 + a] * G[k * 3 + c];
                T1[a * 3 + c] = acc;
            }
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {                 // T2 = T1 U
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += T1[a * 3 + k] * U[k * 3 + c];
                T2[a * 3 + c] = acc;
            }
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {                 // M = U^T T2
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += U[k * 3 + a] * T2[k * 3 + c];
                M[a * 3 + c] = acc;
            }
        double Nm[9];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {
                const double den = st[a] + st[c];
                Nm[a * 3 + c] = (a != c && fabs(den) > 1e-30) ? M[a * 3 + c] / den : 0.0;
            }
        double Sk[9];                                       // N - N^T
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) Sk[a * 3 + c] = Nm[a * 3 + c] - Nm[c * 3 + a];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {                 // T1 = U Sk
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += U[a * 3 + k] * Sk[k * 3 + c];
                T1[a * 3 + c] = acc;
            }
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {                 // T2 = T1 U^T  (= K - K^T)
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += T1[a * 3 + k] * U[c * 3 + k];
                T2[a * 3 + c] = acc;
            }
        double gH[9];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {                 // gA = R T2 ; gH = gA^T
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += R[a * 3 + k] * T2[k * 3 + c];
                gH[c * 3 + a] = acc;
            }
        for (int a = 0; a < 3; ++a) {
            double a1 = 0, a2 = 0;
            for (int c = 0; c < 3; ++c) {
                a1 += gH[a * 3 + c] * ax[30 + c];            // gH * sum w (t-ct)
                a2 += gH[c * 3 + a] * ax[27 + c];            // gH^T * sum w (s-cs)
            }
            g_cs[a] -= a1;
            g_ct[a] -= a2;
        }
        for (int i = 0; i < 9; ++i) sh[i] = (float)gH[i];
        for (int a = 0; a < 3; ++a) { sh[9 + a] = (float)g_cs[a]; sh[12 + a] = (float)g_ct[a]; }
    }
    __syncthreads();
    float gH[9], g_cs[3], g_ct[3], cs[3], ct[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) gH[i] = sh[i];
#pragma unroll
    for (int a = 0; a < 3; ++a) { g_cs[a] = sh[9 + a]; g_ct[a] = sh[12 + a]; cs[a] = ax[a]; ct[a] = ax[3 + a]; }
    const float* s = src + (size_t)b * n * 3;
    const float* t = tgt + (size_t)b * n * 3;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float wi = w[(size_t)b * n + i];
        float sv[3], tv[3], ds[3], dt[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { sv[a] = s[i * 3 + a]; tv[a] = t[i * 3 + a]; ds[a] = sv[a] - cs[a]; dt[a] = tv[a] - ct[a]; }
        float gwi = 0.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float hs = gH[a * 3 + 0] * dt[0] + gH[a * 3 + 1] * dt[1] + gH[a * 3 + 2] * dt[2];      // (gH dt)_a
            const float ht = gH[0 * 3 + a] * ds[0] + gH[1 * 3 + a] * ds[1] + gH[2 * 3 + a] * ds[2];      // (gH^T ds)_a
            if (gsrc) gsrc[((size_t)b * n + i) * 3 + a] = wi * (hs + g_cs[a]);
            if (gtgt) gtgt[((size_t)b * n + i) * 3 + a] = wi * (ht + g_ct[a]);
            gwi += ds[a] * hs + sv[a] * g_cs[a] + tv[a] * g_ct[a];
        }
        if (gw) gw[(size_t)b * n + i] = gwi;
    }
}

}  // namespace eq

extern "C" int equigspr_kabsch_forward(const float* src, const float* tgt, const float* w, int32_t batch, int32_t n,
                                       float* R, float* t, float* aux, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(src && tgt && w && R && t && aux, EQUIGSPR_EINVAL, "kabsch_forward: null pointer");
    EQ_CHECK_ARG(batch >= 1 && n >= 1, EQUIGSPR_EINVAL, "kabsch_forward: need batch>=1, n>=1");
    int nb = (n + 255) / 256;
    if (nb > kKabschMaxBlocks) nb = kKabschMaxBlocks;
    hipLaunchKernelGGL(kabsch_moments_kernel, dim3(nb, batch), dim3(256), 0, (hipStream_t)stream, src, tgt, w, n, aux);
    hipLaunchKernelGGL(kabsch_solve_kernel, dim3(batch), dim3(64), 0, (hipStream_t)stream, nb, R, t, aux);
    EQ_CHECK_LAUNCH("kabsch_fwd_kernel");
    return 0;
}

extern "C" int equigspr_kabsch_backward(const float* src, const float* tgt, const float* w, const float* aux,
                                        const float* grad_R, const float* grad_t, int32_t batch, int32_t n,
                                        float* grad_src, float* grad_tgt, float* grad_w, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(src && tgt && w && aux && grad_R && grad_t, EQUIGSPR_EINVAL, "kabsch_backward: null pointer");
    EQ_CHECK_ARG(batch >= 1 && n >= 1, EQUIGSPR_EINVAL, "kabsch_backward: need batch>=1, n>=1");
    hipLaunchKernelGGL(kabsch_bwd_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, src, tgt, w, aux, grad_R,
                       grad_t, n, grad_src, grad_tgt, grad_w);
    EQ_CHECK_LAUNCH("kabsch_bwd_kernel");
    return 0;
}
