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
    kabsch_bwd_block(src, tgt, w, aux, gR + (size_t)blockIdx.y * 9, gt + (size_t)blockIdx.y * 3, n, gsrc, gtgt, gw, blockIdx.y, blockIdx.x, gridDim.x);
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
    hipLaunchKernelGGL(kabsch_bwd_kernel, dim3(kabsch_bwd_blocks(n), batch), dim3(256), 0, (hipStream_t)stream, src, tgt, w, aux, grad_R,
                       grad_t, n, grad_src, grad_tgt, grad_w);
    EQ_CHECK_LAUNCH("kabsch_bwd_kernel");
    return 0;
}
