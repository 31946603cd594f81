// The reference pose loss per pair (pose_loss() of src/3dmatch_train_egnn_with_batch.py:896-962: geodesic rotation angle :952,
// translation-direction angle :955-960) and its gradient, as device functions shared by csrc/train_ops.hip (the stand-alone loss
// kernels) and csrc/head.hip (the training head evaluates the step loss and its gradient inside its own launches).
#pragma once

namespace eq {

// torch.clamp semantics: NaN propagates (fminf/fmaxf would turn it into a bound) -- e.g. the translation-direction cosine
// of a zero translation (a pair without any valid correspondence gets R = I, t = 0, train :708-711) is NaN in the reference
__device__ __forceinline__ float clamp_unit(float v) { return (v != v) ? v : fminf(fmaxf(v, -1.0f), 1.0f); }

// Rb [9], tb [3], G = gt_pose row [16]
__device__ __forceinline__ void pose_loss_pair(const float* Rb, const float* tb, const float* G, float& rot_loss, float& trans_loss) {
    float tr = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) tr = fmaf(Rb[i * 3 + j], G[i * 4 + j], tr);          // trace(R^T R_gt)
    rot_loss = acosf(clamp_unit((tr - 1.0f) * 0.5f));
    float dp = 0.0f, nt = 0.0f, ng = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const float a = tb[i], g = G[i * 4 + 3]; dp = fmaf(a, g, dp); nt = fmaf(a, a, nt); ng = fmaf(g, g, ng); }
    trans_loss = acosf(clamp_unit(dp / (sqrtf(nt) * sqrtf(ng))));
}

// gR [9], gt [3] = g_rot d rot_loss / dR, g_trans d trans_loss / dt
__device__ __forceinline__ void pose_loss_grad_pair(const float* Rb, const float* tb, const float* G, float g_rot, float g_trans,
                                                    float* gR, float* gt) {
    float tr = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) tr = fmaf(Rb[i * 3 + j], G[i * 4 + j], tr);
    const float x = (tr - 1.0f) * 0.5f;
    // torch.clamp passes the gradient on [min, max] inclusive; acos'(x) = -1/sqrt(1-x^2) (the reference's singularity at x = 1)
    const float dx = (x >= -1.0f && x <= 1.0f) ? -1.0f / sqrtf(1.0f - x * x) : 0.0f;
    const float kr = g_rot * dx * 0.5f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) gR[i * 3 + j] = kr * G[i * 4 + j];
    float dp = 0.0f, nt = 0.0f, ng = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const float a = tb[i], g = G[i * 4 + 3]; dp = fmaf(a, g, dp); nt = fmaf(a, a, nt); ng = fmaf(g, g, ng); }
    const float lt = sqrtf(nt), lg = sqrtf(ng);
    const float c = dp / (lt * lg);
    const float dc = (c >= -1.0f && c <= 1.0f) ? -1.0f / sqrtf(1.0f - c * c) : 0.0f;
    const float kt = g_trans * dc;
#pragma unroll
    for (int i = 0; i < 3; ++i) gt[i] = kt * (G[i * 4 + 3] / (lt * lg) - c * tb[i] / nt);
}

// loss[0] = closs + mean_b rot[b] + mean_b trans[b]: 256 threads, per-thread strided sums then the fixed tree (one block)
__device__ __forceinline__ void step_loss_sum(const float* rot_loss, const float* trans_loss, float closs, int B, float* loss) {
    __shared__ float sr[256], st[256];
    float ar = 0.0f, at = 0.0f;
    for (int b = threadIdx.x; b < B; b += 256) { ar += rot_loss[b]; at += trans_loss[b]; }
    sr[threadIdx.x] = ar; st[threadIdx.x] = at;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) { sr[threadIdx.x] += sr[threadIdx.x + off]; st[threadIdx.x] += st[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = closs + sr[0] / (float)B + st[0] / (float)B;
}

}  // namespace eq
