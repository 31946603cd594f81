// Correspondence construction from descriptors (data_preprocess/3DMatch_Feature.py:157-196, :331-351):
//   distance = sqrt(2 - 2 * (src_desc @ tgt_desc.T) + 1e-6);  source_idx = argmin(distance, axis=1);
//   labels   = || R_gt src[corr0] + t_gt - tgt[corr1] || < inlier_threshold.
// The N x M distance matrix is never materialised: a wave owns 32 source rows (their K-vector is the register-resident
// MFMA A operand), target tiles of 32 rows are staged through LDS once per block and shared by its 4 waves, every
// 32x32 block of dot products comes out of K/2 v_mfma_f32_32x32x2_f32 and is folded straight into a per-lane running
// (distance, index) minimum; the 32 lanes of a row are combined at the end by a lexicographic butterfly (first index wins,
// as np.argmin).
#include "egnn_mfma.h"

namespace eq {

constexpr int kMatchMaxK = 64;

__global__ __launch_bounds__(256) void match_argmin_kernel(const float* __restrict__ S, int N, const float* __restrict__ T, int M, int K,
                                                           long long* __restrict__ idx_out, float* __restrict__ dis_out) {
    __shared__ float tt[32][kMatchMaxK + 1];            // target tile, rows padded to an odd stride
    const int lane = threadIdx.x & 63, el = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const int steps = (K + 1) / 2;
    const int src0 = (blockIdx.x * 4 + wv) * 32;
    // A operand: S[src0 + el][2s + h]
    float a[kMatchMaxK / 2];
    {
        const int i = min(src0 + el, N - 1);
#pragma unroll
        for (int s = 0; s < kMatchMaxK / 2; ++s) {
            const int k = 2 * s + h;
            a[s] = (s < steps && k < K) ? S[(size_t)i * K + k] : 0.0f;
        }
    }
    float best_d[16];
    int best_j[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { best_d[r] = INFINITY; best_j[r] = 0x7fffffff; }
    for (int t0 = 0; t0 < M; t0 += 32) {
        __syncthreads();
        for (int q = threadIdx.x; q < 32 * K; q += 256) {
            const int j = q / K, k = q - j * K;
            tt[j][k] = (t0 + j < M) ? T[(size_t)(t0 + j) * K + k] : 0.0f;
        }
        if (K & 1) { if (threadIdx.x < 32) tt[threadIdx.x][K] = 0.0f; }
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < kMatchMaxK / 2; ++s)
            if (s < steps) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], tt[el][2 * s + h], acc, 0, 0, 0);
        const int j = t0 + el;
        if (j < M) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = sqrtf((2.0f - 2.0f * acc[r]) + 1e-6f);      // the reference's float32 expression
                if (d < best_d[r]) { best_d[r] = d; best_j[r] = j; }
            }
        }
    }
    // argmin over the 32 target columns of each row (lanes of one half): lowest distance, then lowest index
#pragma unroll
    for (int off = 1; off <= 16; off <<= 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float od = __shfl_xor(best_d[r], off, 32);
            const int oj = __shfl_xor(best_j[r], off, 32);
            if (od < best_d[r] || (od == best_d[r] && oj < best_j[r])) { best_d[r] = od; best_j[r] = oj; }
        }
    }
    if (el == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = src0 + fidx(r, h);
            if (i < N) { idx_out[i] = best_j[r]; if (dis_out) dis_out[i] = best_d[r]; }
        }
    }
}

// labels[c] = || R src[corr[c][0]] + t - tgt[corr[c][1]] || < thr      (fp32, as the reference's float32 key points)
__global__ void corr_labels_kernel(const float* __restrict__ src, const float* __restrict__ tgt, const long long* __restrict__ corr,
                                   int n_corr, const float* __restrict__ pose, float thr, int* __restrict__ labels,
                                   float* __restrict__ dist_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_corr) return;
    const float* s = src + (size_t)corr[2 * c] * 3;
    const float* g = tgt + (size_t)corr[2 * c + 1] * 3;
    float d2 = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float w = ((pose[4 * a] * s[0] + pose[4 * a + 1] * s[1]) + pose[4 * a + 2] * s[2]) + pose[4 * a + 3];
        const float e = w - g[a];
        d2 += e * e;
    }
    const float d = sqrtf(d2);
    labels[c] = d < thr ? 1 : 0;
    if (dist_out) dist_out[c] = d;
}

}  // namespace eq

extern "C" int equigspr_match_descriptors(const float* src_desc, int32_t n_src, const float* tgt_desc, int32_t n_tgt, int32_t dim,
                                          int64_t* nearest_idx, float* nearest_dist, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(src_desc && tgt_desc && nearest_idx && n_src >= 1 && n_tgt >= 1, EQUIGSPR_EINVAL, "match_descriptors: bad arguments");
    EQ_CHECK_ARG(dim >= 1 && dim <= kMatchMaxK, EQUIGSPR_ENOTSUP, "match_descriptors: descriptor width must be in [1, 64]");
    hipLaunchKernelGGL(match_argmin_kernel, dim3(ceil_div(n_src, 128)), dim3(256), 0, (hipStream_t)stream, src_desc, n_src, tgt_desc, n_tgt,
                       dim, reinterpret_cast<long long*>(nearest_idx), nearest_dist);
    EQ_CHECK_LAUNCH("match_descriptors");
    return 0;
}

extern "C" int equigspr_correspondence_labels(const float* src_pts, const float* tgt_pts, const int64_t* corr, int32_t n_corr,
                                              const float* gt_pose, float inlier_threshold, int32_t* labels, float* dist, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(src_pts && tgt_pts && corr && gt_pose && labels && n_corr >= 0, EQUIGSPR_EINVAL, "correspondence_labels: bad arguments");
    if (n_corr == 0) return 0;
    hipLaunchKernelGGL(corr_labels_kernel, dim3(ceil_div(n_corr, 256)), dim3(256), 0, (hipStream_t)stream, src_pts, tgt_pts,
                       reinterpret_cast<const long long*>(corr), n_corr, gt_pose, inlier_threshold, labels, dist);
    EQ_CHECK_LAUNCH("correspondence_labels");
    return 0;
}
