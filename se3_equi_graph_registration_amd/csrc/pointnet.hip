// PointNet encoder: PointNetLayer / PointNet of src/3dmatch_train_egnn_with_batch.py:52-96.
//   message(j -> i) = Linear2(ReLU(Linear1([h_j, pos_j - pos_i]))),  out_i = max over incoming edges (0 if none),
//   PointNet = conv1, ReLU, conv2, ReLU, global_max_pool over the batch vector.
// Linear1 is split per node:  W1 [h_j | pos_j - pos_i] + b1 = (Wh h_j + Wp pos_j) + (b1 - Wp pos_i) = A_j + C_i, so the
// per-edge work is one add + ReLU + the out x out second Linear; the max runs over the target's edge segment of the
// inverse index (cptr/cperm), no atomics.  One wave per target node, lane = output channel (<= 64).
#include "common.h"

namespace eq {

constexpr int kPnMax = 64;

// A[j] = Wh h_j + Wp pos_j,  C[j] = b1 - Wp pos_j      (w1: [out, in + 3] row-major as nn.Linear)
__global__ __launch_bounds__(256) void pn_project_kernel(const float* __restrict__ h, const float* __restrict__ pos, int n, int c_in,
                                                         int c_out, const float* __restrict__ w1, const float* __restrict__ b1,
                                                         float* __restrict__ A, float* __restrict__ C) {
    extern __shared__ float sm[];                      // w1^T [c_in + 3][c_out + 1] then 4 feature rows
    const int ld = c_out + 1, kin = c_in + 3;
    for (int t = threadIdx.x; t < c_out * kin; t += 256) { const int o = t / kin, i = t - o * kin; sm[i * ld + o] = w1[t]; }
    float* rows = sm + kin * ld;
    const int wv = threadIdx.x >> 6, f = threadIdx.x & 63;
    __syncthreads();
    for (int j = blockIdx.x * 4 + wv; j < n; j += gridDim.x * 4) {
        float* r = rows + wv * kPnMax;
        for (int i = f; i < c_in; i += 64) r[i] = h[(size_t)j * c_in + i];
        __builtin_amdgcn_wave_barrier();
        if (f < c_out) {
            float a = 0.0f;
            for (int i = 0; i < c_in; ++i) a = fmaf(sm[i * ld + f], r[i], a);
            float p = 0.0f;
            for (int i = 0; i < 3; ++i) p = fmaf(sm[(c_in + i) * ld + f], pos[(size_t)j * 3 + i], p);
            A[(size_t)j * c_out + f] = a + p;
            C[(size_t)j * c_out + f] = b1[f] - p;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// out[i] = relu?( max_e ( b2 + W2 relu(A[src(e)] + C[i]) ) ) over the edges e with target i; 0 when there are none
__global__ __launch_bounds__(256) void pn_aggregate_kernel(const float* __restrict__ A, const float* __restrict__ C,
                                                           const int* __restrict__ src, const int* __restrict__ tptr,
                                                           const int* __restrict__ tperm, int n, int c_out,
                                                           const float* __restrict__ w2, const float* __restrict__ b2, int relu_out,
                                                           float* __restrict__ out) {
    extern __shared__ float sm[];                      // w2^T [c_out][c_out + 1], then 4 hidden rows
    const int ld = c_out + 1;
    for (int t = threadIdx.x; t < c_out * c_out; t += 256) { const int o = t / c_out, i = t - o * c_out; sm[i * ld + o] = w2[t]; }
    float* rows = sm + c_out * ld;
    const int wv = threadIdx.x >> 6, f = threadIdx.x & 63;
    __syncthreads();
    const bool act = f < c_out;
    const float bias = act ? b2[f] : 0.0f;
    for (int i = blockIdx.x * 4 + wv; i < n; i += gridDim.x * 4) {
        float* r = rows + wv * kPnMax;
        const float ci = act ? C[(size_t)i * c_out + f] : 0.0f;
        const int lo = tptr[i], hi = tptr[i + 1];
        float best = -INFINITY;
        for (int p = lo; p < hi; ++p) {
            const int j = src[tperm ? tperm[p] : p];
            if (act) r[f] = fmaxf(A[(size_t)j * c_out + f] + ci, 0.0f);
            __builtin_amdgcn_wave_barrier();
            if (act) {
                float s = bias;
                for (int g = 0; g < c_out; ++g) s = fmaf(sm[g * ld + f], r[g], s);
                best = fmaxf(best, s);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (act) {
            float v = hi > lo ? best : 0.0f;
            if (relu_out) v = fmaxf(v, 0.0f);
            out[(size_t)i * c_out + f] = v;
        }
    }
}

// global_max_pool over a sorted batch vector: out[b][f] = max_{i: batch[i] == b} x[i][f]
__global__ __launch_bounds__(256) void pn_global_max_kernel(const float* __restrict__ x, const long long* __restrict__ batch, int n,
                                                            int channels, float* __restrict__ out) {
    __shared__ float part[256];
    const int b = blockIdx.x;
    int lo = 0, hi = n;                                // first index with batch >= b
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (batch[mid] < b) lo = mid + 1; else hi = mid; }
    const int start = lo;
    hi = n;                                            // first index with batch > b
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (batch[mid] <= b) lo = mid + 1; else hi = mid; }
    const int stop = lo;
    const int f = threadIdx.x % kPnMax, slice = threadIdx.x / kPnMax;      // 4 row slices x 64 channels
    float m = -INFINITY;
    if (f < channels)
        for (int i = start + slice; i < stop; i += 4) m = fmaxf(m, x[(size_t)i * channels + f]);
    part[threadIdx.x] = m;
    __syncthreads();
    if (slice == 0 && f < channels) {
        m = fmaxf(fmaxf(part[f], part[kPnMax + f]), fmaxf(part[2 * kPnMax + f], part[3 * kPnMax + f]));
        out[(size_t)b * channels + f] = m;
    }
}

}  // namespace eq

extern "C" int equigspr_pointnet_layer_forward(const float* h, const float* pos, int32_t n_nodes, int32_t c_in, int32_t c_out,
                                               const int32_t* src, const int32_t* tptr, const int32_t* tperm, const float* w1,
                                               const float* b1, const float* w2, const float* b2, int32_t relu_out, float* work,
                                               float* out, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(h && pos && tptr && w1 && b1 && w2 && b2 && work && out && n_nodes >= 1, EQUIGSPR_EINVAL,
                 "pointnet_layer_forward: bad arguments");           // src may be NULL for a graph without edges
    EQ_CHECK_ARG(c_in >= 1 && c_in <= kPnMax && c_out >= 1 && c_out <= kPnMax, EQUIGSPR_ENOTSUP,
                 "pointnet_layer_forward: channels must be in [1, 64]");
    hipStream_t s = (hipStream_t)stream;
    float* A = work;
    float* C = work + (size_t)n_nodes * c_out;
    int blocks = ceil_div(n_nodes, 4);
    blocks = blocks > 2048 ? 2048 : blocks;
    const size_t lds1 = ((size_t)(c_in + 3) * (c_out + 1) + 4 * kPnMax) * sizeof(float);
    const size_t lds2 = ((size_t)c_out * (c_out + 1) + 4 * kPnMax) * sizeof(float);
    hipLaunchKernelGGL(pn_project_kernel, dim3(blocks), dim3(256), lds1, s, h, pos, n_nodes, c_in, c_out, w1, b1, A, C);
    hipLaunchKernelGGL(pn_aggregate_kernel, dim3(blocks), dim3(256), lds2, s, A, C, src, tptr, tperm, n_nodes, c_out, w2, b2, relu_out, out);
    EQ_CHECK_LAUNCH("pointnet_layer_forward");
    return 0;
}

extern "C" int equigspr_global_max_pool(const float* x, const int64_t* batch_sorted, int32_t n_nodes, int32_t channels,
                                        int32_t n_batches, float* out, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(x && batch_sorted && out && n_nodes >= 1 && n_batches >= 1, EQUIGSPR_EINVAL, "global_max_pool: bad arguments");
    EQ_CHECK_ARG(channels >= 1 && channels <= kPnMax, EQUIGSPR_ENOTSUP, "global_max_pool: channels must be in [1, 64]");
    hipLaunchKernelGGL(pn_global_max_kernel, dim3(n_batches), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<const long long*>(batch_sorted), n_nodes, channels, out);
    EQ_CHECK_LAUNCH("global_max_pool");
    return 0;
}
