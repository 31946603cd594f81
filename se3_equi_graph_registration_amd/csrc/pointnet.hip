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

// ---------------------------------------------------------------------------------------
// backward of the layer (round 6; the reference never optimises its PointNet, train :1034 -- here it is a trainable module).
//   target pass (wave per target i, lane = channel): recompute the messages, the per-channel ARGMAX edge (first maximum in the
//     segment's order: PyG's max aggregation routes the gradient to one edge), then for every edge of the segment
//     dpre_e = ReLU'(pre_e) . W2^T v_e with v_e[f] = g_i[f] where e is channel f's argmax; dPre[e], dC[i] = sum_e dpre_e;
//     dW2 row f += g_i[f] ReLU(pre_{e*(f)}), db2 += g_i: per-wave slabs.
//   source pass (wave per source j): dA_j = sum_{e: src(e) = j} dPre[e]; grad_h_j = Wh^T dA_j, grad_pos_j = Wp^T (dA_j - dC_j);
//     dW1 row g += [dA_j[g] h_j | (dA_j[g] - dC_j[g]) pos_j], db1 += dC_j: per-wave slabs.
//   slabs summed in fixed order (no atomics).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pn_bwd_target_kernel(const float* __restrict__ A, const float* __restrict__ C, const int* __restrict__ src,
                                                            const int* __restrict__ tptr, const int* __restrict__ tperm, int n, int c_out,
                                                            const float* __restrict__ w2, const float* __restrict__ b2, int relu_out,
                                                            const float* __restrict__ gout, float* __restrict__ dPre, float* __restrict__ dC,
                                                            float* __restrict__ slab) {
    extern __shared__ float sm[];                      // w2^T [c_out][c_out + 1], then per wave: hidden row | v row | C row
    const int ld = c_out + 1;
    for (int t = threadIdx.x; t < c_out * c_out; t += 256) { const int o = t / c_out, i = t - o * c_out; sm[i * ld + o] = w2[t]; }
    float* rows = sm + c_out * ld;
    const int wv = threadIdx.x >> 6, f = threadIdx.x & 63;
    __syncthreads();
    const bool act = f < c_out;
    const float bias = act ? b2[f] : 0.0f;
    float* r = rows + wv * 3 * kPnMax;
    float* v = r + kPnMax;
    float* crow = v + kPnMax;
    float dw[kPnMax];                                  // row f of dW2
#pragma unroll
    for (int g = 0; g < kPnMax; ++g) dw[g] = 0.0f;
    float db = 0.0f;
    for (int i = blockIdx.x * 4 + wv; i < n; i += gridDim.x * 4) {
        const float ci = act ? C[(size_t)i * c_out + f] : 0.0f;
        if (act) crow[f] = ci;
        const int lo = tptr[i], hi = tptr[i + 1];
        float best = -INFINITY;
        int arg = -1;
        for (int p = lo; p < hi; ++p) {
            const int j = src[tperm ? tperm[p] : p];
            if (act) r[f] = fmaxf(A[(size_t)j * c_out + f] + ci, 0.0f);
            __builtin_amdgcn_wave_barrier();
            if (act) {
                float s = bias;
                for (int g = 0; g < c_out; ++g) s = fmaf(sm[g * ld + f], r[g], s);
                if (s > best) { best = s; arg = p; }
            }
            __builtin_amdgcn_wave_barrier();
        }
        float gi = (act && hi > lo) ? gout[(size_t)i * c_out + f] : 0.0f;
        if (relu_out && !(best > 0.0f)) gi = 0.0f;
        if (arg < 0) gi = 0.0f;
        // weight gradient of channel f through its argmax edge
        if (act && gi != 0.0f) {
            const int jf = src[tperm ? tperm[arg] : arg];
            db += gi;
#pragma unroll
            for (int g = 0; g < kPnMax; ++g)
                if (g < c_out) dw[g] = fmaf(gi, fmaxf(A[(size_t)jf * c_out + g] + crow[g], 0.0f), dw[g]);
        }
        float dci = 0.0f;
        for (int p = lo; p < hi; ++p) {
            const int e = tperm ? tperm[p] : p;
            const bool mine = act && arg == p && gi != 0.0f;
            float dpre = 0.0f;
            if (__ballot(mine)) {                       // wave-uniform: most edges are no channel's argmax
                if (f < kPnMax) v[f] = mine ? gi : 0.0f;
                __builtin_amdgcn_wave_barrier();
                if (act) {
                    float dr = 0.0f;
                    for (int q = 0; q < c_out; ++q) dr = fmaf(sm[f * ld + q], v[q], dr);      // sum_q W2[q][f] v[q]
                    const float pre = A[(size_t)src[e] * c_out + f] + ci;
                    dpre = pre > 0.0f ? dr : 0.0f;
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (act) { dPre[(size_t)e * c_out + f] = dpre; dci += dpre; }
        }
        if (act) dC[(size_t)i * c_out + f] = dci;
        __builtin_amdgcn_wave_barrier();
    }
    float* my = slab + (size_t)(blockIdx.x * 4 + wv) * (c_out * c_out + c_out);
    if (act) {
#pragma unroll
        for (int g = 0; g < kPnMax; ++g)
            if (g < c_out) my[f * c_out + g] = dw[g];
        my[c_out * c_out + f] = db;
    }
}

__global__ __launch_bounds__(256) void pn_bwd_source_kernel(const float* __restrict__ h, const float* __restrict__ pos, int n, int c_in, int c_out,
                                                            const int* __restrict__ sptr, const int* __restrict__ sperm,
                                                            const float* __restrict__ w1, const float* __restrict__ dPre,
                                                            const float* __restrict__ dC, float* __restrict__ gh, float* __restrict__ gpos,
                                                            float* __restrict__ slab) {
    extern __shared__ float sm[];                      // w1 [c_out][kin + 1] row-major, then per wave: dA row | dA - dC row | h row
    const int kin = c_in + 3, ld = kin + 1;
    for (int t = threadIdx.x; t < c_out * kin; t += 256) { const int o = t / kin, i = t - o * kin; sm[o * ld + i] = w1[t]; }
    float* rows = sm + c_out * ld;
    const int wv = threadIdx.x >> 6, f = threadIdx.x & 63;
    __syncthreads();
    const bool act = f < c_out;
    float* da = rows + wv * 3 * kPnMax;
    float* dd = da + kPnMax;
    float* hr = dd + kPnMax;
    float dw[kPnMax + 3];                              // row f of dW1: [h part | pos part]
#pragma unroll
    for (int g = 0; g < kPnMax + 3; ++g) dw[g] = 0.0f;
    float db = 0.0f;
    for (int j = blockIdx.x * 4 + wv; j < n; j += gridDim.x * 4) {
        float a = 0.0f;
        if (act)
            for (int p = sptr[j]; p < sptr[j + 1]; ++p) a += dPre[(size_t)(sperm ? sperm[p] : p) * c_out + f];
        const float c = act ? dC[(size_t)j * c_out + f] : 0.0f;
        if (act) { da[f] = a; dd[f] = a - c; }
        for (int i = f; i < c_in; i += 64) hr[i] = h[(size_t)j * c_in + i];
        const float px = pos[(size_t)j * 3 + 0], py = pos[(size_t)j * 3 + 1], pz = pos[(size_t)j * 3 + 2];
        __builtin_amdgcn_wave_barrier();
        if (gh && f < c_in) {                          // grad_h_j[f] = sum_g Wh[g][f] dA[g]
            float s = 0.0f;
            for (int g = 0; g < c_out; ++g) s = fmaf(sm[g * ld + f], da[g], s);
            gh[(size_t)j * c_in + f] = s;
        }
        if (gpos && f < 3) {                           // grad_pos_j[f] = sum_g Wp[g][f] (dA[g] - dC[g])
            float s = 0.0f;
            for (int g = 0; g < c_out; ++g) s = fmaf(sm[g * ld + c_in + f], dd[g], s);
            gpos[(size_t)j * 3 + f] = s;
        }
        if (act) {
            db += c;
#pragma unroll
            for (int i = 0; i < kPnMax; ++i)
                if (i < c_in) dw[i] = fmaf(a, hr[i], dw[i]);
            const float d = a - c;
            dw[kPnMax + 0] = fmaf(d, px, dw[kPnMax + 0]);
            dw[kPnMax + 1] = fmaf(d, py, dw[kPnMax + 1]);
            dw[kPnMax + 2] = fmaf(d, pz, dw[kPnMax + 2]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    float* my = slab + (size_t)(blockIdx.x * 4 + wv) * (c_out * kin + c_out);
    if (act) {
#pragma unroll
        for (int i = 0; i < kPnMax; ++i)
            if (i < c_in) my[f * kin + i] = dw[i];
        my[f * kin + c_in + 0] = dw[kPnMax + 0];
        my[f * kin + c_in + 1] = dw[kPnMax + 1];
        my[f * kin + c_in + 2] = dw[kPnMax + 2];
        my[c_out * kin + f] = db;
    }
}

// dst[idx] = sum_s slab[s][idx] in slab order; `split`: the first `split` entries go to dst_a, the rest to dst_b
__global__ void pn_slab_sum_kernel(const float* __restrict__ slab, int n_slabs, int size, int split, float* __restrict__ dst_a,
                                   float* __restrict__ dst_b) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= size) return;
    float a = 0.0f;
    for (int s = 0; s < n_slabs; ++s) a += slab[(size_t)s * size + idx];
    if (idx < split) { if (dst_a) dst_a[idx] = a; }
    else if (dst_b) dst_b[idx - split] = a;
}

// global_max_pool backward: grad_x[i][f] = grad_out[b][f] / #ties where x[i][f] is its graph's maximum, else 0
__global__ __launch_bounds__(256) void pn_global_max_bwd_kernel(const float* __restrict__ x, const long long* __restrict__ batch, int n,
                                                                int channels, const float* __restrict__ gout, float* __restrict__ gx) {
    __shared__ float part[256];
    __shared__ int cnt[256];
    const int b = blockIdx.x;
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (batch[mid] < b) lo = mid + 1; else hi = mid; }
    const int start = lo;
    hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (batch[mid] <= b) lo = mid + 1; else hi = mid; }
    const int stop = lo;
    const int f = threadIdx.x % kPnMax, slice = threadIdx.x / kPnMax;
    float m = -INFINITY;
    if (f < channels)
        for (int i = start + slice; i < stop; i += 4) m = fmaxf(m, x[(size_t)i * channels + f]);
    part[threadIdx.x] = m;
    __syncthreads();
    m = fmaxf(fmaxf(part[f], part[kPnMax + f]), fmaxf(part[2 * kPnMax + f], part[3 * kPnMax + f]));
    int k = 0;
    if (f < channels)
        for (int i = start + slice; i < stop; i += 4) k += x[(size_t)i * channels + f] == m ? 1 : 0;
    cnt[threadIdx.x] = k;
    __syncthreads();
    k = cnt[f] + cnt[kPnMax + f] + cnt[2 * kPnMax + f] + cnt[3 * kPnMax + f];
    if (f < channels) {
        const float g = gout[(size_t)b * channels + f] / (float)(k > 0 ? k : 1);
        for (int i = start + slice; i < stop; i += 4) gx[(size_t)i * channels + f] = x[(size_t)i * channels + f] == m ? g : 0.0f;
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

namespace eq {
static int pn_bwd_blocks(int n) { int b = ceil_div(n, 4); return b < 1 ? 1 : (b > 256 ? 256 : b); }
}
/* floats of `work` for equigspr_pointnet_layer_backward */
extern "C" int64_t equigspr_pointnet_layer_backward_work_size(int32_t n_nodes, int32_t n_edges, int32_t c_in, int32_t c_out) {
    using namespace eq;
    if (n_nodes < 1 || n_edges < 0 || c_in < 1 || c_in > kPnMax || c_out < 1 || c_out > kPnMax) return -1;
    const int64_t S = (int64_t)pn_bwd_blocks(n_nodes) * 4;
    return 3 * (int64_t)n_nodes * c_out + (int64_t)(n_edges > 0 ? n_edges : 1) * c_out + S * ((int64_t)c_out * c_out + c_out)
         + S * ((int64_t)c_out * (c_in + 3) + c_out);
}

/* Backward of equigspr_pointnet_layer_forward (ABI 14).  src / tptr / tperm as in the forward; sptr / sperm: the edges sorted by SOURCE
 * (DeviceGraph.rptr / rperm).  grad_out [n, c_out]; grad_h [n, c_in], grad_pos [n, 3] (each may be NULL), grad_w1 [c_out, c_in + 3],
 * grad_b1 [c_out], grad_w2 [c_out, c_out], grad_b2 [c_out] (each may be NULL; written, not accumulated).  The maximum's gradient goes to
 * ONE edge per channel (the first maximum in the target's segment: torch_geometric's max aggregation).  Deterministic, no atomics. */
extern "C" int equigspr_pointnet_layer_backward(const float* h, const float* pos, int32_t n_nodes, int32_t n_edges, int32_t c_in, int32_t c_out,
                                                const int32_t* src, const int32_t* tptr, const int32_t* tperm, const int32_t* sptr,
                                                const int32_t* sperm, const float* w1, const float* b1, const float* w2, const float* b2,
                                                int32_t relu_out, const float* grad_out, float* work, float* grad_h, float* grad_pos,
                                                float* grad_w1, float* grad_b1, float* grad_w2, float* grad_b2, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(h && pos && tptr && sptr && w1 && b1 && w2 && b2 && grad_out && work && n_nodes >= 1 && n_edges >= 0, EQUIGSPR_EINVAL,
                 "pointnet_layer_backward: bad arguments");
    EQ_CHECK_ARG(n_edges == 0 || src, EQUIGSPR_EINVAL, "pointnet_layer_backward: src is NULL");
    EQ_CHECK_ARG(c_in >= 1 && c_in <= kPnMax && c_out >= 1 && c_out <= kPnMax, EQUIGSPR_ENOTSUP,
                 "pointnet_layer_backward: channels must be in [1, 64]");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = pn_bwd_blocks(n_nodes), S = blocks * 4, kin = c_in + 3;
    float* A = work;
    float* C = A + (size_t)n_nodes * c_out;
    float* dC = C + (size_t)n_nodes * c_out;
    float* dPre = dC + (size_t)n_nodes * c_out;
    float* slab2 = dPre + (size_t)(n_edges > 0 ? n_edges : 1) * c_out;
    float* slab1 = slab2 + (size_t)S * (c_out * c_out + c_out);
    int pblocks = ceil_div(n_nodes, 4);
    pblocks = pblocks > 2048 ? 2048 : pblocks;
    const size_t lds1 = ((size_t)kin * (c_out + 1) + 4 * kPnMax) * sizeof(float);
    hipLaunchKernelGGL(pn_project_kernel, dim3(pblocks), dim3(256), lds1, s, h, pos, n_nodes, c_in, c_out, w1, b1, A, C);
    const size_t ldsT = ((size_t)c_out * (c_out + 1) + 4 * 3 * kPnMax) * sizeof(float);
    hipLaunchKernelGGL(pn_bwd_target_kernel, dim3(blocks), dim3(256), ldsT, s, A, C, src, tptr, tperm, n_nodes, c_out, w2, b2, relu_out,
                       grad_out, dPre, dC, slab2);
    const size_t ldsS = ((size_t)c_out * (kin + 1) + 4 * 3 * kPnMax) * sizeof(float);
    hipLaunchKernelGGL(pn_bwd_source_kernel, dim3(blocks), dim3(256), ldsS, s, h, pos, n_nodes, c_in, c_out, sptr, sperm, w1, dPre, dC,
                       grad_h, grad_pos, slab1);
    const int size2 = c_out * c_out + c_out, size1 = c_out * kin + c_out;
    if (grad_w2 || grad_b2)
        hipLaunchKernelGGL(pn_slab_sum_kernel, dim3(ceil_div(size2, 256)), dim3(256), 0, s, slab2, S, size2, c_out * c_out, grad_w2, grad_b2);
    if (grad_w1 || grad_b1)
        hipLaunchKernelGGL(pn_slab_sum_kernel, dim3(ceil_div(size1, 256)), dim3(256), 0, s, slab1, S, size1, c_out * kin, grad_w1, grad_b1);
    EQ_CHECK_LAUNCH("pointnet_layer_backward");
    return 0;
}

extern "C" int equigspr_global_max_pool_backward(const float* x, const int64_t* batch_sorted, int32_t n_nodes, int32_t channels,
                                                 int32_t n_batches, const float* grad_out, float* grad_x, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(x && batch_sorted && grad_out && grad_x && n_nodes >= 1 && n_batches >= 1, EQUIGSPR_EINVAL, "global_max_pool_backward: bad arguments");
    EQ_CHECK_ARG(channels >= 1 && channels <= kPnMax, EQUIGSPR_ENOTSUP, "global_max_pool_backward: channels must be in [1, 64]");
    hipStream_t s = (hipStream_t)stream;
    hipError_t err = hipMemsetAsync(grad_x, 0, sizeof(float) * (size_t)n_nodes * channels, s);      // nodes of a batch id >= n_batches get no gradient
    if (err != hipSuccess) { set_error("global_max_pool_backward memset: %s", hipGetErrorString(err)); return (int)err; }
    hipLaunchKernelGGL(pn_global_max_bwd_kernel, dim3(n_batches), dim3(256), 0, s, x, reinterpret_cast<const long long*>(batch_sorted), n_nodes,
                       channels, grad_out, grad_x);
    EQ_CHECK_LAUNCH("global_max_pool_backward");
    return 0;
}
