// gcnLayer (src/gcnLayer.py) for gfx950 -- SURVEY §8(a) row A17.
//   build_graph :58-67 (+ _get_weights :28-46, get_normalize_adj :49-56): Gaussian-weighted kNN adjacency, no
//     self loops, symmetrised by max, normalised D^-1/2 A D^-1/2 -- kept SPARSE here (CSR by node, <= k-1 out-
//     plus the unmatched in-neighbours), never as the reference's [N,N] matrix;
//   GraphConvolution.forward :77-80: Conv1d_1x1(bmm(x, adj)) fused into one gather kernel over that CSR;
//   GlobalPooling.forward :90-94: max || mean over the points.
// Edge weights are formed in fp64 from the fp32 coordinates (the reference runs scipy/numpy in fp64) and
// stored as fp32.  HBM-bound integer/gather work: one thread per node, no atomics, deterministic order.
#include "common.h"
#include "scan.h"

namespace eq {

__device__ inline double dist64(const float* __restrict__ x, int a, int b) {
    const double dx = (double)x[3 * a] - (double)x[3 * b], dy = (double)x[3 * a + 1] - (double)x[3 * b + 1],
                 dz = (double)x[3 * a + 2] - (double)x[3 * b + 2];
    return sqrt(dx * dx + dy * dy + dz * dz);
}

// sigma2[c] = mean_i(dist(i, k-th neighbour of i))^2  (_get_weights :36)
__global__ __launch_bounds__(256) void gcn_sigma_kernel(const float* __restrict__ x, const int* __restrict__ nbr, int n, int k,
                                                        double* __restrict__ sigma2) {
    __shared__ double part[256];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int g = c * n + i;
        s += dist64(x, g, nbr[(size_t)g * k + (k - 1)]);
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double m = part[0] / (double)n;
        sigma2[c] = m * m;
    }
}

// true when centre `c` already lists `i` ... i.e. whether j is among the out-neighbours of i
__device__ inline bool is_out_neighbour(const int* __restrict__ nbr_i, int k, int j) {
    bool hit = false;
    for (int q = 0; q < k; ++q) hit |= (nbr_i[q] == j);
    return hit;
}

// pass 0 (kFill = false): cnt[i], row sums -> dinv[i];  pass 1 (kFill = true): write (idx, unnormalised weight)
template <bool kFill>
__global__ void gcn_adj_kernel(const float* __restrict__ x, const int* __restrict__ nbr, const int* __restrict__ col,
                               const int* __restrict__ rptr, const int* __restrict__ rperm, int n_total, int n, int k,
                               const double* __restrict__ sigma2, int* __restrict__ cnt, float* __restrict__ dinv,
                               const int* __restrict__ ptr, int* __restrict__ idx, float* __restrict__ val) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const int cloud = i / n;
    const double inv_s2 = 1.0 / sigma2[cloud];
    const int* nb = nbr + (size_t)i * k;
    int count = 0;
    double rsum = 0.0;
    int w = kFill ? ptr[i] : 0;
    for (int q = 0; q < k; ++q) {                       // out-neighbours (the node itself is dropped: setdiag(0) :41)
        const int j = nb[q];
        if (j == i) continue;
        const double d = dist64(x, i, j), wt = exp(-d * d * inv_s2);
        if (kFill) { idx[w] = j - cloud * n; val[w] = (float)wt; ++w; }
        ++count; rsum += wt;
    }
    for (int p = rptr[i]; p < rptr[i + 1]; ++p) {       // centres that list i (weights.T > weights fills these, :43-44)
        const int j = col[rperm[p]];
        if (j == i || is_out_neighbour(nb, k, j)) continue;
        const double d = dist64(x, i, j), wt = exp(-d * d * inv_s2);
        if (kFill) { idx[w] = j - cloud * n; val[w] = (float)wt; ++w; }
        ++count; rsum += wt;
    }
    if (!kFill) {
        cnt[i] = count;
        dinv[i] = rsum > 0.0 ? (float)(1.0 / sqrt(rsum)) : 0.0f;     // d_inv[isinf] = 0 (:54)
    }
}

// val[p] *= dinv[i] * dinv[j]   (get_normalize_adj :56)
__global__ void gcn_normalize_kernel(const int* __restrict__ ptr, const int* __restrict__ idx, const float* __restrict__ dinv,
                                     int n_total, int n, float* __restrict__ val) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const int base = (i / n) * n;
    const double di = (double)dinv[i];
    for (int p = ptr[i]; p < ptr[i + 1]; ++p) val[p] = (float)((double)val[p] * di * (double)dinv[base + idx[p]]);
}

__global__ void gcn_dense_kernel(const int* __restrict__ ptr, const int* __restrict__ idx, const float* __restrict__ val,
                                 int n_total, int n, float* __restrict__ dense) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    float* row = dense + (size_t)i * n;                 // [cloud][i_local][:]
    for (int p = ptr[i]; p < ptr[i + 1]; ++p) row[idx[p]] = val[p];
}

// y[b][o][i] = bias[o] + sum_c W[o][c] * agg[c],  agg[c] = sum_{(j,v) in adj(i)} v * x[b][c][j]   (kSparse)
//                                                   agg[c] = xin[b][c][i]                          (!kSparse: bmm done by caller)
constexpr int kGcnMaxC = 64;
template <bool kSparse>
__global__ __launch_bounds__(256) void gcn_conv_kernel(const float* __restrict__ xin, int n_batch, int c_in, int c_out, int n,
                                                       const int* __restrict__ ptr, const int* __restrict__ idx,
                                                       const float* __restrict__ val, int adj_clouds,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y) {
    extern __shared__ float wsm[];                      // W [c_out][c_in] + bias
    for (int t = threadIdx.x; t < c_out * c_in; t += blockDim.x) wsm[t] = w[t];
    for (int t = threadIdx.x; t < c_out; t += blockDim.x) wsm[c_out * c_in + t] = bias ? bias[t] : 0.0f;
    __syncthreads();
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= n_batch * n) return;
    const int b = gi / n, i = gi - b * n;
    const float* xb = xin + (size_t)b * c_in * n;
    float agg[kGcnMaxC];
#pragma unroll
    for (int c = 0; c < kGcnMaxC; ++c) agg[c] = 0.0f;
    if (kSparse) {
        const int node = (adj_clouds == 1 ? 0 : b) * n + i;
        for (int p = ptr[node]; p < ptr[node + 1]; ++p) {
            const int j = idx[p];
            const float v = val[p];
#pragma unroll
            for (int c = 0; c < kGcnMaxC; ++c)
                if (c < c_in) agg[c] = fmaf(v, xb[(size_t)c * n + j], agg[c]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < kGcnMaxC; ++c)
            if (c < c_in) agg[c] = xb[(size_t)c * n + i];
    }
    for (int o = 0; o < c_out; ++o) {
        float s = wsm[c_out * c_in + o];
#pragma unroll
        for (int c = 0; c < kGcnMaxC; ++c)
            if (c < c_in) s = fmaf(wsm[o * c_in + c], agg[c], s);
        y[((size_t)b * c_out + o) * n + i] = s;
    }
}

// out[b][c] = max_i x[b][c][i],  out[b][C + c] = mean_i x[b][c][i]
__global__ __launch_bounds__(256) void gcn_pool_kernel(const float* __restrict__ x, int channels, int n, float* __restrict__ out) {
    __shared__ float smax[256];
    __shared__ float ssum[256];
    const int b = blockIdx.x / channels, c = blockIdx.x - b * channels;
    const float* row = x + ((size_t)b * channels + c) * n;
    float m = -INFINITY, s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) { const float v = row[i]; m = fmaxf(m, v); s += v; }
    smax[threadIdx.x] = m; ssum[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + off]);
            ssum[threadIdx.x] += ssum[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[(size_t)b * 2 * channels + c] = smax[0];
        out[(size_t)b * 2 * channels + channels + c] = ssum[0] / (float)n;
    }
}

// ---- backward of the layer (round 6; the reference never trains it, gcnLayer.py:100-107 is a demo: here it is a trainable module) ----------
// agg[b][c][i] = sum_{(j,v) in adj(i)} v x[b][c][j]: the aggregation half of the forward alone (for the weight gradient)
__global__ __launch_bounds__(256) void gcn_agg_kernel(const float* __restrict__ x, int n_batch, int c_in, int n, const int* __restrict__ ptr,
                                                      const int* __restrict__ idx, const float* __restrict__ val, int adj_clouds,
                                                      float* __restrict__ agg) {
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= n_batch * n) return;
    const int b = gi / n, i = gi - b * n;
    const float* xb = x + (size_t)b * c_in * n;
    const int node = (adj_clouds == 1 ? 0 : b) * n + i;
    const int lo = ptr[node], hi = ptr[node + 1];
    for (int c = 0; c < c_in; ++c) {
        float a = 0.0f;
        for (int p = lo; p < hi; ++p) a = fmaf(val[p], xb[(size_t)c * n + idx[p]], a);
        agg[((size_t)b * c_in + c) * n + i] = a;
    }
}
// dW[o][c] = sum_{b,i} dy[b][o][i] agg[b][c][i]; blockIdx.y == c_in: db[o] = sum_{b,i} dy[b][o][i].  One workgroup per entry, fixed tree.
__global__ __launch_bounds__(256) void gcn_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ agg, int n_batch, int c_in,
                                                        int c_out, int n, float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[256];
    const int o = blockIdx.x, c = blockIdx.y;
    const bool bias = c == c_in;
    float a = 0.0f;
    for (int b = 0; b < n_batch; ++b) {
        const float* g = dy + ((size_t)b * c_out + o) * n;
        const float* r = bias ? nullptr : agg + ((size_t)b * c_in + c) * n;
        for (int i = threadIdx.x; i < n; i += 256) a = bias ? a + g[i] : fmaf(g[i], r[i], a);
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (bias) { if (db) db[o] = red[0]; }
        else dw[(size_t)o * c_in + c] = red[0];
    }
}
// GlobalPooling backward: grad_x[b][c][i] = g_mean / n + (x == max ? g_max / #ties : 0)   (torch.amax's even split between ties)
__global__ __launch_bounds__(256) void gcn_pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout, int channels, int n,
                                                           float* __restrict__ gx) {
    __shared__ float smax[256];
    __shared__ int scnt[256];
    const int b = blockIdx.x / channels, c = blockIdx.x - b * channels;
    const float* row = x + ((size_t)b * channels + c) * n;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, row[i]);
    smax[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + off]);
        __syncthreads();
    }
    m = smax[0];
    int k = 0;
    for (int i = threadIdx.x; i < n; i += 256) k += row[i] == m ? 1 : 0;
    scnt[threadIdx.x] = k;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) scnt[threadIdx.x] += scnt[threadIdx.x + off];
        __syncthreads();
    }
    const float gmax = gout[(size_t)b * 2 * channels + c] / (float)scnt[0], gmean = gout[(size_t)b * 2 * channels + channels + c] / (float)n;
    float* out = gx + ((size_t)b * channels + c) * n;
    for (int i = threadIdx.x; i < n; i += 256) out[i] = gmean + (row[i] == m ? gmax : 0.0f);
}

}  // namespace eq

extern "C" int equigspr_gcn_build_graph(const float* coords, int32_t n_clouds, int32_t n, int32_t k, const int32_t* nbr,
                                        const int32_t* col, const int32_t* rptr, const int32_t* rperm, double* sigma2,
                                        int32_t* cnt_scratch, float* dinv, int32_t* adj_ptr, int32_t* adj_idx, float* adj_val,
                                        void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(coords && nbr && col && rptr && rperm && sigma2 && cnt_scratch && dinv && adj_ptr && adj_idx && adj_val,
                 EQUIGSPR_EINVAL, "gcn_build_graph: null pointer");
    EQ_CHECK_ARG(n_clouds >= 1 && n >= 2 && k >= 2 && k <= n && k <= 64, EQUIGSPR_EINVAL, "gcn_build_graph: need 2 <= k <= min(n, 64)");
    hipStream_t s = (hipStream_t)stream;
    const int nt = n_clouds * n;
    hipLaunchKernelGGL(gcn_sigma_kernel, dim3(n_clouds), dim3(256), 0, s, coords, nbr, n, k, sigma2);
    hipLaunchKernelGGL(gcn_adj_kernel<false>, dim3(ceil_div(nt, 128)), dim3(128), 0, s, coords, nbr, col, rptr, rperm, nt, n, k,
                       sigma2, cnt_scratch, dinv, (const int*)nullptr, (int*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(csr_scan_kernel<0>, dim3(1), dim3(1024), 0, s, cnt_scratch, nt, adj_ptr);
    hipLaunchKernelGGL(gcn_adj_kernel<true>, dim3(ceil_div(nt, 128)), dim3(128), 0, s, coords, nbr, col, rptr, rperm, nt, n, k,
                       sigma2, (int*)nullptr, (float*)nullptr, adj_ptr, adj_idx, adj_val);
    hipLaunchKernelGGL(gcn_normalize_kernel, dim3(ceil_div(nt, 128)), dim3(128), 0, s, adj_ptr, adj_idx, dinv, nt, n, adj_val);
    EQ_CHECK_LAUNCH("gcn_build_graph");
    return 0;
}

extern "C" int equigspr_gcn_adjacency_to_dense(const int32_t* adj_ptr, const int32_t* adj_idx, const float* adj_val,
                                               int32_t n_clouds, int32_t n, float* dense_zeroed, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(adj_ptr && adj_idx && adj_val && dense_zeroed && n_clouds >= 1 && n >= 1, EQUIGSPR_EINVAL, "gcn_adjacency_to_dense: bad args");
    const int nt = n_clouds * n;
    hipLaunchKernelGGL(gcn_dense_kernel, dim3(ceil_div(nt, 128)), dim3(128), 0, (hipStream_t)stream, adj_ptr, adj_idx, adj_val, nt, n,
                       dense_zeroed);
    EQ_CHECK_LAUNCH("gcn_adjacency_to_dense");
    return 0;
}

extern "C" int equigspr_gcn_conv_forward(const float* x, int32_t n_batch, int32_t c_in, int32_t c_out, int32_t n,
                                         const int32_t* adj_ptr, const int32_t* adj_idx, const float* adj_val, int32_t adj_clouds,
                                         const float* weight, const float* bias, float* y, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(x && weight && y && n_batch >= 1 && n >= 1 && c_in >= 1 && c_out >= 1, EQUIGSPR_EINVAL, "gcn_conv_forward: bad args");
    EQ_CHECK_ARG(c_in <= kGcnMaxC && (size_t)(c_out * c_in + c_out) * 4 <= 64 * 1024, EQUIGSPR_ENOTSUP,
                 "gcn_conv_forward: in_features <= 64 and (in+1)*out <= 16384");
    const bool sparse = adj_ptr != nullptr;
    EQ_CHECK_ARG(!sparse || (adj_idx && adj_val && (adj_clouds == 1 || adj_clouds == n_batch)), EQUIGSPR_EINVAL,
                 "gcn_conv_forward: adjacency must hold 1 or n_batch clouds");
    const size_t lds = (size_t)(c_out * c_in + c_out) * sizeof(float);
    const dim3 grid(ceil_div((long long)n_batch * n, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (sparse)
        hipLaunchKernelGGL(gcn_conv_kernel<true>, grid, block, lds, s, x, n_batch, c_in, c_out, n, adj_ptr, adj_idx, adj_val, adj_clouds,
                           weight, bias, y);
    else
        hipLaunchKernelGGL(gcn_conv_kernel<false>, grid, block, lds, s, x, n_batch, c_in, c_out, n, adj_ptr, adj_idx, adj_val, 0, weight,
                           bias, y);
    EQ_CHECK_LAUNCH("gcn_conv_forward");
    return 0;
}

extern "C" int equigspr_gcn_global_pool(const float* x, int32_t n_batch, int32_t channels, int32_t n, float* out, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(x && out && n_batch >= 1 && channels >= 1 && n >= 1, EQUIGSPR_EINVAL, "gcn_global_pool: bad args");
    hipLaunchKernelGGL(gcn_pool_kernel, dim3(n_batch * channels), dim3(256), 0, (hipStream_t)stream, x, channels, n, out);
    EQ_CHECK_LAUNCH("gcn_global_pool");
    return 0;
}

/* Backward of equigspr_gcn_conv_forward for the weights: agg_scratch [n_batch, c_in, n] (sparse adjacency: filled here; dense path: the
 * caller passes x @ adj in it and adj_ptr == NULL), grad_w [c_out, c_in], grad_b [c_out] or NULL.  The input gradient is the forward
 * itself on grad_y with the transposed weight (the normalised adjacency is symmetric). */
extern "C" int equigspr_gcn_conv_backward_weights(const float* x, const float* grad_y, int32_t n_batch, int32_t c_in, int32_t c_out, int32_t n,
                                                  const int32_t* adj_ptr, const int32_t* adj_idx, const float* adj_val, int32_t adj_clouds,
                                                  float* agg_scratch, float* grad_w, float* grad_b, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(grad_y && agg_scratch && grad_w && n_batch >= 1 && n >= 1 && c_in >= 1 && c_out >= 1, EQUIGSPR_EINVAL,
                 "gcn_conv_backward_weights: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (adj_ptr) {
        EQ_CHECK_ARG(x && adj_idx && adj_val && (adj_clouds == 1 || adj_clouds == n_batch), EQUIGSPR_EINVAL,
                     "gcn_conv_backward_weights: adjacency must hold 1 or n_batch clouds");
        hipLaunchKernelGGL(gcn_agg_kernel, dim3(ceil_div((long long)n_batch * n, 256)), dim3(256), 0, s, x, n_batch, c_in, n, adj_ptr, adj_idx,
                           adj_val, adj_clouds, agg_scratch);
    }
    hipLaunchKernelGGL(gcn_wgrad_kernel, dim3(c_out, c_in + 1), dim3(256), 0, s, grad_y, agg_scratch, n_batch, c_in, c_out, n, grad_w, grad_b);
    EQ_CHECK_LAUNCH("gcn_conv_backward_weights");
    return 0;
}

extern "C" int equigspr_gcn_global_pool_backward(const float* x, const float* grad_out, int32_t n_batch, int32_t channels, int32_t n,
                                                 float* grad_x, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(x && grad_out && grad_x && n_batch >= 1 && channels >= 1 && n >= 1, EQUIGSPR_EINVAL, "gcn_global_pool_backward: bad args");
    hipLaunchKernelGGL(gcn_pool_bwd_kernel, dim3(n_batch * channels), dim3(256), 0, (hipStream_t)stream, x, grad_out, channels, n, grad_x);
    EQ_CHECK_LAUNCH("gcn_global_pool_backward");
    return 0;
}
