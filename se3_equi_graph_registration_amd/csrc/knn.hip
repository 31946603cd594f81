// k-NN graph build + inverse-index (CSR) build for gfx950.
//
// knn: one 64-lane wavefront per query point, brute force over the query's cloud (N <= 8192
// -> <= 128 candidates per lane held in VGPRs).  Selection is exact and deterministic:
//   fast path : tau = k-th smallest of the 64 per-lane minima (an upper bound of the k-th
//               distance), compact the (usually ~1.2k) candidates <= tau through LDS, one
//               64-lane bitonic sort on (d2,index) keys, emit the first k.
//   slow path : k rounds of wave-wide min extraction (taken when more than 64 candidates
//               tie under tau, e.g. heavily duplicated points; also forced by tests).
// Both paths produce the (d2, index)-lexicographic order of oracle.knn_graph bit for bit.
//
// Replaces torch_cluster.knn_graph at src/3dmatch_train_egnn_with_batch.py:1005-1006.
#include "common.h"
#include "scan.h"

namespace eq {

typedef unsigned long long u64;

__device__ __forceinline__ u64 wave_bitonic_sort(u64 key, int lane) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const u64 other = __shfl_xor(key, j, 64);
            const bool up = (lane & k2) == 0;
            const bool lower = (lane & j) == 0;
            const u64 mn = key < other ? key : other;
            const u64 mx = key < other ? other : key;
            key = (lower == up) ? mn : mx;
        }
    }
    return key;
}

// k-th smallest (1-based) of the wave's 64 non-negative floats given as bit patterns (unsigned order = float order): bisection on
// the value with one ballot + popcount per step -- one vector instruction per step against ~300 for a 64-lane bitonic sort.
__device__ __forceinline__ unsigned wave_kth_smallest_bits(unsigned bits, int k) {
    unsigned lo = 0u, hi = 0x7f800000u;                 // +inf: every lane counts
    while (lo < hi) {                                   // wave-uniform
        const unsigned mid = lo + ((hi - lo) >> 1);
        const int c = __popcll(__ballot(bits <= mid));
        if (c >= k) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const u64 o = __shfl_xor(v, j, 64);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ u64 pack_key(float d2, int idx) {
    return ((u64)__float_as_uint(d2) << 32) | (unsigned)idx;
}

template <int NPL>
__global__ __launch_bounds__(64) void knn_kernel(const float* __restrict__ x, int n_clouds, int n, int k,
                                                 int loop, long long* __restrict__ edges,
                                                 int* __restrict__ row32, int* __restrict__ col32,
                                                 int* __restrict__ row_hist, int force_slow) {
    __shared__ u64 list[64];
    const int lane = threadIdx.x;
    const int qg = blockIdx.x;                 // query id over all clouds
    const int cloud = qg / n;
    const int q = qg - cloud * n;
    const float* __restrict__ xc = x + (size_t)cloud * n * 3;
    const float qx = xc[q * 3 + 0], qy = xc[q * 3 + 1], qz = xc[q * 3 + 2];
    const float INF = __uint_as_float(0x7f800000u);

    float d[NPL];
    float lmin = INF;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int idx = i * 64 + lane;
        float v = INF;
        if (idx < n) {
            // exact, contraction-free arithmetic contract (see oracle.knn_graph)
            const float dx = __fsub_rn(qx, xc[idx * 3 + 0]);
            const float dy = __fsub_rn(qy, xc[idx * 3 + 1]);
            const float dz = __fsub_rn(qz, xc[idx * 3 + 2]);
            v = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            if (!loop && idx == q) v = INF;
        }
        d[i] = v;
        lmin = fminf(lmin, v);
    }

    int out_idx = 0;
    // ---- fast path -------------------------------------------------------------------
    const unsigned tau_bits = wave_kth_smallest_bits(__float_as_uint(lmin), k);
    const float tau = __uint_as_float(tau_bits);
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) cnt += __popcll(__ballot(d[i] <= tau));
    const bool fast = (cnt <= 64) && (cnt >= k) && !force_slow;   // wave-uniform
    if (fast) {
        int base = 0;
        const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const bool pred = d[i] <= tau;
            const u64 mask = __ballot(pred);
            if (pred) list[base + __popcll(mask & lt_mask)] = pack_key(d[i], i * 64 + lane);
            base += __popcll(mask);
        }
        __syncthreads();   // single-wave block: orders the LDS writes before the reads below
        u64 key = lane < cnt ? list[lane] : ~0ull;
        key = wave_bitonic_sort(key, lane);
        out_idx = (int)(unsigned)(key & 0xffffffffu);
    } else {
        // ---- slow path: k rounds of "smallest key strictly greater than the previous one"
        u64 last = 0;
        bool have_last = false;
        for (int t = 0; t < k; ++t) {
            u64 best = ~0ull;
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                const u64 key = pack_key(d[i], i * 64 + lane);
                const bool ok = !have_last || key > last;
                best = (ok && key < best) ? key : best;
            }
            best = wave_min_u64(best);
            if (lane == t) out_idx = (int)(unsigned)(best & 0xffffffffu);
            last = best;
            have_last = true;
        }
    }
    if (lane < k) {
        const size_t e = (size_t)q * k + lane;
        const size_t epc = (size_t)n * k;
        if (edges) {
            edges[(size_t)cloud * 2 * epc + e] = out_idx;
            edges[(size_t)cloud * 2 * epc + epc + e] = q;
        }
        if (row32) row32[(size_t)cloud * epc + e] = cloud * n + out_idx;
        if (col32) col32[(size_t)cloud * epc + e] = cloud * n + q;
        if (row_hist) atomicAdd(&row_hist[cloud * n + out_idx], 1);      // the key histogram build_csr(key = row) starts from
    }
}

// ---------------------------------------------------------------------------------------
// Tiled multi-query form (used for n > 2 048): the per-query kernel above re-reads the whole cloud from L2 for every query
// (1.6 GB of L2 traffic at B=16, and its 128-candidates-per-lane variant spills).  Here a workgroup of 4 waves owns 16
// queries of one cloud; candidate tiles of 2 048 points are staged through LDS once per workgroup and every candidate read
// from LDS is tested against the wave's 4 queries.  Two passes over the tiles (lane minima -> tau; count + compaction);
// same lane <-> candidate assignment, same arithmetic, same key order as knn_kernel, so the results are bit-identical.
// The (rare) slow path of a query re-scans the cloud from global memory with its own wave.
// ---------------------------------------------------------------------------------------
constexpr int kKnnTile = 2048, kKnnQpw = 4;
__global__ __launch_bounds__(256) void knn_tiled_kernel(const float* __restrict__ x, int n_clouds, int n, int k, int loop,
                                                        long long* __restrict__ edges, int* __restrict__ row32,
                                                        int* __restrict__ col32, int* __restrict__ row_hist,
                                                        int force_slow) {
    __shared__ float tile[kKnnTile * 3];
    __shared__ u64 list[4 * kKnnQpw][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int bpc = (n + 4 * kKnnQpw - 1) / (4 * kKnnQpw);      // workgroups per cloud
    const int cloud = blockIdx.x / bpc;
    const int q0 = (blockIdx.x - cloud * bpc) * 4 * kKnnQpw + wv * kKnnQpw;
    const float* __restrict__ xc = x + (size_t)cloud * n * 3;
    const float INF = __uint_as_float(0x7f800000u);
    float qx[kKnnQpw], qy[kKnnQpw], qz[kKnnQpw];
    int qi[kKnnQpw];
#pragma unroll
    for (int u = 0; u < kKnnQpw; ++u) {
        qi[u] = q0 + u;
        const int qq = min(qi[u], n - 1);
        qx[u] = xc[qq * 3 + 0]; qy[u] = xc[qq * 3 + 1]; qz[u] = xc[qq * 3 + 2];
    }
    auto dist2 = [&](int u, float px, float py, float pz) {
        // exact, contraction-free arithmetic contract (see oracle.knn_graph)
        const float dx = __fsub_rn(qx[u], px), dy = __fsub_rn(qy[u], py), dz = __fsub_rn(qz[u], pz);
        return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    };
    auto stage = [&](int t0, int cnt) {
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * 3; i += 256) tile[i] = xc[(size_t)t0 * 3 + i];
        __syncthreads();
    };
    // ---- pass 1: per-lane minima
    float lmin[kKnnQpw];
#pragma unroll
    for (int u = 0; u < kKnnQpw; ++u) lmin[u] = INF;
    for (int t0 = 0; t0 < n; t0 += kKnnTile) {
        const int cnt = min(kKnnTile, n - t0);
        stage(t0, cnt);
        for (int j = lane; j < cnt; j += 64) {
            const float px = tile[3 * j], py = tile[3 * j + 1], pz = tile[3 * j + 2];
            const int idx = t0 + j;
#pragma unroll
            for (int u = 0; u < kKnnQpw; ++u) {
                float v = dist2(u, px, py, pz);
                if (!loop && idx == qi[u]) v = INF;
                lmin[u] = fminf(lmin[u], v);
            }
        }
    }
    float tau[kKnnQpw];
#pragma unroll
    for (int u = 0; u < kKnnQpw; ++u) {
        tau[u] = __uint_as_float(wave_kth_smallest_bits(__float_as_uint(lmin[u]), k));
    }
    // ---- pass 2: count the candidates <= tau and compact the first 64 of them
    int base[kKnnQpw];
#pragma unroll
    for (int u = 0; u < kKnnQpw; ++u) base[u] = 0;
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int t0 = 0; t0 < n; t0 += kKnnTile) {
        const int cnt = min(kKnnTile, n - t0);
        stage(t0, cnt);
        for (int j0 = 0; j0 < cnt; j0 += 64) {             // wave-uniform trip count (ballots inside)
            const int j = j0 + lane;
            const bool in = j < cnt;
            const float px = in ? tile[3 * j] : 0.0f, py = in ? tile[3 * j + 1] : 0.0f, pz = in ? tile[3 * j + 2] : 0.0f;
            const int idx = t0 + j;
#pragma unroll
            for (int u = 0; u < kKnnQpw; ++u) {
                float v = dist2(u, px, py, pz);
                if (!in || (!loop && idx == qi[u])) v = INF;
                const bool pred = v <= tau[u];
                const u64 mask = __ballot(pred);
                const int at = base[u] + __popcll(mask & lt_mask);
                if (pred && at < 64) list[wv * kKnnQpw + u][at] = pack_key(v, idx);
                base[u] += __popcll(mask);
            }
        }
    }
    __syncthreads();                                       // LDS list writes before the reads below (and the tile is free)
#pragma unroll
    for (int u = 0; u < kKnnQpw; ++u) {
        if (qi[u] >= n) continue;                          // wave-uniform
        const int cnt = base[u];
        int out_idx = 0;
        if ((cnt <= 64) && (cnt >= k) && !force_slow) {
            u64 key = lane < cnt ? list[wv * kKnnQpw + u][lane] : ~0ull;
            key = wave_bitonic_sort(key, lane);
            out_idx = (int)(unsigned)(key & 0xffffffffu);
        } else {
            // slow path: k rounds of "smallest key strictly greater than the previous one", distances recomputed from global
            u64 last = 0;
            bool have_last = false;
            for (int t = 0; t < k; ++t) {
                u64 best = ~0ull;
                for (int idx = lane; idx < n; idx += 64) {
                    float v = dist2(u, xc[idx * 3 + 0], xc[idx * 3 + 1], xc[idx * 3 + 2]);
                    if (!loop && idx == qi[u]) v = INF;
                    const u64 key = pack_key(v, idx);
                    const bool ok = !have_last || key > last;
                    best = (ok && key < best) ? key : best;
                }
                best = wave_min_u64(best);
                if (lane == t) out_idx = (int)(unsigned)(best & 0xffffffffu);
                last = best;
                have_last = true;
            }
        }
        if (lane < k) {
            const int q = qi[u];
            const size_t e = (size_t)q * k + lane;
            const size_t epc = (size_t)n * k;
            if (edges) {
                edges[(size_t)cloud * 2 * epc + e] = out_idx;
                edges[(size_t)cloud * 2 * epc + epc + e] = q;
            }
            if (row32) row32[(size_t)cloud * epc + e] = cloud * n + out_idx;
            if (col32) col32[(size_t)cloud * epc + e] = cloud * n + q;
            if (row_hist) atomicAdd(&row_hist[cloud * n + out_idx], 1);
        }
    }
}

// ---------------------------------------------------------------------------------------
// CSR build: stable counting sort of edge ids by key
// ---------------------------------------------------------------------------------------
__global__ void csr_count_kernel(const int* __restrict__ key, int n_edges, int* __restrict__ cnt) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_edges) atomicAdd(&cnt[key[e]], 1);
}

// exclusive scan of cnt[0..n) -> ptr[0..n], cursor[0..n) = ptr[0..n) ; single block of 1024
__global__ void csr_fill_kernel(const int* __restrict__ key, int n_edges, int* __restrict__ cursor,
                                int* __restrict__ perm) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_edges) perm[atomicAdd(&cursor[key[e]], 1)] = e;
}

// each node's segment back into ascending edge order (atomics filled it in arbitrary order):
// one wave per node, <= 64 entries sorted with a wave bitonic network, longer segments serially.  The same wave writes the
// inverse permutation of its segment (inv[perm[p]] = p, optional) and hands the node's cursor word back as zero, so the
// scratch array leaves build_csr the way it must enter it.
__global__ __launch_bounds__(256) void csr_sort_kernel(const int* __restrict__ ptr, int n_nodes, int* __restrict__ perm,
                                                       int* __restrict__ inv, int* __restrict__ scratch) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_nodes) return;                     // wave-uniform
    const int lo = ptr[j], hi = ptr[j + 1];
    const int len = hi - lo;
    if (lane == 0) scratch[j] = 0;
    if (len <= 64) {
        // distinct edge ids: a lane's final position is the number of smaller ids in the segment -- `len` broadcast steps of one
        // v_readlane + compare + add (a k = 20 neighbourhood: ~60 instructions) instead of the 21-stage 64-lane bitonic network
        const int key = lane < len ? perm[lo + lane] : 0x7fffffff;
        int rank = 0;
        for (int t = 0; t < len; ++t) rank += (__builtin_amdgcn_readlane(key, t) < key) ? 1 : 0;      // t is wave-uniform
        if (lane < len) {
            if (len > 1) perm[lo + rank] = key;
            if (inv) inv[key] = lo + rank;
        }
    } else {
        if (lane == 0) {
            for (int a = lo + 1; a < hi; ++a) {
                const int v = perm[a];
                int b = a - 1;
                while (b >= lo && perm[b] > v) { perm[b + 1] = perm[b]; --b; }
                perm[b + 1] = v;
            }
        }
        if (inv) {
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            for (int a = lo + lane; a < hi; a += 64) inv[perm[a]] = a;
        }
    }
}

__global__ void edges_i64_to_i32_kernel(const long long* __restrict__ edges, int n_edges, int off,
                                        int* __restrict__ row32, int* __restrict__ col32) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_edges) {
        row32[e] = (int)edges[e] + off;
        col32[e] = (int)edges[(size_t)n_edges + e] + off;
    }
}

}  // namespace eq

#ifndef EQ_KNN_PER_QUERY_MAX
#define EQ_KNN_PER_QUERY_MAX 2048
#endif
extern "C" int equigspr_knn_graph(const float* x, int32_t n_clouds, int32_t n, int32_t k, int32_t loop,
                                  int64_t* edges_i64, int32_t* row32, int32_t* col32, int32_t* row_hist,
                                  int32_t force_slow_path, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(x != nullptr, EQUIGSPR_EINVAL, "knn_graph: x is null");
    EQ_CHECK_ARG(n_clouds >= 1 && n >= 1, EQUIGSPR_EINVAL, "knn_graph: need n_clouds>=1, n>=1");
    EQ_CHECK_ARG(n <= 8192, EQUIGSPR_ENOTSUP, "knn_graph: n=%d > 8192 points per cloud", n);
    EQ_CHECK_ARG(k >= 1 && k <= 64, EQUIGSPR_ENOTSUP, "knn_graph: k=%d outside [1,64]", k);
    EQ_CHECK_ARG(k <= (loop ? n : n - 1), EQUIGSPR_EINVAL, "knn_graph: k=%d exceeds candidates (n=%d, loop=%d)", k, n, loop);
    EQ_CHECK_ARG((long long)n_clouds * n * k < (1ll << 31), EQUIGSPR_EINVAL, "knn_graph: edge count overflows int32");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(n_clouds * n), block(64);
    long long* e64 = reinterpret_cast<long long*>(edges_i64);
    if (n <= 256) {
        hipLaunchKernelGGL(knn_kernel<4>, grid, block, 0, s, x, n_clouds, n, k, loop, e64, row32, col32, row_hist, force_slow_path);
    } else if (n <= EQ_KNN_PER_QUERY_MAX) {      // measured: the per-query kernel is as fast as the tiled one up to 2 048 points (17.6 us at B=1)
        hipLaunchKernelGGL(knn_kernel<32>, grid, block, 0, s, x, n_clouds, n, k, loop, e64, row32, col32, row_hist, force_slow_path);
    } else {                     // 4 096 / 8 192 points: 534 -> ~120 us for two 8 192-point clouds (no 128-value register arrays)
        const int bpc = ceil_div(n, 4 * kKnnQpw);
        hipLaunchKernelGGL(knn_tiled_kernel, dim3(n_clouds * bpc), dim3(256), 0, s, x, n_clouds, n, k, loop, e64, row32, col32,
                           row_hist, force_slow_path);
    }
    EQ_CHECK_LAUNCH("knn_kernel");
    return 0;
}

extern "C" int equigspr_build_csr(const int32_t* key, int32_t n_edges, int32_t n_nodes, int32_t* ptr,
                                  int32_t* perm, int32_t* inv, int32_t* scratch, int32_t scratch_state, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(ptr && scratch && (n_edges == 0 || (key && perm)), EQUIGSPR_EINVAL, "build_csr: null pointer");
    EQ_CHECK_ARG(n_edges >= 0 && n_nodes >= 1, EQUIGSPR_EINVAL, "build_csr: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    // (a single-block LDS count+scan was tried instead of clear + count + scan: 25 us vs ~12 us -- one block cannot hide
    // the key-load latency.)  The clear is skipped for a scratch the caller knows to be zero: the sort pass re-zeroes every
    // word the count/fill passes touched, so a buffer can be handed from call to call (hipMemsetAsync of N+1 words is
    // two fill launches -- an aligned body and a tail -- i.e. ~10 us of a 0.5 ms step)
    EQ_CHECK_ARG(scratch_state >= 0 && scratch_state <= 2, EQUIGSPR_EINVAL, "build_csr: scratch_state must be 0, 1 or 2");
    if (scratch_state == EQUIGSPR_SCRATCH_DIRTY) {
        hipError_t err = hipMemsetAsync(scratch, 0, sizeof(int) * (size_t)(n_nodes + 1), s);
        if (err != hipSuccess) { set_error("build_csr memset: %s", hipGetErrorString(err)); return (int)err; }
    }
    if (n_edges > 0 && scratch_state != EQUIGSPR_SCRATCH_HISTOGRAM)      // (equigspr_knn_graph can leave the histogram behind)
        hipLaunchKernelGGL(csr_count_kernel, dim3(ceil_div(n_edges, 256)), dim3(256), 0, s, key, n_edges, scratch);
    // large N: two-level scan; its chunk totals borrow the head of `perm`, which is only written by the fill below
    const int chunks = ceil_div(n_nodes, kScanChunk);
    launch_exclusive_scan(scratch, n_nodes, ptr, (n_edges >= chunks && chunks <= 1024) ? perm : nullptr, s);
    if (n_edges > 0) {
        hipLaunchKernelGGL(csr_fill_kernel, dim3(ceil_div(n_edges, 256)), dim3(256), 0, s, key, n_edges, scratch, perm);
        hipLaunchKernelGGL(csr_sort_kernel, dim3(ceil_div(n_nodes, 4)), dim3(256), 0, s, ptr, n_nodes, perm, inv, scratch);
    }
    EQ_CHECK_LAUNCH("build_csr");
    return 0;
}

namespace eq {
__global__ void invert_perm_kernel(const int* __restrict__ perm, int n, int* __restrict__ inv) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) inv[perm[p]] = p;
}
}  // namespace eq

extern "C" int equigspr_invert_perm(const int32_t* perm, int32_t n, int32_t* inv, void* stream) {
    using namespace eq;
    if (n == 0) return 0;
    EQ_CHECK_ARG(perm && inv && n > 0, EQUIGSPR_EINVAL, "invert_perm: bad arguments");
    hipLaunchKernelGGL(invert_perm_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, perm, n, inv);
    EQ_CHECK_LAUNCH("invert_perm");
    return 0;
}

extern "C" int equigspr_edges_i64_to_i32(const int64_t* edges_i64, int32_t n_edges, int32_t node_offset,
                                         int32_t* row32, int32_t* col32, void* stream) {
    using namespace eq;
    if (n_edges == 0) return 0;                      // an empty edge list has no storage to point at
    EQ_CHECK_ARG(edges_i64 && row32 && col32 && n_edges > 0, EQUIGSPR_EINVAL, "edges_i64_to_i32: bad args");
    hipLaunchKernelGGL(edges_i64_to_i32_kernel, dim3(ceil_div(n_edges, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(edges_i64), n_edges, node_offset, row32, col32);
    EQ_CHECK_LAUNCH("edges_i64_to_i32");
    return 0;
}
