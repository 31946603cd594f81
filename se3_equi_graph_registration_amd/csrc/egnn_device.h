// Device-side building blocks shared by the forward and backward EGNN translation units.
#pragma once
#include "common.h"

namespace eq {

template <int F> struct Geom {
    static constexpr int FP = (F + 3) / 4 * 4;     // padded row (16-byte rows)
    static constexpr int G = F <= 32 ? 32 : 64;    // lanes per node in node kernels
    static constexpr int NS = 256 / G;             // node slots per 256-thread block
    static constexpr int PS = 2 * FP;              // P row: [Pr (FP) | Pc (FP)]
};

// y = sum_i W[row][i] * v[i], W row-major with row stride `ld`, v in LDS (broadcast reads)
__device__ __forceinline__ float row_dot(const float* __restrict__ wrow, const float* v, int n) {
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) acc = fmaf(wrow[i], v[i], acc);
    return acc;
}

// ---- latency-tolerant segment sums --------------------------------------------------------
// acc_f += sum over a in [lo,hi) of data[perm[a]][f] (and, for lanes f < 3, of data3[perm[a]][f]),
// strictly in ascending a (the CPU scatter_add_ order).  The G lanes of a node first fetch up to
// G indices with ONE coalesced load, broadcast them with shuffles, and keep 16 independent row
// loads in flight per lane before the (ordered) adds -- instead of a dependent index->row chain.
template <int G, bool kHas3, int D = 16>
__device__ __forceinline__ void segment_sum(const int* __restrict__ perm, int lo, int hi, const float* __restrict__ data,
                                            int stride, const float* __restrict__ data3, int f, bool f_active,
                                            float& acc, float& acc3) {
    for (int base = lo; base < hi; base += G) {
        const int cnt = min(G, hi - base);
        const int my = (f < cnt) ? perm[base + f] : 0;
        for (int a0 = 0; a0 < cnt; a0 += D) {
            float v[D], w[D];
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int e = __shfl(my, (a0 + u) & (G - 1), G);
                const bool in = (a0 + u) < cnt;
                v[u] = (in && f_active) ? data[(size_t)e * stride + f] : 0.0f;
                if (kHas3) w[u] = (in && f < 3) ? data3[(size_t)e * 4 + f] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < D; ++u) {
                acc += v[u];
                if (kHas3) acc3 += w[u];
            }
        }
    }
}

// The same sum over rows that already lie in segment order (the edge kernels store at rinv[e]): rows lo..hi-1 are
// contiguous, so there is no index load and every wave-level load is one or two full 128-byte lines.  Same ascending
// order of additions as segment_sum.
template <int G, int D = 16>
__device__ __forceinline__ void segment_sum_sorted(int lo, int hi, const float* __restrict__ data, int stride,
                                                   const float* __restrict__ data3, int f, bool f_active, float& acc, float& acc3) {
    for (int a0 = lo; a0 < hi; a0 += D) {
        float v[D], w[D];
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const bool in = (a0 + u) < hi;
            v[u] = (in && f_active) ? data[(size_t)(a0 + u) * stride + f] : 0.0f;
            w[u] = (in && f < 3) ? data3[(size_t)(a0 + u) * 4 + f] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < D; ++u) { acc += v[u]; acc3 += w[u]; }
    }
}

// Two segments of the same node at once (gather_bwd: the edges where it is the neighbour and those where it is the centre):
// both index loads and 2 x 8 row loads are in flight together instead of one segment after the other.  Each segment is
// still summed in ascending position order into its own accumulators.
template <int G>
__device__ __forceinline__ void segment_sum_pair(const int* __restrict__ perm_a, int lo_a, int hi_a, const int* __restrict__ perm_b,
                                                 int lo_b, int hi_b, const float* __restrict__ data, int stride,
                                                 const float* __restrict__ data3_a, const float* __restrict__ data3_b, int f,
                                                 bool f_active, float& acc_a, float& acc_b, float& acc3_a, float& acc3_b) {
    const int na = hi_a - lo_a, nb = hi_b - lo_b;
    const int n = max(na, nb);
    for (int base = 0; base < n; base += G) {
        const int ca = max(0, min(G, na - base)), cb = max(0, min(G, nb - base));
        const int my_a = (f < ca) ? perm_a[lo_a + base + f] : 0;
        const int my_b = (f < cb) ? perm_b[lo_b + base + f] : 0;
        const int cm = max(ca, cb);
        for (int a0 = 0; a0 < cm; a0 += 8) {
            float va[8], vb[8], wa[8], wb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ea = __shfl(my_a, (a0 + u) & (G - 1), G), eb = __shfl(my_b, (a0 + u) & (G - 1), G);
                const bool ia = (a0 + u) < ca, ib = (a0 + u) < cb;
                va[u] = (ia && f_active) ? data[(size_t)ea * stride + f] : 0.0f;
                vb[u] = (ib && f_active) ? data[(size_t)eb * stride + f] : 0.0f;
                wa[u] = (ia && f < 3) ? data3_a[(size_t)ea * 4 + f] : 0.0f;
                wb[u] = (ib && f < 3) ? data3_b[(size_t)eb * 4 + f] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc_a += va[u]; acc_b += vb[u]; acc3_a += wa[u]; acc3_b += wb[u]; }
        }
    }
}

// hidden_nf == 32 form of the paired gather (gather_bwd): a 128-byte row is read by 8 lanes as float4, so the 32 lanes of
// a node fetch FOUR rows per load instruction and one round of 8 loads per lane covers 32 rows of each segment -- the
// whole gather is two memory round trips (indices, rows) for all but hub nodes, instead of one per 8 rows.  The 16-byte
// coordinate rows are read one per lane.  Fixed summation order: ascending per lane, then a 4-way (rows) / 32-way (coords)
// butterfly.  Returns in lane f: feature f of both row sums, and (f < 3) component f of both coordinate sums.
__device__ __forceinline__ void gather_pair32(const int* __restrict__ perm_a, int lo_a, int hi_a, const int* __restrict__ perm_b,
                                              int lo_b, int hi_b, const float* __restrict__ data, const float* __restrict__ data3_a,
                                              const float* __restrict__ data3_b, int f, float& sum_a, float& sum_b, float& s3_a,
                                              float& s3_b) {
    constexpr int G = 32;
    const int q = f >> 3, v = f & 7;
    const int na = hi_a - lo_a, nb = hi_b - lo_b, n = max(na, nb);
    float4 A = make_float4(0.f, 0.f, 0.f, 0.f), B = A, PA = A, PB = A;
    for (int base = 0; base < n; base += G) {
        const int ca = max(0, min(G, na - base)), cb = max(0, min(G, nb - base));
        const int my_a = (f < ca) ? perm_a[lo_a + base + f] : 0;
        const int my_b = (f < cb) ? perm_b[lo_b + base + f] : 0;
        float4 ra[8], rb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int a = 4 * u + q;
            const int ea = __shfl(my_a, a, G), eb = __shfl(my_b, a, G);
            ra[u] = (a < ca) ? *reinterpret_cast<const float4*>(data + (size_t)ea * 32 + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[u] = (a < cb) ? *reinterpret_cast<const float4*>(data + (size_t)eb * 32 + 4 * v) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 pa = (f < ca) ? *reinterpret_cast<const float4*>(data3_a + (size_t)my_a * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 pb = (f < cb) ? *reinterpret_cast<const float4*>(data3_b + (size_t)my_b * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            A.x += ra[u].x; A.y += ra[u].y; A.z += ra[u].z; A.w += ra[u].w;
            B.x += rb[u].x; B.y += rb[u].y; B.z += rb[u].z; B.w += rb[u].w;
        }
        PA.x += pa.x; PA.y += pa.y; PA.z += pa.z;
        PB.x += pb.x; PB.y += pb.y; PB.z += pb.z;
    }
#pragma unroll
    for (int off = 8; off <= 16; off <<= 1) {           // the 4 row sub-groups
        A.x += __shfl_xor(A.x, off, G); A.y += __shfl_xor(A.y, off, G); A.z += __shfl_xor(A.z, off, G); A.w += __shfl_xor(A.w, off, G);
        B.x += __shfl_xor(B.x, off, G); B.y += __shfl_xor(B.y, off, G); B.z += __shfl_xor(B.z, off, G); B.w += __shfl_xor(B.w, off, G);
    }
#pragma unroll
    for (int off = 1; off <= 16; off <<= 1) {           // coordinate partials: one edge per lane
        PA.x += __shfl_xor(PA.x, off, G); PA.y += __shfl_xor(PA.y, off, G); PA.z += __shfl_xor(PA.z, off, G);
        PB.x += __shfl_xor(PB.x, off, G); PB.y += __shfl_xor(PB.y, off, G); PB.z += __shfl_xor(PB.z, off, G);
    }
    // lane f wants feature f = 4 * (f >> 2) + (f & 3): held by lane (f >> 2) in component (f & 3)
    const int src = f >> 2, c = f & 3;
    const float ax = __shfl(A.x, src, G), ay = __shfl(A.y, src, G), az = __shfl(A.z, src, G), aw = __shfl(A.w, src, G);
    const float bx = __shfl(B.x, src, G), by = __shfl(B.y, src, G), bz = __shfl(B.z, src, G), bw = __shfl(B.w, src, G);
    sum_a = c == 0 ? ax : (c == 1 ? ay : (c == 2 ? az : aw));
    sum_b = c == 0 ? bx : (c == 1 ? by : (c == 2 ? bz : bw));
    s3_a = f == 0 ? PA.x : (f == 1 ? PA.y : PA.z);
    s3_b = f == 0 ? PB.x : (f == 1 ? PB.y : PB.z);
}

// ---- LDS-resident transposed weights for the node kernels --------------------------------
// A lane owns one OUTPUT feature f and needs row f of W: read straight from global that is a
// 64-way uncoalesced access.  Each workgroup therefore stages W^T once (WT[i][f], row stride
// ROWS+1 -> conflict-free for the staging writes and for the per-lane reads) and every node
// iteration reads it from LDS.
// Weight staging global -> LDS.  All of a thread's loads are issued before its first LDS store (a load/store pair per
// loop iteration exposes one L2 round trip per element: 11 k of node_bwd's 34 k ticks in the trace build).
template <int MAXN>
struct Staged {                                         // split form: load() of several matrices first, then their store()s
    static constexpr int PER = (MAXN + 255) / 256;
    float v[PER];
    template <typename SrcIdx>
    __device__ __forceinline__ void load(const float* __restrict__ src, int n, SrcIdx si) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int t = threadIdx.x + 256 * k;
            v[k] = (t < n) ? src[si(t)] : 0.0f;
        }
    }
    template <typename DstIdx>
    __device__ __forceinline__ void store(float* dst, int n, DstIdx di) const {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int t = threadIdx.x + 256 * k;
            if (t < n) dst[di(t)] = v[k];
        }
    }
};
template <int MAXN, typename SrcIdx, typename DstIdx>
__device__ __forceinline__ void stage_batched(float* dst, const float* __restrict__ src, int n, SrcIdx si, DstIdx di) {
    Staged<MAXN> r;
    r.load(src, n, si);
    r.store(dst, n, di);
}
// index maps of the transposed forms (shared by the one-call and the split staging)
template <int ROWS> struct MapT {
    int cols, ld;
    __device__ __forceinline__ int src(int t) const { const int o = t / cols; return o * ld + (t - o * cols); }
    __device__ __forceinline__ int dst(int t) const { const int o = t / cols; return (t - o * cols) * (ROWS + 1) + o; }
};
template <int F, int H, int A> struct MapW1T {
    int c0;
    __device__ __forceinline__ int src(int t) const { const int o = t / F; return LP<F, H, A>::w1_off(o, c0 + (t - o * F)); }
    __device__ __forceinline__ int dst(int t) const { const int o = t / F; return (t - o * F) * (F + 1) + o; }
};
// src [ROWS][ld] (first `cols` columns, cols <= 64) -> dst transposed [cols][ROWS + 1]
template <int ROWS>
__device__ __forceinline__ void stage_T(float* dst, const float* __restrict__ src, int cols, int ld) {
    stage_batched<ROWS * 64>(dst, src, ROWS * cols,
                             [=](int t) { const int o = t / cols; return o * ld + (t - o * cols); },
                             [=](int t) { const int o = t / cols; return (t - o * cols) * (ROWS + 1) + o; });
}
// W1[:, c0:c0+F] of a layer (head-interleaved rows) transposed
template <int F, int H, int A>
__device__ __forceinline__ void stage_w1_T(float* dst, const float* __restrict__ lw, int c0) {
    using LL = LP<F, H, A>;
    stage_batched<F * F>(dst, lw, F * F,
                         [=](int t) { const int o = t / F; return LL::w1_off(o, c0 + (t - o * F)); },
                         [=](int t) { const int o = t / F; return (t - o * F) * (F + 1) + o; });
}
template <int ROWS>
__device__ __forceinline__ float col_dot(const float* WT, int f, const float* v, int n) {
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) acc = fmaf(WT[i * (ROWS + 1) + f], v[i], acc);
    return acc;
}

// per-node projections for a layer: Pr = W1[:, 0:F] h ; Pc = W1[:, F:2F] h + b1  (weights in LDS)
template <int F, int H, int A>
__device__ __forceinline__ void project_pre(const float* __restrict__ lw, const float* WaT, const float* WbT,
                                            const float* hv, int f, float* __restrict__ prow) {
    using LL = LP<F, H, A>;
    constexpr int FP = Geom<F>::FP;
    if (f < F) {
        prow[f] = col_dot<F>(WaT, f, hv, F);
        prow[FP + f] = lw[LL::b1_off(f)] + col_dot<F>(WbT, f, hv, F);
    }
}

}  // namespace eq
