// Exclusive scan of per-node counts: one 1024-thread block for small N (launch-latency sized), a two-launch
// two-level scan above kScanSingleBlockMax (the single block takes 135 us at N = 131 072).
#pragma once
#include "common.h"

namespace eq {

// cnt_cursor[i] (count) -> ptr[i] = cnt_cursor[i] = exclusive prefix; ptr[n] = total
template <int kUnused = 0>
__global__ __launch_bounds__(1024) void csr_scan_kernel(int* __restrict__ cnt_cursor, int n, int* __restrict__ ptr) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int lo = t * chunk;
    const int hi = min(lo + chunk, n);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += cnt_cursor[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;   // exclusive prefix of this thread's chunk
    for (int i = lo; i < hi; ++i) {
        const int c = cnt_cursor[i];
        ptr[i] = run;
        cnt_cursor[i] = run;
        run += c;
    }
    if (t == 1023) ptr[n] = part[1023];
}

constexpr int kScanSingleBlockMax = 8192;
constexpr int kScanChunk = 4096;                 // elements per block of the two-level scan (4 per thread)

// level 1: block b scans its chunk in place (exclusive, local) and writes the chunk total
template <int kUnused = 0>
__global__ __launch_bounds__(1024) void scan_chunks_kernel(int* __restrict__ cnt_cursor, int n, int* __restrict__ totals) {
    __shared__ int part[1024];
    const int t = threadIdx.x, base = blockIdx.x * kScanChunk + 4 * t;
    int v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (base + k < n) ? cnt_cursor[base + k] : 0; s += v[k]; }
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int a = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += a;
        __syncthreads();
    }
    int run = part[t] - s;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) cnt_cursor[base + k] = run;
        run += v[k];
    }
    if (t == 1023) totals[blockIdx.x] = part[1023];
}
// level 2: add the chunk offsets; ptr[i] = cursor[i] = global exclusive prefix.  Every block sums the totals of the chunks before
// it itself (<= 1024 values, one per thread: a block sum) instead of waiting for a third launch that scans them; the last block
// also publishes the grand total ptr[n].
template <int kUnused = 0>
__global__ __launch_bounds__(1024) void scan_apply_kernel(int* __restrict__ cnt_cursor, int n, const int* __restrict__ totals,
                                                          int* __restrict__ ptr) {
    __shared__ int wsum[16];
    const int t = threadIdx.x, b = blockIdx.x;
    int v = (t < b) ? totals[t] : 0;                    // gridDim.x <= 1024 chunks
    const int mine_total = (b == (int)gridDim.x - 1 && t == b) ? totals[b] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((t & 63) == 0) wsum[t >> 6] = v;
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) off += wsum[w];
    const int base = b * kScanChunk + 4 * t;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < n) { const int x = cnt_cursor[base + k] + off; cnt_cursor[base + k] = x; ptr[base + k] = x; }
    if (b == (int)gridDim.x - 1 && t == b) ptr[n] = off + mine_total;
}

// cnt_cursor[0..n) counts -> exclusive prefixes in cnt_cursor and ptr, ptr[n] = total.  `totals`: >= ceil(n / 4096) ints of
// scratch, needed only when n > kScanSingleBlockMax (n <= 4 M).
static inline void launch_exclusive_scan(int* cnt_cursor, int n, int* ptr, int* totals, hipStream_t s) {
    if (n <= kScanSingleBlockMax || totals == nullptr) {
        hipLaunchKernelGGL(csr_scan_kernel<0>, dim3(1), dim3(1024), 0, s, cnt_cursor, n, ptr);
        return;
    }
    const int chunks = (n + kScanChunk - 1) / kScanChunk;
    hipLaunchKernelGGL(scan_chunks_kernel<0>, dim3(chunks), dim3(1024), 0, s, cnt_cursor, n, totals);
    hipLaunchKernelGGL(scan_apply_kernel<0>, dim3(chunks), dim3(1024), 0, s, cnt_cursor, n, totals, ptr);
}

}  // namespace eq
