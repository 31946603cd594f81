// Single-block exclusive scan of per-node counts (N <= a few 100k: one 1024-thread block is launch-latency sized).
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

}  // namespace eq
