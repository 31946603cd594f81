// Fused training pose head for gfx950: everything CrossAttentionPoseRegression.forward does AFTER the
// EGNN (src/3dmatch_train_egnn_with_batch.py:677-796) and its backward, as a handful of kernels instead
// of ~220 small ATen launches:
//   head_node_kernel   per node: sim = <h_s,h_t> (:681), raw_sim (:682), the two egnn_equi_loss terms
//                      (:860-893); per-block fp64 partial sums for the batch-wide mean/std (:777-778); zeroes the tickets
//   head_pair_kernel   grid (1 + ceil(N/256), B).  Every block evaluates the batch statistics and its pair's masked-softmax
//                      scalars (:717-724) from the K1 partials itself, so the two chains below need nothing from each other:
//                        column 0: top-k (:684) by MSB radix select (early exit) + rank by counting in LDS;
//                        columns 1..: Kabsch weights (:718-724) + similarity-MSE partials (:777-781) of 256 nodes each; the
//                        block that finishes a pair's weights last (ticket) computes the Kabsch moments, solves the 3x3
//                        problem (:726-758) and evaluates the pair's pose-loss terms (:896-962)
//   head_mlp_kernel    the MLP on the top-k rows (:760-770) + BCE (:772), 8 rows per block; the block that finishes last over
//                      all pairs (ticket) writes the loss scalars (corr_loss + sim_loss, equi loss) and the training loop's
//                      step loss (:1117-1120)
//   backward: head_pair_bwd_kernel (BCE / MLP backward + MLP weight-gradient slabs; beside them, as extra grid columns, the
//             Kabsch backward with the step loss's pose gradient folded in -> ticket -> gwu for the softmax backward),
//             head_node_bwd_kernel (d h_s, d h_t, d x_s, d x_t; extra grid columns sum the MLP slabs into the grads, +=)
// Launch structure: independent jobs share a launch as extra grid columns; a many -> one dependency inside a launch is a ticket
// (release fence + device-scope atomic; the last block continues behind an acquire fence; the counter is handed back as zero).
// Layouts: h_* [B,N,F] with row stride F; x_* [B,N,3]; labels [B,N] (float 0/1); gt_pose [B,16].
#include <mutex>

#include "common.h"
#include "kabsch_device.h"
#include "pose_loss_device.h"

namespace eq {

typedef unsigned long long u64;
constexpr int kHeadPartials = 8;     // sum s, sum s^2, sum r, sum r^2, rot, feat, (unused x2)
constexpr int kRowsPerBlock = 8;
constexpr int kMaxIn = 128, kMaxH1 = 64, kMaxH2 = 32;     // MLP in = 2F, h1 = F, h2 = F/2 for F <= 64
#ifndef EQ_HEAD_FUSE_MLP
#define EQ_HEAD_FUSE_MLP 1             // the MLP blocks of small batches ride in head_pair_kernel's launch (A/B knob)
#endif
constexpr int kMaxN = 8192;          // head_pair_kernel keeps a pair's keys in LDS: 12 bytes per node

// bounded waits that ran out (the MLP blocks riding in head_pair_kernel's launch wait for their pair's selection): cannot happen
// while the launch is co-resident (the host only fuses then), kept as a safety net; read back by equigspr_head_wait_failures()
__device__ unsigned g_head_wait_failures = 0u;

__device__ __forceinline__ unsigned orderable(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ double block_sum_d(double v, double* red /*[4]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
// NV (<= 8) block sums at once: per wave a reduce-scatter butterfly (W = 8 or 4 slots: at distance o = 32, 16[, 8] a lane keeps the
// half of its slots selected by lane & o and adds the partner's copy, then the remaining distances on its single slot), ONE pair of
// barriers, and every thread adds the four wave totals in wave order.  Replaces NV sequential block_sum_d (2 NV barriers, 12 NV
// shuffles); the result is the same for every thread, the summation order differs from block_sum_d's.
template <int NV>
__device__ __forceinline__ void block_sum_dn(double (&v)[NV], double (*ws)[8] /* __shared__ [4][8] */) {
    constexpr int W = NV > 4 ? 8 : 4;
    static_assert(NV >= 1 && NV <= 8, "block_sum_dn: 1..8 values");
    const int lane = threadIdx.x & 63;
    double t[W];
#pragma unroll
    for (int q = 0; q < W; ++q) t[q] = q < NV ? v[q] : 0.0;
    int n = W, slot = 0, o = 32;
#pragma unroll
    for (int step = 0; step < (W == 8 ? 3 : 2); ++step, o >>= 1) {
        n >>= 1;
        const bool up = (lane & o) != 0;
#pragma unroll
        for (int i2 = 0; i2 < W / 2; ++i2) {
            if (i2 < n) {
                const double keep = up ? t[i2 + n] : t[i2], send = up ? t[i2] : t[i2 + n];
                t[i2] = keep + __shfl_xor(send, o, 64);
            }
        }
        slot += up ? n : 0;
    }
#pragma unroll
    for (int oo = (W == 8 ? 4 : 8); oo > 0; oo >>= 1) t[0] += __shfl_xor(t[0], oo, 64);
    __syncthreads();
    if ((lane & (W == 8 ? 7 : 15)) == 0 && slot < NV) ws[threadIdx.x >> 6][slot] = t[0];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = ws[0][q] + ws[1][q] + ws[2][q] + ws[3][q];
}
__device__ __forceinline__ float block_max_f(float v, float* red /*[4]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---------------------------------------------------------------------------------------------------
// K1: 8 nodes x 32 lanes per block (lane = feature, and feature+32), grid (ceil(N/8), B)
// per-block fp64 partials: sum s, sum s^2, sum r, sum r^2, rot, feat, max_valid(s), sum_valid exp(s - max)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float grp32_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
    return v;
}

__global__ __launch_bounds__(256) void head_node_kernel(const float* __restrict__ hs, const float* __restrict__ ht,
                                                        const float* __restrict__ rhs, const float* __restrict__ rht,
                                                        const float* __restrict__ xs, const float* __restrict__ xt,
                                                        const float* __restrict__ labels, const float* __restrict__ gt_pose,
                                                        int B, int N, int F, int Fr, float* __restrict__ sim,
                                                        float* __restrict__ rsim, float* __restrict__ cosv,
                                                        int* __restrict__ pos, double* __restrict__ partials,
                                                        unsigned* __restrict__ tickets /*[3 B + 1]*/) {
    __shared__ float nv[8][6];     // s, r, rot, feat, valid, (pad)
    if (blockIdx.x == 0 && threadIdx.x == 0) {           // the tickets of the launches behind this one (K2 per pair, K3 one for all)
        tickets[blockIdx.y] = 0u;
        tickets[B + 1 + blockIdx.y] = 0u; tickets[2 * B + 1 + blockIdx.y] = 0u;
        if (blockIdx.y == 0) tickets[B] = 0u;
    }
    const int slot = threadIdx.x >> 5, f = threadIdx.x & 31;
    const int b = blockIdx.y, n = blockIdx.x * 8 + slot;
    const bool live = n < N;
    const long long i = (long long)b * N + (live ? n : 0);
    float s = 0.0f, na = 0.0f, nc = 0.0f, r = 0.0f;
    if (live) {
        for (int k = f; k < F; k += 32) { const float a = hs[i * F + k], c = ht[i * F + k]; s = fmaf(a, c, s); na = fmaf(a, a, na); nc = fmaf(c, c, nc); }
        for (int k = f; k < Fr; k += 32) r = fmaf(rhs[i * Fr + k], rht[i * Fr + k], r);
    }
    s = grp32_sum(s); na = grp32_sum(na); nc = grp32_sum(nc); r = grp32_sum(r);
    if (f == 0) {
        float rot = 0.0f, feat = 0.0f, y = 0.0f;
        if (live) {
            y = labels[i];
            const float cs = s / (fmaxf(sqrtf(na), 1e-8f) * fmaxf(sqrtf(nc), 1e-8f));       // F.cosine_similarity (:887)
            const float* G = gt_pose + (size_t)b * 16;
            float ch = 0.0f;
#pragma unroll
            for (int r3 = 0; r3 < 3; ++r3) {
                const float d = G[r3 * 4 + 0] * xs[i * 3 + 0] + G[r3 * 4 + 1] * xs[i * 3 + 1] + G[r3 * 4 + 2] * xs[i * 3 + 2] +
                                G[r3 * 4 + 3] - xt[i * 3 + r3];
                ch = fmaf(d, d, ch);
            }
            sim[i] = s; rsim[i] = r; cosv[i] = cs; pos[i] = -1;
            rot = ch * y; feat = (cs - y) * (cs - y);
        }
        nv[slot][0] = live ? s : 0.0f; nv[slot][1] = live ? r : 0.0f; nv[slot][2] = rot; nv[slot][3] = feat;
        nv[slot][4] = (live && y != 0.0f) ? 1.0f : 0.0f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double p[6] = {0, 0, 0, 0, 0, 0};
        float mx = -INFINITY;
        for (int q = 0; q < 8; ++q) {
            const double sv = nv[q][0], rv = nv[q][1];
            p[0] += sv; p[1] += sv * sv; p[2] += rv; p[3] += rv * rv; p[4] += nv[q][2]; p[5] += nv[q][3];
            if (nv[q][4] != 0.0f) mx = fmaxf(mx, nv[q][0]);
        }
        double se = 0;
        for (int q = 0; q < 8; ++q)
            if (nv[q][4] != 0.0f) se += expf(nv[q][0] - mx);
        double* out = partials + ((size_t)b * gridDim.x + blockIdx.x) * kHeadPartials;
#pragma unroll
        for (int q = 0; q < 6; ++q) out[q] = p[q];
        out[6] = mx; out[7] = se;
    }
}

// stats: mean_s, std_s (unbiased), mean_r, std_r, rot sum, feat sum  over ALL pairs
__device__ __forceinline__ void head_stats(const double* __restrict__ partials, int n_rec, long long M, double* red,
                                           double (&st)[6]) {
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int k = threadIdx.x; k < n_rec; k += 256)
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[q] += partials[(size_t)k * kHeadPartials + q];
    __shared__ double ws6[4][8];
    block_sum_dn<6>(acc, ws6);
    const double ms = acc[0] / M, mr = acc[2] / M;
    const double vs = fmax((acc[1] - M * ms * ms) / (M - 1), 0.0), vr = fmax((acc[3] - M * mr * mr) / (M - 1), 0.0);
    st[0] = ms; st[1] = sqrt(vs); st[2] = mr; st[3] = sqrt(vr); st[4] = acc[4]; st[5] = acc[5];
}

struct MlpDims { int in, h1, h2; };

// inp[j][c] = cat(h_s[idx_j], h_t[idx_j])[c]: 8 independent global loads in flight per thread (a single
// workgroup per pair has nothing else to hide the latency with)
__device__ __forceinline__ void gather_rows(const float* __restrict__ hs, const float* __restrict__ ht, const int* idxs,
                                            int b, int N, int F, int K, int in_dim, int ld_in, float* inp) {
    const int total = K * in_dim;
    for (int t0 = threadIdx.x; t0 < total; t0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = t0 + u * 256;
            v[u] = 0.0f;
            if (t < total) {
                const int j = t / in_dim, c = t - j * in_dim;
                const size_t rowoff = ((size_t)b * N + idxs[j]) * F;
                v[u] = (c < F) ? hs[rowoff + c] : ht[rowoff + (c - F)];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = t0 + u * 256;
            if (t < total) { const int j = t / in_dim, c = t - j * in_dim; inp[j * ld_in + c] = v[u]; }
        }
    }
}
     // in = 2F, h1 = F, h2 = F/2, out = 1
constexpr int kCand = 512;              // candidate list of the radix select
constexpr int kCandEarly = 256;         // ... and the size at which the select stops refining digits

// K2 (extra blocks of the head_pair_kernel launch): one thread per node: Kabsch weights (:718-724) and the similarity-MSE
// partial sums (:777-781).  Block (xb, b) covers nodes [256 xb, 256 xb + 256) of pair b; its partials go to record b * nCw + xb.
// The softmax scalars (mx, den, Su) and the batch statistics (mu_s, sd_s, mu_r, sd_r) come from the block's own evaluation of
// the K1 partials (the same float values block 0 of the pair publishes for the backward).
struct WeightsJob {
    const float *sim, *rsim, *labels;
    float* wts;
    double* partC;
};
__device__ __forceinline__ void head_weights_block(const WeightsJob& j, int xb, int b, int N, float mx, float den, float Su,
                                                   float mu_s, float sd_s, float mu_r, float sd_r) {
    const int il = xb * 256 + (int)threadIdx.x;
    double q0 = 0, q1 = 0, q2 = 0;
    if (il < N) {
        const size_t i = (size_t)b * N + il;
        const float s = j.sim[i];
        const float u = (j.labels[i] != 0.0f && den > 0.0f) ? expf(s - mx) / den : 0.0f;
        j.wts[i] = u / (Su + 1e-6f);
        const float zs = (s - mu_s) / (sd_s + 1e-6f), zr = (j.rsim[i] - mu_r) / (sd_r + 1e-6f);
        const float e = zs - zr;
        q0 = (double)e * e; q1 = e; q2 = (double)e * (s - mu_s);
    }
    {
        __shared__ double ws3[4][8];
        double qv[3] = {q0, q1, q2};
        block_sum_dn<3>(qv, ws3);
        q0 = qv[0]; q1 = qv[1]; q2 = qv[2];
    }
    if (threadIdx.x == 0) {
        double* rec = j.partC + ((size_t)b * ((N + 255) / 256) + xb) * 3;
        rec[0] = q0; rec[1] = q1; rec[2] = q2;
    }
}

// top-K of sb[0..N) (:684): rank r of top_idx = the r-th largest value, ties by lower index; pos[idx] = rank for the selected
// nodes (the others keep the -1 written by head_node_kernel).  One 256-thread workgroup.  MSB radix select on the value bits
// finds the K-th value T; candidates {v > T} + {v == T} are ranked exactly by counting; if more than kCand candidates tie
// (degenerate inputs) fall back to a full bitonic sort.  keys: u64 [max(Npad, kCand)], khi: unsigned [N], histp: [16 * 256],
// sel: int [3], wtot: int [4] -- all in LDS.  Ends without a barrier after the last global stores.
constexpr int kHistCopies = 16;
__device__ __forceinline__ void head_select_topk(const float* __restrict__ sb, int N, int Npad, int K, u64* keys, unsigned* khi,
                                                 unsigned* histp, int* sel, int* wtot, long long* __restrict__ top_idx_b,
                                                 int* __restrict__ pos_b) {
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += 256) khi[i] = ~orderable(sb[i]);
    unsigned prefix = 0, mask = 0;
    int need = K;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        // 16 privatised copies of the histogram (by lane & 15): the leading digits of similarity values fall into a
        // handful of bins, and same-address LDS atomics serialise (the four passes were 18 k of this kernel's 34 k ticks)
#pragma unroll
        for (int r = 0; r < kHistCopies; ++r) histp[r * 256 + tid] = 0;
        __syncthreads();
        for (int i = tid; i < N; i += 256) {
            const unsigned v = khi[i];
            if ((v & mask) == prefix) atomicAdd(&histp[(tid & (kHistCopies - 1)) * 256 + ((v >> shift) & 255u)], 1u);
        }
        __syncthreads();
        {   // inclusive scan of the 256 bins (one per thread): wave scan + 4 wave totals; the selected bin is the
            // first whose inclusive count reaches `need`
            int c = 0;
#pragma unroll
            for (int r = 0; r < kHistCopies; ++r) c += (int)histp[r * 256 + tid];
            int inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(inc, o, 64);
                if ((tid & 63) >= o) inc += v;
            }
            if ((tid & 63) == 63) wtot[tid >> 6] = inc;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < (tid >> 6); ++w) base += wtot[w];
            inc += base;
            if (inc >= need && inc - c < need) { sel[0] = tid; sel[1] = inc - c; sel[2] = c; }
        }
        __syncthreads();
        need -= sel[1];
        prefix |= (unsigned)sel[0] << shift;
        mask |= 255u << shift;
        // the keys below the selected bin (K - need of them) plus the bin itself are few enough to rank exactly: stop refining
        // (typically after two of the four digits; block-uniform)
        if ((K - need) + sel[2] <= kCandEarly) break;
    }
    EQ_STAMPK(3, 3);
    const int n_less = K - need, n_tie = sel[2];
    const int total = n_less + n_tie;
    if (total <= kCand) {
        __syncthreads();                    // every thread has read sel[] before it is reused as the candidate counter
        if (tid == 0) sel[0] = 0;
        __syncthreads();
        for (int i = tid; i < N; i += 256) {
            const unsigned v = khi[i];
            if ((v & mask) <= prefix) keys[atomicAdd(&sel[0], 1)] = ((u64)v << 32) | (unsigned)i;
        }
        __syncthreads();
        EQ_STAMPK(3, 4);
        // <= kCand unique keys: rank by counting (one pass of broadcast LDS reads, one barrier) instead of a bitonic
        // sort (36 barrier-separated stages at 256 keys); same order
        for (int t = tid; t < total; t += 256) {
            const u64 mine = keys[t];
            int rank = 0;
            for (int j = 0; j < total; ++j) rank += (keys[j] < mine) ? 1 : 0;
            if (rank < K) {
                const int idx = (int)(unsigned)(mine & 0xffffffffu);
                top_idx_b[rank] = idx;
                pos_b[idx] = rank;
            }
        }
        EQ_STAMPK(3, 5);
        return;
    }
    for (int i = tid; i < N; i += 256) keys[i] = ((u64)khi[i] << 32) | (unsigned)i;
    for (int i = N + tid; i < Npad; i += 256) keys[i] = ~0ull;
    const int P = Npad;
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P / 2; t += 256) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const bool up = (lo & k2) == 0;
                const u64 a = keys[lo], c = keys[hi];
                if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int t = tid; t < K; t += 256) {
        const int idx = (int)(unsigned)(keys[t] & 0xffffffffu);
        top_idx_b[t] = idx;
        pos_b[idx] = t;
    }
}

// (defined below, next to the stand-alone MLP launch)
__device__ __forceinline__ void head_mlp_block(
    float* lds, int b, int jb, const float* __restrict__ hs, const float* __restrict__ ht, const float* __restrict__ labels,
    const long long* __restrict__ top_idx, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
    int B, int N, int F, int K, MlpDims md, float* __restrict__ logits, float* __restrict__ a1_save,
    float* __restrict__ a2_save, double* __restrict__ pair_part, int nblk, const double* __restrict__ partC, int nC,
    float* __restrict__ out_scalars, int flags, unsigned* __restrict__ ticket, unsigned n_arrivals, float* __restrict__ step_loss,
    unsigned* __restrict__ ready, unsigned* __restrict__ seen);
__device__ __forceinline__ void head_finalize_all(const double* __restrict__ partC, int nC, const double* __restrict__ pair_part,
                                                  int nblk, int B, int N, int K, float* __restrict__ out_scalars, int flags,
                                                  float* __restrict__ step_loss);

// K2: one workgroup per pair: softmax scalars, top-K (radix select + small bitonic sort), MLP, BCE
__global__ __launch_bounds__(256) void head_pair_kernel(
    const float* __restrict__ hs, const float* __restrict__ ht, const float* __restrict__ labels,
    const float* __restrict__ sim, const double* __restrict__ partials, int nbp,
    const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
    const float* __restrict__ w3, const float* __restrict__ b3, int B, int N, int Npad, int F, int K, MlpDims md,
    int* __restrict__ pos, long long* __restrict__ top_idx, float* __restrict__ logits,
    float* __restrict__ a1_save, float* __restrict__ a2_save, float* __restrict__ pair_f /*[B][4]*/,
    double* __restrict__ pair_part /*[B][2]*/, float* __restrict__ fin, unsigned* __restrict__ tickets, WeightsJob wj, int nwb,
    const float* __restrict__ xs, const float* __restrict__ xt, float* __restrict__ kab_R, float* __restrict__ kab_t,
    float* __restrict__ kab_aux, const float* __restrict__ gt_pose, float* __restrict__ step_loss /* [1 + 2B] or NULL */,
    int nmlp /* > 0: the MLP blocks ride in this launch as grid columns 1 + nwb .. */, int nC, int hflags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[4];
    __shared__ float redf[4];
    __shared__ unsigned histp[kHistCopies * 256];
    __shared__ int sel[3];
    __shared__ int wtot[4];
    // ticket words: [0, B) per pair (Kabsch weights), [B] the batch-wide arrival count, [B + 1, 2B + 1) "selection published",
    // [2B + 1, 3B + 1) readers of the selection -- all zeroed by K1 and handed back as zero
    if (nmlp > 0 && (int)blockIdx.x > nwb) {
        // ---- MLP columns: stage the weights while the pair's select block (dispatched before this one) works, then its rows.
        // Arrivals over the batch: B * nmlp MLP blocks + the B Kabsch tails; the last one writes the loss scalars.
        const int bb = blockIdx.y;
        head_mlp_block(reinterpret_cast<float*>(smem_raw), bb, (int)blockIdx.x - 1 - nwb, hs, ht, labels, top_idx, w1, b1, w2, b2, w3, b3, B, N, F,
                       K, md, logits, a1_save, a2_save, pair_part, nmlp, wj.partC, nC, fin, hflags, tickets + B, (unsigned)B * (unsigned)(nmlp + 1),
                       step_loss, tickets + B + 1 + bb, tickets + 2 * B + 1 + bb);
        return;
    }
    // grid (1 + nwb, B): column 0 = the pair's select block, columns 1.. = its Kabsch-weight blocks.  EVERY block evaluates the
    // batch statistics and its pair's softmax scalars from the K1 partials itself (a few microseconds, L2-resident), so the weight
    // blocks need nothing from the select block and the two chains -- select -> MLP and weights -> Kabsch -- run side by side.
    const int b = blockIdx.y, tid = threadIdx.x;
    const long long M = (long long)B * N;
    const float* sb = sim + (size_t)b * N;
    const float* yb = labels + (size_t)b * N;
    u64* keys = reinterpret_cast<u64*>(smem_raw);                                   // [Npad] (fallback) / candidates
    unsigned* khi = reinterpret_cast<unsigned*>(smem_raw + sizeof(u64) * (size_t)(Npad > kCand ? Npad : kCand));   // [N]
    double st[6];
    EQ_STAMPK(3, 0);
    head_stats(partials, B * nbp, M, red, st);
    EQ_STAMPK(3, 1);
    // ---- masked softmax scalars from the per-block (max, sum exp) records of this pair (:717-724)
    const double* pp = partials + (size_t)b * nbp * kHeadPartials;
    float mx = -INFINITY;
    for (int k = tid; k < nbp; k += 256) mx = fmaxf(mx, (float)pp[(size_t)k * kHeadPartials + 6]);
    mx = block_max_f(mx, redf);
    double den = 0;
    for (int k = tid; k < nbp; k += 256) {
        const float bm = (float)pp[(size_t)k * kHeadPartials + 6];
        if (bm > -INFINITY) den += pp[(size_t)k * kHeadPartials + 7] * (double)expf(bm - mx);
    }
    den = block_sum_d(den, red);
    if (blockIdx.x > 0) {
        // ---- Kabsch weights + similarity-MSE partials of this block's 256 nodes; the block that finishes the pair's weights last
        // (ticket, zeroed by K1 and handed back as zero) computes the pair's Kabsch moments over them and solves the 3x3 problem
        head_weights_block(wj, (int)blockIdx.x - 1, b, N, mx, (float)den, (den > 0.0) ? 1.0f : 0.0f, (float)st[0], (float)st[1],
                           (float)st[2], (float)st[3]);
        if (!kab_aux) return;
        __shared__ int is_last;
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            is_last = atomicAdd(tickets + b, 1u) == (unsigned)nwb - 1u;
            if (is_last) tickets[b] = 0u;               // the backward's ticket starts from zero
        }
        __syncthreads();
        if (!is_last) return;                            // block-uniform
        __threadfence();
        kabsch_moments_block(xs, xt, wj.wts, N, kab_aux, b, 0, 1);
        __syncthreads();
        if (tid < 64) kabsch_solve_wave(1, kab_R, kab_t, kab_aux, b, tid);
        if (step_loss && tid == 0)       // this pair's terms of the training loop's loss (train :1117-1120); K3 adds them up
            pose_loss_pair(kab_R + (size_t)b * 9, kab_t + (size_t)b * 3, gt_pose + (size_t)b * 16, step_loss[1 + b], step_loss[1 + B + b]);
        if (nmlp > 0) {                  // the MLP blocks ride in this launch: this tail is one of the batch's arrivals
            __syncthreads();
            if (tid == 0) {
                __threadfence();
                is_last = atomicAdd(tickets + B, 1u) == (unsigned)B * (unsigned)(nmlp + 1) - 1u;
                if (is_last) tickets[B] = 0u;
            }
            __syncthreads();
            if (!is_last) return;        // block-uniform
            __threadfence();
            head_finalize_all(wj.partC, nC, pair_part, nmlp, B, N, K, fin, hflags, step_loss);
        }
        return;
    }
    if (tid == 0) {
        float* pf = pair_f + (size_t)b * 4;
        pf[0] = mx; pf[1] = (float)den; pf[2] = (den > 0.0) ? 1.0f : 0.0f; pf[3] = 0.0f;
        if (b == 0) { fin[2] = (float)st[0]; fin[3] = (float)st[1]; fin[4] = (float)st[2]; fin[5] = (float)st[3]; fin[8] = (float)st[4]; fin[9] = (float)st[5]; }
    }
    EQ_STAMPK(3, 2);
    head_select_topk(sb, N, Npad, K, keys, khi, histp, sel, wtot, top_idx + (size_t)b * K, pos + (size_t)b * N);
    if (nmlp > 0) {                      // publish the selection to the pair's MLP blocks
        __syncthreads();
        if (tid == 0) { __threadfence(); atomicExch(tickets + B + 1 + b, 1u); }
    }
}

// ---------------------------------------------------------------------------------------------------
// Evaluation pair model (src/eval_egnn_metrics.py:691-818), the part after the EGNN, per pair one 256-thread workgroup
// (launch 2 of 3; launch 1 = head_node_kernel: sim0 = <h0_s, h0_t> on the INPUT features (:691) arrives as `rsim`, the equi-loss
// partials (:687) as `partials`, pos = -1):
//   top-K of sim0 (:694) -> p = mlp(cat(h_s[idx], h_t[idx])) (:706-742) -> the [K] vs [1,K,1] broadcast of :742,:758-768: the
//   `where` result is [1,K,K] and scatter_(dim=1, index [1,K,1]) consumes column 0 only, so EVERY top-K slot i receives
//   where(cond(p[0], sim0[idx_i]), p[0], sim0[idx_i]) -- only the MLP score of the best-ranked row is ever used, and only that
//   row is evaluated here -> the scalars of W / (sum W + 1e-6) (:771) -> softmax over the N points (:774) -> / (sum + 1e-6) (:783).
//   Launch 3 (head_eval_weights_kernel) writes the weights from these scalars and runs the weighted Kabsch.
// out[1] = equi loss (b == 0 writes it), out[10 + min(b, 5)] = p[0] of pair b (diagnostic).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_eval_pair_kernel(
    const float* __restrict__ hs, const float* __restrict__ ht, const float* __restrict__ sim0, const double* __restrict__ partials,
    int nbp, const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
    const float* __restrict__ w3, const float* __restrict__ b3, int B, int N, int Npad, int F, int K, MlpDims md,
    int* __restrict__ pos, long long* __restrict__ top_idx, float* __restrict__ pair_f /*[B][4]*/, float* __restrict__ pair_g /*[B]*/,
    float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ double red[4];
    __shared__ float redf[4];
    __shared__ unsigned histp[kHistCopies * 256];
    __shared__ int sel[3];
    __shared__ int wtot[4];
    __shared__ float inp[kMaxIn], a1[kMaxH1], a2[kMaxH2];
    __shared__ float p0_s;
    const int b = blockIdx.y, tid = threadIdx.x;
    const long long M = (long long)B * N;
    const float* sb = sim0 + (size_t)b * N;
    int* posb = pos + (size_t)b * N;
    u64* keys = reinterpret_cast<u64*>(smem_raw);
    unsigned* khi = reinterpret_cast<unsigned*>(smem_raw + sizeof(u64) * (size_t)(Npad > kCand ? Npad : kCand));
    {
        double st[6];
        head_stats(partials, B * nbp, M, red, st);
        if (b == 0 && tid == 0) { out[8] = (float)st[4]; out[9] = (float)st[5]; out[1] = (float)((st[4] + st[5]) / (double)M); }
    }
    head_select_topk(sb, N, Npad, K, keys, khi, histp, sel, wtot, top_idx + (size_t)b * K, posb);
    __threadfence_block();
    __syncthreads();
    // ---- p[0]: the MLP on the best-ranked row (Linear(2F,F)-ReLU-Linear(F,F/2)-ReLU-Linear(F/2,1), train :594-600)
    const int i0 = (int)top_idx[(size_t)b * K];
    for (int c = tid; c < md.in; c += 256) {
        const size_t rowoff = ((size_t)b * N + i0) * F;
        inp[c] = (c < F) ? hs[rowoff + c] : ht[rowoff + (c - F)];
    }
    __syncthreads();
    {   // four threads per output: partial dots over the input columns q, q + 4, ...; fixed-order sum of the four
        const int o = tid >> 2, q = tid & 3;
        float acc = 0.0f;
        if (o < md.h1)
            for (int c = q; c < md.in; c += 4) acc = fmaf(w1[(size_t)o * md.in + c], inp[c], acc);
        const float s1 = acc + __shfl_xor(acc, 1, 64);
        const float s2 = s1 + __shfl_xor(s1, 2, 64);
        if (o < md.h1 && q == 0) a1[o] = fmaxf(s2 + b1[o], 0.0f);
    }
    __syncthreads();
    {
        const int o = tid >> 2, q = tid & 3;
        float acc = 0.0f;
        if (o < md.h2)
            for (int c = q; c < md.h1; c += 4) acc = fmaf(w2[(size_t)o * md.h1 + c], a1[c], acc);
        const float s1 = acc + __shfl_xor(acc, 1, 64);
        const float s2 = s1 + __shfl_xor(s1, 2, 64);
        if (o < md.h2 && q == 0) a2[o] = fmaxf(s2 + b2[o], 0.0f);
    }
    __syncthreads();
    if (tid == 0) {
        float z = b3[0];
        for (int c = 0; c < md.h2; ++c) z = fmaf(w3[c], a2[c], z);
        p0_s = z;
        out[10 + (b < 5 ? b : 5)] = z;
    }
    __syncthreads();
    const float p0 = p0_s;
    // ---- W (:758-768), its sum, the softmax of W / (sum + 1e-6) and the final renormalisation.  One sweep over the global
    // similarities (W into LDS -- the select's key array is free now --, sum and maximum), the rest on the LDS copy.
    float* Wl = reinterpret_cast<float*>(smem_raw);
    __syncthreads();                                      // every thread is done with the keys / top_idx reads above
    double sw = 0.0;
    float wmax = -INFINITY, wmin = INFINITY;
    for (int i = tid; i < N; i += 256) {
        const float s = sb[i];
        float w = s;
        if (posb[i] >= 0) {
            const bool cond = (p0 > 0.5f && fabsf(p0 - 1.0f) < s) || (p0 > 0.5f && p0 < s);
            w = cond ? p0 : s;
        }
        Wl[i] = w;
        sw += (double)w;
        wmax = fmaxf(wmax, w);
        wmin = fminf(wmin, w);
    }
    sw = block_sum_d(sw, red);
    wmax = block_max_f(wmax, redf);
    wmin = -block_max_f(-wmin, redf);
    const float denom = (float)sw + 1e-6f;
    // the maximum of W / denom: division by a constant is monotonic (increasing for a positive, decreasing for a negative one)
    const float mx = (denom < 0.0f ? wmin : wmax) / denom;
    double se = 0.0;
    for (int i = tid; i < N; i += 256) {
        const float e = __builtin_amdgcn_exp2f((Wl[i] / denom - mx) * 1.4426950408889634f);      // exp(v - max) on the hardware exponential
        Wl[i] = e;
        se += (double)e;
    }
    se = block_sum_d(se, red);
    const float sef = (float)se;
    double sn = 0.0;
    for (int i = tid; i < N; i += 256) {
        const float w = Wl[i] / sef;
        Wl[i] = w;
        sn += (double)w;
    }
    sn = block_sum_d(sn, red);
    const float renorm = (float)sn + 1e-6f;
    // the pair's scalars for the weight blocks of the next launch (head_eval_weights_kernel)
    if (tid == 0) {
        float* pf = pair_f + (size_t)b * 4;
        pf[0] = denom; pf[1] = mx; pf[2] = sef; pf[3] = renorm;
        pair_g[b] = p0;
    }
}

// Launch 3 of the evaluation head, grid (nkb <= 8, B): block xb writes the Kabsch weights of its interleaved share of pair b's points
// (i = 256 xb + tid + 256 nkb j) from the pair's scalars, accumulates their fp64 moments (kabsch_moments_block: same partition),
// and the block that finishes the pair last (ticket) solves the 3 x 3 problem on the RAW coordinates of all N points (:786-818).
__global__ __launch_bounds__(256) void head_eval_weights_kernel(const float* __restrict__ sim0, const int* __restrict__ pos,
                                                                const float* __restrict__ pair_f, const float* __restrict__ pair_g,
                                                                int N, int nkb, float* __restrict__ wts,
                                                                const float* __restrict__ raw_xs, const float* __restrict__ raw_xt,
                                                                float* __restrict__ kab_R, float* __restrict__ kab_t,
                                                                float* __restrict__ kab_aux, unsigned* __restrict__ tickets) {
    const int b = blockIdx.y, xb = blockIdx.x, tid = threadIdx.x;
    const float denom = pair_f[(size_t)b * 4], mx = pair_f[(size_t)b * 4 + 1], sef = pair_f[(size_t)b * 4 + 2], renorm = pair_f[(size_t)b * 4 + 3];
    const float p0 = pair_g[b];
    const float* sb = sim0 + (size_t)b * N;
    const int* posb = pos + (size_t)b * N;
    float* wb = wts + (size_t)b * N;
    for (int i0 = xb * 256 + tid; i0 < N; i0 += nkb * 256 * 4) {         // four independent points per step: their loads overlap
        float sv[4];
        int pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * nkb * 256; sv[u] = i < N ? sb[i] : 0.0f; pv[u] = i < N ? posb[i] : -1; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * nkb * 256;
            if (i >= N) continue;
            const float sq = sv[u];
            float w = sq;
            if (pv[u] >= 0) {
                const bool cond = (p0 > 0.5f && fabsf(p0 - 1.0f) < sq) || (p0 > 0.5f && p0 < sq);
                w = cond ? p0 : sq;
            }
            const float e = __builtin_amdgcn_exp2f((w / denom - mx) * 1.4426950408889634f);       // as in head_eval_pair_kernel
            wb[i] = (e / sef) / renorm;
        }
    }
    kabsch_moments_block(raw_xs, raw_xt, wts, N, kab_aux, b, xb, nkb);     // reads back this thread's own stores (same index set)
    __shared__ int is_last;
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        is_last = atomicAdd(tickets + b, 1u) == (unsigned)nkb - 1u;
        if (is_last) tickets[b] = 0u;
    }
    __syncthreads();
    if (!is_last) return;                                  // block-uniform
    __threadfence();
    if (tid < 64) kabsch_solve_wave(nkb, kab_R, kab_t, kab_aux, b, tid);
}

// out[0] = corr_loss + sim_loss, out[1] = equi loss, out[2..5] = mu_s, sd_s, mu_r, sd_r (written by K2),
// out[6] = sum e, out[7] = sum e (s - mu_s), out[8], out[9] = rot / feat sums (written by K2)
__device__ __forceinline__ void head_finalize_block(const double* __restrict__ partC, int nC,
                                                    const double* __restrict__ pair_part, int nblk, int B, int N, int K,
                                                    float* __restrict__ out, int flags) {
    __shared__ double red[4];
    const long long M = (long long)B * N;
    double a0 = 0, a1 = 0, a2 = 0, bce = 0;
    for (int k = threadIdx.x; k < nC; k += 256) { a0 += partC[(size_t)k * 3]; a1 += partC[(size_t)k * 3 + 1]; a2 += partC[(size_t)k * 3 + 2]; }
    for (int b = threadIdx.x; b < B * nblk; b += 256) bce += pair_part[b];
    {
        __shared__ double ws4[4][8];
        double av[4] = {a0, a1, a2, bce};
        block_sum_dn<4>(av, ws4);
        a0 = av[0]; a1 = av[1]; a2 = av[2]; bce = av[3];
    }
    if (threadIdx.x == 0) {
        // EQUIGSPR_HEAD_NO_SIM_LOSS: the KITTI script's forward returns corr_loss alone (src/kitti_train_egnn_with_batch.py:766)
        out[0] = (float)(bce / ((double)B * K) + ((flags & EQUIGSPR_HEAD_NO_SIM_LOSS) ? 0.0 : a0 / (double)M));
        out[1] = (float)(((double)out[8] + (double)out[9]) / (double)M);
        out[6] = (float)a1; out[7] = (float)a2;
    }
}

// K2b: the correspondence MLP on the K selected rows (:760-773), kRowsPerBlock rows per workgroup so the
// work spreads over the chip instead of one CU per pair.  grid (ceil(K/kRowsPerBlock), B)

// global -> LDS staging with 8 loads of a thread in flight before its 8 stores (a load/store pair per iteration exposes
// one L2 round trip per element; the MLP blocks are latency-bound single workgroups)
template <typename SrcIdx, typename DstIdx>
__device__ __forceinline__ void stage8(float* dst, const float* __restrict__ src, int n, SrcIdx si, DstIdx di) {
    for (int t0 = threadIdx.x; t0 < n; t0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int t = t0 + 256 * k; v[k] = (t < n) ? src[si(t)] : 0.0f; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int t = t0 + 256 * k; if (t < n) dst[di(t)] = v[k]; }
    }
}

// LDS of one MLP block (floats): inp | a1 | a2 | W1^T | W2^T | W3 | biases | idxs | red (fp64 x 4)
constexpr int kMlpLdsFloats = kRowsPerBlock * (kMaxIn + 1) + kRowsPerBlock * (kMaxH1 + 1) + kRowsPerBlock * (kMaxH2 + 1) +
                              kMaxIn * (kMaxH1 + 1) + kMaxH1 * (kMaxH2 + 1) + kMaxH2 + (kMaxH1 + kMaxH2 + 1) + kRowsPerBlock + 1 + 8;
constexpr size_t kMlpLdsBytes = sizeof(float) * (size_t)kMlpLdsFloats + 16;

// the loss scalars from the partial sums of the whole batch (+ the training loop's step loss): run by the block that arrives last
__device__ __forceinline__ void head_finalize_all(const double* __restrict__ partC, int nC, const double* __restrict__ pair_part,
                                                  int nblk, int B, int N, int K, float* __restrict__ out_scalars, int flags,
                                                  float* __restrict__ step_loss) {
    head_finalize_block(partC, nC, pair_part, nblk, B, N, K, out_scalars, flags);
    if (step_loss) {                                     // loss = (corr_loss + sim_loss) + mean rot + mean trans (train :1120)
        __syncthreads();
        step_loss_sum(step_loss + 1, step_loss + 1 + B, out_scalars[0], B, step_loss);
    }
}

// One MLP block: rows j0 .. j0 + 7 of pair b's selection.  `ready` != NULL (the block rides in head_pair_kernel's launch): the
// weights are staged first, then the block waits for its pair's select block to publish top_idx; the pair's last reader hands
// `ready` / `seen` back as zero.  `ticket` counts `n_arrivals` blocks over the whole batch; the last one writes the loss scalars.
__device__ __forceinline__ void head_mlp_block(
    float* lds, int b, int jb, const float* __restrict__ hs, const float* __restrict__ ht, const float* __restrict__ labels,
    const long long* __restrict__ top_idx, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
    int B, int N, int F, int K, MlpDims md, float* __restrict__ logits, float* __restrict__ a1_save,
    float* __restrict__ a2_save, double* __restrict__ pair_part, int nblk, const double* __restrict__ partC, int nC,
    float* __restrict__ out_scalars, int flags, unsigned* __restrict__ ticket, unsigned n_arrivals, float* __restrict__ step_loss,
    unsigned* __restrict__ ready, unsigned* __restrict__ seen) {
    float* inp = lds;
    float* a1 = inp + kRowsPerBlock * (kMaxIn + 1);
    float* a2 = a1 + kRowsPerBlock * (kMaxH1 + 1);
    float* W1 = a2 + kRowsPerBlock * (kMaxH2 + 1);      // transposed [in][h1+1]
    float* W2 = W1 + kMaxIn * (kMaxH1 + 1);             // transposed [h1][h2+1]
    float* W3 = W2 + kMaxH1 * (kMaxH2 + 1);
    float* Bs = W3 + kMaxH2;
    int* idxs = reinterpret_cast<int*>(Bs + (kMaxH1 + kMaxH2 + 1));
    double* red = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(idxs + kRowsPerBlock) + 7) & ~(uintptr_t)7);
    const int tid = threadIdx.x;
    const int j0 = jb * kRowsPerBlock;
    const int Kb = min(kRowsPerBlock, K - j0);
    const int ld_in = md.in + 1, ld1 = md.h1 + 1, ld2 = md.h2 + 1;
    const float* yb = labels + (size_t)b * N;
    stage8(W1, w1, md.h1 * md.in, [](int t) { return t; }, [&](int t) { const int o = t / md.in; return (t - o * md.in) * ld1 + o; });
    stage8(W2, w2, md.h2 * md.h1, [](int t) { return t; }, [&](int t) { const int o = t / md.h1; return (t - o * md.h1) * ld2 + o; });
    for (int t = tid; t < md.h2; t += 256) W3[t] = w3[t];
    for (int t = tid; t < md.h1; t += 256) Bs[t] = b1[t];
    for (int t = tid; t < md.h2; t += 256) Bs[md.h1 + t] = b2[t];
    if (tid == 0) Bs[md.h1 + md.h2] = b3[0];
    __shared__ int wait_failed;
    if (tid == 0) wait_failed = 0;
    if (ready) {                                         // the select block of this pair is in the same launch, dispatched before this one
        if (tid == 0) {
            int it = 0;
            while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && it < (1 << 22)) { __builtin_amdgcn_s_sleep(2); ++it; }
            wait_failed = it >= (1 << 22);   // bounded (seconds): never a hung GPU; a wait that ran out poisons this block's logits and the loss with NaN
            if (wait_failed) atomicAdd(&g_head_wait_failures, 1u);      // ... and is counted, so that it can be told from a genuinely non-finite loss
            __threadfence();
        }
        __syncthreads();
    }
    if (tid < Kb) {
        const int ix = (int)top_idx[(size_t)b * K + j0 + tid];
        idxs[tid] = wait_failed ? 0 : min(max(ix, 0), N - 1);
    }
    __syncthreads();
    if (ready && tid == 0) {                             // the pair's last reader resets the flag for the next forward
        if (atomicAdd(seen, 1u) == (unsigned)nblk - 1u) { *seen = 0u; atomicExch(ready, 0u); }
    }
    gather_rows(hs, ht, idxs, b, N, F, Kb, md.in, ld_in, inp);
    __syncthreads();
    for (int t = tid; t < Kb * md.h1; t += 256) {
        const int j = t / md.h1, o = t - j * md.h1;
        float acc = Bs[o];
#pragma unroll 8
        for (int i = 0; i < md.in; ++i) acc = fmaf(W1[i * ld1 + o], inp[j * ld_in + i], acc);
        acc = fmaxf(acc, 0.0f);
        a1[j * ld1 + o] = acc;
        a1_save[((size_t)b * K + j0 + j) * md.h1 + o] = acc;
    }
    __syncthreads();
    for (int t = tid; t < Kb * md.h2; t += 256) {
        const int j = t / md.h2, o = t - j * md.h2;
        float acc = Bs[md.h1 + o];
#pragma unroll 8
        for (int i = 0; i < md.h1; ++i) acc = fmaf(W2[i * ld2 + o], a1[j * ld1 + i], acc);
        acc = fmaxf(acc, 0.0f);
        a2[j * ld2 + o] = acc;
        a2_save[((size_t)b * K + j0 + j) * md.h2 + o] = acc;
    }
    __syncthreads();
    double bce = 0;
    if (tid < Kb) {
        float z = Bs[md.h1 + md.h2];
        for (int i = 0; i < md.h2; ++i) z = fmaf(W3[i], a2[tid * ld2 + i], z);
        if (wait_failed) z = __uint_as_float(0x7fc00000u);
        logits[(size_t)b * K + j0 + tid] = z;
        const float y = yb[idxs[tid]];
        bce = fmaxf(z, 0.0f) - z * y + log1pf(expf(-fabsf(z)));                          // BCEWithLogits (:772-773)
    }
    bce = block_sum_d(bce, red);
    // ---- the block that arrives last over ALL pairs (ticket) turns the partial sums into the loss scalars
    __shared__ int is_last_m;
    if (tid == 0) {
        pair_part[(size_t)b * nblk + jb] = bce;
        __threadfence();
        is_last_m = atomicAdd(ticket, 1u) == n_arrivals - 1u;
        if (is_last_m) *ticket = 0u;
    }
    __syncthreads();
    if (!is_last_m) return;                              // block-uniform
    __threadfence();
    head_finalize_all(partC, nC, pair_part, nblk, B, N, K, out_scalars, flags, step_loss);
}

// K3 (its own launch when the MLP blocks do not ride in K2): grid (ceil(K/kRowsPerBlock), B)
__global__ __launch_bounds__(256) void head_mlp_kernel(
    const float* __restrict__ hs, const float* __restrict__ ht, const float* __restrict__ labels,
    const long long* __restrict__ top_idx, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
    int B, int N, int F, int K, MlpDims md, float* __restrict__ logits, float* __restrict__ a1_save,
    float* __restrict__ a2_save, double* __restrict__ pair_part, int nblk, const double* __restrict__ partC, int nC,
    float* __restrict__ out_scalars, int flags, unsigned* __restrict__ ticket /* one word: zeroed by K1, handed back as zero */,
    float* __restrict__ step_loss) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];
    head_mlp_block(reinterpret_cast<float*>(mlp_smem), blockIdx.y, blockIdx.x, hs, ht, labels, top_idx, w1, b1, w2, b2, w3, b3, B, N, F, K, md,
                   logits, a1_save, a2_save, pair_part, nblk, partC, nC, out_scalars, flags, ticket, gridDim.x * gridDim.y, step_loss,
                   nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
// KB1: grid (ceil(K/kRowsPerBlock), B): BCE + MLP backward on this block's rows, its MLP weight-gradient slab;
// one extra block per pair reduces gwu = sum_i gw_i u_i for the softmax backward.
// mlp slab layout: dW1 [h1*in] | db1 [h1] | dW2 [h2*h1] | db2 [h2] | dW3 [h2] | db3 [1]
// The Kabsch backward of the head's own Kabsch (grad_R, grad_t -> coordinate and weight gradients) rides in this launch as extra
// grid columns: it only needs the upstream pose gradient, like the MLP blocks, so the two run side by side instead of one after
// the other.  The block that finishes a pair's weight gradients last (ticket) also sums gwu for the softmax backward.
struct KabschBwdJob {
    const float *xs, *xt, *wts, *aux, *gR, *gt;
    const float *g_step, *gt_pose, *kab_t;   // g_step != NULL: + d(step loss)/d(R, t) of the loss the forward evaluated (device scalar)
    float *gxs, *gxt, *gw;
    unsigned* tickets;
    int nkb;                              // 0: gw comes from the caller (equigspr_kabsch_backward ran before)
};
__global__ __launch_bounds__(256) void head_pair_bwd_kernel(
    const float* __restrict__ hs, const float* __restrict__ ht, const float* __restrict__ labels,
    const float* __restrict__ sim, const float* __restrict__ pair_f, const float* __restrict__ gw,
    const long long* __restrict__ top_idx, const float* __restrict__ logits, const float* __restrict__ a1_save,
    const float* __restrict__ a2_save, const float* __restrict__ w1, const float* __restrict__ w2,
    const float* __restrict__ w3, const float* __restrict__ g_cs_ptr, int B, int N, int F, int K, MlpDims md,
    float* __restrict__ pair_g /*[B]*/, float* __restrict__ dinp /*[B][K][in]*/, float* __restrict__ mlp_slab, int nblk,
    KabschBwdJob kj) {
    constexpr int R = kRowsPerBlock;
    __shared__ float inp[R * (kMaxIn + 1)];
    __shared__ float a1[R * (kMaxH1 + 1)];
    __shared__ float a2[R * (kMaxH2 + 1)];
    __shared__ float d1[R * (kMaxH1 + 1)];
    __shared__ float d2[R * (kMaxH2 + 1)];
    __shared__ float dz[R];
    __shared__ float W1[kMaxH1 * kMaxIn];     // row-major [h1][in]
    __shared__ float W2[kMaxH2 * kMaxH1];     // row-major [h2][h1]
    __shared__ float W3[kMaxH2];
    __shared__ int idxs[R];
    __shared__ double red[4];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int j0 = blockIdx.x * R;
    const int Kb = min(R, K - j0);
    const float g_cs = (g_cs_ptr ? g_cs_ptr[0] : 0.0f) + (kj.g_step ? kj.g_step[0] : 0.0f);     // the step loss contains corr_loss + sim_loss once
    const float* sb = sim + (size_t)b * N;
    const float* yb = labels + (size_t)b * N;
    const float* gwb = gw + (size_t)b * N;
    const int ld_in = md.in + 1, ld1 = md.h1 + 1, ld2 = md.h2 + 1;
    if ((int)blockIdx.x >= nblk) {       // extra blocks per pair (not on the MLP blocks' critical path)
        bool do_gwu = (int)blockIdx.x == nblk && kj.nkb == 0;         // gw given by the caller: one block sums gwu = sum_i gw_i u_i
        if ((int)blockIdx.x > nblk) {                                   // the Kabsch backward shares of this pair
            __shared__ float gRt[12];                                   // this pair's upstream grad_R | grad_t
            if (tid == 0) {
#pragma unroll
                for (int q = 0; q < 9; ++q) gRt[q] = kj.gR ? kj.gR[(size_t)b * 9 + q] : 0.0f;
#pragma unroll
                for (int q = 0; q < 3; ++q) gRt[9 + q] = kj.gt ? kj.gt[(size_t)b * 3 + q] : 0.0f;
                if (kj.g_step) {                                        // the step loss's own pose terms: mean over the batch
                    float sR[9], sT[3];
                    const float up = kj.g_step[0] / (float)B;
                    pose_loss_grad_pair(kj.aux + (size_t)b * EQUIGSPR_KABSCH_AUX + 18, kj.kab_t + (size_t)b * 3, kj.gt_pose + (size_t)b * 16, up, up, sR, sT);
#pragma unroll
                    for (int q = 0; q < 9; ++q) gRt[q] += sR[q];
#pragma unroll
                    for (int q = 0; q < 3; ++q) gRt[9 + q] += sT[q];
                }
            }
            kabsch_bwd_block(kj.xs, kj.xt, kj.wts, kj.aux, gRt, gRt + 9, N, kj.gxs, kj.gxt, kj.gw, b, (int)blockIdx.x - nblk - 1, kj.nkb);
            __shared__ int is_last;
            __syncthreads();
            if (tid == 0) {
                __threadfence();
                is_last = atomicAdd(kj.tickets + b, 1u) == (unsigned)kj.nkb - 1u;
                if (is_last) kj.tickets[b] = 0u;                        // a second backward over the same forward starts from zero again
            }
            __syncthreads();
            do_gwu = is_last != 0;                                      // block-uniform
            if (do_gwu) __threadfence();
        }
        if (!do_gwu) return;
        const float mx = pair_f[b * 4], den = pair_f[b * 4 + 1];
        double gwu = 0;
        for (int i = tid; i < N; i += 256)
            if (yb[i] != 0.0f && den > 0.0f) gwu += (double)gwb[i] * (double)(expf(sb[i] - mx) / den);
        gwu = block_sum_d(gwu, red);
        if (tid == 0) pair_g[b] = (float)gwu;
        return;
    }
    stage8(W1, w1, md.h1 * md.in, [](int t) { return t; }, [](int t) { return t; });
    stage8(W2, w2, md.h2 * md.h1, [](int t) { return t; }, [](int t) { return t; });
    for (int t = tid; t < md.h2; t += 256) W3[t] = w3[t];
    if (tid < Kb) idxs[tid] = (int)top_idx[(size_t)b * K + j0 + tid];
    __syncthreads();
    gather_rows(hs, ht, idxs, b, N, F, Kb, md.in, ld_in, inp);
    for (int t = tid; t < Kb * md.h1; t += 256) { const int j = t / md.h1, o = t - j * md.h1; a1[j * ld1 + o] = a1_save[((size_t)b * K + j0) * md.h1 + t]; }
    for (int t = tid; t < Kb * md.h2; t += 256) { const int j = t / md.h2, o = t - j * md.h2; a2[j * ld2 + o] = a2_save[((size_t)b * K + j0) * md.h2 + t]; }
    if (tid < Kb) {
        const float z = logits[(size_t)b * K + j0 + tid];
        const float y = yb[idxs[tid]];
        dz[tid] = g_cs * (1.0f / (1.0f + expf(-z)) - y) / ((float)B * K);
    }
    __syncthreads();
    for (int t = tid; t < Kb * md.h2; t += 256) {
        const int j = t / md.h2, o = t - j * md.h2;
        d2[j * ld2 + o] = (a2[j * ld2 + o] > 0.0f) ? dz[j] * W3[o] : 0.0f;
    }
    __syncthreads();
    for (int t = tid; t < Kb * md.h1; t += 256) {
        const int j = t / md.h1, o = t - j * md.h1;
        float acc = 0.0f;
#pragma unroll 8
        for (int q = 0; q < md.h2; ++q) acc = fmaf(d2[j * ld2 + q], W2[q * md.h1 + o], acc);
        d1[j * ld1 + o] = (a1[j * ld1 + o] > 0.0f) ? acc : 0.0f;
    }
    __syncthreads();
    for (int t = tid; t < Kb * md.in; t += 256) {
        const int j = t / md.in, c = t - j * md.in;
        float acc = 0.0f;
#pragma unroll 8
        for (int q = 0; q < md.h1; ++q) acc = fmaf(d1[j * ld1 + q], W1[q * md.in + c], acc);
        dinp[((size_t)b * K + j0 + j) * md.in + c] = acc;
    }
    // ---- weight-gradient slab of this block (thread-owned outputs, sums over its rows)
    const int msz = md.h1 * md.in + md.h1 + md.h2 * md.h1 + md.h2 + md.h2 + 1;
    float* sl = mlp_slab + ((size_t)b * nblk + blockIdx.x) * msz;
    for (int t = tid; t < md.h1 * md.in; t += 256) {
        const int o = t / md.in, c = t - o * md.in;
        float acc = 0.0f;
#pragma unroll 8
        for (int j = 0; j < Kb; ++j) acc = fmaf(d1[j * ld1 + o], inp[j * ld_in + c], acc);
        sl[t] = acc;
    }
    float* p_db1 = sl + md.h1 * md.in;
    float* p_w2 = p_db1 + md.h1;
    for (int t = tid; t < md.h2 * md.h1; t += 256) {
        const int o = t / md.h1, c = t - o * md.h1;
        float acc = 0.0f;
#pragma unroll 8
        for (int j = 0; j < Kb; ++j) acc = fmaf(d2[j * ld2 + o], a1[j * ld1 + c], acc);
        p_w2[t] = acc;
    }
    float* p_db2 = p_w2 + md.h2 * md.h1;
    float* p_w3 = p_db2 + md.h2;
    float* p_db3 = p_w3 + md.h2;
    for (int t = tid; t < md.h1 + 2 * md.h2 + 1; t += 256) {      // the four small column sums (<= 16 rows each)
        float acc = 0.0f;
        if (t < md.h1) { for (int j = 0; j < Kb; ++j) acc += d1[j * ld1 + t]; p_db1[t] = acc; }
        else if (t < md.h1 + md.h2) { const int o = t - md.h1; for (int j = 0; j < Kb; ++j) acc += d2[j * ld2 + o]; p_db2[o] = acc; }
        else if (t < md.h1 + 2 * md.h2) { const int o = t - md.h1 - md.h2; for (int j = 0; j < Kb; ++j) acc = fmaf(dz[j], a2[j * ld2 + o], acc); p_w3[o] = acc; }
        else { for (int j = 0; j < Kb; ++j) acc += dz[j]; p_db3[0] = acc; }
    }
}

// KB2: 8 nodes x 32 lanes per block: d sim (softmax + similarity-MSE), then d h_s, d h_t, d x_s, d x_t
__global__ __launch_bounds__(256) void head_node_bwd_kernel(
    const float* __restrict__ hs, const float* __restrict__ ht, const float* __restrict__ xs, const float* __restrict__ xt,
    const float* __restrict__ labels, const float* __restrict__ gt_pose, const float* __restrict__ sim,
    const float* __restrict__ rsim, const float* __restrict__ cosv, const int* __restrict__ pos,
    const float* __restrict__ dinp, const float* __restrict__ gw, const float* __restrict__ pair_f,
    const float* __restrict__ pair_g, const float* __restrict__ fin, const float* __restrict__ g_cs_ptr,
    const float* __restrict__ g_equi_ptr, int B, int N, int F, int K, int in_dim, float* __restrict__ ghs,
    float* __restrict__ ght, float* __restrict__ gxs /* += */, float* __restrict__ gxt /* += */, int nbp,
    const float* __restrict__ mlp_slab, int n_slabs, int msz, float* __restrict__ grad_mlp /* += */, int flags,
    const float* __restrict__ g_step_ptr) {
    if ((int)blockIdx.x >= nbp) {        // block-uniform: the MLP weight-gradient slabs of KB1, summed in fixed order, ride in this launch
        // 32 consecutive entries x 8 slab partitions per block (partition p sums slabs p, p + 8, ...; the 8 partials are then
        // added in order): K = 2048 leaves thousands of slabs, a one-thread-per-entry loop over them is a serial latency chain
        __shared__ float red[8][33];
        const int lane = threadIdx.x & 31, part = threadIdx.x >> 5;
        const int t = ((int)blockIdx.x - nbp) * 32 + lane;
        if (blockIdx.y != 0) return;
        float a = 0.0f;
        if (t < msz) {
#pragma unroll 4
            for (int q = part; q < n_slabs; q += 8) a += mlp_slab[(size_t)q * msz + t];
        }
        red[part][lane] = a;
        __syncthreads();
        if (part == 0 && t < msz) {
            float v = red[0][lane];
#pragma unroll
            for (int q = 1; q < 8; ++q) v += red[q][lane];
            grad_mlp[t] += v;
        }
        return;
    }
    const int slot = threadIdx.x >> 5, f = threadIdx.x & 31;
    const int b = blockIdx.y, n = blockIdx.x * 8 + slot;
    if (n >= N) return;
    const long long i = (long long)b * N + n;
    const long long M = (long long)B * N;
    const float g_eq = g_equi_ptr ? g_equi_ptr[0] : 0.0f;
    const float g_cs = (g_cs_ptr ? g_cs_ptr[0] : 0.0f) + (g_step_ptr ? g_step_ptr[0] : 0.0f);
    const float y = labels[i], s = sim[i];
    // softmax-normalised weights backward
    const float mx = pair_f[b * 4], den = pair_f[b * 4 + 1], Su = pair_f[b * 4 + 2], gwu = pair_g[b];
    const float sc = Su + 1e-6f;
    const float u = (y != 0.0f && den > 0.0f) ? expf(s - mx) / den : 0.0f;
    const float gu = gw[i] / sc - gwu / (sc * sc);
    const float guu = gwu / sc - gwu * Su / (sc * sc);
    float gs = u * (gu - guu);
    // similarity-MSE backward (:777-781): zs = (s - mu)/(sd + eps), sd unbiased over all B*N
    {
        const float mu_s = fin[2], sd_s = fin[3], mu_r = fin[4], sd_r = fin[5], sum_e = fin[6], sum_es = fin[7];
        const float inv_sd = 1.0f / (sd_s + 1e-6f);
        const float zs = (s - mu_s) * inv_sd, zr = (rsim[i] - mu_r) / (sd_r + 1e-6f);
        const float e = zs - zr;
        const float k2 = (sd_s > 0.0f) ? sum_es * inv_sd * inv_sd / ((float)(M - 1) * sd_s) : 0.0f;
        if (!(flags & EQUIGSPR_HEAD_NO_SIM_LOSS)) gs += (2.0f / (float)M * g_cs) * ((e - sum_e / (float)M) * inv_sd - k2 * (s - mu_s));
    }
    const int pj = pos[i];
    const float* a = hs + i * F;
    const float* c = ht + i * F;
    float ka = 0.0f, kc = 0.0f, kx = 0.0f;     // feature-loss (cosine) terms
    if (g_eq != 0.0f) {
        float na = 0.0f, nc = 0.0f;
        for (int k = f; k < F; k += 32) { na = fmaf(a[k], a[k], na); nc = fmaf(c[k], c[k], nc); }
        na = grp32_sum(na); nc = grp32_sum(nc);
        const float la = fmaxf(sqrtf(na), 1e-8f), lc = fmaxf(sqrtf(nc), 1e-8f);
        const float cs = cosv[i];
        const float gcos = g_eq * 2.0f * (cs - y) / (float)M;
        kx = gcos / (la * lc);                 // d cos/da = c/(|a||c|) - cos a/|a|^2
        ka = gcos * cs / (la * la);
        kc = gcos * cs / (lc * lc);
    }
    const float* dj = (pj >= 0) ? dinp + ((size_t)b * K + pj) * in_dim : nullptr;
    for (int k = f; k < F; k += 32) {
        float ga = gs * c[k] + kx * c[k] - ka * a[k];
        float gc = gs * a[k] + kx * a[k] - kc * c[k];
        if (dj) { ga += dj[k]; gc += dj[F + k]; }
        ghs[i * F + k] = ga;
        ght[i * F + k] = gc;
    }
    if (g_eq != 0.0f && f < 3) {               // rotation-consistency term (:881-884)
        const float* G = gt_pose + (size_t)b * 16;
        float d[3];
#pragma unroll
        for (int r3 = 0; r3 < 3; ++r3)
            d[r3] = G[r3 * 4 + 0] * xs[i * 3 + 0] + G[r3 * 4 + 1] * xs[i * 3 + 1] + G[r3 * 4 + 2] * xs[i * 3 + 2] + G[r3 * 4 + 3] - xt[i * 3 + r3];
        const float k = g_eq * 2.0f * y / (float)M;
        gxs[i * 3 + f] += k * (G[0 * 4 + f] * d[0] + G[1 * 4 + f] * d[1] + G[2 * 4 + f] * d[2]);
        gxt[i * 3 + f] -= k * d[f];
    }
}

static inline int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }
static inline size_t fwd_smem(int N, int Npad, int K, MlpDims md) {
    const size_t keys = sizeof(u64) * (size_t)(Npad > kCand ? Npad : kCand);
    return keys + sizeof(unsigned) * (size_t)N + 64;
}
struct HeadWork { double *partA, *partC, *pair_part; float *pair_f, *pair_g; unsigned* tickets; int nbp, nC; size_t bytes; };
static inline HeadWork head_work(void* base, int B, int N) {   // pair_part sized for K = N: ceil(N / kRowsPerBlock) row blocks
    HeadWork w;
    w.nbp = ceil_div(N, 8);
    w.nC = B * ceil_div(N, 256);
    char* p = reinterpret_cast<char*>(base);
    size_t o = 0;
    w.partA = reinterpret_cast<double*>(p + o); o += sizeof(double) * (size_t)B * w.nbp * kHeadPartials;
    w.partC = reinterpret_cast<double*>(p + o); o += sizeof(double) * (size_t)w.nC * 3;
    w.pair_part = reinterpret_cast<double*>(p + o); o += sizeof(double) * (size_t)B * ceil_div(N, kRowsPerBlock);
    w.pair_f = reinterpret_cast<float*>(p + o); o += sizeof(float) * (size_t)B * 4;
    w.pair_g = reinterpret_cast<float*>(p + o); o += sizeof(float) * (size_t)B;
    w.tickets = reinterpret_cast<unsigned*>(p + o); o += sizeof(unsigned) * (size_t)(3 * B + 1);  // per-pair tickets (K2 / the backward) + the batch-wide one + selection flags / reader counts; zeroed by K1, handed back as zero
    w.bytes = o + 64;
    return w;
}

}  // namespace eq

extern "C" int64_t equigspr_head_mlp_size(int32_t F) {
    const int in = 2 * F, h1 = F, h2 = F / 2;
    return (int64_t)h1 * in + h1 + (int64_t)h2 * h1 + h2 + h2 + 1;
}

extern "C" int64_t equigspr_head_slab_count(int32_t B, int32_t K) { return (int64_t)B * eq::ceil_div(K, eq::kRowsPerBlock); }

extern "C" int64_t equigspr_head_work_bytes(int32_t B, int32_t N) {
    return (int64_t)eq::head_work(nullptr, B, N).bytes;
}

namespace eq {
// workgroups of head_pair_kernel (256 threads, `smem` bytes of dynamic LDS) the current device holds at once; cached per device and
// LDS size (the occupancy query costs ~10 us).  0 on error: the caller then does not fuse.
static long long resident_pair_blocks(size_t smem) {
    static std::mutex mu;
    struct Entry { int dev; size_t smem; long long blocks; };
    static Entry cache[16];
    static int n_cache = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < n_cache; ++i)
        if (cache[i].dev == dev && cache[i].smem == smem) return cache[i].blocks;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&head_pair_kernel), 256, smem) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    const long long blocks = (long long)per_cu * cus;
    if (n_cache < 16) cache[n_cache++] = Entry{dev, smem, blocks};
    return blocks;
}
}  // namespace eq

extern "C" int equigspr_head_fuses_mlp(int32_t B, int32_t N, int32_t F, int32_t K) {
    using namespace eq;
    if (B < 1 || N < 1 || F < 1 || K < 1 || N > kMaxN) return 0;
    const MlpDims md{2 * F, F, F / 2};
    const size_t sm_sel = fwd_smem(N, next_pow2(N), K, md);
    const size_t sm_fused = kMlpLdsBytes > sm_sel ? kMlpLdsBytes : sm_sel;
    const long long fused_grid = (long long)B * (1 + ceil_div(N, 256) + ceil_div(K, kRowsPerBlock));
    return (EQ_HEAD_FUSE_MLP && fused_grid <= 512 && fused_grid <= resident_pair_blocks(sm_fused)) ? 1 : 0;
}

extern "C" int64_t equigspr_head_wait_failures(void) {
    unsigned v = 0u;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(eq::g_head_wait_failures), sizeof(v)) != hipSuccess) return -1;
    return (int64_t)v;
}

extern "C" int equigspr_head_forward(const float* hs, const float* ht, const float* raw_hs, const float* raw_ht,
                                     const float* xs, const float* xt, const float* labels, const float* gt_pose,
                                     const float* mlp /* flat: w1,b1,w2,b2,w3,b3 */, int32_t B, int32_t N, int32_t F,
                                     int32_t F_raw, int32_t K, int32_t flags, float* sim, float* rsim, float* cosv, int32_t* pos,
                                     float* wts, int64_t* top_idx, float* logits, float* a1_save, float* a2_save,
                                     void* work, float* out_scalars /* [16] device */, float* kabsch_R, float* kabsch_t,
                                     float* kabsch_aux, float* step_loss /* [1 + 2B] or NULL */, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(hs && ht && raw_hs && raw_ht && xs && xt && labels && gt_pose && mlp && sim && rsim && cosv && pos && wts &&
                     top_idx && logits && a1_save && a2_save && work && out_scalars, EQUIGSPR_EINVAL, "head_forward: null pointer");
    EQ_CHECK_ARG(B >= 1 && N >= 2 && F >= 2 && F <= 64 && F_raw >= 1 && F_raw <= 64 && K >= 1 && K <= N && N <= kMaxN,
                 EQUIGSPR_ENOTSUP, "head_forward: need 2<=F<=64, 1<=K<=N, N<=8192");
    hipStream_t s = (hipStream_t)stream;
    const MlpDims md{2 * F, F, F / 2};
    const HeadWork W = head_work(work, B, N);
    const float* w1 = mlp;
    const float* b1 = w1 + md.h1 * md.in;
    const float* w2 = b1 + md.h1;
    const float* b2 = w2 + md.h2 * md.h1;
    const float* w3 = b2 + md.h2;
    const float* b3 = w3 + md.h2;
    hipLaunchKernelGGL(head_node_kernel, dim3(W.nbp, B), dim3(256), 0, s, hs, ht, raw_hs, raw_ht, xs, xt, labels, gt_pose, B, N,
                       F, F_raw, sim, rsim, cosv, pos, W.partA, W.tickets);
    const int Npad = next_pow2(N);
    EQ_CHECK_ARG(!kabsch_aux || (kabsch_R && kabsch_t), EQUIGSPR_EINVAL, "head_forward: kabsch_R / kabsch_t are null");
    EQ_CHECK_ARG(!step_loss || kabsch_aux, EQUIGSPR_EINVAL, "head_forward: step_loss needs the head's own Kabsch (kabsch_aux)");
    const size_t sm_sel = fwd_smem(N, Npad, K, md);
    const size_t sm_fused = kMlpLdsBytes > sm_sel ? kMlpLdsBytes : sm_sel;
    // The MLP blocks may ride in the pair kernel's launch only if the WHOLE fused grid is resident at once for the LDS this launch
    // really asks for (N = 8 192: ~115 KB, one workgroup per CU): then every workgroup runs whatever the dispatch order is
    // (MI355X_MICROARCH.md, contract [G]: HIP promises nothing about that order), and the consumers' wait for their pair's select
    // block cannot starve it.  Otherwise the MLP is its own launch.
    const long long fused_grid = (long long)B * (1 + ceil_div(N, 256) + ceil_div(K, kRowsPerBlock));
    const bool fuse_mlp = EQ_HEAD_FUSE_MLP && kabsch_aux && fused_grid <= 512 && fused_grid <= resident_pair_blocks(sm_fused);
    const size_t sm = fuse_mlp ? sm_fused : sm_sel;
    {   // the dynamic-LDS limit is a per-device function attribute: keep the high-water mark per device, under a lock
        static std::mutex mu;
        static size_t sm_set[64] = {0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || sm > sm_set[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&head_pair_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
            if (e != hipSuccess) { set_error("head_forward: %s", hipGetErrorString(e)); return (int)e; }
            if (dev >= 0 && dev < 64) sm_set[dev] = sm;
        }
    }
    const int nwb = ceil_div(N, 256);
    const int nblk = ceil_div(K, kRowsPerBlock);
    const WeightsJob wj{sim, rsim, labels, wts, W.partC};
    // K2: per pair the select block and, beside it, the weight blocks -> (ticket) Kabsch moments + solve (= equigspr_kabsch_forward
    // on the weights, when kabsch_aux is given) and -- small batches with the head's own Kabsch -- the MLP blocks of K3 as further
    // grid columns: they stage their weights while the select block works and wait for its selection.  Forward progress rests on
    // co-residency, not on dispatch order: fuse_mlp holds only when the whole fused grid fits the device at this launch's LDS size
    // (resident_pair_blocks), so the select block a consumer waits for is running or done.  Larger batches: K3 as its own launch.
    hipLaunchKernelGGL(head_pair_kernel, dim3(1 + nwb + (fuse_mlp ? nblk : 0), B), dim3(256), sm, s, hs, ht, labels, sim, W.partA, W.nbp,
                       w1, b1, w2, b2, w3, b3, B, N, Npad, F, K, md, pos, reinterpret_cast<long long*>(top_idx), logits, a1_save, a2_save,
                       W.pair_f, W.pair_part, out_scalars, W.tickets, wj, nwb, xs, xt, kabsch_R, kabsch_t, kabsch_aux, gt_pose, step_loss,
                       fuse_mlp ? nblk : 0, W.nC, (int)flags);
    // K3: the MLP on the selected rows; the last block over all pairs (ticket) writes the loss scalars
    if (!fuse_mlp)
        hipLaunchKernelGGL(head_mlp_kernel, dim3(nblk, B), dim3(256), kMlpLdsBytes, s, hs, ht, labels,
                           reinterpret_cast<const long long*>(top_idx), w1, b1, w2, b2, w3, b3, B, N, F, K, md, logits, a1_save, a2_save,
                           W.pair_part, nblk, W.partC, W.nC, out_scalars, (int)flags, W.tickets + B, step_loss);
    EQ_CHECK_LAUNCH("head_forward");
    return 0;
}

extern "C" int equigspr_head_backward(const float* hs, const float* ht, const float* xs, const float* xt,
                                      const float* labels, const float* gt_pose, const float* mlp, const float* sim,
                                      const float* rsim, const float* cosv, const int32_t* pos, const float* wts,
                                      const int64_t* top_idx, const float* logits, const float* a1_save,
                                      const float* a2_save, const float* fin_scalars, const float* g_corr_sim,
                                      const float* g_equi, float* gw, const float* kabsch_aux, const float* grad_R,
                                      const float* grad_t, const float* kabsch_t, const float* g_step_loss,
                                      int32_t B, int32_t N, int32_t F, int32_t K,
                                      int32_t flags, void* work /* the forward's */, float* dinp, float* mlp_slab, float* grad_mlp /* += */,
                                      float* ghs, float* ght, float* gxs /* += */, float* gxt /* += */, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(hs && ht && xs && xt && labels && gt_pose && mlp && sim && rsim && cosv && pos && wts && top_idx && logits &&
                     a1_save && a2_save && fin_scalars && gw && work && dinp && mlp_slab && grad_mlp && ghs && ght && gxs && gxt,
                 EQUIGSPR_EINVAL, "head_backward: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const MlpDims md{2 * F, F, F / 2};
    const HeadWork W = head_work(work, B, N);
    const float* w1 = mlp;
    const float* w2 = w1 + md.h1 * md.in + md.h1;
    const float* w3 = w2 + md.h2 * md.h1 + md.h2;
    const int nblk = ceil_div(K, kRowsPerBlock);
    EQ_CHECK_ARG((grad_R != nullptr) == (grad_t != nullptr), EQUIGSPR_EINVAL, "head_backward: grad_R and grad_t go together");
    EQ_CHECK_ARG(!(grad_R || g_step_loss) || kabsch_aux, EQUIGSPR_EINVAL, "head_backward: grad_R / g_step_loss need kabsch_aux");
    EQ_CHECK_ARG(!g_step_loss || kabsch_t, EQUIGSPR_EINVAL, "head_backward: g_step_loss needs kabsch_t");
    KabschBwdJob kj{xs, xt, wts, kabsch_aux, grad_R, grad_t, g_step_loss, gt_pose, kabsch_t, gxs, gxt, gw, W.tickets,
                    kabsch_aux ? kabsch_bwd_blocks(N) : 0};
    hipLaunchKernelGGL(head_pair_bwd_kernel, dim3(nblk + 1 + kj.nkb, B), dim3(256), 0, s, hs, ht, labels, sim, W.pair_f, gw,
                       reinterpret_cast<const long long*>(top_idx), logits, a1_save, a2_save, w1, w2, w3, g_corr_sim, B, N, F, K,
                       md, W.pair_g, dinp, mlp_slab, nblk, kj);
    const int msz = (int)equigspr_head_mlp_size(F);
    hipLaunchKernelGGL(head_node_bwd_kernel, dim3(W.nbp + ceil_div(msz, 32), B), dim3(256), 0, s, hs, ht, xs, xt, labels, gt_pose,
                       sim, rsim, cosv, pos, dinp, gw, W.pair_f, W.pair_g, fin_scalars, g_corr_sim, g_equi, B, N, F, K, md.in, ghs, ght,
                       gxs, gxt, W.nbp, mlp_slab, B * nblk, msz, grad_mlp, (int)flags, g_step_loss);
    EQ_CHECK_LAUNCH("head_backward");
    return 0;
}

extern "C" int equigspr_head_eval_forward(const float* hs, const float* ht, const float* raw_hs, const float* raw_ht,
                                          const float* xs, const float* xt, const float* raw_xs, const float* raw_xt,
                                          const float* labels, const float* gt_pose, const float* mlp, int32_t B, int32_t N,
                                          int32_t F, int32_t F_raw, int32_t K, float* sim, float* sim0, float* cosv, int32_t* pos,
                                          float* wts, int64_t* top_idx, void* work, float* out_scalars, float* kabsch_R,
                                          float* kabsch_t, float* kabsch_aux, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(hs && ht && raw_hs && raw_ht && xs && xt && raw_xs && raw_xt && labels && gt_pose && mlp && sim && sim0 && cosv &&
                     pos && wts && top_idx && work && out_scalars && kabsch_R && kabsch_t && kabsch_aux,
                 EQUIGSPR_EINVAL, "head_eval_forward: null pointer");
    EQ_CHECK_ARG(B >= 1 && N >= 2 && F >= 2 && F <= 64 && F_raw >= 1 && F_raw <= 64 && K >= 1 && K <= N && N <= kMaxN,
                 EQUIGSPR_ENOTSUP, "head_eval_forward: need 2<=F<=64, 1<=K<=N, N<=8192");
    hipStream_t s = (hipStream_t)stream;
    const MlpDims md{2 * F, F, F / 2};
    const HeadWork W = head_work(work, B, N);
    const float* w1 = mlp;
    const float* b1 = w1 + md.h1 * md.in;
    const float* w2 = b1 + md.h1;
    const float* b2 = w2 + md.h2 * md.h1;
    const float* w3 = b2 + md.h2;
    const float* b3 = w3 + md.h2;
    // launch 1: per node <h_s, h_t> (unused by this head), sim0 = <h0_s, h0_t> (:691), the equi-loss terms on the EGNN outputs (:687)
    hipLaunchKernelGGL(head_node_kernel, dim3(W.nbp, B), dim3(256), 0, s, hs, ht, raw_hs, raw_ht, xs, xt, labels, gt_pose, B, N,
                       F, F_raw, sim, sim0, cosv, pos, W.partA, W.tickets);
    const int Npad = next_pow2(N);
    const size_t sm = fwd_smem(N, Npad, K, md);
    {
        static std::mutex mu;
        static size_t sm_set[64] = {0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || sm > sm_set[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&head_eval_pair_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
            if (e != hipSuccess) { set_error("head_eval_forward: %s", hipGetErrorString(e)); return (int)e; }
            if (dev >= 0 && dev < 64) sm_set[dev] = sm;
        }
    }
    hipLaunchKernelGGL(head_eval_pair_kernel, dim3(1, B), dim3(256), sm, s, hs, ht, sim0, W.partA, W.nbp, w1, b1, w2, b2, w3, b3, B, N,
                       Npad, F, K, md, pos, reinterpret_cast<long long*>(top_idx), W.pair_f, W.pair_g, out_scalars);
    // launch 3: the weights and their Kabsch moments spread over up to 8 blocks per pair (one per 1 024 points), ticket -> solve
    const int nkb = min(kKabschMaxBlocks, ceil_div(N, 1024));
    hipLaunchKernelGGL(head_eval_weights_kernel, dim3(nkb, B), dim3(256), 0, s, sim0, pos, W.pair_f, W.pair_g, N, nkb, wts, raw_xs, raw_xt,
                       kabsch_R, kabsch_t, kabsch_aux, W.tickets);
    EQ_CHECK_LAUNCH("head_eval_forward");
    return 0;
}

#ifdef EQ_TRACE
extern "C" int equigspr_trace_set_head(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(eq_trace_ptr), &buf, sizeof(buf));
}
#endif
