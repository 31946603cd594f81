// Typed irrep-basis message passing (SURVEY section 8(f) row N4 "proper": degree-0 AND degree-1 features carried through the layer,
// the equivariant kernels K_J of tools/basis.py::get_basis :153-205 inside it).  The layer (definition: oracle/irreps_oracle.py):
//
//   m0[e, o]    = sum_i R00[e, o, i] K00[e] f0[col, i]  +  sum_i R10[e, o, i] (K10[e, :] . f1[col, i, :])
//   m1[e, o, a] = sum_i R01[e, o, i] K01[e, a] f0[col, i]  +  sum_i sum_J R11[e, o, i, J] (K11[e, a, :, J] . f1[col, i, :])
//   out0[n]     = W_s0 f0[n] + sum_{e: row = n} m0[e]            out1[n] = W_s1 f1[n] + sum_{e: row = n} m1[e]
//
// Division of labour: the radial profile R [E, n_w] = W_r2 SiLU(w_r1 |r| + b_r1) + b_r2 is a plain dense GEMM over all edges and
// goes to the library (rocBLAS / hipBLASLt through ATen, MFMA); the basis tensors come from the existing SH kernels (csrc/sh.hip);
// THIS kernel does what no library does: per edge the contraction of radial weights x basis x gathered features for both degrees,
// and the aggregation at `row` -- a wave per destination node walking its incoming edges in ascending edge order (the inverse
// index of the EGNN path: no atomics, the CPU scatter_add order, bitwise reproducible), the per-edge operands (gathered features,
// basis, K.f contractions) in wave-private LDS, one output (o or (o, a)) per lane and accumulator.
// Edges with r = 0 (self loops of knn_graph(..., loop=True)): the J >= 1 kernels are zero there (Y_J(0) is not covariant).
#include "common.h"

namespace eq {

constexpr int kIrMaxC = 32;                    // channels per degree, in and out
constexpr int kIrWaves = 4;

struct IrrepConvArgs {
    const float* f0; const float* f1; const float* dist;          // [N, C0i], [N, C1i, 3], [E]
    const float* R;                                               // [E, n_w]: R00 | R01 | R10 | R11
    const float* k00; const float* k01; const float* k10; const float* k11;      // [E], [E,3], [E,3], [E,3(out),3(in),3(J)]
    const float* ws0; const float* ws1;                           // [C0o, C0i], [C1o, C1i]
    const int* col; const int* rptr; const int* rperm;
    int n_nodes, c0i, c1i, c0o, c1o;
    float* out0; float* out1;                                     // [N, C0o], [N, C1o, 3]
};

__global__ __launch_bounds__(64 * kIrWaves) void irrep_conv_kernel(const IrrepConvArgs a) {
    __shared__ float sm[kIrWaves][kIrMaxC + 3 * kIrMaxC + 40 + kIrMaxC + 9 * kIrMaxC];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = blockIdx.x * kIrWaves + wv;
    if (n >= a.n_nodes) return;                                   // wave-uniform
    float* g0 = sm[wv];                        // f0[col, :]                       [C0i]
    float* g1 = g0 + kIrMaxC;                  // f1[col, :, :]                    [C1i * 3]
    float* kb = g1 + 3 * kIrMaxC;              // k00 | k01[3] | k10[3] | k11[27]  (J >= 1 parts zeroed on a self loop)
    float* s1 = kb + 40;                       // K10 . f1[col, i]                 [C1i]
    float* t1 = s1 + kIrMaxC;                  // K11[a, :, J] . f1[col, i]        [C1i][3 a][3 J]
    const int c0i = a.c0i, c1i = a.c1i, c0o = a.c0o, c1o = a.c1o;
    const int n_out = c0o + 3 * c1o;           // <= 128: two outputs per lane at most
    const int o_r01 = c0o * c0i, o_r10 = o_r01 + c1o * c0i, o_r11 = o_r10 + c0o * c1i, n_w = o_r11 + c1o * c1i * 3;
    float acc[2] = {0.0f, 0.0f};
    const int lo = a.rptr[n], hi = a.rptr[n + 1];
    for (int p = lo; p < hi; ++p) {
        const int e = a.rperm[p];
        const int c = a.col[e];
        const float live = a.dist[e] > 0.0f ? 1.0f : 0.0f;
        for (int t = lane; t < c0i; t += 64) g0[t] = a.f0[(size_t)c * c0i + t];
        for (int t = lane; t < 3 * c1i; t += 64) g1[t] = a.f1[(size_t)c * 3 * c1i + t];
        if (lane < 34) {
            float v;
            if (lane == 0) v = a.k00[e];
            else if (lane < 4) v = a.k01[(size_t)e * 3 + lane - 1] * live;
            else if (lane < 7) v = a.k10[(size_t)e * 3 + lane - 4] * live;
            else { const int q = lane - 7; v = a.k11[(size_t)e * 27 + q] * ((q % 3) == 0 ? 1.0f : live); }      // [a][b][J]: J = q % 3
            kb[lane] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int t = lane; t < c1i; t += 64) s1[t] = kb[4] * g1[3 * t] + kb[5] * g1[3 * t + 1] + kb[6] * g1[3 * t + 2];
        for (int t = lane; t < 9 * c1i; t += 64) {
            const int i = t / 9, aj = t - 9 * i, aa = aj / 3, J = aj - 3 * aa;
            const float* k = kb + 7 + aa * 9 + J;                  // K11[aa][b][J], b stride 3
            t1[t] = k[0] * g1[3 * i] + k[3] * g1[3 * i + 1] + k[6] * g1[3 * i + 2];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float* Re = a.R + (size_t)e * n_w;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int o = lane + 64 * u;
            if (o >= n_out) break;
            float m = 0.0f;
            if (o < c0o) {
                const float* r00 = Re + o * c0i;
                const float* r10 = Re + o_r10 + o * c1i;
                float s = 0.0f;
                for (int i = 0; i < c0i; ++i) s = fmaf(r00[i], g0[i], s);
                m = s * kb[0];
                for (int i = 0; i < c1i; ++i) m = fmaf(r10[i], s1[i], m);
            } else {
                const int oo = (o - c0o) / 3, aa = (o - c0o) - 3 * oo;
                const float* r01 = Re + o_r01 + oo * c0i;
                const float* r11 = Re + o_r11 + oo * c1i * 3;
                float s = 0.0f;
                for (int i = 0; i < c0i; ++i) s = fmaf(r01[i], g0[i], s);
                m = s * kb[1 + aa];
                for (int i = 0; i < c1i; ++i) {
                    const float* tt = t1 + 9 * i + 3 * aa;
                    m = fmaf(r11[3 * i], tt[0], m);
                    m = fmaf(r11[3 * i + 1], tt[1], m);
                    m = fmaf(r11[3 * i + 2], tt[2], m);
                }
            }
            acc[u] += m;
        }
        __builtin_amdgcn_wave_barrier();                          // the next edge overwrites the scratch
    }
    // self-interaction (channel mixing of the node's own features) + store
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int o = lane + 64 * u;
        if (o >= n_out) break;
        float v = acc[u];
        if (o < c0o) {
            for (int i = 0; i < c0i; ++i) v = fmaf(a.ws0[o * c0i + i], a.f0[(size_t)n * c0i + i], v);
            a.out0[(size_t)n * c0o + o] = v;
        } else {
            const int oo = (o - c0o) / 3, aa = (o - c0o) - 3 * oo;
            for (int i = 0; i < c1i; ++i) v = fmaf(a.ws1[oo * c1i + i], a.f1[((size_t)n * c1i + i) * 3 + aa], v);
            a.out1[((size_t)n * c1o + oo) * 3 + aa] = v;
        }
    }
}


// ---- aggregate-then-transform form (the fast path) --------------------------------------------------------------------------
// The radial profile is LINEAR in the hidden radial activations: R[e, w] = sum_h W_r2[w, h] hid[e, h] + b_r2[w].  So the sum over a
// node's incoming edges can be taken BEFORE the radial weights are applied:
//     out0[n, o] = sum_{h, i} W_r2[(o, i) of R00, h] * M00[n, i, h] + ...,    M00[n, i, h] = sum_{e -> n} hid~[e, h] * (K00[e] f0[col, i])
// with hid~ = [hid, 1] (the 1 carries the bias), and likewise for the other three paths with their per-edge operands
//     c[e, :] = [ K00 f0 (C0i) | K10 . f1 (C1i) | K01[a] f0 (3 C0i) | K11[a, :, J] . f1 (9 C1i) ]            (Lc = 4 C0i + 10 C1i values).
// This kernel accumulates the per-node moment matrix M[n, j, h] = sum_e c[e, j] hid~[e, h] (an outer product per edge: Lc (H + 1) FMAs
// instead of the n_w H of the radial GEMM -- 6.9 k against 55 k at 32 + 8 channels), in ascending edge order; the radial weights are then
// applied to M by dense GEMMs over NODES (library).  Nothing of size [E, n_w] is ever formed (566 MB at one 2 048-point pair).
// A wave per node; lane l owns c-entries j = l, l + 64, l + 128, l + 192 (Lc <= 256) and all H + 1 <= 33 columns of their rows.
constexpr int kIrMaxH = 32;
constexpr int kIrMaxLc = 256;

struct IrrepMomentsArgs {
    const float* f0; const float* f1; const float* dist; const float* hid;      // hid [E, 32]: H columns, zero padded
    const float* k00; const float* k01; const float* k10; const float* k11;
    const int* col; const int* rptr; const int* rperm;
    int n_nodes, c0i, c1i, H;
    float* M;                                                                   // [N, Lc, H + 1]
};

constexpr int kIrRowsPerPass = 16;
#ifndef EQ_IR_MOM_OCC
#define EQ_IR_MOM_OCC 8                     // workgroups per CU = waves per SIMD the register budget is set for (the per-edge chain index -> rows -> LDS -> FMAs is latency: 194 us at four waves, 137 at eight, one pair)
#endif
__global__ __launch_bounds__(64 * kIrWaves, EQ_IR_MOM_OCC) void irrep_moments_kernel(const IrrepMomentsArgs a) {
    __shared__ float sm[kIrWaves][40];
    __shared__ float tiles[kIrWaves][kIrRowsPerPass * (kIrMaxH + 1)];      // transposing buffer of the output: 16 rows at a time (a 64-row buffer, 8.4 KB per wave, held the kernel at four waves per SIMD)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c0i = a.c0i, c1i = a.c1i, H = a.H, HA = H + 1;
    const int o10 = c0i, o01 = o10 + c1i, o11 = o01 + 3 * c0i, Lc = o11 + 9 * c1i;
    const int nq = (Lc + 63) >> 6;             // 64-entry slices of the operand vector c: one WAVE per (node, slice)
    // The node, its edge range and everything indexed by an edge are wave-uniform: told to the compiler (readfirstlane) they live in
    // scalar registers, and hid~[e, h] is the SCALAR operand of the outer product's FMAs.  One slice per wave keeps a wave at 33
    // accumulators (~60 VGPRs, eight waves per SIMD): the per-edge chain  basis -> LDS -> operand -> FMAs  is latency, and occupancy
    // is what hides it (all four slices in one wave: 181 VGPRs, two waves per SIMD, 286 us at one pair; with the hid~ row read from
    // LDS instead of scalar registers: 0.8 ms).
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * kIrWaves + wv);
    const int n = wid / nq, q = wid - n * nq;
    if (n >= a.n_nodes) return;
    float* kb = sm[wv];                        // k00 | k01[3] | k10[3] | k11[27]  (J >= 1 parts zeroed on a self loop)
    // this lane's entry of the operand vector, decoded once: c[j] = kb[ks] * f0[foff] (dot == 0) or sum_b kb[ks + b * kst] * f1[foff + b] (dot == 1)
    int dot = 0, foff = 0, ks = 0, kst = 0;
    const int t = lane + 64 * q;
    if (t < o10) { foff = t; ks = 0; }
    else if (t < o01) { dot = 1; foff = 3 * (t - o10); ks = 4; kst = 1; }
    else if (t < o11) { const int u = t - o01, aa = u / c0i; foff = u - aa * c0i; ks = 1 + aa; }
    else if (t < Lc) { const int u = t - o11, aj = u / c1i, i = u - aj * c1i, aa = aj / 3, J = aj - 3 * aa;      // [a][J][i]
                       dot = 1; foff = 3 * i; ks = 7 + aa * 9 + J; kst = 3; }
    else dot = -1;                                                      // no entry: contributes zeros
    float acc[kIrMaxH + 1];
#pragma unroll
    for (int h = 0; h <= kIrMaxH; ++h) acc[h] = 0.0f;
    const int lo = a.rptr[n], hi = a.rptr[n + 1];
    for (int p = lo; p < hi; ++p) {
        const int e = a.rperm[p];
        const int c = a.col[e];
        const float live = a.dist[e] > 0.0f ? 1.0f : 0.0f;
        if (lane < 34) {
            float v;
            if (lane == 0) v = a.k00[e];
            else if (lane < 4) v = a.k01[(size_t)e * 3 + lane - 1] * live;
            else if (lane < 7) v = a.k10[(size_t)e * 3 + lane - 4] * live;
            else { const int u = lane - 7; v = a.k11[(size_t)e * 27 + u] * ((u % 3) == 0 ? 1.0f : live); }
            kb[lane] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float* f0r = a.f0 + (size_t)c * c0i;
        const float* f1r = a.f1 + (size_t)c * 3 * c1i;
        float cj = 0.0f;
        if (dot == 0) cj = kb[ks] * f0r[foff];
        else if (dot == 1) cj = kb[ks] * f1r[foff] + kb[ks + kst] * f1r[foff + 1] + kb[ks + 2 * kst] * f1r[foff + 2];
        // the hid row: always kIrMaxH columns (the caller pads with zeros), fetched as whole 16-byte scalar loads -- a per-column
        // `if (h < H)` made every column its own s_load + s_waitcnt + branch: 32 serial scalar-memory round trips per edge
        const float4* h4 = reinterpret_cast<const float4*>(a.hid + (size_t)e * kIrMaxH);
        float hv[kIrMaxH];
#pragma unroll
        for (int u = 0; u < kIrMaxH / 4; ++u) { const float4 v = h4[u]; hv[4 * u] = v.x; hv[4 * u + 1] = v.y; hv[4 * u + 2] = v.z; hv[4 * u + 3] = v.w; }
#pragma unroll
        for (int h = 0; h < kIrMaxH; ++h) acc[h] = fmaf(cj, hv[h], acc[h]);
        acc[kIrMaxH] += cj;                                             // the constant 1 of hid~ (column H of the output)
        __builtin_amdgcn_wave_barrier();                                // the next edge overwrites kb
    }
    // The wave's 64 rows (H + 1 values each) are transposed through LDS and streamed out row by row (a lane writing its own row
    // stores 4 bytes at a stride of 132: 0.27 ms for the 112 MB of one pair).  Destination: FOUR arrays, one per path, each the
    // contiguous A operand of its node-level GEMM (one [N, Lc, H + 1] block made the caller copy two of them, 90 MB at one pair):
    //   S00 [N][C0i (H+1)] | S10 [N][C1i (H+1)] | S01 [N * 3][C0i (H+1)] (rows (n, a)) | S11 [N * 3][3 C1i (H+1)] (rows (n, a), columns (J, i, h))
    float* tile = tiles[wv];
    const int rows = min(64, Lc - 64 * q);
    // entry t = 64 q + lane of the operand vector -> its row's first word in the sectioned output (entries of a section are contiguous)
    __shared__ long long rowbase[kIrWaves][64];
    {
        const size_t N_ = (size_t)a.n_nodes;
        size_t base;
        if (t < o10) base = ((size_t)n * c0i + t) * HA;
        else if (t < o01) base = N_ * c0i * HA + ((size_t)n * c1i + (t - o10)) * HA;
        else if (t < o11) base = N_ * (c0i + c1i) * HA + ((size_t)n * 3 * c0i + (t - o01)) * HA;
        else base = N_ * (c0i + c1i + 3 * c0i) * HA + ((size_t)n * 9 * c1i + (t - o11)) * HA;
        rowbase[wv][lane] = (long long)base;
    }
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto path_of = [&](int tt) { return tt < o10 ? 0 : (tt < o01 ? 1 : (tt < o11 ? 2 : 3)); };
    for (int r0 = 0; r0 < rows; r0 += kIrRowsPerPass) {                  // wave-uniform
        const int nr = min(kIrRowsPerPass, rows - r0);
        const int lr = lane - r0;                                        // this lane's row inside the pass
        if (lr >= 0 && lr < nr) {
#pragma unroll
            for (int h = 0; h < kIrMaxH; ++h)
                if (h < H) tile[lr * HA + h] = acc[h];
            tile[lr * HA + H] = acc[kIrMaxH];
        }
        wave_sync();
        if (path_of(64 * q + r0) == path_of(64 * q + r0 + nr - 1)) {     // the pass lies inside ONE path's array: one contiguous block
            float* dst = a.M + rowbase[wv][r0];
            for (int k = lane; k < nr * HA; k += 64) dst[k] = tile[k];
        } else {                                                         // two rows per step: lanes 0..31 / 32..63 (H + 1 <= 33)
            for (int rr = 0; rr < nr; rr += 2) {
                const int r = rr + (lane >> 5), hh = lane & 31;
                if (r < nr) {
                    float* dst = a.M + rowbase[wv][r0 + r];
                    if (hh < HA) dst[hh] = tile[r * HA + hh];
                    if (hh == 0 && HA > 32) dst[32] = tile[r * HA + 32];
                }
            }
        }
        wave_sync();                                                     // the next pass overwrites the buffer
    }
}


}  // namespace eq

extern "C" int equigspr_irrep_conv_forward(const float* f0, const float* f1, const float* dist, const float* radial, const float* k00,
                                           const float* k01, const float* k10, const float* k11, const float* w_self0,
                                           const float* w_self1, const int32_t* col, const int32_t* rptr, const int32_t* rperm,
                                           int32_t n_nodes, int32_t n_edges, int32_t c0_in, int32_t c1_in, int32_t c0_out, int32_t c1_out,
                                           float* out0, float* out1, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(f0 && f1 && dist && radial && k00 && k01 && k10 && k11 && w_self0 && w_self1 && col && rptr && rperm && out0 && out1,
                 EQUIGSPR_EINVAL, "irrep_conv_forward: null argument");
    EQ_CHECK_ARG(n_nodes >= 1 && n_edges >= 0, EQUIGSPR_EINVAL, "irrep_conv_forward: n_nodes >= 1, n_edges >= 0");
    EQ_CHECK_ARG(c0_in >= 1 && c1_in >= 1 && c0_out >= 1 && c1_out >= 1 && c0_in <= kIrMaxC && c1_in <= kIrMaxC && c0_out <= kIrMaxC &&
                     c1_out <= kIrMaxC, EQUIGSPR_ENOTSUP, "irrep_conv_forward: 1..%d channels per degree", kIrMaxC);
    EQ_CHECK_ARG(c0_out + 3 * c1_out <= 128, EQUIGSPR_ENOTSUP, "irrep_conv_forward: c0_out + 3 c1_out must be <= 128 (two outputs per lane)");
    IrrepConvArgs a{f0, f1, dist, radial, k00, k01, k10, k11, w_self0, w_self1, col, rptr, rperm, n_nodes, c0_in, c1_in, c0_out, c1_out, out0, out1};
    hipLaunchKernelGGL(irrep_conv_kernel, dim3(ceil_div(n_nodes, kIrWaves)), dim3(64 * kIrWaves), 0, (hipStream_t)stream, a);
    EQ_CHECK_LAUNCH("irrep_conv_forward");
    return 0;
}

extern "C" int equigspr_irrep_moments_forward(const float* f0, const float* f1, const float* dist, const float* hid, const float* k00,
                                              const float* k01, const float* k10, const float* k11, const int32_t* col, const int32_t* rptr,
                                              const int32_t* rperm, int32_t n_nodes, int32_t n_edges, int32_t c0_in, int32_t c1_in,
                                              int32_t hidden, float* moments, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(f0 && f1 && dist && hid && k00 && k01 && k10 && k11 && col && rptr && rperm && moments, EQUIGSPR_EINVAL,
                 "irrep_moments_forward: null argument");
    EQ_CHECK_ARG(n_nodes >= 1 && n_edges >= 0, EQUIGSPR_EINVAL, "irrep_moments_forward: n_nodes >= 1, n_edges >= 0");
    EQ_CHECK_ARG(c0_in >= 1 && c1_in >= 1 && c0_in <= kIrMaxC && c1_in <= kIrMaxC && hidden >= 1 && hidden <= kIrMaxH &&
                     4 * c0_in + 10 * c1_in <= kIrMaxLc, EQUIGSPR_ENOTSUP,
                 "irrep_moments_forward: 1..%d channels per degree with 4 c0_in + 10 c1_in <= %d, hidden <= %d", kIrMaxC, kIrMaxLc, kIrMaxH);
    IrrepMomentsArgs a{f0, f1, dist, hid, k00, k01, k10, k11, col, rptr, rperm, n_nodes, c0_in, c1_in, hidden, moments};
    const int nq = ceil_div(4 * c0_in + 10 * c1_in, 64);           // one wave per (node, 64-entry slice of the operand vector)
    hipLaunchKernelGGL(irrep_moments_kernel, dim3(ceil_div((long long)n_nodes * nq, kIrWaves)), dim3(64 * kIrWaves), 0, (hipStream_t)stream, a);
    EQ_CHECK_LAUNCH("irrep_moments_forward");
    return 0;
}
