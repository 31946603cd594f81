// MFMA formulation of the per-edge MLP chain (hidden_nf == 32), shared by the forward and
// backward edge kernels.
//
// A wave works on a TILE of 32 edges.  Lane l = (e = l & 31, h = l >> 5) owns edge e of the tile and
// HALF of its 32 features: register r (0..15) holds feature fidx(r,h) = (r&3) + 8*(r>>2) + 4*h.
// That is exactly the C/D layout of v_mfma_f32_32x32x2_f32 with D[i = feature][j = edge], and
// -- with the K index of step s mapped to feature fidx(s,h) -- register s of such a tile is
// ALSO the B operand of step s of the next product.  So a whole chain
//      t1 = (P_r + P_c) + Wgeo*geo -> SiLU -> W2 -> LayerNorm -> W6 -> SiLU -> w7.
// runs as 7 + 16 + 16 MFMAs per 32 edges with the weights held in VGPRs (the A operands, loaded
// once per wave), no LDS, no cross-lane traffic except one lane-half exchange for the LayerNorm
// and the final dot product.  fp32 MFMA is an exact fmaf chain (guide §3), so numerics match
// the VALU formulation.  The backward uses the transposed weights as A operands the same way.
#pragma once
#include "egnn_device.h"

namespace eq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ constexpr int fidx(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }   // partner lane of the other half

// Per-workgroup LDS image of the layer's edge-MLP weights (filled with coalesced global reads,
// rows padded to odd strides so the per-lane operand fetches below are bank-conflict-free).
// B = 0: hidden_nf = 32.  B = 1: hidden_nf = 33 -- the 32x32 cores stay on the matrix cores; feature 32 (the "border")
// is one extra k-step per product for its column and a VALU dot product for its row (accessors x_* below).
template <int B>
struct TileWeightsLdsT {
    static constexpr int R = 32 + B;                   // rows / columns of the square matrices
    static constexpr int GS = 15;                      // Wgeo row stride: 13 used columns
    static constexpr int WS = 33 + 2 * B;              // W2full / W6 row stride (odd)
    static constexpr int VS = 32 + 4 * B;              // stride of the per-feature vectors
    static constexpr int o_geo = 0, o_w2 = R * GS, o_w6 = o_w2 + R * WS, o_vec = o_w6 + R * WS;   // vec: b2|gam|bet|b6|w7
    static constexpr int size = o_vec + 5 * VS;
};
using TileWeightsLds = TileWeightsLdsT<0>;

template <int H, int A, int B = 0>
__device__ __forceinline__ void stage_tile_weights(float* sm, const float* __restrict__ lw) {
    constexpr int F = 32 + B;
    using LL = LP<F, H, A>;
    using T = TileWeightsLdsT<B>;
    constexpr int FH = F / H;
    constexpr int NG = (T::R * T::GS + 255) / 256, NW = (T::R * T::R + 255) / 256;
    // all global loads of a thread first, then its LDS stores (one exposed L2 round trip instead of one per element)
    float vg[NG], v2[NW], v6[NW];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const int t = threadIdx.x + 256 * u, i = t / T::GS, k = t - i * T::GS;
        vg[u] = (t < T::R * T::GS && k < kGeo + A) ? lw[LL::w1_off(i, 2 * F + k)] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
        const int t = threadIdx.x + 256 * u, i = t / T::R, k = t - i * T::R;
        const bool in = t < T::R * T::R;
        v2[u] = (in && i / FH == k / FH) ? lw[LL::w2_off(i, k)] : 0.0f;
        v6[u] = in ? lw[LL::w6 + t] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < NG; ++u) {
        const int t = threadIdx.x + 256 * u;
        if (t < T::R * T::GS) sm[T::o_geo + t] = vg[u];
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
        const int t = threadIdx.x + 256 * u, i = t / T::R, k = t - i * T::R;
        if (t < T::R * T::R) {
            sm[T::o_w2 + i * T::WS + k] = v2[u];
            sm[T::o_w6 + i * T::WS + k] = v6[u];
        }
    }
    if (threadIdx.x < T::R) {
        const int k = threadIdx.x;
        sm[T::o_vec + k] = lw[LL::b2_off(k)];
        sm[T::o_vec + T::VS + k] = lw[LL::ln_g + k];
        sm[T::o_vec + 2 * T::VS + k] = lw[LL::ln_b + k];
        sm[T::o_vec + 3 * T::VS + k] = lw[LL::b6 + k];
        sm[T::o_vec + 4 * T::VS + k] = lw[LL::w7 + k];
    }
}

// Operand accessors.  kCached = false: the MFMA A operands and the per-feature vectors are fetched from
// the LDS image right where they are used (one conflict-free ds_read per MFMA) instead of living in ~170
// VGPRs -- that lets four forward waves share a SIMD; `refresh()` re-materialises the base pointer so the
// compiler cannot hoist the loop-invariant reads back into registers.  kCached = true: everything is read
// once per wave into registers (the backward kernel runs at one wave per SIMD anyway, where exposed LDS
// latency costs more than the registers).  The border accessors (x_*, B = 1) always read LDS.
template <int H, int A, bool kCached = false, int B = 0>
struct TileWeights {
    using T = TileWeightsLdsT<B>;
    const float* base;     // the workgroup's LDS image (address space known to the compiler: ds_read, not flat_load)
    const float* sm;       // base + an opaque zero offset (refresh()): keeps the reads inside the tile loop
    int i, h;
    float c_wg[kCached ? 7 : 1], c_w2[kCached ? 16 : 1], c_w6[kCached ? 16 : 1], c_vec[kCached ? 80 : 1];
    float c_w6t[kCached ? 16 : 1], c_w2t[kCached ? 16 : 1], c_wgt[kCached ? 16 : 1];
    __device__ __forceinline__ float l_wg(int s) const { return sm[T::o_geo + i * T::GS + 2 * s + h]; }
    __device__ __forceinline__ float l_w2(int s) const { return sm[T::o_w2 + i * T::WS + fidx(s, h)]; }
    __device__ __forceinline__ float l_w6(int s) const { return sm[T::o_w6 + i * T::WS + fidx(s, h)]; }
    __device__ __forceinline__ float l_vec(int which, int r) const { return sm[T::o_vec + T::VS * which + fidx(r, h)]; }
    __device__ __forceinline__ float l_w6t(int s) const { return sm[T::o_w6 + fidx(s, h) * T::WS + i]; }
    __device__ __forceinline__ float l_w2t(int s) const { return sm[T::o_w2 + fidx(s, h) * T::WS + i]; }
    __device__ __forceinline__ float l_wgt(int s) const { return (i < kGeo) ? sm[T::o_geo + fidx(s, h) * T::GS + i] : 0.0f; }
    __device__ __forceinline__ void load(const float* s_, int lane) {
        base = s_; sm = s_; i = lane & 31; h = lane >> 5;
        if (kCached) {
#pragma unroll
            for (int s = 0; s < 7; ++s) c_wg[kCached ? s : 0] = l_wg(s);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                c_w2[kCached ? s : 0] = l_w2(s); c_w6[kCached ? s : 0] = l_w6(s);
                c_w6t[kCached ? s : 0] = l_w6t(s); c_w2t[kCached ? s : 0] = l_w2t(s); c_wgt[kCached ? s : 0] = l_wgt(s);
#pragma unroll
                for (int w = 0; w < 5; ++w) c_vec[kCached ? w * 16 + s : 0] = l_vec(w, s);
            }
        }
    }
    __device__ __forceinline__ void refresh() {
        if (!kCached || B) {
            int off = 0;
            asm volatile("" : "+v"(off));          // laundering the POINTER would lose its address space (flat loads)
            sm = base + off;
        }
    }
    __device__ __forceinline__ float wg(int s) const { return kCached ? c_wg[kCached ? s : 0] : l_wg(s); }
    __device__ __forceinline__ float w2(int s) const { return kCached ? c_w2[kCached ? s : 0] : l_w2(s); }
    __device__ __forceinline__ float w6(int s) const { return kCached ? c_w6[kCached ? s : 0] : l_w6(s); }
    __device__ __forceinline__ float vec(int which, int r) const { return kCached ? c_vec[kCached ? which * 16 + r : 0] : l_vec(which, r); }
    __device__ __forceinline__ float b2(int r) const { return vec(0, r); }
    __device__ __forceinline__ float gam(int r) const { return vec(1, r); }
    __device__ __forceinline__ float bet(int r) const { return vec(2, r); }
    __device__ __forceinline__ float b6(int r) const { return vec(3, r); }
    __device__ __forceinline__ float w7(int r) const { return vec(4, r); }
    // transposed products of the backward
    __device__ __forceinline__ float w6t(int s) const { return kCached ? c_w6t[kCached ? s : 0] : l_w6t(s); }
    __device__ __forceinline__ float w2t(int s) const { return kCached ? c_w2t[kCached ? s : 0] : l_w2t(s); }
    __device__ __forceinline__ float wgt(int s) const { return kCached ? c_wgt[kCached ? s : 0] : l_wgt(s); }
    // ---- border (feature 32) accessors, B = 1 only; mat = o_w2 or o_w6
    __device__ __forceinline__ float x_col_i(int mat) const { return sm[mat + i * T::WS + 32]; }            // W[i][32], lane i
    __device__ __forceinline__ float x_row_i(int mat) const { return sm[mat + 32 * T::WS + i]; }            // W[32][i], lane i
    __device__ __forceinline__ float x_row(int mat, int r) const { return sm[mat + 32 * T::WS + fidx(r, h)]; }   // W[32][fidx(r,h)]
    __device__ __forceinline__ float x_col(int mat, int r) const { return sm[mat + fidx(r, h) * T::WS + 32]; }   // W[fidx(r,h)][32]
    __device__ __forceinline__ float x_cc(int mat) const { return sm[mat + 32 * T::WS + 32]; }
    __device__ __forceinline__ float x_wg(int k) const { return sm[T::o_geo + 32 * T::GS + k]; }            // Wgeo[32][k]
    __device__ __forceinline__ float x_wg_i() const { return (i < kGeo + A) ? sm[T::o_geo + 32 * T::GS + i] : 0.0f; }   // lane i
    __device__ __forceinline__ float x_vec(int which) const { return sm[T::o_vec + T::VS * which + 32]; }
};

// y = W v (+ bias) for a (32+B)-square W held as LDS image `mat`: the 32x32 core is 16 MFMA steps with the cached/LDS A
// operands `aop(s)`; with B = 1 one more step adds W[:, 32] * vb to the 32 core outputs, and the 33rd output is a 16-term
// per-lane dot product + one lane-half exchange.  kT: transposed product (y = W^T v).
template <bool kT, int B, typename W_t, typename AOp>
__device__ __forceinline__ void mv_border(const W_t& W, int mat, AOp aop, const float (&v)[16], float vb, f32x16& acc, float& yb,
                                          float yb_init, int h) {
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aop(s), v[s], acc, 0, 0, 0);
    if constexpr (B == 1) {
        const float ax = kT ? W.x_row_i(mat) : W.x_col_i(mat);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? 0.0f : ax, h ? 0.0f : vb, acc, 0, 0, 0);
        float d = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) d = fmaf(kT ? W.x_col(mat, r) : W.x_row(mat, r), v[r], d);
        d += xhalf(d);
        yb = fmaf(W.x_cc(mat), vb, d + yb_init);
    } else {
        yb = 0.0f;
    }
}

// gather 16 of the 32 floats of a 128-byte row: features fidx(r,h), r = 0..15 (4 x 16-byte loads)
__device__ __forceinline__ void load_half_row(const float* __restrict__ row, int h, float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(row + 8 * q + 4 * h);
        v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
    }
}
__device__ __forceinline__ void store_half_row(float* __restrict__ row, int h, const float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(row + 8 * q + 4 * h) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// rows of (32 + B) floats with a padded stride: the 32 core columns as before, column 32 separately
__device__ __forceinline__ void store_border(float* __restrict__ row, int h, float vb) {
    if (h == 0) *reinterpret_cast<float4*>(row + 32) = make_float4(vb, 0.0f, 0.0f, 0.0f);
}

// everything the backward needs from the forward of one tile (the *b members: feature 32, B = 1)
struct TileFwd {
    float t1[16], xhat[16], m[16], q[16];
    float sg1[16], sgq[16];      // sigmoid(t1), sigmoid(q): SiLU = v*sg, SiLU' = sg*(1 + v*(1-sg)) without a second exp
    float rstd, s;
    float t1b, xhatb, mb, qb, sg1b, sgqb;
};

template <int H, int A, bool kCached, int B = 0>
__device__ __forceinline__ void tile_forward(const TileWeights<H, A, kCached, B>& W, const float* __restrict__ prow,
                                             const float* __restrict__ pcrow, const EdgeGeo& eg, float attr, int h,
                                             TileFwd& o) {
    using T = TileWeightsLdsT<B>;
    constexpr float kInvF = 1.0f / (32 + B);
    float pr[16], pc[16];
    load_half_row(prow, h, pr);
    load_half_row(pcrow, h, pc);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = pr[r] + pc[r];
    EQ_STAMP(4);
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        // B operand: z-geometry entry k = 2s+h of this lane's edge (12 geo values, then edge_attr)
        const float g0 = (2 * s < kGeo) ? eg.g[(2 * s < kGeo) ? 2 * s : 0] : ((2 * s == kGeo) ? attr : 0.0f);
        const float g1 = (2 * s + 1 < kGeo) ? eg.g[(2 * s + 1 < kGeo) ? 2 * s + 1 : 0] : ((2 * s + 1 == kGeo) ? attr : 0.0f);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W.wg(s), h ? g1 : g0, acc, 0, 0, 0);
    }
    o.t1b = 0.0f; o.sg1b = 0.0f;
    float a1b = 0.0f;
    if constexpr (B == 1) {
        float t = prow[32] + pcrow[32];
#pragma unroll
        for (int k = 0; k < kGeo; ++k) t = fmaf(W.x_wg(k), eg.g[k], t);
        if (A) t = fmaf(W.x_wg(kGeo), attr, t);
        o.t1b = t;
        o.sg1b = sigmoid_fast(t);
        a1b = t * o.sg1b;
    }
    f32x16 acc2, acc6;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o.t1[r] = acc[r]; acc2[r] = W.b2(r); acc6[r] = W.b6(r); }
    float a1[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        o.sg1[s] = sigmoid_fast(o.t1[s]);
        a1[s] = o.t1[s] * o.sg1[s];
    }
    float ub;
    mv_border<false, B>(W, T::o_w2, [&](int s) { return W.w2(s); }, a1, a1b, acc2, ub, B ? W.x_vec(0) : 0.0f, h);
    float sum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc2[r];
    EQ_STAMP(5);
    sum += xhalf(sum);
    if (B) sum += ub;
    const float mean = sum * kInvF;
    float var = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float c = acc2[r] - mean; var = fmaf(c, c, var); }
    var += xhalf(var);
    if (B) var = fmaf(ub - mean, ub - mean, var);
    var *= kInvF;
    o.rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        o.xhat[r] = (acc2[r] - mean) * o.rstd;
        o.m[r] = fmaf(o.xhat[r], W.gam(r), W.bet(r));
    }
    o.xhatb = 0.0f; o.mb = 0.0f;
    if constexpr (B == 1) {
        o.xhatb = (ub - mean) * o.rstd;
        o.mb = fmaf(o.xhatb, W.x_vec(1), W.x_vec(2));
    }
    mv_border<false, B>(W, T::o_w6, [&](int s) { return W.w6(s); }, o.m, o.mb, acc6, o.qb, B ? W.x_vec(3) : 0.0f, h);
    float sp = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o.q[r] = acc6[r]; o.sgq[r] = sigmoid_fast(acc6[r]); sp = fmaf(W.w7(r), acc6[r] * o.sgq[r], sp); }
    o.s = sp + xhalf(sp);
    o.sgqb = 0.0f;
    if constexpr (B == 1) {
        o.sgqb = sigmoid_fast(o.qb);
        o.s = fmaf(W.x_vec(4), o.qb * o.sgqb, o.s);
    }
    EQ_STAMP(6);
}

static inline int64_t saved_node_floats(int FP, int L, int64_t N) {      // h_0..h_L, x_0..x_L, agg_0..agg_{L-1}; padded to 16 bytes
    return ((int64_t)(L + 1) * N * FP + (int64_t)(L + 1) * N * 3 + (int64_t)L * N * FP + 3) / 4 * 4;
}

}  // namespace eq
