// Edge kernels for the hidden widths WITHOUT an MFMA formulation (any hidden_nf <= 64 other than 32 / 33; the reference takes
// --hidden_node_nf and any num_heads dividing it: src/3dmatch_train_egnn_with_batch.py:1547, :186, :202-209).
//
// Layout: lane = feature.  A wave walks over a contiguous range of edges, one edge at a time; lane f < F holds feature f of
// every per-edge vector (t1, SiLU(t1), u, xhat, m, q, ...), the vectors a matrix product consumes are published through a
// wave-private LDS row (uniform-address reads = broadcasts), and the weights are read from a per-workgroup LDS image laid out
// so that the 64 lanes of a read touch consecutive words.  The backward keeps ROW f of every weight gradient in registers of
// lane f (F + F/H + 16 + 5 accumulators), sums them over the wave's edges, and the four waves of a workgroup merge into one
// slab (fixed order: bitwise-reproducible gradients, like the MFMA kernels).  Same math as E_GCL.edge_model / coord_model
// (:231-268) and the same interfaces as edge_fwd16_kernel / edge_bwd16_kernel: m rows at the edge's aggregation position, dt1
// rows [E][FP], gxr / gxc [E][4], Slabs<F>::edge.  An order of magnitude slower than the MFMA kernels -- the widths the
// reference actually runs (32, 33) never come here.
#pragma once
#include "egnn_bwd16.h"

namespace eq {

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LDS image of a layer's edge weights, lane-friendly: every matrix both ways
template <int F, int H>
struct LaneImage {
    static constexpr int FH = F / H;
    static constexpr int LDW = 64;                        // row stride of all matrices (lanes read consecutive words of a row)
    static constexpr int o_w2t = 0;                       // W2T[il][o]  = W2[o][head(o) FH + il]          (FH rows)
    static constexpr int o_w2 = o_w2t + FH * LDW;         // W2R[ol][i]  = W2[head(i) FH + ol][i]          (FH rows)
    static constexpr int o_w6t = o_w2 + FH * LDW;         // W6T[i][o]   = W6[o][i]                        (F rows)
    static constexpr int o_w6 = o_w6t + F * LDW;          // W6R[o][i]   = W6[o][i]                        (F rows)
    static constexpr int size = o_w6 + F * LDW;
    static constexpr int size_fwd = o_w6t + F * LDW;      // the forward needs W2T and W6T only (W2R staged anyway: layout shared)
};

template <int F, int H, int A, bool kBwd>
__device__ __forceinline__ void stage_lane_image(float* sm, const float* __restrict__ lw) {
    using LL = LP<F, H, A>;
    using IM = LaneImage<F, H>;
    constexpr int FH = F / H;
    for (int t = threadIdx.x; t < FH * 64; t += 256) {
        const int il = t >> 6, o = t & 63;
        const float v = o < F ? lw[LL::w2_off(o, il)] : 0.0f;               // W2[o][head(o) FH + il]
        sm[IM::o_w2t + il * IM::LDW + o] = v;
        if (kBwd) {
            // W2R[ol][i] = W2[head(i) FH + ol][i]: the element of row (head(i), ol) in column i
            const int ol = il, i = o;
            sm[IM::o_w2 + ol * IM::LDW + i] = i < F ? lw[LL::w2_off((i / FH) * FH + ol, i % FH)] : 0.0f;
        }
    }
    for (int t = threadIdx.x; t < F * 64; t += 256) {
        const int a = t >> 6, b = t & 63;
        sm[IM::o_w6t + a * IM::LDW + b] = b < F ? lw[LL::w6 + b * F + a] : 0.0f;                 // [i = a][o = b]
        if (kBwd) sm[IM::o_w6 + a * IM::LDW + b] = b < F ? lw[LL::w6 + a * F + b] : 0.0f;        // [o = a][i = b]
    }
}

// per-lane constants of a layer: row f of W_geo (12 geometry columns + the attribute column), biases, LayerNorm, w7
template <int F, int H, int A>
struct LaneConsts {
    float wg[kGeo + 1], b2, gam, bet, b6, w7;
    __device__ __forceinline__ void load(const float* __restrict__ lw, int f) {
        using LL = LP<F, H, A>;
        const bool on = f < F;
        const int ff = on ? f : 0;
#pragma unroll
        for (int g = 0; g < kGeo + 1; ++g) wg[g] = (on && g < kGeo + A) ? lw[LL::w1_off(ff, 2 * F + g)] : 0.0f;
        b2 = on ? lw[LL::b2_off(ff)] : 0.0f;
        gam = on ? lw[LL::ln_g + ff] : 0.0f;
        bet = on ? lw[LL::ln_b + ff] : 0.0f;
        b6 = on ? lw[LL::b6 + ff] : 0.0f;
        w7 = on ? lw[LL::w7 + ff] : 0.0f;
    }
};

// forward state of one edge, lane f = feature f (lanes >= F hold zeros)
struct LaneEdge {
    float t1, sg1, xhat, rstd, m, q, sgq, s;
};

// vec: wave-private LDS row [64].  Returns the forward of edge (r, c) in `o`; geo / attr are wave-uniform.
template <int F, int H, int A>
__device__ __forceinline__ void lane_forward(const float* sm, float* vec, const LaneConsts<F, H, A>& k, const float* __restrict__ P,
                                             int r, int c, const float (&geo)[kGeo], float attr, int f, LaneEdge& o) {
    using IM = LaneImage<F, H>;
    constexpr int FP = Geom<F>::FP, FH = F / H;
    const bool on = f < F;
    float t1 = 0.0f;
    if (on) {
        t1 = P[(size_t)r * 2 * FP + f] + P[(size_t)c * 2 * FP + FP + f];
#pragma unroll
        for (int g = 0; g < kGeo; ++g) t1 = fmaf(k.wg[g], geo[g], t1);
        if (A) t1 = fmaf(k.wg[kGeo], attr, t1);
    }
    o.t1 = t1;
    o.sg1 = sigmoid_fast(t1);
    __builtin_amdgcn_wave_barrier();
    vec[f] = on ? t1 * o.sg1 : 0.0f;                     // SiLU (:245)
    __builtin_amdgcn_wave_barrier();
    float u = k.b2;
    {
        const int hb = on ? (f / FH) * FH : 0;
#pragma unroll 8
        for (int il = 0; il < FH; ++il) u = fmaf(sm[IM::o_w2t + il * IM::LDW + f], vec[hb + il], u);
    }
    u = on ? u : 0.0f;
    // LayerNorm over the F features (biased variance, eps 1e-5; :249)
    const float mean = wave_sum64(u) * (1.0f / F);
    const float cen = on ? u - mean : 0.0f;
    const float var = wave_sum64(cen * cen) * (1.0f / F);
    o.rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
    o.xhat = cen * o.rstd;
    o.m = on ? fmaf(o.xhat, k.gam, k.bet) : 0.0f;
    __builtin_amdgcn_wave_barrier();
    vec[f] = o.m;
    __builtin_amdgcn_wave_barrier();
    float q = k.b6;
#pragma unroll 8
    for (int i = 0; i < F; ++i) q = fmaf(sm[IM::o_w6t + i * IM::LDW + f], vec[i], q);
    o.q = on ? q : 0.0f;
    o.sgq = sigmoid_fast(o.q);
    o.s = wave_sum64(on ? k.w7 * (o.q * o.sgq) : 0.0f);   // :264
}

// contiguous share of the edges for wave w of nw
__device__ __forceinline__ void lane_range(int n_edges, int w, int nw, int& lo, int& hi) {
    const int per = (n_edges + nw - 1) / nw;
    lo = min(n_edges, w * per);
    hi = min(n_edges, lo + per);
}

template <int F, int H, int A>
__global__ __launch_bounds__(256) void edge_fwd_lane_kernel(const float* __restrict__ lw, const int* __restrict__ row,
                                                            const int* __restrict__ col, const float* __restrict__ edge_attr,
                                                            const float* __restrict__ x, const float* __restrict__ P,
                                                            const int* __restrict__ rinv, int n_edges, int basis,
                                                            float* __restrict__ m_out, float* __restrict__ trans_out) {
    using IM = LaneImage<F, H>;
    constexpr int FP = Geom<F>::FP;
    __shared__ float sm[IM::size_fwd];
    __shared__ float vecs[4][64];
    const int f = threadIdx.x & 63, wv = threadIdx.x >> 6;
    stage_lane_image<F, H, A, false>(sm, lw);
    LaneConsts<F, H, A> k;
    k.load(lw, f);
    __syncthreads();
    float* vec = vecs[wv];
    int lo, hi;
    lane_range(n_edges, blockIdx.x * 4 + wv, gridDim.x * 4, lo, hi);
    for (int e = lo; e < hi; ++e) {
        const int r = row[e], c = col[e], at = rinv[e];
        float xr[3], xc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { xr[i] = x[(size_t)r * 3 + i]; xc[i] = x[(size_t)c * 3 + i]; }
        EdgeGeo eg;
        edge_geometry_mode(basis, xr, xc, eg);
        const float attr = A ? (edge_attr ? edge_attr[e] : 1.0f) : 0.0f;
        LaneEdge o;
        lane_forward<F, H, A>(sm, vec, k, P, r, c, eg.g, attr, f, o);
        if (f < FP) m_out[(size_t)at * FP + f] = o.m;                     // the padding columns (f >= F) hold zeros
        if (f < 4) trans_out[(size_t)at * 4 + f] = f < 3 ? eg.d[f] * o.s : 0.0f;      // trans = coord_diff * s (:264)
    }
}

template <int F, int H, int A>
__global__ __launch_bounds__(256) void edge_bwd_lane_kernel(const float* __restrict__ lw, const int* __restrict__ row,
                                                            const int* __restrict__ col, const float* __restrict__ edge_attr,
                                                            const float* __restrict__ x, const float* __restrict__ P,
                                                            const float* __restrict__ dagg, const float* __restrict__ Gx, int n_edges,
                                                            int basis, float* __restrict__ dt1_out, float* __restrict__ gxr_out,
                                                            float* __restrict__ gxc_out, float* __restrict__ slab) {
    using IM = LaneImage<F, H>;
    using SB = Slabs<F>;
    constexpr int FP = Geom<F>::FP, FH = F / H;
    extern __shared__ __attribute__((aligned(16))) float dyn[];          // image | 4 x 2 wave rows; reused for the slab merge
    float* sm = dyn;
    float* vecs = dyn + IM::size;
    const int f = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool on = f < F;
    stage_lane_image<F, H, A, true>(sm, lw);
    LaneConsts<F, H, A> k;
    k.load(lw, f);
    __syncthreads();
    float* vec = vecs + wv * 128;
    float* vec2 = vec + 64;
    // row f of the weight gradients, summed over this wave's edges
    float a6[F], a2[FH], ag[16], ab6 = 0.0f, ab2 = 0.0f, agam = 0.0f, abet = 0.0f, aw7 = 0.0f;
#pragma unroll
    for (int i = 0; i < F; ++i) a6[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < FH; ++i) a2[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) ag[i] = 0.0f;
    int lo, hi;
    lane_range(n_edges, blockIdx.x * 4 + wv, gridDim.x * 4, lo, hi);
    for (int e = lo; e < hi; ++e) {
        const int r = row[e], c = col[e];
        float xr[3], xc[3], gxv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { xr[i] = x[(size_t)r * 3 + i]; xc[i] = x[(size_t)c * 3 + i]; gxv[i] = Gx[(size_t)r * 4 + i]; }
        EdgeGeo eg;
        edge_geometry_mode(basis, xr, xc, eg);
        const float attr = A ? (edge_attr ? edge_attr[e] : 1.0f) : 0.0f;
        LaneEdge o;
        lane_forward<F, H, A>(sm, vec, k, P, r, c, eg.g, attr, f, o);          // leaves m in vec
        // ---- coord head backward: trans = d s, s = w7 . SiLU(q)
        const float ds = dot3(eg.d, gxv);
        const float dq = on ? k.w7 * ds * (o.sgq * (1.0f + o.q * (1.0f - o.sgq))) : 0.0f;
        aw7 = fmaf(o.q * o.sgq, ds, aw7);
        ab6 += dq;
#pragma unroll
        for (int i = 0; i < F; ++i) a6[i] = fmaf(dq, vec[i], a6[i]);            // dW6[f][i] += dq[f] m[i]
        vec2[f] = dq;
        __builtin_amdgcn_wave_barrier();
        // ---- dm = dagg[row] + W6^T dq
        float dm = on ? dagg[(size_t)r * FP + f] : 0.0f;
#pragma unroll 8
        for (int oo = 0; oo < F; ++oo) dm = fmaf(sm[IM::o_w6 + oo * IM::LDW + f], vec2[oo], dm);
        dm = on ? dm : 0.0f;
        agam = fmaf(dm, o.xhat, agam);
        abet += dm;
        // ---- LayerNorm backward
        const float dxh = dm * k.gam;
        const float m1 = wave_sum64(dxh) * (1.0f / F), m2 = wave_sum64(dxh * o.xhat) * (1.0f / F);
        const float du = on ? o.rstd * (dxh - m1 - o.xhat * m2) : 0.0f;
        ab2 += du;
        __builtin_amdgcn_wave_barrier();
        vec[f] = on ? o.t1 * o.sg1 : 0.0f;                                      // SiLU(t1) again (m is consumed)
        vec2[f] = du;
        __builtin_amdgcn_wave_barrier();
        const int hb = on ? (f / FH) * FH : 0;
#pragma unroll
        for (int il = 0; il < FH; ++il) a2[il] = fmaf(du, vec[hb + il], a2[il]);       // dW2[f][hb + il] += du[f] SiLU(t1)[hb + il]
        // ---- dt1 = (W2^T du) SiLU'(t1): column f of W2 lives in rows hb .. hb + FH - 1
        float dt1 = 0.0f;
#pragma unroll 8
        for (int ol = 0; ol < FH; ++ol) dt1 = fmaf(sm[IM::o_w2 + ol * IM::LDW + f], vec2[hb + ol], dt1);
        dt1 = on ? dt1 * (o.sg1 * (1.0f + o.t1 * (1.0f - o.sg1))) : 0.0f;
        if (f < FP) dt1_out[(size_t)e * FP + f] = dt1;
#pragma unroll
        for (int g = 0; g < kGeo; ++g) ag[g] = fmaf(dt1, eg.g[g], ag[g]);        // dWgeo[f][g] += dt1[f] geo[g]
        ag[kGeo] = fmaf(dt1, attr, ag[kGeo]);
        // ---- dgeo = Wgeo^T dt1 (12 wave sums), geometry backward on every lane (wave-uniform values)
        float gg[kGeo];
#pragma unroll
        for (int g = 0; g < kGeo; ++g) gg[g] = wave_sum64(k.wg[g] * dt1);
        const float gd[3] = {o.s * gxv[0], o.s * gxv[1], o.s * gxv[2]};
        float gxr[3], gxc[3];
        edge_geometry_backward_mode(basis, xr, xc, eg, gg, gd, gxr, gxc);
        if (f < 4) {
            gxr_out[(size_t)e * 4 + f] = f == 0 ? gxr[0] : f == 1 ? gxr[1] : f == 2 ? gxr[2] : 0.0f;
            gxc_out[(size_t)e * 4 + f] = f == 0 ? gxc[0] : f == 1 ? gxc[1] : f == 2 ? gxc[2] : 0.0f;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // ---- the four waves' partial rows -> one slab per workgroup, waves added in order
    __syncthreads();
    float* sl = dyn;                                        // every wave is past its last image / row read
    for (int t = threadIdx.x; t < SB::edge; t += 256) sl[t] = 0.0f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wv == w && on) {
#pragma unroll
            for (int i = 0; i < F; ++i) sl[SB::e_w6 + f * F + i] += a6[i];
            const int hb = (f / FH) * FH;                   // the dense F x F block of W2 stays zero outside the head's diagonal block
#pragma unroll
            for (int q = 0; q < FH; ++q) sl[SB::e_w2 + f * F + hb + q] += a2[q];
#pragma unroll
            for (int g = 0; g < 16; ++g) sl[SB::e_geo + f * 16 + g] += ag[g];
            sl[SB::e_b6 + f] += ab6; sl[SB::e_b2 + f] += ab2; sl[SB::e_gam + f] += agam; sl[SB::e_bet + f] += abet; sl[SB::e_w7 + f] += aw7;
        }
        __syncthreads();
    }
    for (int t = threadIdx.x; t < SB::edge; t += 256) slab[(size_t)blockIdx.x * SB::edge + t] = sl[t];
}

template <int F, int H>
static inline size_t edge_bwd_lane_smem() {
    const size_t a = (size_t)LaneImage<F, H>::size + 4 * 128, b = (size_t)Slabs<F>::edge;
    return sizeof(float) * (a > b ? a : b);
}

}  // namespace eq
