// Feature-parallel MFMA edge kernels for the hidden widths that are whole multiples of 16 other than 32 (16, 48, 64; the reference
// takes --hidden_node_nf and any num_heads dividing it: src/3dmatch_train_egnn_with_batch.py:1547, :186, :202-209).
//
// The 16-edge formulation of egnn_mfma16.h keeps ALL features of an edge in one wave: at hidden_nf 64 its backward would need 144
// registers of weight-gradient accumulators on top of 96 of tile state -- it does not fit.  Here the NB = F / 16 waves of a tile
// group each own ONE 16-feature block of every vector (lane (e = l & 15, g = l >> 4): edge e, features 16 w + 4 g + r -- the same
// C/D layout, one f32x4 per vector instead of NB) and one block ROW of every matrix product and weight gradient:
//   y_w = sum_bi W[16 w .., 16 bi ..] v_bi     needs the other waves' blocks v_bi: every wave publishes its block in an LDS tile
//   V [16 edges][F] (ds_write_b128), a workgroup barrier, then reads all blocks back as MFMA B operands (ds_read_b128);
//   LayerNorm / the coord scalar reduce over the F features: per-wave partial (hsum4 + two lane swaps) -> LDS [NB][16] -> barrier ->
//   summed in fixed wave order by everyone (bitwise reproducible);
//   weight gradients: wave w accumulates rows 16 w .. 16 w + 15 of dW6, dW2, dWgeo (NB + NB + 1 accumulators) -- no merge at the end,
//   every wave writes its own rows of the workgroup's slab.
// Tile groups per workgroup: fp_tpb below.
// Same math as E_GCL.edge_model / coord_model (:231-268), same interfaces as the other edge kernels: m rows at the edge's
// aggregation position, trans [E][4], dt1 rows [E][FP], gxr / gxc [E][4], Slabs<F>::edge.  (Round 4; the lane-per-feature kernels
// of rounds 1-3 (csrc/egnn_edge_lane.h, removed in round 5) were ~5x slower per flop: 506 / 124 us backward / forward per launch at hidden_nf 64, one 2 048-point pair.)
#pragma once
#include "egnn_bwd16.h"

namespace eq {

// tile groups per workgroup (measured at hidden_nf 64, one pair): the forward likes two (36 us against 44 with one), the backward one
// (83 us against 93 with two once its image had shrunk to 43 KB) -- a workgroup barrier then synchronises exactly the NB waves that
// exchange data, and the workgroups a CU holds run out of step.  hidden_nf 16: four one-wave groups, no workgroup barrier in the tile loop.
constexpr int fp_tpb(int F, bool bwd) { return F / 16 >= 3 ? (bwd ? 1 : 2) : 4; }
constexpr int fp_threads(int F, bool bwd) { return 64 * (F / 16) * fp_tpb(F, bwd); }

template <int F, int H, int A, bool kBwd>
struct FpLayout {
    static constexpr int NB = F / 16, FH = F / H;
    static constexpr int kTpb = fp_tpb(F, kBwd);
    static constexpr int NW = NB * kTpb;
    static constexpr int WS = F + 8;                                     // row stride of the F-column matrices (b128 A-operand reads conflict-free)
    static constexpr int GS = 24;                                        // Wgeo [F][16 (+8)]
    static constexpr int VS = F + 4;                                     // row stride of the exchange tiles [16 edges][F]
    // weight image
    static constexpr int o_w2 = 0;
    static constexpr int o_w6 = o_w2 + F * WS;
    static constexpr int o_wg = o_w6 + F * WS;
    static constexpr int o_vec = o_wg + F * GS;                          // b2 | gamma | beta | b6 | w7
    static constexpr int image = o_vec + 5 * F;                          // (no transposed copies: the backward reads W^T operands column-wise,
                                                                         //  four ds_read_b32 instead of one b128 -- the image stays at 43 KB at
                                                                         //  width 64 and two workgroups share a CU)
    // per tile group
    static constexpr int g_v = 0;                                        // V [16][VS]
    static constexpr int g_x = g_v + 16 * VS;                            // backward: X (dq, then du), Z (dt1)
    static constexpr int g_z = g_x + (kBwd ? 16 * VS : 0);
    static constexpr int g_geo = g_z + (kBwd ? 16 * VS : 0);             // [16][16]: the edges' geometry rows
    static constexpr int g_dgeo = g_geo + 16 * 16;                       // backward: [16][16] dL/dgeo
    static constexpr int g_red = g_dgeo + (kBwd ? 16 * 16 : 0);          // [2][NB][16]
    static constexpr int g_idx = g_red + 2 * NB * 16;                    // row, col, pos (ints), ds / s (floats): [4][16]
    static constexpr int g_front = g_idx + 4 * 16;                       // backward: x_r, x_c, Gx[row] of the 16 edges [16][12]
    static constexpr int group = g_front + (kBwd ? 16 * 12 : 0);
    static constexpr int total = image + kTpb * group;
    // a 16 x 16 block (bo, bi) of W2 is all zero when its rows and columns lie in different heads
    // (heads need not be multiples of 16 wide -- hidden_nf 48 with 2 heads: 24 -- so compare the head RANGES the block's rows and columns touch)
    static constexpr __device__ bool w2_block_live(int bo, int bi) {
        return (16 * bo) / FH <= (16 * bi + 15) / FH && (16 * bi) / FH <= (16 * bo + 15) / FH;
    }
};

template <int F, int H, int A, bool kBwd>
__device__ __forceinline__ void fp_stage_image(float* sm, const float* __restrict__ lw, int nthreads) {
    using L = FpLayout<F, H, A, kBwd>;
    using LL = LP<F, H, A>;
    constexpr int FH = F / H;
    for (int t = threadIdx.x; t < F * F; t += nthreads) {
        const int i = t / F, k = t - i * F;
        const float v2 = (i / FH == k / FH) ? lw[LL::w2_off(i, k % FH)] : 0.0f;       // block-diagonal over the heads
        const float v6 = lw[LL::w6 + i * F + k];
        sm[L::o_w2 + i * L::WS + k] = v2;
        sm[L::o_w6 + i * L::WS + k] = v6;
    }
    for (int t = threadIdx.x; t < F * 16; t += nthreads) {
        const int i = t >> 4, k = t & 15;
        const float v = k < kGeo + A ? lw[LL::w1_off(i, 2 * F + k)] : 0.0f;
        sm[L::o_wg + i * L::GS + k] = v;
    }
    for (int t = threadIdx.x; t < 5 * F; t += nthreads) {
        const int w = t / F, k = t - w * F;
        sm[L::o_vec + t] = lw[w == 0 ? LL::b2_off(k) : (w == 1 ? LL::ln_g + k : (w == 2 ? LL::ln_b + k : (w == 3 ? LL::b6 + k : LL::w7 + k)))];
    }
}

// forward state of a tile as seen by wave w: its 16-feature block of every vector
struct FpSub {
    f32x4 t1, sg1, xhat, m, q, sgq;
    float rstd, s;
};

// y_w (+)= sum over the live input blocks of  W[16 w + i][16 bi + 4 g + r] * V[e][16 bi + 4 g + r]
template <int F, int H, int A, bool kBwd, bool kBlockDiag>
__device__ __forceinline__ f32x4 fp_matvec(const float* mat, const float* V, int w, int i, int g, f32x4 acc) {
    using L = FpLayout<F, H, A, kBwd>;
#pragma unroll
    for (int bi = 0; bi < L::NB; ++bi) {
        if (kBlockDiag && !L::w2_block_live(w, bi)) continue;            // wave-uniform
        const f32x4 a = lds4(mat + (16 * w + i) * L::WS + 16 * bi + 4 * g);
        const f32x4 b = lds4(V + i * L::VS + 16 * bi + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma16(a[r], b[r], acc);
    }
    return acc;
}

// y_w (+)= sum over the live input blocks of  W^T[16 w + i][16 bi + 4 g + r] * V[e][16 bi + 4 g + r]  =  W[16 bi + 4 g + r][16 w + i] * ...
// read column-wise from the un-transposed image (lanes i = 0..15 touch consecutive words: conflict-free)
template <int F, int H, int A, bool kBlockDiag>
__device__ __forceinline__ f32x4 fp_matvec_t(const float* mat, const float* V, int w, int i, int g, f32x4 acc) {
    using L = FpLayout<F, H, A, true>;
#pragma unroll
    for (int bi = 0; bi < L::NB; ++bi) {
        if (kBlockDiag && !L::w2_block_live(bi, w)) continue;            // wave-uniform
        const float* col = mat + (16 * bi + 4 * g) * L::WS + 16 * w + i;
        const f32x4 b = lds4(V + i * L::VS + 16 * bi + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma16(col[r * L::WS], b[r], acc);
    }
    return acc;
}

// the exchange barrier of a tile group: its NB waves must meet.  One wave per group (hidden_nf 16): nobody else to wait for -- a
// wave-level fence instead of stalling on the workgroup's other, unrelated tile groups
template <int NB>
__device__ __forceinline__ void fp_sync() {
    if constexpr (NB == 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// sum over the NB waves' partials of the tile group (fixed order), after the caller's barrier
template <int NB>
__device__ __forceinline__ float fp_total(const float* red, int e) {
    float s = red[e];
#pragma unroll
    for (int w = 1; w < NB; ++w) s += red[w * 16 + e];
    return s;
}

// One tile's forward for wave w of its group.  Barriers: every call site runs the same number for all waves of the workgroup.
// On entry the group's idx / geo rows are in LDS (front done + barrier).  Leaves m in V (published) and returns the state.
template <int F, int H, int A, bool kBwd>
__device__ __forceinline__ void fp_tile_forward(const float* img, float* grp, const float* __restrict__ P, int w, int i, int g, FpSub& o) {
    using L = FpLayout<F, H, A, kBwd>;
    constexpr int FP = Geom<F>::FP, NB = L::NB;
    float* V = grp + L::g_v;
    float* red = grp + L::g_red;
    const int* idx = reinterpret_cast<const int*>(grp + L::g_idx);
    const int rr = idx[i], cc = idx[16 + i];
    f32x4 acc = ldg4(P + (size_t)rr * 2 * FP + 16 * w + 4 * g) + ldg4(P + (size_t)cc * 2 * FP + FP + 16 * w + 4 * g);
    {
        const f32x4 gv = lds4(grp + L::g_geo + i * 16 + 4 * g);
        const f32x4 a = lds4(img + L::o_wg + (16 * w + i) * L::GS + 4 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma16(a[s], gv[s], acc);
    }
    o.t1 = acc;
    o.sg1 = sigmoid4(acc);
    st4(V + i * L::VS + 16 * w + 4 * g, acc * o.sg1);                    // SiLU (:245)
    fp_sync<NB>();
    f32x4 u = fp_matvec<F, H, A, kBwd, true>(img + L::o_w2, V, w, i, g, lds4(img + L::o_vec + 0 * F + 16 * w + 4 * g));
    // LayerNorm over the F features (biased variance, eps 1e-5; :249): mean first, then the centred second moment
    {
        const float p = gsum(hsum4(u));
        if (g == 0) red[w * 16 + i] = p;
    }
    fp_sync<NB>();
    const float mean = fp_total<NB>(red, i) * (1.0f / F);
    const f32x4 c = u - splat4(mean);
    {
        const float p = gsum(hsum4(c * c));
        if (g == 0) red[NB * 16 + w * 16 + i] = p;
    }
    fp_sync<NB>();
    const float var = fp_total<NB>(red + NB * 16, i) * (1.0f / F);
    o.rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
    o.xhat = c * splat4(o.rstd);
    o.m = fma4(o.xhat, lds4(img + L::o_vec + 1 * F + 16 * w + 4 * g), lds4(img + L::o_vec + 2 * F + 16 * w + 4 * g));
    st4(V + i * L::VS + 16 * w + 4 * g, o.m);                            // every wave is past its W2 product (two barriers ago)
    fp_sync<NB>();
    o.q = fp_matvec<F, H, A, kBwd, false>(img + L::o_w6, V, w, i, g, lds4(img + L::o_vec + 3 * F + 16 * w + 4 * g));
    o.sgq = sigmoid4(o.q);
    {
        const float p = gsum(hsum4(lds4(img + L::o_vec + 4 * F + 16 * w + 4 * g) * (o.q * o.sgq)));
        if (g == 0) red[w * 16 + i] = p;                                 // the mean's partials were consumed before the second barrier
    }
    fp_sync<NB>();
    o.s = fp_total<NB>(red, i);                                          // :264
}

template <int F, int H, int A>
__global__ __launch_bounds__(fp_threads(F, false)) void edge_fwd_fp_kernel(
    const float* __restrict__ lw, const int* __restrict__ row, const int* __restrict__ col, const float* __restrict__ edge_attr,
    const float* __restrict__ x, const float* __restrict__ P, const int* __restrict__ rinv, int n_edges, int basis,
    float* __restrict__ m_out, float* __restrict__ trans_out) {
    using L = FpLayout<F, H, A, false>;
    constexpr int FP = Geom<F>::FP, NB = L::NB;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tg = wave / NB, w = wave - tg * NB, i = lane & 15, g = lane >> 4;
    fp_stage_image<F, H, A, false>(sm, lw, 64 * L::NW);
    float* grp = sm + L::image + tg * L::group;
    int* idx = reinterpret_cast<int*>(grp + L::g_idx);
    const int n_tiles = (n_edges + 15) >> 4;
    const int stride = gridDim.x * L::kTpb;
    const int iters = (n_tiles + stride - 1) / stride;                   // the same trip count for every tile group: barriers inside
    const bool stream_out = stream_out_edges(n_edges);                   // small graphs: outputs written through (OutRows, egnn_mfma16.h)
    const OutRows m_rows(m_out), trans_rows(trans_out);
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        const int tile = it * stride + blockIdx.x * L::kTpb + tg;
        const int e = tile * 16 + lane;                                  // front lanes: wave w == 0, lanes 0..15
        const bool front = w == 0 && lane < 16;
        const bool live = tile < n_tiles && e < n_edges;
        float d[3] = {0.0f, 0.0f, 0.0f};
        int at = 0;
        if (front) {
            const int ee = live ? e : 0;
            const int r = row[ee], c = col[ee];
            at = rinv[ee];
            float xr[3], xc[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { xr[k] = x[(size_t)r * 3 + k]; xc[k] = x[(size_t)c * 3 + k]; }
            EdgeGeo eg;
            edge_geometry_mode(basis, xr, xc, eg);
            const float attr = A ? (edge_attr ? edge_attr[ee] : 1.0f) : 0.0f;
            store_geo_row(grp + L::g_geo + lane * 16, eg, attr);
            idx[lane] = r; idx[16 + lane] = c; idx[32 + lane] = live ? at : -1;
#pragma unroll
            for (int k = 0; k < 3; ++k) d[k] = eg.d[k];
        }
        fp_sync<NB>();
        FpSub o;
        fp_tile_forward<F, H, A, false>(sm, grp, P, w, i, g, o);
        const int pos = idx[32 + i];
        if (pos >= 0) m_rows.st4((size_t)pos * FP + 16 * w + 4 * g, o.m, stream_out);            // at the edge's aggregation position
        if (front && live) trans_rows.st4(4 * (size_t)at, f32x4{d[0] * o.s, d[1] * o.s, d[2] * o.s, 0.0f}, stream_out);      // trans = coord_diff * s (:264)
        fp_sync<NB>();                                                 // the next tile's front overwrites idx / geo / red
    }
}

template <int F, int H, int A>
static inline size_t edge_fwd_fp_smem() { return sizeof(float) * (size_t)FpLayout<F, H, A, false>::total; }

// acc[by] += X_w^T Y_by over the 16 edges of the tiles (K step s, lane group g <-> edge s + 4 g, as pass16): rows 16 w .. of the
// outer product.  kBlockDiag: only the blocks a head of W2 touches.
template <int F, int H, int A, int NBY, bool kBlockDiag>
__device__ __forceinline__ void fp_outer(const float* X, const float* Y, int ldy, int w, int i, int g, f32x4 (&acc)[NBY]) {
    using L = FpLayout<F, H, A, true>;
    float a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = X[(s + 4 * g) * L::VS + 16 * w + i];
#pragma unroll
    for (int by = 0; by < NBY; ++by) {
        if (kBlockDiag && !L::w2_block_live(w, by)) continue;
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[by] = mfma16(a[s], Y[(s + 4 * g) * ldy + 16 * by + i], acc[by]);
    }
}

template <int F, int H, int A>
__global__ __launch_bounds__(fp_threads(F, true)) void edge_bwd_fp_kernel(
    const float* __restrict__ lw, const int* __restrict__ row, const int* __restrict__ col, const float* __restrict__ edge_attr,
    const float* __restrict__ x, const float* __restrict__ P, const float* __restrict__ dagg, const float* __restrict__ Gx, int n_edges,
    int basis, float* __restrict__ dt1_out, float* __restrict__ gxr_out, float* __restrict__ gxc_out, float* __restrict__ slab) {
    using L = FpLayout<F, H, A, true>;
    using SB = Slabs<F>;
    constexpr int FP = Geom<F>::FP, NB = L::NB, FH = F / H;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tg = wave / NB, w = wave - tg * NB, i = lane & 15, g = lane >> 4;
    fp_stage_image<F, H, A, true>(sm, lw, 64 * L::NW);
    float* grp = sm + L::image + tg * L::group;
    float* V = grp + L::g_v;
    float* X = grp + L::g_x;
    float* Z = grp + L::g_z;
    float* geo = grp + L::g_geo;
    float* dgeo = grp + L::g_dgeo;
    float* red = grp + L::g_red;
    int* idx = reinterpret_cast<int*>(grp + L::g_idx);
    float* dsb = grp + L::g_idx + 32;                                    // dL/ds of the 16 edges (0 for padding edges)
    float* fr = grp + L::g_front + (lane & 15) * 12;
    f32x4 acc6[NB], acc2[NB], accg[1];
    f32x4 ls_w7 = splat4(0.0f), ls_gam = splat4(0.0f), ls_bet = splat4(0.0f), ls_b6 = splat4(0.0f), ls_b2 = splat4(0.0f);
#pragma unroll
    for (int b = 0; b < NB; ++b) { acc6[b] = splat4(0.0f); acc2[b] = splat4(0.0f); }
    accg[0] = splat4(0.0f);
    const int n_tiles = (n_edges + 15) >> 4;
    const int stride = gridDim.x * L::kTpb;
    const int iters = (n_tiles + stride - 1) / stride;
    const bool stream_out = stream_out_edges(n_edges);
    const OutRows dt1_rows(dt1_out), gxr_rows(gxr_out), gxc_rows(gxc_out), slab_rows(slab);
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        const int tile = it * stride + blockIdx.x * L::kTpb + tg;
        const int e = tile * 16 + lane;
        const bool front = w == 0 && lane < 16;
        const bool live = tile < n_tiles && e < n_edges;
        if (front) {
            const int ee = live ? e : 0;
            const int r = row[ee], c = col[ee];
            float xr[3], xc[3], gxv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { xr[k] = x[(size_t)r * 3 + k]; xc[k] = x[(size_t)c * 3 + k]; gxv[k] = Gx[(size_t)r * 4 + k]; }
            EdgeGeo eg;
            edge_geometry_mode(basis, xr, xc, eg);
            const float attr = A ? (edge_attr ? edge_attr[ee] : 1.0f) : 0.0f;
            store_geo_row(geo + lane * 16, eg, attr);
            idx[lane] = r; idx[16 + lane] = c;
            dsb[lane] = live ? dot3(eg.d, gxv) : 0.0f;                   // dL/ds of the coord head (:264); 0 masks padding edges
            st4(fr, f32x4{xr[0], xr[1], xr[2], xc[0]});
            st4(fr + 4, f32x4{xc[1], xc[2], gxv[0], gxv[1]});
            fr[8] = gxv[2];
        }
        fp_sync<NB>();
        const bool valid = tile < n_tiles && tile * 16 + i < n_edges;    // this lane's edge (i) exists
        const int eg_ = tile * 16 + i;
        FpSub o;
        fp_tile_forward<F, H, A, true>(sm, grp, P, w, i, g, o);          // recompute; leaves m in V
        // ---- coord head backward: s = w7 . SiLU(q)
        const float ds = dsb[i];
        ls_w7 = fma4(o.q * o.sgq, splat4(ds), ls_w7);
        const f32x4 dq = lds4(sm + L::o_vec + 4 * F + 16 * w + 4 * g) * splat4(ds) * silu_grad4(o.q, o.sgq);
        ls_b6 += dq;
        st4(X + i * L::VS + 16 * w + 4 * g, dq);
        const f32x4 da = valid ? ldg4(dagg + (size_t)idx[i] * FP + 16 * w + 4 * g) : splat4(0.0f);
        fp_sync<NB>();
        fp_outer<F, H, A, NB, false>(X, V, L::VS, w, i, g, acc6);        // dW6 rows += dq^T m
        // ---- dm = dagg[row] + W6^T dq
        const f32x4 dm = fp_matvec_t<F, H, A, false>(sm + L::o_w6, X, w, i, g, da);
        ls_gam = fma4(dm, o.xhat, ls_gam);
        ls_bet += dm;
        // ---- LayerNorm backward
        f32x4 du = dm * lds4(sm + L::o_vec + 1 * F + 16 * w + 4 * g);
        {
            float m1 = hsum4(du), m2 = hsum4(du * o.xhat);
            gsum2(m1, m2);
            if (g == 0) { red[w * 16 + i] = m1; red[NB * 16 + w * 16 + i] = m2; }
        }
        fp_sync<NB>();                                                 // ... and every wave is past its reads of X (dq) and V (m)
        {
            const float m1 = fp_total<NB>(red, i) * -(1.0f / F), m2 = fp_total<NB>(red + NB * 16, i) * -(1.0f / F);
            du = splat4(o.rstd) * fma4(o.xhat, splat4(m2), du + splat4(m1));
        }
        ls_b2 += du;
        st4(X + i * L::VS + 16 * w + 4 * g, du);
        st4(V + i * L::VS + 16 * w + 4 * g, o.t1 * o.sg1);               // SiLU(t1) again
        fp_sync<NB>();
        fp_outer<F, H, A, NB, true>(X, V, L::VS, w, i, g, acc2);         // dW2 rows += du^T SiLU(t1)
        // ---- dt1 = (W2^T du) * SiLU'(t1)
        f32x4 dt1 = fp_matvec_t<F, H, A, true>(sm + L::o_w2, X, w, i, g, splat4(0.0f));
        dt1 *= silu_grad4(o.t1, o.sg1);
        if (valid) dt1_rows.st4((size_t)eg_ * FP + 16 * w + 4 * g, dt1, stream_out);
        st4(Z + i * L::VS + 16 * w + 4 * g, dt1);
        fp_sync<NB>();
        fp_outer<F, H, A, 1, false>(Z, geo, 16, w, i, g, accg);          // dWgeo | dw_attr rows += dt1^T [geo, attr]
        if (w == 0) {                                                    // dgeo = Wgeo^T dt1: one 16-row block, this wave's job
            f32x4 dg = splat4(0.0f);
#pragma unroll
            for (int bi = 0; bi < NB; ++bi) {
                const float* colg = sm + L::o_wg + (16 * bi + 4 * g) * L::GS + i;          // Wgeo^T[i][16 bi + 4 g + r] = Wgeo[16 bi + 4 g + r][i]
                const f32x4 b = lds4(Z + i * L::VS + 16 * bi + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) dg = mfma16(colg[r * L::GS], b[r], dg);
            }
            st4(dgeo + i * 16 + 4 * g, dg);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- back: one lane per edge, geometry backward from the stashed coordinates
            if (front && live) {
                const f32x4 f0 = lds4(fr), f1 = lds4(fr + 4);
                const float xr2[3] = {f0[0], f0[1], f0[2]}, xc2[3] = {f0[3], f1[0], f1[1]}, gx2[3] = {f1[2], f1[3], fr[8]};
                EdgeGeo eg;
                edge_geometry_mode(basis, xr2, xc2, eg);
                const f32x4 g0 = lds4(dgeo + lane * 16), g1 = lds4(dgeo + lane * 16 + 4), g2 = lds4(dgeo + lane * 16 + 8);
                const float gg[kGeo] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3], g2[0], g2[1], g2[2], g2[3]};
                const float gd[3] = {o.s * gx2[0], o.s * gx2[1], o.s * gx2[2]};
                float gxr[3], gxc[3];
                edge_geometry_backward_mode(basis, xr2, xc2, eg, gg, gd, gxr, gxc);
                gxr_rows.st4(4 * (size_t)e, f32x4{gxr[0], gxr[1], gxr[2], 0.0f}, stream_out);
                gxc_rows.st4(4 * (size_t)e, f32x4{gxc[0], gxc[1], gxc[2], 0.0f}, stream_out);
            }
        }
        fp_sync<NB>();                                                 // the next tile's front overwrites idx / geo / the tiles
    }
    // ---- per-lane running sums: over the 16 edges (lanes) of the group
    auto lane16_sum = [](float v) { return row16_sum(v); };      // over the 16 lanes (edges) of a group
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ls_w7[k] = lane16_sum(ls_w7[k]); ls_gam[k] = lane16_sum(ls_gam[k]); ls_bet[k] = lane16_sum(ls_bet[k]);
        ls_b6[k] = lane16_sum(ls_b6[k]); ls_b2[k] = lane16_sum(ls_b2[k]);
    }
    // ---- the tile groups of the workgroup own the same rows: summed in LDS in fixed order, one slab per workgroup
    __syncthreads();
    float* sl = sm;                                                      // every wave is past its last image read
    for (int t = threadIdx.x; t < SB::edge; t += 64 * L::NW) sl[t] = 0.0f;
    __syncthreads();
    for (int q = 0; q < L::kTpb; ++q) {
        if (tg == q) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int orow = 16 * w + 4 * g + k;
#pragma unroll
                for (int by = 0; by < NB; ++by) {
                    const int ocol = 16 * by + i;
                    sl[SB::e_w6 + orow * F + ocol] += acc6[by][k];
                    if (orow / FH == ocol / FH) sl[SB::e_w2 + orow * F + ocol] += acc2[by][k];      // the dense F x F block stays zero outside the heads' diagonal blocks
                }
                sl[SB::e_geo + orow * 16 + i] += accg[0][k];
                if (i == 0) {
                    sl[SB::e_gam + orow] += ls_gam[k]; sl[SB::e_bet + orow] += ls_bet[k]; sl[SB::e_w7 + orow] += ls_w7[k];
                    sl[SB::e_b6 + orow] += ls_b6[k]; sl[SB::e_b2 + orow] += ls_b2[k];
                }
            }
        }
        __syncthreads();
    }
    for (int t = threadIdx.x; t < SB::edge; t += 64 * L::NW) slab_rows.st1((size_t)blockIdx.x * SB::edge + t, sl[t], stream_out);
}

template <int F, int H, int A>
static inline size_t edge_bwd_fp_smem() {
    const size_t a = (size_t)FpLayout<F, H, A, true>::total, b = (size_t)Slabs<F>::edge;
    return sizeof(float) * (a > b ? a : b);
}

}  // namespace eq
