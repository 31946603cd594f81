// EGNN forward for gfx950: embedding_in -> L x E_GCL -> embedding_out, all clouds of a batch
// in one launch per stage.
//
// Per layer two kernels (reference E_GCL.forward :280-289):
//   edge kernel : a wave per chunk of <= 64 edges, 16-edge sub-tiles on v_mfma_f32_16x16x4_f32 (egnn_mfma16.h).  The
//                 77-wide concat [h_r|h_c|geo|attr] (:238-242) is never built: W1 is split into
//                 [W_src | W_tgt | W_geo | w_attr]; W_src*h and W_tgt*h + b1 are projected ONCE PER NODE (P buffer) by
//                 the previous node kernel, so an edge only adds two gathered rows and a 13-column geometry
//                 product.  Then SiLU, block-diagonal W2, LayerNorm, coord MLP as matrix products with the weights
//                 read from an LDS image.  Writes m_e and trans_e once, at the edge's position in aggregation order.
//   node kernel : G lanes per node (lane = feature).  Sums m_e / trans_e over the node's
//                 incoming-edge list in ascending edge id (atomic-free, bitwise reproducible,
//                 the CPU scatter_add_ order of unsorted_segment_sum :343-348), node MLP with
//                 residual (:252-260), coordinate update (:262-268), and the NEXT layer's
//                 per-node projections (or embedding_out after the last layer).
#include "egnn_mfma.h"
#include "egnn_mfma16.h"
#include "egnn_edge_fp.h"
#include <mutex>


// hidden widths on the feature-parallel MFMA kernels (egnn_edge_fp.h): 16, 48, 64 with 1, 2, 4 or 8 heads (in_edge_nf 1; 0 with one head)
#define EQ_DISPATCH_LANE_WIDTHS \
    EQ_DISPATCH(16, 1, 1) EQ_DISPATCH(16, 1, 0) EQ_DISPATCH(16, 2, 1) EQ_DISPATCH(16, 4, 1) EQ_DISPATCH(16, 8, 1) \
    EQ_DISPATCH(16, 2, 0) EQ_DISPATCH(16, 4, 0) EQ_DISPATCH(16, 8, 0) \
    EQ_DISPATCH(48, 1, 1) EQ_DISPATCH(48, 1, 0) EQ_DISPATCH(48, 2, 1) EQ_DISPATCH(48, 4, 1) EQ_DISPATCH(48, 8, 1) \
    EQ_DISPATCH(48, 2, 0) EQ_DISPATCH(48, 4, 0) EQ_DISPATCH(48, 8, 0) \
    EQ_DISPATCH(64, 1, 1) EQ_DISPATCH(64, 1, 0) EQ_DISPATCH(64, 2, 1) EQ_DISPATCH(64, 4, 1) EQ_DISPATCH(64, 8, 1) \
    EQ_DISPATCH(64, 2, 0) EQ_DISPATCH(64, 4, 0) EQ_DISPATCH(64, 8, 0)

namespace eq {

// ---------------------------------------------------------------------------------------
// embedding_in (+ layer-0 projections).  h0 -> hs0, x -> xs0
// ---------------------------------------------------------------------------------------
template <int F, int H, int A>
__global__ __launch_bounds__(256) void embed_in_kernel(const float* __restrict__ params, int in_nf, int w_in, int b_in,
                                                       int layer0, const float* __restrict__ h_in,
                                                       const float* __restrict__ x_in, int n_nodes,
                                                       float* __restrict__ h0, float* __restrict__ x0,
                                                       float* __restrict__ P) {
    constexpr int FP = Geom<F>::FP, G = Geom<F>::G, NS = Geom<F>::NS, PS = Geom<F>::PS;
    __shared__ float vin[NS][64];
    __shared__ float hv[NS][FP];
    __shared__ float WinT[64 * (F + 1)];
    __shared__ float WaT[F * (F + 1)];
    __shared__ float WbT[F * (F + 1)];
    const int slot = threadIdx.x / G, f = threadIdx.x % G;
    stage_T<F>(WinT, params + w_in, in_nf, in_nf);
    stage_w1_T<F, H, A>(WaT, params + layer0, 0);
    stage_w1_T<F, H, A>(WbT, params + layer0, F);
    const int groups = (n_nodes + NS - 1) / NS;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int j = grp * NS + slot;
        const bool live = j < n_nodes;
        __syncthreads();
        if (live) {
            for (int i = f; i < in_nf; i += G) vin[slot][i] = h_in[(size_t)j * in_nf + i];
            if (f < 3) x0[(size_t)j * 3 + f] = x_in[(size_t)j * 3 + f];
        }
        __syncthreads();
        if (live && f < F) {
            const float v = params[b_in + f] + col_dot<F>(WinT, f, vin[slot], in_nf);
            h0[(size_t)j * FP + f] = v;
            hv[slot][f] = v;
        }
        __syncthreads();
        if (live) project_pre<F, H, A>(params + layer0, WaT, WbT, hv[slot], f, P + (size_t)j * PS);
    }
}

// ---------------------------------------------------------------------------------------
// edge kernel, 16-edge MFMA formulation (hidden_nf == 32; egnn_mfma16.h): a wave per chunk of <= 64 edges.
// Latency chain of a chunk: indices (requested one chunk ahead) -> coordinate gathers + the first sub-tile's projection
// rows together -> geometry -> sub-tiles (each requests the next one's projection rows before it computes).
// ---------------------------------------------------------------------------------------
constexpr int kFwd16Waves = 4;
#ifndef EQ_FWD16_OCC
#define EQ_FWD16_OCC 3                 // 4-wave workgroups per CU (= waves per SIMD) the register budget is set for
#endif

// workgroups per CU the register budget is set for: hidden_nf 33 carries the border scalars of two row sets and spills at 168
#ifndef EQ_FWD16_OCC33
#define EQ_FWD16_OCC33 2
#endif
constexpr int fwd16_occ(int kB) { return kB ? EQ_FWD16_OCC33 : EQ_FWD16_OCC; }

// kBf (experimental, EQ_EDGE_BF16X3 = 1 builds only; egnn_mfma16.h): the W2 / W6 products on the bf16 matrix pipe with 3-way split
// operands; the fragments are staged with the fp32 image (threads 0..127 fetch their eight weights per matrix straight from global
// memory: no second barrier).  ONE path for every graph size, so that a cloud's outputs do not depend on what else is in the batch.
template <int H, int A, int kB = 0, bool kBf = W16::use_bf16(false, kB)>
__global__ __launch_bounds__(64 * kFwd16Waves, fwd16_occ(kB)) void edge_fwd16_kernel(const float* __restrict__ lw, const int* __restrict__ row,
                                                         const int* __restrict__ col, const float* __restrict__ edge_attr,
                                                         const float* __restrict__ x, const float* __restrict__ P,
                                                         const int* __restrict__ rinv, int n_edges, int chunk, int basis,
                                                         float* __restrict__ m_out, float* __restrict__ trans_out) {
    using C = Chunk16;
    const bool stream_out = stream_out_edges(n_edges);                  // write the outputs through (small graphs: egnn_mfma16.h)
    const OutRows m_rows(m_out), trans_rows(trans_out);
    constexpr int FP = 32 + 4 * kB;                    // hidden_nf = 32 + kB, rows padded to 16 bytes
    __shared__ __attribute__((aligned(16))) float wsm_[kBf ? W16::image_size(false, kB) : W16::image_size_fp32(false, kB)];
    __shared__ __attribute__((aligned(16))) float scratch[kFwd16Waves][C::fwd_size];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int wave = blockIdx.x * kFwd16Waves + wv, n_waves = gridDim.x * kFwd16Waves;
    EQ_STAMP(0);
    float* sc = scratch[wv];
    int* idx_r = reinterpret_cast<int*>(sc + C::o_r);
    int* idx_c = reinterpret_cast<int*>(sc + C::o_c);
    int* pos = reinterpret_cast<int*>(sc + C::o_a);
    float* s_buf = sc + C::o_b;
    // A wave's unit of work: <= 4 sixteen-edge tiles behind one front (one lane per edge) -- `chunk` / 16 consecutive tiles, or,
    // chunk == 16 (small graphs: one or two tiles per wave in all), the tiles ch, ch + S, ch + 2 S, ch + 3 S the wave would have
    // taken one by one (S = the number of waves), so that it pays the front once (same scheme as the backward, egnn_bwd.hip).
    const int n_tiles = (n_edges + 15) >> 4;
    const bool strided = chunk != 64;       // every size below the 64-edge regime: tile-granular shares (+- one tile per wave)
    const int tpc = chunk >> 4;
    const int n_chunks = strided ? n_tiles : (n_edges + chunk - 1) / chunk;
    // the four waves of a workgroup take four consecutive chunks (shared centre nodes: their rows hit in L1/L2)
    const int ch_step = n_waves;
    const int ts = strided ? ch_step : 1;
    const int unit_step = strided ? 4 * ch_step : ch_step;
    int ch = wave;
    auto edge_id = [&](int ch_) { return ((strided ? ch_ : ch_ * tpc) + (lane >> 4) * ts) * 16 + (lane & 15); };     // this lane's edge of unit ch_ (>= n_edges: none)
    auto lane_live = [&](int e_) { return e_ < n_edges && (strided || lane < chunk); };
    auto edge_of = [&](int ch_) { const int e_ = edge_id(ch_); return lane_live(e_) ? e_ : 0; };
    // Start-up order (a wave lives for one to three tiles on a small graph): the first unit's indices are requested BEFORE the weight
    // image, its coordinates between the image's loads and its LDS stores -- index -> coordinate -> geometry and global -> LDS
    // run side by side instead of one after the other.
    int e = ch < n_chunks ? edge_of(ch) : 0;
    int r = row[e], c = col[e], at = rinv[e];
    StageW16<H, A, false, 64 * kFwd16Waves, kB, kBf> stage;
    stage.load(lw);
    float xr0[3], xc0[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { xr0[k] = x[(size_t)r * 3 + k]; xc0[k] = x[(size_t)c * 3 + k]; }
    stage.store(wsm_);
    __syncthreads();
    EQ_STAMP(1);
    if (ch >= n_chunks) return;
    bool first = true;
    for (; ch < n_chunks; ch += unit_step) {
        int off = 0;
        asm volatile("" : "+v"(off));                  // an opaque zero offset keeps the (loop-invariant) weight reads inside the loop
        const float* wsm = wsm_ + off;
        const float* bord = wsm + W16::border_base(false);
        const int t0 = strided ? ch : ch * tpc;
        const int n_sub = strided ? min(4, (n_tiles - t0 + ts - 1) / ts) : min(tpc, n_tiles - t0);
        const bool mine = lane_live(edge_id(ch));
        EQ_STAMP(2);
        // ---- front: one lane per edge
        idx_r[lane] = r; idx_c[lane] = c; pos[lane] = at;
        float xr[3], xc[3];
        if (first) {                                   // gathered during the start-up above
#pragma unroll
            for (int k = 0; k < 3; ++k) { xr[k] = xr0[k]; xc[k] = xc0[k]; }
            first = false;
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) { xr[k] = x[(size_t)r * 3 + k]; xc[k] = x[(size_t)c * 3 + k]; }
        }
        const float attr = A ? (edge_attr ? edge_attr[e] : 1.0f) : 0.0f;
        __builtin_amdgcn_wave_barrier();
        PRowsT<kB> pw;                                 // first sub-tile's projection rows: in flight with the coordinate gathers
        pw.load(P, idx_r[i], idx_c[i], g);
        const int chn = ch + unit_step;                // the next unit's indices
        int rn = r, cn = c, atn = at, en = e;
        if (chn < n_chunks) { en = edge_of(chn); rn = row[en]; cn = col[en]; atn = rinv[en]; }
        EdgeGeo eg;
        edge_geometry_mode(basis, xr, xc, eg);
        store_geo_row(sc + C::o_geo + lane * C::GS, eg, attr);
        __builtin_amdgcn_wave_barrier();
        EQ_STAMP(3);
        // ---- the chunk's 16-edge sub-tiles
        for (int sub = 0; sub < n_sub; ++sub) {
            const int el = sub * 16 + i;
            PRowsT<kB> pn;                             // next sub-tile's rows (the last one re-requests its own: harmless)
            {
                const int eln = (sub + 1 < n_sub ? sub + 1 : sub) * 16 + i;
                pn.load(P, idx_r[eln], idx_c[eln], g);
            }
            Sub16 o;
            tile_forward16<kB, false, kBf>(wsm, bord, sc + C::o_geo + el * C::GS, pw, i, g, o);
            if ((t0 + sub * ts) * 16 + i < n_edges) {
                // stored at the edge's position in the by-row (aggregation) order: node_fwd then streams its segment contiguously
                // 128-byte rows for every hidden_nf: the 33rd feature of m leaves as the 4th component of the edge's `trans` item (back
                // phase) -- 144-byte rows straddled two cache lines in every read of the node kernel (66 us against 47 at width 32)
                const size_t dst = (size_t)pos[el] * 32 + 4 * g;
                m_rows.st4(dst, o.m[0], stream_out);
                m_rows.st4(dst + 16, o.m[1], stream_out);
                if (g == 0) {
                    s_buf[el] = o.s;
                    if (kB) reinterpret_cast<float*>(pos)[el] = o.mb;      // the edge's position word is consumed (every lane group read it above, in the same instruction)
                }
            }
            pw = pn;
            EQ_STAMP(10);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- back: one lane per edge, trans = coord_diff * s   (:264)
        if (mine) {
            const float s = s_buf[lane];
            const float mb = kB ? reinterpret_cast<const float*>(pos)[lane] : 0.0f;
            trans_rows.st4(4 * (size_t)at, f32x4{eg.d[0] * s, eg.d[1] * s, eg.d[2] * s, mb}, stream_out);
        }
        __builtin_amdgcn_wave_barrier();               // the next chunk's front overwrites the scratch
        r = rn; c = cn; at = atn; e = en;
        EQ_STAMP(11);
    }
}

template <int F, int H, int A>
static void launch_edge_fwd(const float* lw, const equigspr_graph* g, const float* edge_attr, const float* x, const float* P,
                            float* m, float* trans, int basis, hipStream_t s) {
    const int E = g->n_edges;
    if constexpr (F == 32 || (F == 33 && H == 1)) {       // the 16-edge MFMA formulation (hidden_nf 33: with the border feature); EQ_FWD16_OCC 4-wave workgroups per CU
        constexpr int occ = fwd16_occ(F - 32);
        const int chunk = edge_chunk_for(E, 256 * 4 * occ);
        int blocks = ceil_div(ceil_div(E, chunk), kFwd16Waves);
        const int cap = edge_grid_cap(256 * occ);
        blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
        if (W16::use_bf16(false, F - 32) && !edge_products_fp32())       // split-bf16 products (default) or fp32 (equigspr_set_edge_products)
            hipLaunchKernelGGL((edge_fwd16_kernel<H, A, F - 32, true>), dim3(blocks), dim3(64 * kFwd16Waves), 0, s, lw, g->row, g->col, edge_attr, x, P,
                               g->rinv, E, chunk, basis, m, trans);
        else
            hipLaunchKernelGGL((edge_fwd16_kernel<H, A, F - 32, false>), dim3(blocks), dim3(64 * kFwd16Waves), 0, s, lw, g->row, g->col, edge_attr, x, P,
                               g->rinv, E, chunk, basis, m, trans);
    } else if constexpr (F % 16 == 0) {  // 16 / 48 / 64: the feature-parallel MFMA kernels (egnn_edge_fp.h)
        using L = FpLayout<F, H, A, false>;
        const size_t smem = edge_fwd_fp_smem<F, H, A>();
        {   // more than 64 KB of dynamic LDS needs the attribute (per device and instantiation)
            static std::mutex mu;
            static bool done[64] = {false};
            int dev = 0;
            (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> lk(mu);
            if (dev < 0 || dev >= 64 || !done[dev]) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_fwd_fp_kernel<F, H, A>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                if (dev >= 0 && dev < 64) done[dev] = true;
            }
        }
        const int tiles = ceil_div(E, 16);
        int blocks = ceil_div(tiles, L::kTpb);
        blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
        hipLaunchKernelGGL((edge_fwd_fp_kernel<F, H, A>), dim3(blocks), dim3(64 * L::NW), smem, s, lw, g->row, g->col, edge_attr, x, P, g->rinv, E,
                           basis, m, trans);
    } else {
        static_assert(F % 16 == 0, "hidden widths: 32 / 33 (16-edge MFMA kernels) and 16 / 48 / 64 (feature-parallel MFMA kernels)");
    }
}

// ---------------------------------------------------------------------------------------
// node kernel
// ---------------------------------------------------------------------------------------
template <int F, int H, int A>
__global__ __launch_bounds__(256, 3) void node_fwd_kernel(
    const float* __restrict__ lw,        // this layer's block
    const float* __restrict__ next_lw,   // next layer's block (projections) or nullptr on the last layer
    const float* __restrict__ w_out, const float* __restrict__ b_out, int out_nf,   // embedding_out (last layer)
    const int* __restrict__ rptr, const int* __restrict__ rperm, const float* __restrict__ m_in,
    const float* __restrict__ trans_in, const float* __restrict__ h, const float* __restrict__ x, int n_nodes,
    float* __restrict__ h_next, float* __restrict__ x_next, float* __restrict__ agg_save,
    float* __restrict__ P, float* __restrict__ h_out) {
    using LL = LP<F, H, A>;
    constexpr int FP = Geom<F>::FP, G = Geom<F>::G, NS = Geom<F>::NS, PS = Geom<F>::PS;
    __shared__ float vin[NS][2 * FP];
    __shared__ float hid[NS][FP];
    __shared__ float hnew[NS][FP];
    __shared__ float W4T[2 * F * (F + 1)];
    __shared__ float W5T[F * (F + 1)];
    __shared__ float WaT[64 * (F + 1) > F * 65 ? 64 * (F + 1) : F * 65];   // next layer's W_src^T, or W_out^T [F][out_nf+1 <= 65]
    __shared__ float WbT[F * (F + 1)];
    const int slot = threadIdx.x / G, f = threadIdx.x % G;
    EQ_STAMPK(1, 0);
    {   // all matrices' loads in flight together, then the LDS stores
        const MapT<F> m4{2 * F, 2 * F}, m5{F, F};
        const MapW1T<F, H, A> ma{0}, mb{F};
        Staged<2 * F * F> s4;
        Staged<F * F> s5, sa, sb;
        s4.load(lw + LL::w4, 2 * F * F, [=](int t) { return m4.src(t); });
        s5.load(lw + LL::w5, F * F, [=](int t) { return m5.src(t); });
        if (next_lw) {
            sa.load(next_lw, F * F, [=](int t) { return ma.src(t); });
            sb.load(next_lw, F * F, [=](int t) { return mb.src(t); });
        }
        s4.store(W4T, 2 * F * F, [=](int t) { return m4.dst(t); });
        s5.store(W5T, F * F, [=](int t) { return m5.dst(t); });
        if (next_lw) {
            sa.store(WaT, F * F, [=](int t) { return ma.dst(t); });
            sb.store(WbT, F * F, [=](int t) { return mb.dst(t); });
        }
    }
    if (!next_lw) {
        for (int t = threadIdx.x; t < out_nf * F; t += 256) {       // W_out^T: [i][o], row stride 65
            const int o = t / F, i = t - o * F;
            WaT[i * 65 + o] = w_out[t];
        }
    }
    EQ_STAMPK(1, 1);
    const int groups = (n_nodes + NS - 1) / NS;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int j = grp * NS + slot;
        const bool live = j < n_nodes;
        const bool act = live && f < F;
        float hj = 0.0f, agg = 0.0f, xagg = 0.0f;
        if (live) {
            const int lo = rptr[j], hi = rptr[j + 1];
            if (f < F) hj = h[(size_t)j * FP + f];
            segment_sum_sorted<G>(lo, hi, m_in, FP, trans_in, f, f < F, agg, xagg);         // rows lo..hi-1 = ascending edge id
            if (f < 3) x_next[(size_t)j * 3 + f] = x[(size_t)j * 3 + f] + xagg;              // :267
        }
        EQ_STAMPK(1, 2);
        __syncthreads();
        if (act) {
            vin[slot][f] = hj;
            vin[slot][F + f] = agg;
            if (agg_save) agg_save[(size_t)j * FP + f] = agg;
        }
        __syncthreads();
        if (act) hid[slot][f] = silu(lw[LL::b4 + f] + col_dot<F>(W4T, f, vin[slot], 2 * F));
        __syncthreads();
        if (act) {
            const float hn = hj + lw[LL::b5 + f] + col_dot<F>(W5T, f, hid[slot], F);   // :257-259
            h_next[(size_t)j * FP + f] = hn;
            hnew[slot][f] = hn;
        }
        __syncthreads();
        EQ_STAMPK(1, 3);
        if (live) {
            if (next_lw) {
                project_pre<F, H, A>(next_lw, WaT, WbT, hnew[slot], f, P + (size_t)j * PS);
            } else {
                for (int o = f; o < out_nf; o += G) {
                    float a = b_out[o];
                    for (int i = 0; i < F; ++i) a = fmaf(WaT[i * 65 + o], hnew[slot][i], a);
                    h_out[(size_t)j * out_nf + o] = a;                                                    // :337
                }
            }
        }
        EQ_STAMPK(1, 4);
    }
}

// ---------------------------------------------------------------------------------------
// node kernel, 16-node MFMA formulation (hidden_nf == 32): ONE wave per 16 nodes, no LDS, no barriers.
// Lane l = (node n = l & 15, g = l >> 4) owns features 16 b + 4 g + r of its node (the C/D layout of v_mfma_f32_16x16x4_f32 and,
// as in egnn_mfma16.h, the B operand of the next product).  The segment sums stream the node's rows (stored by the edge kernel in
// aggregation order) as float4 in ascending edge id -- the same order of additions as the lane-per-feature kernel.  The node
// MLP (:252-260), the next layer's projections and embedding_out are 32 + 16 + 32 (or <= 64) MFMAs whose A operands are read
// straight from the parameter block (L2-resident, 16-byte rows) -- the lane-per-feature kernel spent a third of its time
// staging four matrices into LDS per 8 nodes and the rest in barrier-separated LDS dot products.
// ---------------------------------------------------------------------------------------
#ifndef EQ_NODE_FWD16
#define EQ_NODE_FWD16 1
#endif
#ifndef EQ_NODE_FWD16_NPT_SMALL
#define EQ_NODE_FWD16_NPT_SMALL 16     // nodes per wave below EQ_NODE_FWD16_SMALL nodes (4: four copies share a node's segment -- another summation order; timing knob)
#define EQ_NODE_FWD16_D_SMALL 16
#endif
#ifndef EQ_NODE_FWD16_D_LARGE
#define EQ_NODE_FWD16_D_LARGE 4          // segment rows in flight per lane above EQ_NODE_FWD16_SMALL nodes
#endif
#ifndef EQ_NODE_FWD16_SMALL
#define EQ_NODE_FWD16_SMALL 16384
#endif
// four consecutive matrix entries W[row][col .. col + 3] as an MFMA A-operand quad: one 16-byte load when the row stride keeps
// them aligned (hidden_nf 32), four 4-byte loads otherwise (hidden_nf 33: 33- / 66- / 79-float rows)
template <bool kAligned>
__device__ __forceinline__ f32x4 wquad(const float* __restrict__ w, int row, int ld, int col) {
    if constexpr (kAligned) return ldg4(w + (size_t)row * ld + col);
    const float* p = w + (size_t)row * ld + col;
    return f32x4{p[0], p[1], p[2], p[3]};
}

// NPT = nodes per wave: 16, or 4 when there are too few nodes to occupy the chip -- the 16 columns then hold every node four
// times and the copies share out the node's segment (copy p takes rows p, p + 4, ...), so a k = 20 neighbourhood is ONE
// round trip of 5 rows per lane and four times as many waves are in flight.
template <int H, int A, int D, int kB = 0, int NPT = 16>      // D = segment rows in flight per lane
__global__ __launch_bounds__(64) void node_fwd16_kernel(
    const float* __restrict__ lw, const float* __restrict__ next_lw, const float* __restrict__ w_out, const float* __restrict__ b_out,
    int out_nf, const int* __restrict__ rptr, const float* __restrict__ m_in, const float* __restrict__ trans_in,
    const float* __restrict__ h, const float* __restrict__ x, int n_nodes, float* __restrict__ h_next, float* __restrict__ x_next,
    float* __restrict__ agg_save, float* __restrict__ P, float* __restrict__ h_out) {
    constexpr int F = 32 + kB, FP = 32 + 4 * kB;      // hidden_nf 33: feature 32 is a scalar per node held by its four lanes (the "border")
    using LL = LP<F, H, A>;
    constexpr bool kAl = (kB == 0);
    constexpr int NP = 16 / NPT;
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4, part = i / NPT;
    const int j = min((int)blockIdx.x * NPT + (i % NPT), n_nodes - 1);
    const bool live = (int)blockIdx.x * NPT + (i % NPT) < n_nodes && i < NPT;       // the copy that stores
    const int lo = rptr[j], hi = rptr[j + 1];
    // weights of the node MLP: requested before the segment loop, consumed after it.  W4 = [F][2F] over [h | agg]
    f32x4 w4a[2][4], w5a[2][2], b4v[2], b5v[2];
#pragma unroll
    for (int bo = 0; bo < 2; ++bo) {
#pragma unroll
        for (int bi = 0; bi < 4; ++bi) w4a[bo][bi] = wquad<kAl>(lw + LL::w4, 16 * bo + i, 2 * F, (bi < 2 ? 0 : F) + 16 * (bi & 1) + 4 * g);
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) w5a[bo][bi] = wquad<kAl>(lw + LL::w5, 16 * bo + i, F, 16 * bi + 4 * g);
        b4v[bo] = wquad<kAl>(lw + LL::b4, 0, 0, 16 * bo + 4 * g);
        b5v[bo] = wquad<kAl>(lw + LL::b5, 0, 0, 16 * bo + 4 * g);
    }
    // next layer's projection weights (W1 rows are 2F + 13 floats apart: 4-byte aligned only -> scalar loads), also requested up front
    float as[2][2][4], at[2][2][4];
    f32x4 pcb[2];
    if (next_lw) {
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bo = 0; bo < 2; ++bo)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    as[bi][bo][r] = next_lw[LL::w1_off(16 * bo + i, 16 * bi + 4 * g + r)];
                    at[bi][bo][r] = next_lw[LL::w1_off(16 * bo + i, F + 16 * bi + 4 * g + r)];
                }
#pragma unroll
        for (int bo = 0; bo < 2; ++bo)
#pragma unroll
            for (int r = 0; r < 4; ++r) pcb[bo][r] = next_lw[LL::b1_off(16 * bo + 4 * g + r)];
    }
    f32x4 hj[2], agg[2] = {splat4(0.0f), splat4(0.0f)};
#pragma unroll
    for (int b = 0; b < 2; ++b) hj[b] = ldg4(h + (size_t)j * FP + 16 * b + 4 * g);
    float hjb = kB ? h[(size_t)j * FP + 32] : 0.0f, aggb = 0.0f;
    float4 xs = make_float4(0.f, 0.f, 0.f, 0.f);
    // ---- segment sums over rows lo..hi-1 (ascending edge id)
    const int len = hi - lo;
    int maxlen = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
    for (int a0 = 0; a0 < maxlen; a0 += D * NP) {
        f32x4 v[D][2];
        float4 tv[D];
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int aa = a0 + part + u * NP;
            const bool in = aa < len;
            const size_t a = (size_t)(lo + (in ? aa : 0));
#pragma unroll
            for (int b = 0; b < 2; ++b) v[u][b] = in ? ldg4(m_in + a * 32 + 16 * b + 4 * g) : splat4(0.0f);      // [E][32] rows; the 33rd feature is tv.w
            tv[u] = (in && g == 0) ? reinterpret_cast<const float4*>(trans_in)[a] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < D; ++u) {
            agg[0] += v[u][0]; agg[1] += v[u][1];
            xs.x += tv[u].x; xs.y += tv[u].y; xs.z += tv[u].z;
            if (kB) xs.w += tv[u].w;                   // lane group 0 sums the 33rd feature of m
        }
    }
    if constexpr (NP > 1) {                      // the copies of a node add up their parts (every copy gets the same sums)
#pragma unroll
        for (int o = NPT; o < 16; o <<= 1) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) agg[b][r] += __shfl_xor(agg[b][r], o, 64);
            if (kB) xs.w += __shfl_xor(xs.w, o, 64);
            xs.x += __shfl_xor(xs.x, o, 64); xs.y += __shfl_xor(xs.y, o, 64); xs.z += __shfl_xor(xs.z, o, 64);
        }
    }
    if constexpr (kB == 1) aggb = __shfl(xs.w, i, 64);      // ... and hands the sum to the node's other three lanes
    if (live && g == 0) {                                                             // :267
        x_next[(size_t)j * 3 + 0] = x[(size_t)j * 3 + 0] + xs.x;
        x_next[(size_t)j * 3 + 1] = x[(size_t)j * 3 + 1] + xs.y;
        x_next[(size_t)j * 3 + 2] = x[(size_t)j * 3 + 2] + xs.z;
    }
    if (live && agg_save) {
        st4(agg_save + (size_t)j * FP + 4 * g, agg[0]);
        st4(agg_save + (size_t)j * FP + 16 + 4 * g, agg[1]);
        if (kB && g == 0) st4(agg_save + (size_t)j * FP + 32, f32x4{aggb, 0.0f, 0.0f, 0.0f});
    }
    // border helpers: column c of a matrix adds W[o][c] * vb to every output o; row 32 is a per-lane partial dot + lane-group sum
    auto add_col = [&](f32x4 (&acc)[2], const float* __restrict__ w, int ld, int c, float vb) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[b][r] = fmaf(w[(size_t)(16 * b + 4 * g + r) * ld + c], vb, acc[b][r]);
    };
    auto row_dot = [&](const float* __restrict__ wrow, int c0, const f32x4 (&v)[2]) {
        float p = 0.0f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) p = fmaf(wrow[c0 + 16 * b + 4 * g + r], v[b][r], p);
        return p;
    };
    // ---- hid = SiLU(W4 [h | agg] + b4)
    f32x4 acc[2] = {b4v[0], b4v[1]};
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
        const f32x4 bv = bi < 2 ? hj[bi] : agg[bi - 2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[0] = mfma16(w4a[0][bi][r], bv[r], acc[0]);
            acc[1] = mfma16(w4a[1][bi][r], bv[r], acc[1]);
        }
    }
    float vb4 = 0.0f;
    if constexpr (kB == 1) {
        add_col(acc, lw + LL::w4, 2 * F, 32, hjb);
        add_col(acc, lw + LL::w4, 2 * F, F + 32, aggb);
        const float* w32 = lw + LL::w4 + (size_t)32 * 2 * F;
        vb4 = gsum(row_dot(w32, 0, hj) + row_dot(w32, F, agg)) + w32[32] * hjb + w32[F + 32] * aggb + lw[LL::b4 + 32];
    }
    f32x4 hid[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) hid[b] = acc[b] * sigmoid4(acc[b]);
    const float hidb = kB ? vb4 * sigmoid_fast(vb4) : 0.0f;
    // ---- h' = h + W5 hid + b5   (:257-259)
    f32x4 hn[2] = {hj[0] + b5v[0], hj[1] + b5v[1]};
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hn[0] = mfma16(w5a[0][bi][r], hid[bi][r], hn[0]);
            hn[1] = mfma16(w5a[1][bi][r], hid[bi][r], hn[1]);
        }
    float hnb = 0.0f;
    if constexpr (kB == 1) {
        add_col(hn, lw + LL::w5, F, 32, hidb);
        const float* w32 = lw + LL::w5 + (size_t)32 * F;
        hnb = hjb + lw[LL::b5 + 32] + gsum(row_dot(w32, 0, hid)) + w32[32] * hidb;
    }
    if (live) {
        st4(h_next + (size_t)j * FP + 4 * g, hn[0]);
        st4(h_next + (size_t)j * FP + 16 + 4 * g, hn[1]);
        if (kB && g == 0) st4(h_next + (size_t)j * FP + 32, f32x4{hnb, 0.0f, 0.0f, 0.0f});
    }
    if (next_lw) {
        // ---- next layer's projections: P = [W_src h' | W_tgt h' + b1]
        f32x4 pr[2] = {splat4(0.0f), splat4(0.0f)}, pc[2] = {pcb[0], pcb[1]};
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pr[0] = mfma16(as[bi][0][r], hn[bi][r], pr[0]);
                pr[1] = mfma16(as[bi][1][r], hn[bi][r], pr[1]);
                pc[0] = mfma16(at[bi][0][r], hn[bi][r], pc[0]);
                pc[1] = mfma16(at[bi][1][r], hn[bi][r], pc[1]);
            }
        float prb = 0.0f, pcbb = 0.0f;
        if constexpr (kB == 1) {          // H == 1 here: W1 is one [33][79] block
            const float* w1 = next_lw + LL::w1_off(0, 0);
            add_col(pr, w1, LL::D, 32, hnb);
            add_col(pc, w1, LL::D, F + 32, hnb);
            const float* w32 = w1 + (size_t)32 * LL::D;
            prb = gsum(row_dot(w32, 0, hn)) + w32[32] * hnb;
            pcbb = gsum(row_dot(w32, F, hn)) + w32[F + 32] * hnb + next_lw[LL::b1_off(32)];
        }
        if (live) {
            float* prow = P + (size_t)j * 2 * FP + 4 * g;
            st4(prow, pr[0]); st4(prow + 16, pr[1]); st4(prow + FP, pc[0]); st4(prow + FP + 16, pc[1]);
            if (kB && g == 0) {
                st4(P + (size_t)j * 2 * FP + 32, f32x4{prb, 0.0f, 0.0f, 0.0f});
                st4(P + (size_t)j * 2 * FP + FP + 32, f32x4{pcbb, 0.0f, 0.0f, 0.0f});
            }
        }
    } else {
        // ---- embedding_out (:337): out_nf <= 64 rows of W_out [out_nf][F]
        for (int bo = 0; bo * 16 < out_nf; ++bo) {
            const int orow = 16 * bo + i;
            f32x4 o4;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int oo = 16 * bo + 4 * g + r; o4[r] = oo < out_nf ? b_out[oo] : 0.0f; }
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {
                const f32x4 a = orow < out_nf ? wquad<kAl>(w_out, orow, F, 16 * bi + 4 * g) : splat4(0.0f);
#pragma unroll
                for (int r = 0; r < 4; ++r) o4 = mfma16(a[r], hn[bi][r], o4);
            }
            if constexpr (kB == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int oo = 16 * bo + 4 * g + r; if (oo < out_nf) o4[r] = fmaf(w_out[(size_t)oo * F + 32], hnb, o4[r]); }
            }
            if (live) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int oo = 16 * bo + 4 * g + r; if (oo < out_nf) h_out[(size_t)j * out_nf + oo] = o4[r]; }
            }
        }
    }
}

// embedding_in in the same form (in_nf == hidden_nf, 32 or 33): h_0 = W_in h_in + b_in, x_0 = x_in, P_0 = [W_src h_0 | W_tgt h_0 + b1]
// for the first layer's edge kernel; one wave per 16 nodes, 16 + 32 MFMAs, all weights requested up front.  (The lane-per-feature
// embed_in_kernel remains for other input widths.)
template <int H, int A, int kB>
__global__ __launch_bounds__(64) void embed_in16_kernel(const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                        const float* __restrict__ lw0, const float* __restrict__ h_in,
                                                        const float* __restrict__ x_in, int n_nodes, float* __restrict__ h0,
                                                        float* __restrict__ x0, float* __restrict__ P) {
    constexpr int F = 32 + kB, FP = 32 + 4 * kB;
    using LL = LP<F, H, A>;
    constexpr bool kAl = (kB == 0);
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const int j = min((int)blockIdx.x * 16 + i, n_nodes - 1);
    const bool live = (int)blockIdx.x * 16 + i < n_nodes;
    f32x4 wia[2][2], hin[2], acc[2], pcb[2];
    float as[2][2][4], at[2][2][4];
#pragma unroll
    for (int bo = 0; bo < 2; ++bo) {
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) wia[bo][bi] = wquad<kAl>(w_in, 16 * bo + i, F, 16 * bi + 4 * g);
        acc[bo] = wquad<kAl>(b_in, 0, 0, 16 * bo + 4 * g);
    }
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bo = 0; bo < 2; ++bo)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                as[bi][bo][r] = lw0[LL::w1_off(16 * bo + i, 16 * bi + 4 * g + r)];
                at[bi][bo][r] = lw0[LL::w1_off(16 * bo + i, F + 16 * bi + 4 * g + r)];
            }
#pragma unroll
    for (int bo = 0; bo < 2; ++bo)
#pragma unroll
        for (int r = 0; r < 4; ++r) pcb[bo][r] = lw0[LL::b1_off(16 * bo + 4 * g + r)];
#pragma unroll
    for (int b = 0; b < 2; ++b) hin[b] = wquad<kAl>(h_in, j, F, 16 * b + 4 * g);
    const float hinb = kB ? h_in[(size_t)j * F + 32] : 0.0f;
    if (live && g == 0) {
        x0[(size_t)j * 3 + 0] = x_in[(size_t)j * 3 + 0]; x0[(size_t)j * 3 + 1] = x_in[(size_t)j * 3 + 1]; x0[(size_t)j * 3 + 2] = x_in[(size_t)j * 3 + 2];
    }
    auto add_col = [&](f32x4 (&a_)[2], const float* __restrict__ w, int ld, int c, float vb) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) a_[b][r] = fmaf(w[(size_t)(16 * b + 4 * g + r) * ld + c], vb, a_[b][r]);
    };
    auto row_dot = [&](const float* __restrict__ wrow, int c0, const f32x4 (&v)[2]) {
        float p = 0.0f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) p = fmaf(wrow[c0 + 16 * b + 4 * g + r], v[b][r], p);
        return p;
    };
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[0] = mfma16(wia[0][bi][r], hin[bi][r], acc[0]);
            acc[1] = mfma16(wia[1][bi][r], hin[bi][r], acc[1]);
        }
    float h0b = 0.0f;
    if constexpr (kB == 1) {
        add_col(acc, w_in, F, 32, hinb);
        const float* w32 = w_in + (size_t)32 * F;
        h0b = gsum(row_dot(w32, 0, hin)) + w32[32] * hinb + b_in[32];
    }
    if (live) {
        st4(h0 + (size_t)j * FP + 4 * g, acc[0]);
        st4(h0 + (size_t)j * FP + 16 + 4 * g, acc[1]);
        if (kB && g == 0) st4(h0 + (size_t)j * FP + 32, f32x4{h0b, 0.0f, 0.0f, 0.0f});
    }
    f32x4 pr[2] = {splat4(0.0f), splat4(0.0f)}, pc[2] = {pcb[0], pcb[1]};
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pr[0] = mfma16(as[bi][0][r], acc[bi][r], pr[0]);
            pr[1] = mfma16(as[bi][1][r], acc[bi][r], pr[1]);
            pc[0] = mfma16(at[bi][0][r], acc[bi][r], pc[0]);
            pc[1] = mfma16(at[bi][1][r], acc[bi][r], pc[1]);
        }
    float prb = 0.0f, pcbb = 0.0f;
    if constexpr (kB == 1) {          // H == 1 here: W1 is one [33][79] block
        const float* w1 = lw0 + LL::w1_off(0, 0);
        add_col(pr, w1, LL::D, 32, h0b);
        add_col(pc, w1, LL::D, F + 32, h0b);
        const float* w32 = w1 + (size_t)32 * LL::D;
        prb = gsum(row_dot(w32, 0, acc)) + w32[32] * h0b;
        pcbb = gsum(row_dot(w32, F, acc)) + w32[F + 32] * h0b + lw0[LL::b1_off(32)];
    }
    if (live) {
        float* prow = P + (size_t)j * 2 * FP + 4 * g;
        st4(prow, pr[0]); st4(prow + 16, pr[1]); st4(prow + FP, pc[0]); st4(prow + FP + 16, pc[1]);
        if (kB && g == 0) {
            st4(P + (size_t)j * 2 * FP + 32, f32x4{prb, 0.0f, 0.0f, 0.0f});
            st4(P + (size_t)j * 2 * FP + FP + 32, f32x4{pcbb, 0.0f, 0.0f, 0.0f});
        }
    }
}

// ---------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------
struct FwdBuffers {
    float *P, *m, *trans;        // work
    float *hs, *xs, *aggs;       // saved (or ping-pong in work when not training)
    bool training;
};

static inline int64_t saved_floats(int F, int L, int64_t N) {       // node-level activations only: the edge backward recomputes
    return saved_node_floats(fpad(F), L, N);
}
static inline int64_t work_floats(int F, int64_t N, int64_t E) {
    const int FP = fpad(F);
    // P | m | trans | ping-pong h (2) | ping-pong x (2)
    return N * 2 * FP + E * FP + E * 4 + 2 * N * FP + 2 * N * 3 + 64;
}

// measurement hook: average duration of the layer-0 edge kernel, HIP events on `s`
template <int F, int H, int A>
static int time_edge_fwd_impl(const equigspr_egnn_config* cfg, const float* params, const equigspr_graph* g,
                              const float* h_in, const float* x_in, const float* edge_attr, float* work, int iters,
                              float* ms_out, hipStream_t s) {
    const ModelLayout ML = make_model_layout(cfg->in_nf, F, cfg->out_nf, A, cfg->n_layers, H);
    const int N = g->n_nodes, E = g->n_edges;
    constexpr int FP = Geom<F>::FP, NS = Geom<F>::NS;
    float* P = work;
    float* m = P + (size_t)N * 2 * FP;
    float* trans = m + (size_t)E * FP;
    float* pp_h = trans + (size_t)E * 4;
    float* pp_x = pp_h + (size_t)2 * N * FP;
    const int node_groups = ceil_div(N, NS);
    hipLaunchKernelGGL((embed_in_kernel<F, H, A>), dim3(node_groups < 2048 ? node_groups : 2048), dim3(256), 0, s, params,
                       cfg->in_nf, ML.w_in, ML.b_in, ML.layer0, h_in, x_in, N, pp_h, pp_x, P);
    const float* lw = params + ML.layer0;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch_edge_fwd<F, H, A>(lw, g, edge_attr, pp_x, P, m, trans, cfg->edge_basis, s);      // warm
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) launch_edge_fwd<F, H, A>(lw, g, edge_attr, pp_x, P, m, trans, cfg->edge_basis, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_out = ms / iters;
    EQ_CHECK_LAUNCH("time_edge_fwd");
    return 0;
}

template <int F, int H, int A>
static int forward_impl(const equigspr_egnn_config* cfg, const float* params, const equigspr_graph* g,
                        const float* h_in, const float* x_in, const float* edge_attr, float* h_out, float* x_out,
                        float* saved, float* work, hipStream_t s) {
    const ModelLayout ML = make_model_layout(cfg->in_nf, F, cfg->out_nf, A, cfg->n_layers, H);
    const int N = g->n_nodes, E = g->n_edges, L = cfg->n_layers;
    constexpr int FP = Geom<F>::FP, NS = Geom<F>::NS;
    float* P = work;
    float* m = P + (size_t)N * 2 * FP;
    float* trans = m + (size_t)E * FP;
    float* pp_h = trans + (size_t)E * 4;
    float* pp_x = pp_h + (size_t)2 * N * FP;
    const bool training = saved != nullptr;
    float* hs = saved;
    float* xs = training ? saved + (size_t)(L + 1) * N * FP : nullptr;
    float* aggs = training ? xs + (size_t)(L + 1) * N * 3 : nullptr;
    auto hbuf = [&](int l) { return training ? hs + (size_t)l * N * FP : pp_h + (size_t)(l & 1) * N * FP; };
    auto xbuf = [&](int l) { return training ? xs + (size_t)l * N * 3 : pp_x + (size_t)(l & 1) * N * 3; };

    const int node_groups = ceil_div(N, NS);
#ifndef EQ_NODE_FWD_BLOCKS
#define EQ_NODE_FWD_BLOCKS 2048
#endif
    const int node_grid = node_groups < EQ_NODE_FWD_BLOCKS ? node_groups : EQ_NODE_FWD_BLOCKS;
    bool embed16 = false;
    if constexpr (EQ_NODE_FWD16 && (F == 32 || (F == 33 && H == 1))) embed16 = cfg->in_nf == F;
    if (embed16) {
        if constexpr (EQ_NODE_FWD16 && (F == 32 || (F == 33 && H == 1)))
            hipLaunchKernelGGL((embed_in16_kernel<H, A, F - 32>), dim3(ceil_div(N, 16)), dim3(64), 0, s, params + ML.w_in, params + ML.b_in,
                               params + ML.layer0, h_in, x_in, N, hbuf(0), xbuf(0), P);
    } else
    hipLaunchKernelGGL((embed_in_kernel<F, H, A>), dim3(node_grid), dim3(256), 0, s, params, cfg->in_nf, ML.w_in,
                       ML.b_in, ML.layer0, h_in, x_in, N, hbuf(0), xbuf(0), P);
    for (int l = 0; l < L; ++l) {
        const float* lw = params + ML.layer0 + (size_t)l * ML.layer_size;
        const bool last = (l == L - 1);
        if (E > 0) {
            launch_edge_fwd<F, H, A>(lw, g, edge_attr, xbuf(l), P, m, trans, cfg->edge_basis, s);
        }
        float* xn = last ? x_out : xbuf(l + 1);
        static_assert(EQ_NODE_FWD16 || F != 33, "hidden_nf 33: the edge kernel writes 128-byte message rows (33rd feature in trans.w), which only node_fwd16_kernel reads");
        if constexpr (EQ_NODE_FWD16 && (F == 32 || (F == 33 && H == 1))) {
            // few nodes: one wave per CU or less, latency decides -> 16 rows in flight; many nodes: 4 (registers for occupancy)
            // few nodes: one wave per CU or less, latency decides -> 16 rows in flight; many nodes: 4 (registers for occupancy).  Both sum
            // a node's rows in ascending order, so a cloud's outputs do not depend on what else is in the batch
            // (tests/test_gpu_fullsize.py::test_batching_is_exact_fullsize).  The 4-node tiles (NPT = 4: four copies share a segment,
            // 9.0 -> 8.3 us at one pair) add partial sums in another order and are therefore not used.
            if (N <= EQ_NODE_FWD16_SMALL)
                hipLaunchKernelGGL((node_fwd16_kernel<H, A, EQ_NODE_FWD16_D_SMALL, F - 32, EQ_NODE_FWD16_NPT_SMALL>), dim3(ceil_div(N, EQ_NODE_FWD16_NPT_SMALL)), dim3(64), 0, s, lw, last ? nullptr : lw + ML.layer_size,
                                   params + ML.w_out, params + ML.b_out, cfg->out_nf, g->rptr, m, trans, hbuf(l), xbuf(l), N, hbuf(l + 1), xn,
                                   training ? aggs + (size_t)l * N * FP : nullptr, P, h_out);
            else
                hipLaunchKernelGGL((node_fwd16_kernel<H, A, EQ_NODE_FWD16_D_LARGE, F - 32>), dim3(ceil_div(N, 16)), dim3(64), 0, s, lw, last ? nullptr : lw + ML.layer_size,
                                   params + ML.w_out, params + ML.b_out, cfg->out_nf, g->rptr, m, trans, hbuf(l), xbuf(l), N, hbuf(l + 1), xn,
                                   training ? aggs + (size_t)l * N * FP : nullptr, P, h_out);
        } else
        hipLaunchKernelGGL((node_fwd_kernel<F, H, A>), dim3(node_grid), dim3(256), 0, s, lw,
                           last ? nullptr : lw + ML.layer_size, params + ML.w_out, params + ML.b_out, cfg->out_nf,
                           g->rptr, g->rperm, m, trans, hbuf(l), xbuf(l), N, hbuf(l + 1), xn,
                           training ? aggs + (size_t)l * N * FP : nullptr, P, h_out);
        if (last && training) {   // keep x_L with the saved activations too (backward reads xs[L] never; cheap copy skipped)
        }
    }
    EQ_CHECK_LAUNCH("egnn_forward");
    return 0;
}

}  // namespace eq

extern "C" int64_t equigspr_egnn_saved_size(const equigspr_egnn_config* cfg, int32_t n_nodes, int32_t n_edges) {
    if (!eq::config_supported(cfg) || n_nodes < 0) return -1;
    return eq::saved_floats(cfg->hidden_nf, cfg->n_layers, n_nodes);
}

extern "C" int64_t equigspr_egnn_work_size(const equigspr_egnn_config* cfg, int32_t n_nodes, int32_t n_edges) {
    if (!eq::config_supported(cfg) || n_nodes < 0 || n_edges < 0) return -1;
    return eq::work_floats(cfg->hidden_nf, n_nodes, n_edges);
}

extern "C" int equigspr_egnn_forward(const equigspr_egnn_config* cfg, const float* params, const equigspr_graph* g,
                                     const float* h_in, const float* x_in, const float* edge_attr, float* h_out,
                                     float* x_out, float* saved, float* work, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(config_supported(cfg), EQUIGSPR_ENOTSUP,
                 "egnn_forward: unsupported config (hidden_nf in {32,33}, heads in {1,2,4,8} (33: 1), in/out<=64, in_edge_nf in {0,1})");
    EQ_CHECK_ARG(params && g && h_in && x_in && h_out && x_out && work, EQUIGSPR_EINVAL, "egnn_forward: null pointer");
    EQ_CHECK_ARG(g->n_nodes >= 1 && g->n_edges >= 0 && g->rptr && (g->n_edges == 0 || (g->row && g->col && g->rperm && g->rinv)), EQUIGSPR_EINVAL,
                 "egnn_forward: incomplete graph");
    hipStream_t s = (hipStream_t)stream;
    const int F = cfg->hidden_nf, H = cfg->num_heads, A = cfg->in_edge_nf;
#define EQ_DISPATCH(FF, HH, AA) \
    if (F == FF && H == HH && A == AA) return forward_impl<FF, HH, AA>(cfg, params, g, h_in, x_in, edge_attr, h_out, x_out, saved, work, s);
    EQ_DISPATCH(32, 1, 1) EQ_DISPATCH(32, 1, 0) EQ_DISPATCH(32, 4, 1) EQ_DISPATCH(32, 4, 0)
    EQ_DISPATCH(32, 2, 1) EQ_DISPATCH(32, 2, 0) EQ_DISPATCH(32, 8, 1) EQ_DISPATCH(32, 8, 0)
    EQ_DISPATCH_LANE_WIDTHS
    EQ_DISPATCH(33, 1, 1) EQ_DISPATCH(33, 1, 0)
#undef EQ_DISPATCH
    set_error("egnn_forward: no kernel for F=%d H=%d A=%d", F, H, A);
    return EQUIGSPR_ENOTSUP;
}

extern "C" int equigspr_time_edge_fwd(const equigspr_egnn_config* cfg, const float* params, const equigspr_graph* g,
                                      const float* h_in, const float* x_in, const float* edge_attr, float* work,
                                      int32_t iters, float* ms_out_host, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(config_supported(cfg) && params && g && h_in && x_in && work && ms_out_host && iters >= 1 && g->n_edges > 0,
                 EQUIGSPR_EINVAL, "time_edge_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int F = cfg->hidden_nf, H = cfg->num_heads, A = cfg->in_edge_nf;
#define EQ_DISPATCH(FF, HH, AA) \
    if (F == FF && H == HH && A == AA) return time_edge_fwd_impl<FF, HH, AA>(cfg, params, g, h_in, x_in, edge_attr, work, iters, ms_out_host, s);
    EQ_DISPATCH(32, 1, 1) EQ_DISPATCH(32, 1, 0) EQ_DISPATCH(32, 4, 1) EQ_DISPATCH(32, 4, 0)
    EQ_DISPATCH(32, 2, 1) EQ_DISPATCH(32, 2, 0) EQ_DISPATCH(32, 8, 1) EQ_DISPATCH(32, 8, 0)
    EQ_DISPATCH_LANE_WIDTHS
    EQ_DISPATCH(33, 1, 1) EQ_DISPATCH(33, 1, 0)
#undef EQ_DISPATCH
    return EQUIGSPR_ENOTSUP;
}

#ifdef EQ_TRACE
extern "C" int equigspr_trace_set_fwd(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(eq_trace_ptr), &buf, sizeof(buf));
}
#endif
