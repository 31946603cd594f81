// Node-level EGNN backward on the matrix cores (hidden_nf 32, and 33 with the border feature): the gather backward of layer l
// and the node-MLP backward of layer l - 1 (reference: autograd through E_GCL.node_model / unsorted_segment_sum,
// src/3dmatch_train_egnn_with_batch.py:254-261, 96-100; derivation tests/manual_backward.py).
//
// A wave owns a tile of NPT nodes (16, or 4 when the problem has too few nodes to occupy the chip: the 16 columns then
// hold every node four times and the copies share out the node's segment).  Lane l = (i = l & 15, g = l >> 4) holds features
// 16 b + 4 g + r of node i -- the C/D layout of v_mfma_f32_16x16x4_f32 with D[feature][node], which is also the B operand
// of the next product (egnn_mfma16.h), so the chain
//     dP_r, dP_c (segment sums of dt1)  ->  G_h = dh_part + W_src^T dP_r + W_tgt^T dP_c  ->  v = W4 [h | agg] + b4
//     ->  dv = (W5^T G_h) SiLU'(v)  ->  dh_part' = G_h + W4h^T dv,  dagg = W4a^T dv,  P = [W_src h | W_tgt h + b1]
// needs no cross-lane traffic.  The A operands come from a per-workgroup LDS image of the nine 32 x 32 operand matrices
// (transposes staged as their own copies), one ds_read_b128 per four MFMAs; an operand fetched from global memory right
// before its MFMA exposes an L2 round trip per instruction (130 per tile in the first version of this kernel).
// Weight gradients are outer products summed over nodes: the vectors are transposed through wave-private LDS tiles and
// accumulated with K = nodes (pass16, egnn_bwd16.h) in registers across the wave's tiles; at the end the waves of a workgroup
// add their copies in LDS in wave order and the workgroup writes ONE slab copy (fixed order: bitwise reproducible).
#include "egnn_bwd16.h"

namespace eq {

#define NB_STAMP(k) do { if constexpr (kG && kN) EQ_STAMPK(4, k); } while (0)     // developer trace build only (tools/trace_edge.py)
#ifndef EQ_NB16_DBG
#define EQ_NB16_DBG 0                // timing experiments only (wrong results): 1 = no segment loads, 2 = no outer products, 4 = no slab copies, 8 = three col-side rows per node
#endif
#ifndef EQ_NB16_OCC
#define EQ_NB16_OCC 1                // waves per SIMD the register allocation is held to
#endif
// kE: the embedding backward rides along (hidden_nf == in_nf == out_nf).  Node-only launch (the backward's first): G_h = gh_out W_out
// instead of a G_h read, G_x = gx_out, dW_out += gh_out (x) h_L, db_out.  Gather-only launch (its last): dW_in += G_h (x) h_in, db_in,
// optionally grad_h_in = G_h W_in, instead of a G_h write.  (Replaces embed_out_bwd_kernel / embed_in_bwd_kernel, which cost
// 52 + 59 us per step at the KITTI shard in their 64-lanes-per-node form for hidden_nf 33.)
template <int H, int A, int kB, bool kG, bool kN, int NW, int NPT, int D, bool kE = false>
__global__ __launch_bounds__(64 * NW, EQ_NB16_OCC) void node_bwd16_kernel(const NodeBwd16Args a) {
    static_assert(!kE || (kG != kN), "embedding backward rides with the node-only or the gather-only launch");
    constexpr bool kEO = kE && kN, kEI = kE && kG;
    constexpr int F = 32 + kB, FP = 32 + 4 * kB, LD = Chunk16::LD, NP = 16 / NPT;
    using LL = LP<F, H, A>;
    using SB = Slabs<F>;
    constexpr bool kAl = (kB == 0);
    constexpr int kTile = 16 * LD;                                   // one [16 nodes][LD] operand tile
    constexpr int kWave = 5 * kTile;
    constexpr int kEmb = F * F + F;                                    // dW_out | db_out  or  dW_in | db_in
    constexpr int kGat = kG ? SB::gather : 0, kNod = kN ? SB::node : 0, kSlab = kGat + kNod + (kE ? kEmb : 0);
    constexpr int NC = NW > 4 ? 4 : NW;                               // slab copies in LDS: wave w adds into copy w % NC
    constexpr int kLds = NW * kWave > NC * kSlab ? NW * kWave : NC * kSlab;       // operand tiles, then the slab copies
    // operand images: m = 0 W_in^T (kEI) | 1 W_src^T, 2 W_tgt^T of layer l | 3 W4[:, :F], 4 W4[:, F:], 5 W5^T, 6 W4[:, :F]^T, 7 W4[:, F:]^T,
    // 8 W_src, 9 W_tgt of the node layer | 10 W_out^T (kEO).  Each [32][WS]; hidden_nf 33: + column 32 | row 32 | corner per matrix.
    // Then b4 | b1 (FP each).  A launch stages the contiguous range of them it uses.
    constexpr int MS = 32 * W16::WS, NM = 11, BS = 72;
    enum { mWinT = 0, mWsT = 1, mWtT = 2, mW4h = 3, mW4a = 4, mW5T = 5, mW4hT = 6, mW4aT = 7, mWs = 8, mWt = 9, mWoutT = 10 };
    constexpr int o_bord = NM * MS, o_vec = o_bord + kB * NM * BS, kImg = o_vec + 2 * FP;
    __shared__ __attribute__((aligned(16))) float img[kImg];
    __shared__ __attribute__((aligned(16))) float lds[kLds];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    float* X0 = lds + wv * kWave;
    float *X1 = X0 + kTile, *Y0 = X0 + 2 * kTile, *Y1 = X0 + 3 * kTile, *Y2 = X0 + 4 * kTile;
    // segment bounds of the wave's first tile: requested before the operand staging (and of every next tile one tile ahead), so the
    // index loads behind them start right after the barrier
    const int n_tiles = (a.n_nodes + NPT - 1) / NPT;
    auto node_of = [&](int tile_) { const int jr_ = tile_ * NPT + (i % NPT); return jr_ < a.n_nodes ? jr_ : a.n_nodes - 1; };
    int p_rlo = 0, p_rhi = 0, p_clo = 0, p_chi = 0;
    auto request_bounds = [&](int tile_) {
        if (kG && tile_ < n_tiles) {
            const int j_ = node_of(tile_);
            p_rlo = a.rptr[j_]; p_rhi = a.rptr[j_ + 1]; p_clo = a.cptr[j_]; p_chi = a.cptr[j_ + 1];
        }
    };
    request_bounds(blockIdx.x * NW + wv);
    {   // ---- stage: element (i, k) of operand m; all loads of a thread before its first store
        auto el = [&](int m, int r, int c) -> const float* {
            switch (m) {
                case mWinT: return a.w_in + c * F + r;                 // A[i][f] = W_in[f][i]
                case mWsT: return a.lw_g + LL::w1_off(c, r);
                case mWtT: return a.lw_g + LL::w1_off(c, F + r);
                case mW4h: return a.lw_n + LL::w4 + r * 2 * F + c;
                case mW4a: return a.lw_n + LL::w4 + r * 2 * F + F + c;
                case mW5T: return a.lw_n + LL::w5 + c * F + r;
                case mW4hT: return a.lw_n + LL::w4 + c * 2 * F + r;
                case mW4aT: return a.lw_n + LL::w4 + c * 2 * F + F + r;
                case mWs: return a.lw_n + LL::w1_off(r, c);
                case mWt: return a.lw_n + LL::w1_off(r, F + c);
                default: return a.w_out + c * F + r;                   // A[f][o] = W_out[o][f]
            }
        };
        constexpr int NT = 64 * NW, PER = 1024 / NT, M0 = kEI ? 0 : (kG ? 1 : 3), M1 = kEO ? 11 : (kN ? 10 : 3);
        constexpr int NBE = (M1 - M0) * BS, PB = (NBE + NT - 1) / NT;
        auto transposed = [](int m) { return m <= mWtT || (m >= mW5T && m <= mW4aT) || m == mWoutT; };      // walk the source rows: coalesced global reads
        float v[NM][PER], vb[kB ? PB : 1], vv = 0.0f;
#pragma unroll
        for (int m = M0; m < M1; ++m)
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int t = threadIdx.x + NT * u;
                const int r = transposed(m) ? (t & 31) : (t >> 5), c = transposed(m) ? (t >> 5) : (t & 31);
                v[m][u] = *el(m, r, c);
            }
        if constexpr (kB == 1) {
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int t = threadIdx.x + NT * u, m = M0 + t / BS, q = t % BS;
                vb[u] = 0.0f;
                if (t < NBE && q <= 64) {
                    const float* p = nullptr;
#pragma unroll
                    for (int mm = M0; mm < M1; ++mm)
                        if (mm == m) p = q < 32 ? el(mm, q, 32) : (q < 64 ? el(mm, 32, q - 32) : el(mm, 32, 32));
                    vb[u] = *p;
                }
            }
        }
        if (kN && threadIdx.x < 2 * FP) {
            const int q = threadIdx.x % FP;
            vv = q < F ? (threadIdx.x < FP ? a.lw_n[LL::b4 + q] : a.lw_n[LL::b1_off(q)]) : 0.0f;
        }
#pragma unroll
        for (int m = M0; m < M1; ++m)
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int t = threadIdx.x + NT * u;
                const int r = transposed(m) ? (t & 31) : (t >> 5), c = transposed(m) ? (t >> 5) : (t & 31);
                img[m * MS + r * W16::WS + c] = v[m][u];
            }
        if constexpr (kB == 1) {
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int t = threadIdx.x + NT * u;
                if (t < NBE) img[o_bord + M0 * BS + t] = vb[u];
            }
        }
        if (kN && threadIdx.x < 2 * FP) img[o_vec + threadIdx.x] = vv;
    }
    __syncthreads();
    // acc += A_m v (+ border); returns the border output's sum
    const float* im = img;                               // re-derived per tile through an opaque offset: keeps the operand reads in the loop
    auto prod = [&](int m, const f32x4 (&vin)[2], float vinb, f32x4 (&acc)[2]) -> float {
        mv16(im + m * MS, i, g, vin, acc);
        if constexpr (kB == 1) {
            const float* bd = im + o_bord + m * BS;
            return mv16_border(bd, bd + 32, bd[64], g, vin, vinb, acc);
        }
        return 0.0f;
    };

    f32x4 accA[2][2], accB[2][2], acc4a[2][2], acc4b[2][2], acc5[2][2], accE[2][2];      // accE: dW_out or dW_in (kE)
    f32x2 cs_be = {0.0f, 0.0f};
    PassBorder pbE;
    pbE.init();
#pragma unroll
    for (int x_ = 0; x_ < 2; ++x_)
#pragma unroll
        for (int y_ = 0; y_ < 2; ++y_) accE[x_][y_] = splat4(0.0f);
    // a [N][F] row of the caller's tensors (row stride F: 16-byte rows only for hidden_nf 32)
    auto load_rowF = [&](const float* __restrict__ base, int j_, f32x4 (&v_)[2], float& vb_) {
        const float* p_ = base + (size_t)j_ * F;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if constexpr (kAl) v_[b] = ldg4(p_ + 16 * b + 4 * g);
            else v_[b] = f32x4{p_[16 * b + 4 * g], p_[16 * b + 4 * g + 1], p_[16 * b + 4 * g + 2], p_[16 * b + 4 * g + 3]};
        }
        vb_ = kB ? p_[32] : 0.0f;
    };
    f32x2 cs_b1 = {0.0f, 0.0f}, cs_b4 = {0.0f, 0.0f}, cs_b5 = {0.0f, 0.0f}, cs_none = {0.0f, 0.0f};
    PassBorder pbA, pbB, pb4a, pb4b, pb5;
    pbA.init(); pbB.init(); pb4a.init(); pb4b.init(); pb5.init();
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) accA[x][y] = accB[x][y] = acc4a[x][y] = acc4b[x][y] = acc5[x][y] = splat4(0.0f);

    auto put = [&](float* T, const f32x4 (&v)[2], float vb) {
        st4(T + i * LD + 4 * g, v[0]);
        st4(T + i * LD + 16 + 4 * g, v[1]);
        if (kB && g == 0) T[i * LD + 32] = vb;
    };

    NB_STAMP(0);
    for (int tile = blockIdx.x * NW + wv; tile < n_tiles; tile += gridDim.x * NW) {
        const int jr = tile * NPT + (i % NPT), part = i / NPT;
        const bool live = jr < a.n_nodes, keep = live && i < NPT;      // keep: the copy of the node that stores and enters the sums
        const int j = live ? jr : a.n_nodes - 1;
        const f32x4 zero = splat4(0.0f);
        f32x4 gh[2];
        float ghb = 0.0f;
        {
            int off = 0;
            asm volatile("" : "+v"(off));
            im = img + off;
        }
        NB_STAMP(1);
        // inputs of the node part.  Few nodes (one tile per wave, latency decides): requested first, consumed after the gather;
        // many nodes (two waves per SIMD hide the latency, registers are short): requested when needed
        constexpr bool kEarly = NPT < 16;
        f32x4 hn[2], an[2];
        float hnb = 0.0f, anb = 0.0f;
        auto load_node_inputs = [&]() {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                hn[b] = ldg4(a.h_n + (size_t)j * FP + 16 * b + 4 * g);
                an[b] = ldg4(a.agg_n + (size_t)j * FP + 16 * b + 4 * g);
            }
            if (kB) { hnb = a.h_n[(size_t)j * FP + 32]; anb = a.agg_n[(size_t)j * FP + 32]; }
        };
        if constexpr (kN && kEarly) load_node_inputs();
        if constexpr (kG) {
            const int rlo = p_rlo, rlen = p_rhi - p_rlo, clo = p_clo, clen = (EQ_NB16_DBG & 8) ? min(3, p_chi - p_clo) : p_chi - p_clo;
            request_bounds(tile + gridDim.x * NW);
            f32x4 hg[2];
            float hgb = 0.0f;
            auto load_gather_inputs = [&]() {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    hg[b] = ldg4(a.h_g + (size_t)j * FP + 16 * b + 4 * g);
                    gh[b] = ldg4(a.dh_part + (size_t)j * FP + 16 * b + 4 * g);          // residual + node-MLP path of layer l
                }
                if (kB) { hgb = a.h_g[(size_t)j * FP + 32]; ghb = a.dh_part[(size_t)j * FP + 32]; }
            };
            if constexpr (kEarly) load_gather_inputs();
            const f32x4 gx0 = g == 0 ? ldg4(a.Gx + (size_t)j * 4) : zero;
            int maxlen = (EQ_NB16_DBG & 1) ? 0 : max(rlen, clen);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
            // ---- segment sums: dP_r over the edges with row == j, dP_c over those with col == j; lane groups 0 / 1 also sum the
            // coordinate gradients of the row / col side.  Copy `part` of a node takes entries part, part + NP, ...
            f32x4 dpr[2] = {zero, zero}, dpc[2] = {zero, zero};
            float dprb = 0.0f, dpcb = 0.0f;
            f32x4 xs = zero;
            const float* d3 = g == 0 ? a.gxr : a.gxc;
            NB_STAMP(2);
            int er[D], ec[D];
            auto load_idx = [&](int base, int (&r_)[D], int (&c_)[D]) {
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    const int aa = base + part + u * NP;
                    r_[u] = aa < rlen ? a.rperm[rlo + aa] : -1;
                    c_[u] = aa < clen ? a.cperm[clo + aa] : -1;
                }
            };
            load_idx(0, er, ec);
            for (int base = 0; base < maxlen; base += D * NP) {
                f32x4 vr[D][2], vc[D][2], tv[D];
#pragma unroll
                for (int u = 0; u < D; ++u) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        vr[u][b] = er[u] >= 0 ? ldg4(a.dt1 + (size_t)er[u] * 32 + 16 * b + 4 * g) : zero;      // [E][32] rows + [E] border column
                        vc[u][b] = ec[u] >= 0 ? ldg4(a.dt1 + (size_t)ec[u] * 32 + 16 * b + 4 * g) : zero;
                    }
                    // (hidden_nf 33: the border column of dt1 arrives as tv.w from the edge backward and is summed in xs[3])
                    const int e3 = g == 0 ? er[u] : ec[u];
                    tv[u] = (g < 2 && e3 >= 0) ? ldg4(d3 + (size_t)e3 * 4) : zero;
                }
                load_idx(base + D * NP, er, ec);         // the next round's indices: in flight with this round's rows
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    dpr[0] += vr[u][0]; dpr[1] += vr[u][1];
                    dpc[0] += vc[u][0]; dpc[1] += vc[u][1];
                    xs += tv[u];
                }
            }
            NB_STAMP(3);
            if constexpr (!kEarly) load_gather_inputs();
            if constexpr (NP > 1) {                     // the copies of a node add up their parts (all copies get the same sum)
#pragma unroll
                for (int o = NPT; o < 16; o <<= 1) {
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            dpr[b][r] += __shfl_xor(dpr[b][r], o, 64);
                            dpc[b][r] += __shfl_xor(dpc[b][r], o, 64);
                        }
#pragma unroll
                    for (int r = 0; r < (kB ? 4 : 3); ++r) xs[r] += __shfl_xor(xs[r], o, 64);
                }
            }
            if constexpr (kB == 1) {                    // lane group 0 summed the row side's border column, group 1 the col side's
                dprb = __shfl(xs[3], i, 64);
                dpcb = __shfl(xs[3], 16 + i, 64);
            }
            {   // G_x = identity + row-segment terms + col-segment terms
                f32x4 xc;
#pragma unroll
                for (int r = 0; r < 3; ++r) xc[r] = __shfl(xs[r], (lane + 16) & 63, 64);
                if (g == 0 && keep) {
                    const f32x4 gx = f32x4{(gx0[0] + xs[0]) + xc[0], (gx0[1] + xs[1]) + xc[1], (gx0[2] + xs[2]) + xc[2], 0.0f};
                    st4(a.Gx_new + (size_t)j * 4, gx);
                    if (a.gx_out3) { a.gx_out3[(size_t)j * 3 + 0] = gx[0]; a.gx_out3[(size_t)j * 3 + 1] = gx[1]; a.gx_out3[(size_t)j * 3 + 2] = gx[2]; }
                }
            }
            // ---- G_h = dh_part + W_src^T dP_r + W_tgt^T dP_c   (W1 = [W_src | W_tgt | W_geo] per output row)
            ghb += prod(mWsT, dpr, dprb, gh);
            ghb += prod(mWtT, dpc, dpcb, gh);
            NB_STAMP(4);
            // ---- dW_src += dP_r (x) h, dW_tgt += dP_c (x) h, db1 += dP_c
            if (!keep) { dpr[0] = dpr[1] = dpc[0] = dpc[1] = zero; dprb = dpcb = 0.0f; }
            put(X0, dpr, dprb); put(X1, dpc, dpcb); put(Y0, hg, hgb);
            __builtin_amdgcn_wave_barrier();
            if (!(EQ_NB16_DBG & 2)) {
            pass16<2, false, kB>(X0, Y0, LD, i, g, accA, cs_none, pbA);
            pass16<2, true, kB>(X1, Y0, LD, i, g, accB, cs_b1, pbB);
            }
            __builtin_amdgcn_wave_barrier();
            if constexpr (!kN && !kEI) {
                if (keep) {
                    st4(a.Gh_out + (size_t)j * FP + 4 * g, gh[0]);
                    st4(a.Gh_out + (size_t)j * FP + 16 + 4 * g, gh[1]);
                    if (kB && g == 0) st4(a.Gh_out + (size_t)j * FP + 32, f32x4{ghb, 0.0f, 0.0f, 0.0f});
                }
            }
            if constexpr (kEI) {
                // ---- embedding_in backward: grad_h_in = G_h W_in (optional), dW_in += G_h (x) h_in, db_in += G_h
                f32x4 hin[2];
                float hinb;
                load_rowF(a.h_in, j, hin, hinb);
                if (a.grad_h_in) {
                    f32x4 gi[2] = {zero, zero};
                    const float gib = prod(mWinT, gh, ghb, gi);
                    if (keep) {
                        float* q_ = a.grad_h_in + (size_t)j * F;
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            if constexpr (kAl) st4(q_ + 16 * b + 4 * g, gi[b]);
                            else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) q_[16 * b + 4 * g + r] = gi[b][r];
                            }
                        }
                        if (kB && g == 0) q_[32] = gib;
                    }
                }
                if (!keep) { gh[0] = gh[1] = zero; ghb = 0.0f; }
                put(X0, gh, ghb); put(Y0, hin, hinb);
                __builtin_amdgcn_wave_barrier();
                pass16<2, true, kB>(X0, Y0, LD, i, g, accE, cs_be, pbE);
                __builtin_amdgcn_wave_barrier();
            }
        } else if constexpr (kEO) {
            // ---- embedding_out backward: G_h = gh_out W_out, G_x = gx_out, dW_out += gh_out (x) h_L, db_out += gh_out
            f32x4 go[2], hl[2];
            float gob, hlb = 0.0f;
            load_rowF(a.gh_out, j, go, gob);
#pragma unroll
            for (int b = 0; b < 2; ++b) hl[b] = ldg4(a.hL + (size_t)j * FP + 16 * b + 4 * g);
            if (kB) hlb = a.hL[(size_t)j * FP + 32];
            if (g == 0 && keep) st4(a.Gx_new + (size_t)j * 4, f32x4{a.gx_out[(size_t)j * 3], a.gx_out[(size_t)j * 3 + 1], a.gx_out[(size_t)j * 3 + 2], 0.0f});
            gh[0] = gh[1] = zero;
            ghb = prod(mWoutT, go, gob, gh);
            if (!keep) { go[0] = go[1] = zero; gob = 0.0f; }
            put(X0, go, gob); put(Y0, hl, hlb);
            __builtin_amdgcn_wave_barrier();
            pass16<2, true, kB>(X0, Y0, LD, i, g, accE, cs_be, pbE);
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int b = 0; b < 2; ++b) gh[b] = ldg4(a.Gh_in + (size_t)j * FP + 16 * b + 4 * g);
            if (kB) ghb = a.Gh_in[(size_t)j * FP + 32];
        }
        NB_STAMP(5);
        if constexpr (kN) {
            if constexpr (!kEarly) load_node_inputs();
            // ---- v = W4 [h | agg] + b4 ; hid = SiLU(v)
            f32x4 v[2] = {lds4(im + o_vec + 4 * g), lds4(im + o_vec + 16 + 4 * g)};
            float vb = kB ? im[o_vec + 32] : 0.0f;
            vb += prod(mW4h, hn, hnb, v);
            vb += prod(mW4a, an, anb, v);
            f32x4 hid[2], sp[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const f32x4 sg = sigmoid4(v[b]), one = splat4(1.0f);
                hid[b] = v[b] * sg;
                sp[b] = sg * (one + v[b] * (one - sg));                       // SiLU'(v)
            }
            const float sgb = kB ? sigmoid_fast(vb) : 0.0f, hidb = vb * sgb, spb = sgb * (1.0f + vb * (1.0f - sgb));
            // ---- dv = (W5^T G_h) SiLU'(v)
            f32x4 dv[2] = {zero, zero};
            float dvb = prod(mW5T, gh, ghb, dv) * spb;
#pragma unroll
            for (int b = 0; b < 2; ++b) dv[b] *= sp[b];
            // ---- dh_part = G_h + W4[:, :F]^T dv ; dagg = W4[:, F:]^T dv
            f32x4 dh[2] = {gh[0], gh[1]}, da[2] = {zero, zero};
            const float dhb = ghb + prod(mW4hT, dv, dvb, dh), dab = prod(mW4aT, dv, dvb, da);
            if (keep) {
                st4(a.dh_part + (size_t)j * FP + 4 * g, dh[0]); st4(a.dh_part + (size_t)j * FP + 16 + 4 * g, dh[1]);
                st4(a.dagg + (size_t)j * FP + 4 * g, da[0]); st4(a.dagg + (size_t)j * FP + 16 + 4 * g, da[1]);
                if (kB && g == 0) {
                    st4(a.dh_part + (size_t)j * FP + 32, f32x4{dhb, 0.0f, 0.0f, 0.0f});
                    st4(a.dagg + (size_t)j * FP + 32, f32x4{dab, 0.0f, 0.0f, 0.0f});
                }
            }
            // ---- the layer's projection rows for the edge recompute: P = [W_src h | W_tgt h + b1]
            if (a.P) {
                f32x4 pr[2] = {zero, zero}, pc[2] = {lds4(im + o_vec + FP + 4 * g), lds4(im + o_vec + FP + 16 + 4 * g)};
                const float prb = prod(mWs, hn, hnb, pr), pcb = (kB ? im[o_vec + FP + 32] : 0.0f) + prod(mWt, hn, hnb, pc);
                if (keep) {
                    float* prow = a.P + (size_t)j * 2 * FP + 4 * g;
                    st4(prow, pr[0]); st4(prow + 16, pr[1]); st4(prow + FP, pc[0]); st4(prow + FP + 16, pc[1]);
                    if (kB && g == 0) {
                        st4(a.P + (size_t)j * 2 * FP + 32, f32x4{prb, 0.0f, 0.0f, 0.0f});
                        st4(a.P + (size_t)j * 2 * FP + FP + 32, f32x4{pcb, 0.0f, 0.0f, 0.0f});
                    }
                }
            }
            NB_STAMP(6);
            // ---- dW4 += dv (x) [h | agg], db4 += dv, dW5 += G_h (x) hid, db5 += G_h
            if (!keep) { dv[0] = dv[1] = gh[0] = gh[1] = zero; dvb = ghb = 0.0f; }
            put(X0, dv, dvb); put(X1, gh, ghb); put(Y0, hn, hnb); put(Y1, an, anb); put(Y2, hid, hidb);
            __builtin_amdgcn_wave_barrier();
            if (!(EQ_NB16_DBG & 2)) {
            pass16<2, true, kB>(X0, Y0, LD, i, g, acc4a, cs_b4, pb4a);
            pass16<2, false, kB>(X0, Y1, LD, i, g, acc4b, cs_none, pb4b);
            pass16<2, true, kB>(X1, Y2, LD, i, g, acc5, cs_b5, pb5);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    NB_STAMP(7);
    // ---- slab copies.  Column sums and border partials: over the four lane groups first
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        cs_b1[b] = gsum(cs_b1[b]); cs_b4[b] = gsum(cs_b4[b]); cs_b5[b] = gsum(cs_b5[b]); cs_be[b] = gsum(cs_be[b]);
        if constexpr (kB == 1 && kE) { pbE.bcol[b] = gsum(pbE.bcol[b]); pbE.brow[b] = gsum(pbE.brow[b]); }
        if constexpr (kB == 1) {
            PassBorder* pbs[5] = {&pbA, &pbB, &pb4a, &pb4b, &pb5};
#pragma unroll
            for (int k = 0; k < 5; ++k) { pbs[k]->bcol[b] = gsum(pbs[k]->bcol[b]); pbs[k]->brow[b] = gsum(pbs[k]->brow[b]); }
        }
    }
    if constexpr (kB == 1) {
        PassBorder* pbs[5] = {&pbA, &pbB, &pb4a, &pb4b, &pb5};
#pragma unroll
        for (int k = 0; k < 5; ++k) { pbs[k]->corner = gsum(pbs[k]->corner); pbs[k]->xbs = gsum(pbs[k]->xbs); }
    }
    if constexpr (kB == 1 && kE) { pbE.corner = gsum(pbE.corner); pbE.xbs = gsum(pbE.xbs); }
    auto emit_all = [&](float* tg, float* tn, bool add) {
        // add: a second wave's values on top of the first's (LDS float add without return; two addends -> order-free)
        auto put1 = [&](float* p, float v) {
            if (add) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else *p = v;
        };
        auto emit_mat = [&](float* dst, int ld, int c0, const f32x4 (&acc)[2][2], const PassBorder& pb) {
#pragma unroll
            for (int bx = 0; bx < 2; ++bx) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int by = 0; by < 2; ++by) put1(dst + (16 * bx + 4 * g + r) * ld + c0 + 16 * by + i, acc[bx][by][r]);
                if (kB && g == 0) {
                    put1(dst + (16 * bx + i) * ld + c0 + 32, pb.bcol[bx]);
                    put1(dst + 32 * ld + c0 + 16 * bx + i, pb.brow[bx]);
                }
            }
            if (kB && g == 0 && i == 0) put1(dst + 32 * ld + c0 + 32, pb.corner);
        };
        auto emit_vec = [&](float* dst, const f32x2& cs, const PassBorder& pb) {
            if (g == 0) {
                put1(dst + i, cs[0]); put1(dst + 16 + i, cs[1]);
                if (kB && i == 0) put1(dst + 32, pb.xbs);
            }
        };
        if constexpr (kG) {
            emit_mat(tg, F, 0, accA, pbA);
            emit_mat(tg + F * F, F, 0, accB, pbB);
            emit_vec(tg + 2 * F * F, cs_b1, pbB);
        }
        if constexpr (kN) {
            emit_mat(tn, 2 * F, 0, acc4a, pb4a);
            emit_mat(tn, 2 * F, F, acc4b, pb4b);
            emit_vec(tn + F * 2 * F, cs_b4, pb4a);
            emit_mat(tn + F * 2 * F + F, F, 0, acc5, pb5);
            emit_vec(tn + F * 2 * F + F + F * F, cs_b5, pb5);
        }
        if constexpr (kE) {                              // behind the gather / node part of the copy
            float* te = tg + kGat + kNod;
            emit_mat(te, F, 0, accE, pbE);
            emit_vec(te + F * F, cs_be, pbE);
        }
    };
    if (EQ_NB16_DBG & 4) return;
    float* gat_dst = kG ? a.gat_slab + (size_t)blockIdx.x * SB::gather : nullptr;
    float* node_dst = kN ? a.node_slab + (size_t)blockIdx.x * SB::node : nullptr;
    float* emb_dst = kE ? a.emb_slab + (size_t)blockIdx.x * kEmb : nullptr;
    static_assert(NW > 1, "one slab copy per workgroup is assembled in LDS");
    {
        __syncthreads();                                 // the operand tiles are dead: the LDS becomes the slab copies
        NB_STAMP(8);
        if (wv < NC) emit_all(lds + wv * kSlab, lds + wv * kSlab + kGat, false);
        __syncthreads();
        if constexpr (NW > NC) {
            if (wv >= NC) emit_all(lds + (wv - NC) * kSlab, lds + (wv - NC) * kSlab + kGat, true);
            __syncthreads();
        }
        NB_STAMP(9);
        if constexpr (kGat % 4 == 0 && kSlab % 4 == 0 && SB::gather % 4 == 0 && SB::node % 4 == 0 && kEmb % 4 == 0) {
            // copies added in fixed order, four entries per thread and step (hidden_nf 32: every slab boundary is 16-byte aligned)
            for (int t = 4 * threadIdx.x; t < kSlab; t += 4 * 64 * NW) {
                f32x4 v = lds4(lds + t);
#pragma unroll
                for (int w = 1; w < NC; ++w) v += lds4(lds + w * kSlab + t);
                st4(t < kGat ? gat_dst + t : (t < kGat + kNod ? node_dst + (t - kGat) : emb_dst + (t - kGat - kNod)), v);
            }
        } else {
            for (int t = threadIdx.x; t < kSlab; t += 64 * NW) {
                float v = lds[t];
#pragma unroll
                for (int w = 1; w < NC; ++w) v += lds[w * kSlab + t];
                if (t < kGat) gat_dst[t] = v;
                else if (t < kGat + kNod) node_dst[t - kGat] = v;
                else emb_dst[t - kGat - kNod] = v;
            }
        }
    }
    NB_STAMP(10);
}

#ifndef EQ_NB16_SMALL
#define EQ_NB16_SMALL 16384          // nodes up to which the 4-node tiles are used (one wave per SIMD still has work)
#endif
#ifndef EQ_NB16_D
#define EQ_NB16_D 4
#endif

int node_bwd16_max_slabs(int n_nodes) { return n_nodes < 1 ? 1 : (n_nodes + 15) / 16 < 512 ? (n_nodes + 15) / 16 : 512; }

#ifndef EQ_NB16_DL
#define EQ_NB16_DL 4                 // segment rows in flight per lane on the many-node path (2 .. 12 measured equal: the phase is bandwidth-bound)
#endif
#ifndef EQ_NB16_NW
#define EQ_NB16_NW 4                 // waves per workgroup of the many-node path (8 = two per SIMD measured slower: 130 vs 121 us at 65 536 nodes)
#endif
template <int H, int A, int kB, bool kG, bool kN, bool kE = false>
static int launch_mode(const NodeBwd16Args& args, hipStream_t s, int* n_slabs) {
    constexpr int NW = 4, NWL = EQ_NB16_NW;
    const int N = args.n_nodes;
    if (N <= EQ_NB16_SMALL) {
        const int tiles = ceil_div(N, 4);
        int blocks = ceil_div(tiles, NW);
        blocks = blocks > 256 ? 256 : blocks;
        hipLaunchKernelGGL((node_bwd16_kernel<H, A, kB, kG, kN, NW, 4, EQ_NB16_D, kE>), dim3(blocks), dim3(64 * NW), 0, s, args);
        *n_slabs = blocks;
    } else {
        const int tiles = ceil_div(N, 16);
        int blocks = ceil_div(tiles, NWL);
        blocks = blocks > 2048 / NWL ? 2048 / NWL : blocks;
        hipLaunchKernelGGL((node_bwd16_kernel<H, A, kB, kG, kN, NWL, 16, EQ_NB16_DL, kE>), dim3(blocks), dim3(64 * NWL), 0, s, args);
        *n_slabs = blocks;
    }
    return 0;
}

template <int H, int A, int kB>
static int launch_cfg(int mode, const NodeBwd16Args& args, hipStream_t s, int* n_slabs) {
    switch (mode) {
        case kNodeBwdGather | kNodeBwdNode: return launch_mode<H, A, kB, true, true>(args, s, n_slabs);
        case kNodeBwdGather: return launch_mode<H, A, kB, true, false>(args, s, n_slabs);
        case kNodeBwdNode: return launch_mode<H, A, kB, false, true>(args, s, n_slabs);
        case kNodeBwdGather | kNodeBwdEmbed: return launch_mode<H, A, kB, true, false, true>(args, s, n_slabs);
        case kNodeBwdNode | kNodeBwdEmbed: return launch_mode<H, A, kB, false, true, true>(args, s, n_slabs);
    }
    return EQUIGSPR_EINVAL;
}

int launch_node_bwd16(int F, int H, int A, int mode, const NodeBwd16Args& args, hipStream_t s, int* n_slabs) {
    if (F == 32 && H == 1 && A == 1) return launch_cfg<1, 1, 0>(mode, args, s, n_slabs);
    if (F == 32 && H == 1 && A == 0) return launch_cfg<1, 0, 0>(mode, args, s, n_slabs);
    if (F == 32 && H == 4 && A == 1) return launch_cfg<4, 1, 0>(mode, args, s, n_slabs);
    if (F == 32 && H == 4 && A == 0) return launch_cfg<4, 0, 0>(mode, args, s, n_slabs);
    if (F == 32 && H == 2 && A == 1) return launch_cfg<2, 1, 0>(mode, args, s, n_slabs);
    if (F == 32 && H == 2 && A == 0) return launch_cfg<2, 0, 0>(mode, args, s, n_slabs);
    if (F == 32 && H == 8 && A == 1) return launch_cfg<8, 1, 0>(mode, args, s, n_slabs);
    if (F == 32 && H == 8 && A == 0) return launch_cfg<8, 0, 0>(mode, args, s, n_slabs);
    if (F == 33 && H == 1 && A == 1) return launch_cfg<1, 1, 1>(mode, args, s, n_slabs);
    if (F == 33 && H == 1 && A == 0) return launch_cfg<1, 0, 1>(mode, args, s, n_slabs);
    set_error("node_bwd16: no 16-node kernel for hidden_nf=%d, num_heads=%d, in_edge_nf=%d", F, H, A);
    return EQUIGSPR_ENOTSUP;
}

}  // namespace eq

#ifdef EQ_TRACE
extern "C" int equigspr_trace_set_node_bwd(unsigned long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(eq_trace_ptr), &buf, sizeof(buf));
}
#endif
