// Typed irrep-basis message passing (SURVEY section 8(f) row N4 "proper"): the tensor-field layer the reference's basis
// (tools/basis.py::get_basis :153-205, tools/irr_repr.py:103-104, tools/spherical_harmonics.py:75-123) was written for and its
// scripts never build.  Degree-0 features f0 [N, C0] and degree-1 features f1 [N, C1, 3] are both carried through; every
// (d_in, d_out) path is a pairwise kernel sum_J R_J(|r|) K_J(r / |r|); messages go from `col` to `row` (train :252-256, :270-278).
// Definition: oracle/irreps_oracle.py.
//
//   d = |r|, r = x[row] - x[col], u = r / d          hid = SiLU(w_r1 d + b_r1)  [H]         R = W_r2 hid + b_r2
//   m0[e, o]    = sum_i R00[o, i] K00 f0[col, i]    + sum_i R10[o, i] (K10 . f1[col, i])
//   m1[e, o, a] = sum_i R01[o, i] K01[a] f0[col, i] + sum_i sum_J R11[o, i, J] (K11[a, :, J] . f1[col, i])
//   out0[n] = W_s0 f0[n] + sum_{e: row = n} m0[e]        out1[n] = W_s1 f1[n] + sum_{e: row = n} m1[e]
//
// MI355X formulation (round 5: one first-class kernel path, no basis tensors in HBM, no library GEMMs):
//  * The degree <= 1 basis in CLOSED FORM, in registers (the constants -- Q_J's signs included -- are fitted on the host to the
//    package's own get_basis, sh.degree1_closed_form):  K00 = k00,  K01[a] = k01 u_a,  K10[b] = k10 u_b,
//    K11[a, b, 0] = k110 delta_ab,  K11[a, b, 1] = k111 eps_abc u_c,  K11[a, b, 2] = k112 (u_a u_b - delta_ab / 3);  on a self
//    loop (r = 0) the J >= 1 kernels are zero (Y_J(0) is not covariant).
//  * AGGREGATE, THEN TRANSFORM: R is linear in hid~ = [hid, 1], so the sum over a node's incoming edges is taken before the
//    radial weights act.  Per edge the operand vector
//        c[e] = [ K00 f0 (C0) | K01[a] f0 (3 C0) | K10 . f1 (C1) | K11[a, :, J] . f1 ((a, J)-major, 9 C1) ]
//    and the per-node moments M[n, j, h] = sum_{e -> n} c[e, j] hid[e, h], Mb[n, j] = sum_e c[e, j] (the bias column).
//    `irrep_moments16_kernel`: a wave per (node, half of c); the moments are OUTER PRODUCTS with K = the node's edges, on
//    v_mfma_f32_16x16x4_f32: A = hid [16 h x 4 edges], B = c [4 edges x 16 entries]; both operands are computed in the lane
//    that feeds them (lane (idx = l & 15, k = l >> 4)): the gathered feature rows are read as 64-byte segments, the per-edge
//    geometry once per edge by the one-lane-per-edge front phase (wave-private LDS, broadcast reads).
//  * `irrep_node16_kernel`: out = W M (+ bias, + self-interaction) for 16 nodes per workgroup, the waves splitting K; the
//    radial weights are read IN PLACE from the parameter tensor (address = out * so + in * si + h: the same kernel applies the
//    transposed weights of the backward by swapping the strides).
//  * Backward: the adjoint of a tensor-field convolution is a tensor-field convolution on the reversed edges with transposed
//    weights -- the SAME two kernels give d f0 / d f1 (aggregate at `col`, gather the output gradients from `row`);
//    `irrep_expand16_kernel` (dM = W^T g) + `irrep_geom_bwd_kernel` (per edge: d hid, d u -> d x, d w_r1, d b_r1) +
//    `irrep_wgrad16_kernel` (dW = g^T M over node ranges, slabs summed in fixed order) complete it.  No atomics anywhere:
//    bitwise reproducible.
#include <atomic>
#include <type_traits>

#include "common.h"
#include "egnn_mfma16.h"

namespace eq {
namespace ir {

constexpr int kMaxC = 32;

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// channel padding of the operand vector: degree-0 slots in multiples of 16 entries, degree-1 slots 8 / 16 / 32
static inline int pad0(int c) { return round_up(c, 16); }
static inline int pad1(int c) { return c <= 8 ? 8 : (c <= 16 ? 16 : 32); }

struct Consts { float k00, k01, k10, k110, k111, k112; };

// ------------------------------------------------------------------------------------------------ moments
// Per-edge geometry in AGGREGATION order (irrep_edge_table_kernel): position p of node n's segment [ptr[n], ptr[n + 1]) holds the
// unit vector and length of r = x[n] - x[src] and the source node -- one coalesced read per wave instead of the dependent chain
// index -> source -> coordinates -> sqrt / divide in every launch of every layer (the coordinates are fixed through a stack).
// a loaded value where `ok`, zero elsewhere -- by a MULTIPLICATION with 1 / 0, so the load itself stays unconditional: a select (or an
// AND with a 0 / -1 mask, which the compiler folds back into a select) lets it move the load under a branch, and every branch
// around a load costs the s_waitcnt pass its exact count -- it falls back to vmcnt(0), which also waits for the prefetch of the
// next step.  (The masked lanes read clamped, valid addresses: finite values, so x * 0 is a zero.)
__device__ __forceinline__ float keep_if(float v, bool ok) { return v * (ok ? 1.0f : 0.0f); }

struct EdgeTab { const float4* geo; const int* src; };       // [E] (u0, u1, u2, d), [E]

struct MomArgs {
    const float* f0; const float* f1;                       // [N, c0], [N, c1, 3]: the features are gathered at tab.src[p]
    const int* ptr; EdgeTab tab;                            // node n aggregates the positions ptr[n] .. ptr[n + 1] (ascending edge id)
    const float* w1; const float* b1;                       // radial.0: [H], [H]
    Consts k;
    int n_nodes, c0, c1, c0p, c1p, lc;
    float* M; float* Mb;                                    // [N, lc, H], [N, lc]
};

__global__ void irrep_edge_table_kernel(const float* __restrict__ x, const int* __restrict__ agg, const int* __restrict__ src,
                                       const int* __restrict__ perm, int n_edges, float4* __restrict__ geo, int* __restrict__ srcs) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_edges) return;
    const int e = perm[p], n = agg[e], s = src[e];
    const float r0 = x[(size_t)n * 3] - x[(size_t)s * 3], r1 = x[(size_t)n * 3 + 1] - x[(size_t)s * 3 + 1], r2 = x[(size_t)n * 3 + 2] - x[(size_t)s * 3 + 2];
    const float d = sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
    const float inv = d > 0.0f ? 1.0f / d : 0.0f;            // self loop / duplicated point: u = 0, only the J = 0 kernels act
    geo[p] = make_float4(r0 * inv, r1 * inv, r2 * inv, d);
    srcs[p] = s;
}

// One wave per (node, unit): unit 0 = the 4 c0p entries built from f0, units 1.. = 80-entry groups of the 10 c1p built from f1.
// r = x[n] - x[src]: the caller flips k01 / k10 when it aggregates at the other end of the edges (the adjoint).
#ifndef EQ_IR_MOM_OCC
#define EQ_IR_MOM_OCC 5                 // workgroups per CU (= waves per SIMD) the register budget is set for; one pair: 43.3 us at three (140 registers), 35.7-36.9 at four (105), 33.1 at five
#endif
template <int NH, int B0, int C1P>
__global__ __launch_bounds__(256, EQ_IR_MOM_OCC) void irrep_moments16_kernel(const MomArgs a) {
    __shared__ float4 geo[4][2][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wv;
    if (n >= a.n_nodes) return;                               // wave-uniform
    constexpr int H = 16 * NH;
    const int j = lane & 15, kq = lane >> 4;
    float w1v[NH], b1v[NH];
#pragma unroll
    for (int hb = 0; hb < NH; ++hb) { w1v[hb] = a.w1[16 * hb + j]; b1v[hb] = a.b1[16 * hb + j]; }
    const int lo = a.ptr[n], hi = a.ptr[n + 1];
    float4* gA = geo[wv][0];
    float4* gB = geo[wv][1];
    // front phase, one lane per edge of the chunk: (u, d) and the edge's scalar factors (zero for the padding slots of the last
    // k-step: such a slot then contributes c = 0 to every moment)
    auto front = [&](int base, int cnt) {
        const bool valid = lane < cnt;
        const float4 gv = valid ? a.tab.geo[base + lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const int s = valid ? a.tab.src[base + lane] : n;
        const bool live = valid && gv.w > 0.0f;              // self loop / duplicated point: only the J = 0 kernels act
        gA[lane] = gv;
        gB[lane] = make_float4(valid ? a.k.k00 : 0.0f, valid ? a.k.k110 : 0.0f, live ? a.k.k112 : 0.0f, __int_as_float(s));
    };
    const int unit = blockIdx.y;
    if (unit == 0) {
        // ---- c00 = K00 f0 | c01[a] = K01[a] f0: blocks (slot s = 0..3, sub): entry 16 (s B0 + sub) + j
        f32x4 acc[4 * B0][NH];
        float bs[4 * B0];
#pragma unroll
        for (int t = 0; t < 4 * B0; ++t) {
            bs[t] = 0.0f;
#pragma unroll
            for (int hb = 0; hb < NH; ++hb) acc[t][hb] = splat4(0.0f);
        }
        for (int base = lo; base < hi; base += 64) {
            const int cnt = min(64, hi - base);
            front(base, cnt);
            wave_sync();
            const int nks = (cnt + 3) >> 2;
            // the gathered feature rows of k-step ks + 1 are requested before the products of k-step ks (index -> row is a
            // dependent chain of two memory round trips; with three waves per SIMD nothing else hides it)
            auto gather = [&](int ks, float (&gg)[B0]) {
                const float* f0r = a.f0 + (size_t)__float_as_int(gB[4 * ks + kq].w) * a.c0;
#pragma unroll
                for (int sub = 0; sub < B0; ++sub) gg[sub] = (16 * sub + j < a.c0) ? f0r[16 * sub + j] : 0.0f;
            };
            float gnext[B0];
            gather(0, gnext);
            for (int ks = 0; ks < nks; ++ks) {
                const float4 A = gA[4 * ks + kq], B = gB[4 * ks + kq];
                float g[B0];
#pragma unroll
                for (int sub = 0; sub < B0; ++sub) g[sub] = gnext[sub];
                if (ks + 1 < nks) gather(ks + 1, gnext);
                float hid[NH];
#pragma unroll
                for (int hb = 0; hb < NH; ++hb) hid[hb] = silu(fmaf(w1v[hb], A.w, b1v[hb]));
                const float ku[4] = {B.x, a.k.k01 * A.x, a.k.k01 * A.y, a.k.k01 * A.z};
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int sub = 0; sub < B0; ++sub) {
                        const float c = ku[s] * g[sub];
                        bs[s * B0 + sub] += c;
#pragma unroll
                        for (int hb = 0; hb < NH; ++hb) acc[s * B0 + sub][hb] = mfma16(hid[hb], c, acc[s * B0 + sub][hb]);
                    }
            }
            wave_sync();                                          // the next chunk overwrites the geometry
        }
        // D[h][entry]: lane (entry = j, g = kq) holds h = 16 hb + 4 g + r
#pragma unroll
        for (int t = 0; t < 4 * B0; ++t) {
            const size_t ent = (size_t)n * a.lc + 16 * t + j;
#ifdef EQ_IR_DBG_NOSTORE
            if (acc[t][0][0] == 123.456f)
#endif
#pragma unroll
            for (int hb = 0; hb < NH; ++hb) st4(a.M + ent * H + 16 * hb + 4 * kq, acc[t][hb]);
            const float tot = gsum(bs[t]);
            if (kq == 0) a.Mb[ent] = tot;
        }
        return;
    }
    // ---- c10 = K10 . f1 (slot 0) | c11[a, J] = K11[a, :, J] . f1 (slot 1 + 3 a + J): five 16-entry blocks per wave
    f32x4 acc[5][NH];
    float bs[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        bs[t] = 0.0f;
#pragma unroll
        for (int hb = 0; hb < NH; ++hb) acc[t][hb] = splat4(0.0f);
    }
    // C1P = 8: lane halves hold slots 2 t | 2 t + 1 of channel j & 7; C1P >= 16: slots S0 + t (S0 = 0 / 5) of channel 16 sub + j
    const int q = unit - 1;
    const int S0 = (C1P == 8) ? 0 : 5 * (q & 1), sub = (C1P == 8) ? 0 : (q >> 1);
    const int ch = (C1P == 8) ? (j & 7) : 16 * sub + j;
    const bool upper = (C1P == 8) && (j >= 8);
    const bool ch_ok = ch < a.c1;
    for (int base = lo; base < hi; base += 64) {
        const int cnt = min(64, hi - base);
        front(base, cnt);
        wave_sync();
        const int nks = (cnt + 3) >> 2;
        auto gather = [&](int ks, float (&gg)[3]) {
            const float* f1r = a.f1 + ((size_t)__float_as_int(gB[4 * ks + kq].w) * a.c1 + ch) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) gg[c] = ch_ok ? f1r[c] : 0.0f;
        };
        float gnext[3];
        gather(0, gnext);
        for (int ks = 0; ks < nks; ++ks) {
            const float4 A = gA[4 * ks + kq], B = gB[4 * ks + kq];
            const float g0 = gnext[0], g1 = gnext[1], g2 = gnext[2];
            if (ks + 1 < nks) gather(ks + 1, gnext);
            float hid[NH];
#pragma unroll
            for (int hb = 0; hb < NH; ++hb) hid[hb] = silu(fmaf(w1v[hb], A.w, b1v[hb]));
            const float sd = A.x * g0 + A.y * g1 + A.z * g2;                       // u . g
            const float cr[3] = {g1 * A.z - g2 * A.y, g2 * A.x - g0 * A.z, g0 * A.y - g1 * A.x};      // g x u = eps_abc g_b u_c
            const float gv[3] = {g0, g1, g2}, uv[3] = {A.x, A.y, A.z};
            float ent[10];
            ent[0] = a.k.k10 * sd;
#pragma unroll
            for (int aa = 0; aa < 3; ++aa) {
                ent[1 + 3 * aa] = B.y * gv[aa];
                ent[2 + 3 * aa] = a.k.k111 * cr[aa];
                ent[3 + 3 * aa] = B.z * (uv[aa] * sd - gv[aa] * (1.0f / 3.0f));
            }
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                float c;
                if (C1P == 8) c = upper ? ent[2 * t + 1] : ent[2 * t];
                else c = S0 ? ent[5 + t] : ent[t];
                bs[t] += c;
#pragma unroll
                for (int hb = 0; hb < NH; ++hb) acc[t][hb] = mfma16(hid[hb], c, acc[t][hb]);
            }
        }
        wave_sync();
    }
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int e_in = (C1P == 8) ? 16 * t + j : (S0 + t) * a.c1p + 16 * sub + j;
        const size_t ent = (size_t)n * a.lc + 4 * a.c0p + e_in;
#ifdef EQ_IR_DBG_NOSTORE
        if (acc[t][0][0] == 123.456f)
#endif
#pragma unroll
        for (int hb = 0; hb < NH; ++hb) st4(a.M + ent * H + 16 * hb + 4 * kq, acc[t][hb]);
        const float tot = gsum(bs[t]);
        if (kq == 0) a.Mb[ent] = tot;
    }
}

// ------------------------------------------------------------------------------------------------ node GEMM: out = W M
// One path's radial weights inside the parameter tensors: W(out, in, h) = w[out * so + in * si + h], bias(out, in) = b[(out * so
// + in * si) / H]  (path 11: + J * H, + J).  Forward: so = C_in * H (x 3 for path 11), si = H (x 3); the adjoint swaps the roles.
struct PathW { const float* w; const float* b; int so, si; };
struct SelfW { const float* w; int so, si; };
struct NodeArgs {
    const float* M; const float* Mb; const float* f0; const float* f1;     // moments of the c0 / c1 INPUT channels; node features
    PathW p00, p10, p01, p11;       // p10: degree-1 input -> degree-0 output; p01: degree-0 input -> degree-1 output
    SelfW s0, s1;
    int n_nodes, c0, c1, c0p, c1p, lc, o0, o1;
    float* out0; float* out1;       // [N, o0], [N, o1, 3]
};

#ifndef EQ_IR_NODE_WAVES
#define EQ_IR_NODE_WAVES 16             // one pair has 256 node tiles: 16 waves per workgroup = four per SIMD (32.3 -> 29.1 us against 8)
#endif
constexpr int kNodeWaves = EQ_IR_NODE_WAVES;      // irrep_node16_kernel: waves per workgroup (they split K)
constexpr int kExpWaves = 8;                       // irrep_expand16_kernel: waves per workgroup (one entry each per group)

template <int NH, int NCB0, int NCB1>
__global__ __launch_bounds__(64 * kNodeWaves) void irrep_node16_kernel(const NodeArgs a) {
    constexpr int H = 16 * NH, NT = NCB0 + 3 * NCB1;
    __shared__ f32x4 part[kNodeWaves][NT][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int nA = min(n0 + i16, a.n_nodes - 1);                 // the A operand's node (rows beyond the end repeat the last one)
    f32x4 acc0[NCB0], acc1[3][NCB1];
#pragma unroll
    for (int cb = 0; cb < NCB0; ++cb) acc0[cb] = splat4(0.0f);
#pragma unroll
    for (int aa = 0; aa < 3; ++aa)
#pragma unroll
        for (int cb = 0; cb < NCB1; ++cb) acc1[aa][cb] = splat4(0.0f);
    const float* Mn = a.M + (size_t)nA * a.lc * H;
    const float* Mbn = a.Mb + (size_t)nA * a.lc;
    const int f1base = 4 * a.c0p;
    // B operand of a weight row: W(o, in, 16 t + 4 kq + r); rows beyond the output count are zero
    auto wrow = [&](const float* w, int so, int si, int o, int on, int in, f32x4 (&bv)[NH]) {
        const float* p = w + (size_t)min(o, on - 1) * so + (size_t)in * si;
#pragma unroll
        for (int t = 0; t < NH; ++t) bv[t] = o < on ? ldg4(p + 16 * t + 4 * kq) : splat4(0.0f);
    };
    auto mrow = [&](int entry, f32x4 (&av)[NH]) {
#pragma unroll
        for (int t = 0; t < NH; ++t) av[t] = ldg4(Mn + (size_t)entry * H + 16 * t + 4 * kq);
    };
    // ---- degree-0 outputs: K = (i, h) over c00 and c10.  The operands of step i + kNodeWaves are requested before the products of
    // step i (two waves per SIMD: an exposed L2 / HBM round trip per step was most of this kernel's time)
    {
        const int n0s = a.c0 + a.c1;
        f32x4 avn[NH], bvn[NCB0][NH];
        auto fetch0 = [&](int i) {
            const bool d0 = i < a.c0;
            const int in = d0 ? i : i - a.c0;
            mrow(d0 ? in : f1base + in, avn);
            const PathW& p = d0 ? a.p00 : a.p10;
#pragma unroll
            for (int cb = 0; cb < NCB0; ++cb) wrow(p.w, p.so, p.si, 16 * cb + i16, a.o0, in, bvn[cb]);
        };
        if (wv < n0s) fetch0(wv);
        for (int i = wv; i < n0s; i += kNodeWaves) {
            f32x4 av[NH], bv[NCB0][NH];
#pragma unroll
            for (int t = 0; t < NH; ++t) {
                av[t] = avn[t];
#pragma unroll
                for (int cb = 0; cb < NCB0; ++cb) bv[cb][t] = bvn[cb][t];
            }
            if (i + kNodeWaves < n0s) fetch0(i + kNodeWaves);
#pragma unroll
            for (int cb = 0; cb < NCB0; ++cb)
#pragma unroll
                for (int t = 0; t < NH; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc0[cb] = mfma16(av[t][r], bv[cb][t][r], acc0[cb]);
        }
    }
    // ---- degree-1 outputs: per component a the same weights act on the moments of c01[a] / c11[a, J]
    {
        const int n1s = a.c0 + 3 * a.c1;
        f32x4 avn[3][NH], bvn[NCB1][NH];
        auto fetch1 = [&](int i) {
            const bool d0 = i < a.c0;
            const int in = d0 ? i : (i - a.c0) / 3, J = d0 ? 0 : (i - a.c0) - 3 * in;
#pragma unroll
            for (int cb = 0; cb < NCB1; ++cb) {
                if (d0) wrow(a.p01.w, a.p01.so, a.p01.si, 16 * cb + i16, a.o1, in, bvn[cb]);
                else wrow(a.p11.w + J * H, a.p11.so, a.p11.si, 16 * cb + i16, a.o1, in, bvn[cb]);
            }
#pragma unroll
            for (int aa = 0; aa < 3; ++aa) mrow(d0 ? a.c0p * (1 + aa) + in : f1base + a.c1p * (1 + 3 * aa + J) + in, avn[aa]);
        };
        if (wv < n1s) fetch1(wv);
        for (int i = wv; i < n1s; i += kNodeWaves) {
            f32x4 av[3][NH], bv[NCB1][NH];
#pragma unroll
            for (int t = 0; t < NH; ++t) {
#pragma unroll
                for (int aa = 0; aa < 3; ++aa) av[aa][t] = avn[aa][t];
#pragma unroll
                for (int cb = 0; cb < NCB1; ++cb) bv[cb][t] = bvn[cb][t];
            }
            if (i + kNodeWaves < n1s) fetch1(i + kNodeWaves);
#pragma unroll
            for (int aa = 0; aa < 3; ++aa)
#pragma unroll
                for (int cb = 0; cb < NCB1; ++cb)
#pragma unroll
                    for (int t = 0; t < NH; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc1[aa][cb] = mfma16(av[aa][t][r], bv[cb][t][r], acc1[aa][cb]);
        }
    }
    // ---- bias column (K = entries, Mb) and self-interaction (K = input channels, the node's own features): four entries per step
    for (int i0 = 4 * wv; i0 < a.c0 + a.c1; i0 += 4 * kNodeWaves) {          // degree-0 outputs
        const int i = i0 + kq;                                               // c0 % 4 == 0 is not required: per-lane path choice
        const bool live = i < a.c0 + a.c1, d0 = i < a.c0;
        const int in = d0 ? i : i - a.c0;
        const float am = live ? Mbn[d0 ? in : f1base + in] : 0.0f;
        const float as = live ? (d0 ? a.f0[(size_t)nA * a.c0 + in] : 0.0f) : 0.0f;
        const PathW& p = d0 ? a.p00 : a.p10;
#pragma unroll
        for (int cb = 0; cb < NCB0; ++cb) {
            const int o = 16 * cb + i16;
            const bool ok = live && o < a.o0;
            const float bm = ok ? p.b[((size_t)o * p.so + (size_t)in * p.si) / H] : 0.0f;
            const float bsf = (ok && d0) ? a.s0.w[(size_t)o * a.s0.so + (size_t)in * a.s0.si] : 0.0f;
            acc0[cb] = mfma16(am, bm, acc0[cb]);
            acc0[cb] = mfma16(as, bsf, acc0[cb]);
        }
    }
    for (int i0 = 4 * wv; i0 < a.c0 + 3 * a.c1; i0 += 4 * kNodeWaves) {      // degree-1 outputs
        const int i = i0 + kq;
        const bool live = i < a.c0 + 3 * a.c1, d0 = i < a.c0;
        const int in = d0 ? i : (i - a.c0) / 3, J = d0 ? 0 : (i - a.c0) - 3 * in;
        const bool self = live && !d0 && J == 0;                             // W_s1 f1[n, in, a]: once per input channel
        float bm[NCB1], bsf[NCB1];
#pragma unroll
        for (int cb = 0; cb < NCB1; ++cb) {
            const int o = 16 * cb + i16;
            const bool ok = live && o < a.o1;
            bm[cb] = !ok ? 0.0f : (d0 ? a.p01.b[((size_t)o * a.p01.so + (size_t)in * a.p01.si) / H]
                                      : a.p11.b[((size_t)o * a.p11.so + (size_t)in * a.p11.si) / H + J]);
            bsf[cb] = (self && o < a.o1) ? a.s1.w[(size_t)o * a.s1.so + (size_t)in * a.s1.si] : 0.0f;
        }
#pragma unroll
        for (int aa = 0; aa < 3; ++aa) {
            const float am = live ? Mbn[d0 ? a.c0p * (1 + aa) + in : f1base + a.c1p * (1 + 3 * aa + J) + in] : 0.0f;
            const float as = self ? a.f1[((size_t)nA * a.c1 + in) * 3 + aa] : 0.0f;
#pragma unroll
            for (int cb = 0; cb < NCB1; ++cb) {
                acc1[aa][cb] = mfma16(am, bm[cb], acc1[aa][cb]);
                acc1[aa][cb] = mfma16(as, bsf[cb], acc1[aa][cb]);
            }
        }
    }
    // ---- the waves' partial tiles are summed in wave order (fixed: bitwise reproducible) and stored; D[node][o]: lane (o = i16,
    // g = kq) holds nodes 4 g + r
#pragma unroll
    for (int cb = 0; cb < NCB0; ++cb) part[wv][cb][lane] = acc0[cb];
#pragma unroll
    for (int aa = 0; aa < 3; ++aa)
#pragma unroll
        for (int cb = 0; cb < NCB1; ++cb) part[wv][NCB0 + aa * NCB1 + cb][lane] = acc1[aa][cb];
    __syncthreads();
    for (int t = wv; t < NT; t += kNodeWaves) {
        f32x4 s = part[0][t][lane];
#pragma unroll
        for (int w = 1; w < kNodeWaves; ++w) s += part[w][t][lane];
        const bool deg0 = t < NCB0;
        const int aa = deg0 ? 0 : (t - NCB0) / NCB1, cb = deg0 ? t : (t - NCB0) - aa * NCB1;
        const int o = 16 * cb + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + 4 * kq + r;
            if (n >= a.n_nodes) continue;
            if (deg0) { if (o < a.o0) a.out0[(size_t)n * a.o0 + o] = s[r]; }
            else if (o < a.o1) a.out1[((size_t)n * a.o1 + o) * 3 + aa] = s[r];
        }
    }
}

// (Round 5's opt-in `irrep_conv_fused_kernel` -- both launches above in one, the moments kept in an LDS tile -- was deleted in round 6:
// 77 us against 39 + 30 us at one pair, equal at 16 pairs (its phase A re-walks every node's edges once per operand chunk at two
// waves per SIMD), a third code path that nothing used by default.  DESIGN.md section 9 keeps the measurement.)

// ------------------------------------------------------------------------------------------------ backward: dM = W^T g
// The gradient of the moments (for the geometry backward): dM[n, j, h] = sum_o W(o, j, h) g[n, o] with the path's output gradient
// (g0 for c00 / c10, g1[:, :, a] for c01[a] / c11[a, J]); dMb likewise with the biases.  Padded entries are written as zeros.
// D[node][h] = sum_o A[node][o] B[o][h]: the 16 nodes' gradient rows stay in registers as A operands for the whole tile.
struct ExpArgs {
    const float* g0; const float* g1;                    // [N, o0], [N, o1, 3]
    PathW p00, p10, p01, p11;                            // forward descriptors
    int n_nodes, c0, c1, c0p, c1p, lc, o0, o1;
    float* dM; float* dMb;                               // [N, lc, H], [N, lc]
};

constexpr int kExpLd = kExpWaves * 32 + 4;      // floats between nodes of the staging buffer (= 4 mod 32: conflict-free 16-byte writes of 8 nodes)
constexpr int kExpSplit = 2;            // workgroups per node tile (each takes a contiguous share of the entry groups): four waves per SIMD at one pair instead of two

// NKS0 / NKS1: K steps (four output channels each) of the paths with degree-0 / degree-1 OUTPUT gradients -- compile-time, so every
// entry of a run requests the same number of loads and the next entry's weights can stay in flight across this entry's products.
template <int NH, int NKS0, int NKS1>
__global__ __launch_bounds__(64 * kExpWaves) void irrep_expand16_kernel(const ExpArgs a) {
    constexpr int H = 16 * NH;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    // B operands: g[node][4 s + kq] (output channels beyond the count: zero)
    const int nA = min(n0 + i16, a.n_nodes - 1);
    float g0r[NKS0], g1r[3][NKS1];
#pragma unroll
    for (int s = 0; s < NKS0; ++s) {
        const int o = 4 * s + kq;
        g0r[s] = keep_if(a.g0[(size_t)nA * a.o0 + min(o, a.o0 - 1)], o < a.o0);
    }
#pragma unroll
    for (int s = 0; s < NKS1; ++s) {
        const int o = 4 * s + kq;
#pragma unroll
        for (int aa = 0; aa < 3; ++aa) g1r[aa][s] = keep_if(a.g1[((size_t)nA * a.o1 + min(o, a.o1 - 1)) * 3 + aa], o < a.o1);
    }
    // this workgroup's share of the groups of 8 entries
    const int n_grp = a.lc / kExpWaves, per = (n_grp + kExpSplit - 1) / kExpSplit;
    const int e_lo = (int)blockIdx.y * per * kExpWaves, e_hi = min(a.lc, e_lo + per * kExpWaves);
    // The tiles leave through LDS: a wave's own tile is 16 nodes x 64 bytes per store instruction, 27 KB apart (measured: 110 MB in
    // 92 us = 1.2 TB/s); staged as [node][8 entries][H] the workgroup writes every node's 8 entries as ONE contiguous 1 KB row per
    // instruction.  Two buffers: one barrier per group of 8 entries.
    __shared__ __attribute__((aligned(16))) float stage[2][16 * kExpLd];
    int buf = 0;
    // One RUN = the entries of one path (c00 | c01 x 3 | c10 | c11 x 9): descriptor, output-gradient degree and K-step count fixed.
    // blockw: entries per slot (c0p / c1p); slots: aa (c01) or 3 aa + J (c11).
    auto run = [&](auto nks_tag, const bool deg1, const bool hasJ, int rb, int re, const float* pw, const float* pb, int so, int si, int on,
                   int cin, int blockw) {
        constexpr int NKS = decltype(nks_tag)::value;
        struct Ent { float w[NKS][NH], b[NKS]; };
        const int lo = max(rb, e_lo), hi = min(re, e_hi);
        if (lo >= hi) return;                                               // workgroup-uniform
        const int soh = so / H;
        auto fetch = [&](int e, Ent& E_) {                                  // e wave-uniform; NKS * (NH + 1) loads, always
            const int q = e - rb, slot = q / blockw;
            const int in = min(q - slot * blockw, cin - 1);                 // pad entries read the last real row (never used)
            const int J = hasJ ? slot % 3 : 0;
            const float* wp = pw + (size_t)in * si + J * H + i16;
            const float* bp = pb + ((size_t)in * si) / H + J;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                const int o = 4 * s + kq, oc = min(o, on - 1);
#pragma unroll
                for (int t = 0; t < NH; ++t) E_.w[s][t] = wp[(size_t)oc * so + 16 * t];      // (rows past `on`: clamped, and their
                E_.b[s] = bp[(size_t)oc * soh];                                               //  B operand g is zero)
            }
        };
        Ent nxt;
        const int last = hi - kExpWaves + wv;                               // this wave's last entry of the run
        fetch(lo + wv, nxt);
        for (int e0 = lo; e0 < hi; e0 += kExpWaves, buf ^= 1) {
            const int e = e0 + wv;
            const Ent cur = nxt;
            fetch(min(e + kExpWaves, last), nxt);                           // (the last one again at the end: unconditional)
            const int q = e - rb, slot = q / blockw;
            const bool real = q - slot * blockw < cin;
            const int aa = hasJ ? slot / 3 : slot;                          // component of g1 (deg1 runs)
            f32x4 acc[NH];
            float bsum = 0.0f;
#pragma unroll
            for (int t = 0; t < NH; ++t) acc[t] = splat4(0.0f);
            if (real) {
                // D[h][node] = sum_o A[h][o] B[o][node]
#pragma unroll
                for (int s = 0; s < NKS; ++s) {
                    // (the component's row by masks, not by `?:` -- the compiler turns a select chain over g1r into an indexed
                    // scratch array, and a scratch load is one more vmcnt(0))
                    float gvv;
                    const float g1v = __int_as_float((__float_as_int(g1r[0][s]) & (aa == 0 ? -1 : 0)) | (__float_as_int(g1r[1][s]) & (aa == 1 ? -1 : 0)) |
                                                     (__float_as_int(g1r[2][s]) & (aa == 2 ? -1 : 0)));
                    if constexpr (NKS == NKS0 && NKS == NKS1) gvv = deg1 ? g1v : g0r[s];
                    else if constexpr (NKS == NKS0) gvv = g0r[s];
                    else gvv = g1v;
#pragma unroll
                    for (int t = 0; t < NH; ++t) acc[t] = mfma16(cur.w[s][t], gvv, acc[t]);
                    bsum = fmaf(gvv, cur.b[s], bsum);
                }
            }
            // lane (node = i16, g = kq) holds h = 16 t + 4 g + r  ->  stage[node][wv][h]
            float* st = stage[buf];
            const float bt = gsum(bsum);
#pragma unroll
            for (int t = 0; t < NH; ++t) st4(st + i16 * kExpLd + wv * H + 16 * t + 4 * kq, acc[t]);
            if (kq == 0 && n0 + i16 < a.n_nodes) a.dMb[(size_t)(n0 + i16) * a.lc + e] = bt;
            __syncthreads();
            // rows out: wave w takes nodes w, w + 8; 8 entries x H floats = one contiguous run per node
            constexpr int RUN = kExpWaves * H;                             // floats per node and group (256 at H = 32: one 1 KB instruction)
            for (int nl = wv; nl < 16; nl += kExpWaves) {
                const int n = n0 + nl;
                if (n >= a.n_nodes) continue;                               // wave-uniform
                for (int c = 4 * lane; c < RUN; c += 256)
                    st4(a.dM + ((size_t)n * a.lc + e0) * H + c, lds4(st + nl * kExpLd + c));
            }
        }
    };
    const int f1base = 4 * a.c0p;
    run(std::integral_constant<int, NKS0>{}, false, false, 0, a.c0p, a.p00.w, a.p00.b, a.p00.so, a.p00.si, a.o0, a.c0, a.c0p);
    run(std::integral_constant<int, NKS1>{}, true, false, a.c0p, f1base, a.p01.w, a.p01.b, a.p01.so, a.p01.si, a.o1, a.c0, a.c0p);
    if (a.c1 > 0) {
        run(std::integral_constant<int, NKS0>{}, false, false, f1base, f1base + a.c1p, a.p10.w, a.p10.b, a.p10.so, a.p10.si, a.o0, a.c1, a.c1p);
        run(std::integral_constant<int, NKS1>{}, true, true, f1base + a.c1p, a.lc, a.p11.w, a.p11.b, a.p11.so, a.p11.si, a.o1, a.c1, a.c1p);
    } else {
        // no active degree-1 input: its entries' gradients are never read (the geometry kernel skips the section)
    }
}

// ------------------------------------------------------------------------------------------------ backward: geometry + radial.0
// Per edge e (row = n):  d c[e, j] = sum_h dM[n, j, h] hid[e, h] + dMb[n, j],   d hid[e, h] = sum_j dM[n, j, h] c[e, j];  from d c the
// gradient of the unit vector u (the closed-form basis is linear / quadratic in u), from d hid the gradient of d = |r| (through
// SiLU and w_r1) and the sums d w_r1, d b_r1;  d r = dd u + (du - (du . u) u) / d  (zero on a self loop).
// A wave per node; its edges in blocks of 16: lane (e = l & 15, g = l >> 4) owns edge e and the four entries 16 blk + 4 g + r of
// every 16-entry block -- the C/D layout of D[entry][edge] (d c) AND the B operand of D[h][edge] = sum_j dM[j][h] c[e][j] (d hid),
// so c is computed once, in the lane that consumes it both ways.  The block's dM tile goes through a wave-private LDS tile (read as
// rows for d c, as columns for d hid).
struct GeomArgs {
    const float* f0; const float* f1;
    const int* ptr; const int* perm; EdgeTab tab;
    const float* w1; const float* b1;
    Consts k;
    int n_nodes, c0, c1, c0p, c1p, lc;
    const float* dM; const float* dMb;
    float* dr;          // [E, 4]: d loss / d r_e (the caller scatters +dr to row, -dr to col)
    float* dxr;         // [N, 4]: sum of dr over the node's row segment
    float* slab;        // [ceil(N / 4), 2 H]: per-workgroup sums of d w_r1 | d b_r1
};

// contribution of one f1-part slot to the operand c and (given D = d c) to du, for one channel's vector gv; SLOT: 0 = K10 . f1,
// 1 + 3 a + J = K11[a, :, J] . f1
template <int SLOT>
__device__ __forceinline__ float slot_c(const float (&gv)[3], const float (&u)[3], float sd, const Consts& k, float k110v, float k112l) {
    if (SLOT == 0) return k.k10 * sd;
    constexpr int aa = (SLOT - 1) / 3, J = (SLOT - 1) % 3;
    if (J == 0) return k110v * gv[aa];
    if (J == 1) return k.k111 * (gv[(aa + 1) % 3] * u[(aa + 2) % 3] - gv[(aa + 2) % 3] * u[(aa + 1) % 3]);
    return k112l * (u[aa] * sd - gv[aa] * (1.0f / 3.0f));
}
template <int SLOT>
__device__ __forceinline__ void slot_du(const float (&gv)[3], const float (&u)[3], float sd, const Consts& k, float k112l, float D,
                                        float (&du)[3]) {
    if (SLOT == 0) {
        const float t = k.k10 * D;
#pragma unroll
        for (int c = 0; c < 3; ++c) du[c] = fmaf(t, gv[c], du[c]);
        return;
    }
    constexpr int aa = (SLOT - 1) / 3, J = (SLOT - 1) % 3;
    if (J == 1) {                 // c = k111 (g x u)_a: d/du = k111 D (e_a x g)
        const float t = k.k111 * D;
        du[(aa + 1) % 3] = fmaf(-t, gv[(aa + 2) % 3], du[(aa + 1) % 3]);
        du[(aa + 2) % 3] = fmaf(t, gv[(aa + 1) % 3], du[(aa + 2) % 3]);
    } else if (J == 2) {          // c = k112 (u_a (u . g) - g_a / 3): d/du = k112 D ((u . g) e_a + u_a g)
        const float t = k112l * D;
        du[aa] = fmaf(t, sd, du[aa]);
        const float tu = t * u[aa];
#pragma unroll
        for (int c = 0; c < 3; ++c) du[c] = fmaf(tu, gv[c], du[c]);
    }
}

constexpr int kTileStride = 36;      // dM tile [16 entries][H <= 32] in LDS: rows 36 floats apart (16-byte aligned rows, column reads spread over the banks)

#ifndef EQ_IR_GEOM_OCC
#define EQ_IR_GEOM_OCC 2
#endif
template <int NH, int C1P>
__global__ __launch_bounds__(256, EQ_IR_GEOM_OCC) void irrep_geom_bwd_kernel(const GeomArgs a) {
    constexpr int H = 16 * NH;
    __shared__ float tile_s[4][16 * kTileStride];
    __shared__ float red[4][2 * 32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wv;
    const int i16 = lane & 15, g = lane >> 4;
    float* tile = tile_s[wv];
    // lane-owned radial hidden units in the two layouts: B operand of d c (h = 16 t + 4 g + r) = C/D rows of d hid
    float w1v[NH][4], b1v[NH][4], aw[NH][4], ab[NH][4];
#pragma unroll
    for (int t = 0; t < NH; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            w1v[t][r] = a.w1[16 * t + 4 * g + r];
            b1v[t][r] = a.b1[16 * t + 4 * g + r];
            aw[t][r] = ab[t][r] = 0.0f;
        }
    const bool node_ok = n < a.n_nodes;                      // wave-uniform
    const int nn = node_ok ? n : a.n_nodes - 1;
    const int lo = node_ok ? a.ptr[nn] : 0, hi = node_ok ? a.ptr[nn + 1] : 0;
    const float* dMn = a.dM + (size_t)nn * a.lc * H;
    const float* dMbn = a.dMb + (size_t)nn * a.lc;
    const int f1base = 4 * a.c0p, B0 = a.c0p >> 4;
    float xs0 = 0.0f, xs1 = 0.0f, xs2 = 0.0f;               // sum of dr over the node's edges
    // the table rows of a block's edges: unconditional loads (clamped position, masked below), the NEXT block's requested while this
    // one computes (three dependent round trips per block otherwise: position -> source -> feature row)
    int eid_n = 0, s_n = 0;
    float4 gq_n = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto front = [&](int base_) {
        const int p = min(base_ + i16, hi - 1);
        eid_n = a.perm[p]; gq_n = a.tab.geo[p]; s_n = a.tab.src[p];
    };
    if (lo < hi) front(lo);
    for (int base = lo; base < hi; base += 16) {
        const int cnt = min(16, hi - base);
        const bool valid = i16 < cnt;
        const float vf = valid ? 1.0f : 0.0f;
        const int eid = eid_n, s = s_n;                               // (a padding lane keeps a real source row: its factors are zero)
        const float4 gq = make_float4(gq_n.x * vf, gq_n.y * vf, gq_n.z * vf, gq_n.w * vf);
        front(base + 16);
        const float d = gq.w;
        const bool live = valid && d > 0.0f;
        const float inv = live ? 1.0f / d : 0.0f;
        const float u[3] = {gq.x, gq.y, gq.z};
        const float k00v = valid ? a.k.k00 : 0.0f, k110v = valid ? a.k.k110 : 0.0f, k112l = live ? a.k.k112 : 0.0f;
        // hid[e][h] for the lane's h = 16 t + 4 g + r (B operand of d c), and SiLU' for the d hid epilogue
        f32x4 hid[NH], dsl[NH];
#pragma unroll
        for (int t = 0; t < NH; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = fmaf(w1v[t][r], d, b1v[t][r]);
                hid[t][r] = silu(z);
                dsl[t][r] = silu_grad(z);
            }
        f32x4 dh[NH];
#pragma unroll
        for (int t = 0; t < NH; ++t) dh[t] = splat4(0.0f);
        float du[3] = {0.0f, 0.0f, 0.0f};
        f32x4 apre[NH], bpre;
        auto fetch = [&](int blk) {
#pragma unroll
            for (int t = 0; t < NH; ++t) apre[t] = ldg4(dMn + ((size_t)(16 * blk + i16)) * H + 16 * t + 4 * g);
            bpre = ldg4(dMbn + 16 * blk + 4 * g);
        };
        fetch(0);
        const int f1b = f1base >> 4;
        // one 16-entry block: stage the dM tile, d c (optional), d hid; `cv` = the lane's four operand values c[e][16 blk + 4 g + r]
        // (the tile of block `next` is requested while this block computes: tile rows: lane (j = i16, kq = g) loads
        // dM[n][16 blk + j][16 t + 4 g ..]; with two waves per SIMD nothing else covers the round trip)
        auto block = [&](int blk, int next, const f32x4& cv, bool want_dc, f32x4& dc) {
            f32x4 arow[NH];
#pragma unroll
            for (int t = 0; t < NH; ++t) arow[t] = apre[t];
            const f32x4 brow = bpre;
            fetch(next);
#pragma unroll
            for (int t = 0; t < NH; ++t) st4(tile + i16 * kTileStride + 16 * t + 4 * g, arow[t]);
            wave_sync();
            if (want_dc) {
                dc = brow;                                                        // D[entry][edge]: lane (e, g) holds entries 4 g + r
#pragma unroll
                for (int t = 0; t < NH; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dc = mfma16(arow[t][r], hid[t][r], dc);
            }
            // d hid: D[h][e] += sum_j dM[j][h] c[e][j]; k-step r: A lane (h = i16, kq = g) = dM[16 blk + 4 g + r][16 t + h]
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < NH; ++t) dh[t] = mfma16(tile[(4 * g + r) * kTileStride + 16 * t + i16], cv[r], dh[t]);
            wave_sync();                                                          // the next block overwrites the tile
        };
        // ---- entries built from f0: c00 = K00 f0 (no u in it: d c not needed), c01[a] = k01 u_a f0
        const float* f0r = a.f0 + (size_t)s * a.c0;
        for (int sub = 0; sub < B0; ++sub) {
            const int i0 = 16 * sub + 4 * g;
            f32x4 gv;
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] = keep_if(f0r[min(i0 + r, a.c0 - 1)], i0 + r < a.c0);
            f32x4 dc;
            block(sub, B0 + sub, gv * splat4(k00v), false, dc);
#pragma unroll
            for (int aa = 0; aa < 3; ++aa) {
                block((1 + aa) * B0 + sub, aa < 2 ? (2 + aa) * B0 + sub : (sub + 1 < B0 ? sub + 1 : f1b), gv * splat4(a.k.k01 * u[aa]), true, dc);
                du[aa] = fmaf(a.k.k01, dc[0] * gv[0] + dc[1] * gv[1] + dc[2] * gv[2] + dc[3] * gv[3], du[aa]);
            }
        }
        // ---- entries built from f1 (none active: the first layer of a stack whose input has no degree-1 features)
        if (a.c1 == 0) {
        } else if (C1P == 8) {
            // block t holds slots 2 t (lane groups g = 0, 1) and 2 t + 1 (g = 2, 3) of channels 4 (g & 1) + r
            const int ch0 = 4 * (g & 1);
            const bool up = g >= 2;
            float gv[4][3], sd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = ch0 + r < a.c1;
                const float* f1r = a.f1 + ((size_t)s * a.c1 + min(ch0 + r, a.c1 - 1)) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) gv[r][c] = keep_if(f1r[c], ok);
                sd[r] = u[0] * gv[r][0] + u[1] * gv[r][1] + u[2] * gv[r][2];
            }
            auto pair = [&](auto tag) {
                constexpr int t = decltype(tag)::value;
                f32x4 cv, dc;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float lo_ = slot_c<2 * t>(gv[r], u, sd[r], a.k, k110v, k112l);
                    const float hi_ = slot_c<2 * t + 1>(gv[r], u, sd[r], a.k, k110v, k112l);
                    cv[r] = up ? hi_ : lo_;
                }
                block(f1b + t, t < 4 ? f1b + t + 1 : 0, cv, true, dc);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    slot_du<2 * t>(gv[r], u, sd[r], a.k, k112l, up ? 0.0f : dc[r], du);
                    slot_du<2 * t + 1>(gv[r], u, sd[r], a.k, k112l, up ? dc[r] : 0.0f, du);
                }
            };
            pair(std::integral_constant<int, 0>{}); pair(std::integral_constant<int, 1>{}); pair(std::integral_constant<int, 2>{});
            pair(std::integral_constant<int, 3>{}); pair(std::integral_constant<int, 4>{});
        } else {
            constexpr int SB = C1P / 16;
            for (int sub = 0; sub < SB; ++sub) {
                const int ch0 = 16 * sub + 4 * g;
                float gv[4][3], sd[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = ch0 + r < a.c1;
                    const float* f1r = a.f1 + ((size_t)s * a.c1 + min(ch0 + r, a.c1 - 1)) * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) gv[r][c] = keep_if(f1r[c], ok);
                    sd[r] = u[0] * gv[r][0] + u[1] * gv[r][1] + u[2] * gv[r][2];
                }
                auto one = [&](auto tag) {
                    constexpr int SL = decltype(tag)::value;
                    f32x4 cv, dc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) cv[r] = slot_c<SL>(gv[r], u, sd[r], a.k, k110v, k112l);
                    constexpr bool needs = (SL == 0) || ((SL - 1) % 3 != 0);       // J = 0 has no u in it
                    block(f1b + SL * SB + sub, SL < 9 ? f1b + (SL + 1) * SB + sub : (sub + 1 < SB ? f1b + sub + 1 : 0), cv, needs, dc);
                    if (needs) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) slot_du<SL>(gv[r], u, sd[r], a.k, k112l, dc[r], du);
                    }
                };
                one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{});
                one(std::integral_constant<int, 3>{}); one(std::integral_constant<int, 4>{}); one(std::integral_constant<int, 5>{});
                one(std::integral_constant<int, 6>{}); one(std::integral_constant<int, 7>{}); one(std::integral_constant<int, 8>{});
                one(std::integral_constant<int, 9>{});
            }
        }
        // ---- the edge's totals: d hid is D[h][e] with lane (e, g) holding h = 16 t + 4 g + r
        float dd = 0.0f;
#pragma unroll
        for (int t = 0; t < NH; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float q = dh[t][r] * dsl[t][r];          // d loss / d z[e, h]
                dd = fmaf(q, w1v[t][r], dd);
                aw[t][r] = fmaf(q, d, aw[t][r]);
                ab[t][r] += q;
            }
        dd = gsum(dd);
        du[0] = gsum(du[0]); du[1] = gsum(du[1]); du[2] = gsum(du[2]);
        const float dun = du[0] * u[0] + du[1] * u[1] + du[2] * u[2];
        float dr[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) dr[c] = live ? fmaf(dd, u[c], (du[c] - dun * u[c]) * inv) : 0.0f;
        if (valid && g == 0) reinterpret_cast<float4*>(a.dr)[eid] = make_float4(dr[0], dr[1], dr[2], 0.0f);
        xs0 += row16_sum(dr[0]); xs1 += row16_sum(dr[1]); xs2 += row16_sum(dr[2]);
    }
    if (node_ok && lane == 0) reinterpret_cast<float4*>(a.dxr)[n] = make_float4(xs0, xs1, xs2, 0.0f);
    // ---- d w_r1 / d b_r1: sum over the wave's 16 edge lanes, then over the workgroup's waves in wave order
#pragma unroll
    for (int t = 0; t < NH; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sw = row16_sum(aw[t][r]), sb = row16_sum(ab[t][r]);
            if (i16 == 0) { red[wv][16 * t + 4 * g + r] = sw; red[wv][32 + 16 * t + 4 * g + r] = sb; }
        }
    __syncthreads();
    if (threadIdx.x < 2 * H) {
        const int which = threadIdx.x / H, h = threadIdx.x - which * H;
        const float v = red[0][32 * which + h] + red[1][32 * which + h] + red[2][32 * which + h] + red[3][32 * which + h];
        a.slab[(size_t)blockIdx.x * 2 * H + threadIdx.x] = v;
    }
}

// d x[n] += (sum of dr over the edges with row = n) - (sum over the edges with col = n): r = x[row] - x[col]
__global__ void irrep_dx_kernel(const float* __restrict__ dr, const float* __restrict__ dxr, const int* __restrict__ cptr,
                                const int* __restrict__ cperm, int n_nodes, float* __restrict__ dx) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    float4 acc = reinterpret_cast<const float4*>(dxr)[n];
    for (int p = cptr[n]; p < cptr[n + 1]; ++p) {
        const float4 v = reinterpret_cast<const float4*>(dr)[cperm[p]];
        acc.x -= v.x; acc.y -= v.y; acc.z -= v.z;
    }
    dx[(size_t)n * 3] += acc.x; dx[(size_t)n * 3 + 1] += acc.y; dx[(size_t)n * 3 + 2] += acc.z;
}

// ------------------------------------------------------------------------------------------------ backward: weight gradients
// d W(o, j, h) = sum_n g[n, o] M[n, j, h] (+ the three components a for the degree-1 outputs), d bias likewise with Mb, d W_self
// with the node's own features: GEMMs with K = nodes.  A wave owns up to four weight rows ("group": same path) and a node range;
// D[h][o] = sum_n A[h][n] B[n][o] with A = the moment rows (lane (h = l & 15, k = l >> 4): M[n0 + k][entry][16 t + h]) and
// B = the output gradients; every node range writes its own slab in the parameter layout, summed later in fixed order.
struct WgradArgs {
    const float* M; const float* Mb; const float* f0; const float* f1; const float* g0; const float* g1;
    int n_nodes, c0, c1, c0p, c1p, lc, o0, o1, H;
    int off00, off01, off10, off11, n_w;      // row offsets inside radial.2
    int nodes_per_split;
    float* slab;                                // [splits][P]: d W_r2 [n_w, H] | d b_r2 [n_w] | d W_s0 [o0, c0] | d W_s1 [o1, c1]
    int P;
};

// groups: [0, G00): R00 rows i (4 per group) | [.., +G10): R10 | [.., +G01): R01 | [.., +G11): R11 rows (i, J) | self0 blocks | self1 blocks
#ifndef EQ_IR_WG_ROWS
#define EQ_IR_WG_ROWS 4                 // weight rows per wave (accumulator tiles = rows x NH x 2)
#endif
constexpr int kWgRows = EQ_IR_WG_ROWS;
#ifndef EQ_IR_WG_OCC
#define EQ_IR_WG_OCC 4                  // waves per SIMD the register budget is held to
#endif
// The radial.2 rows of one wave: D[h][o] += sum over (node, component) of M[node][entry(row, component)][h] g[node][o][component].
// One STEP = four nodes (the MFMA's K) x one component.  Straight-line body: the rows past the path's end read its last row (valid
// memory, results never stored), so every load of a step is requested before the first MFMA, and the NEXT step's loads before this
// step's MFMAs (the version with a break per row made the compiler wait for each 64-byte load in turn: 8 memory latencies per step,
// 2.2 us per step, 90 us per launch).
template <int NH, int NCB, int NAA>
__device__ __forceinline__ void wgrad_rows(const WgradArgs& a, int kind, int grp, int nb, int ne, float* __restrict__ slab) {
    constexpr int H = 16 * NH;
    const int lane = threadIdx.x, i16 = lane & 15, kq = lane >> 4;
    const int cin = (kind == 0 || kind == 2) ? a.c0 : a.c1;
    const int on = NAA == 3 ? a.o1 : a.o0;
    // rows q = 0 .. kWgRows - 1 are CONSECUTIVE entries of one 16-padded block of the operand vector (R11: consecutive input channels
    // of one J), so one address per node serves all of them by immediate offsets; rows past the path's end read pad entries of the same
    // block (allocated; whatever they hold stays in accumulators that are never stored)
    int in0, J = 0;
    if (kind == 3) { const int gin = (a.c1 + kWgRows - 1) / kWgRows; J = grp / gin; in0 = kWgRows * (grp - J * gin); }
    else in0 = kWgRows * grp;
    const int ent0 = (kind == 0 ? 0 : kind == 1 ? 4 * a.c0p : kind == 2 ? a.c0p : 4 * a.c0p + a.c1p * (1 + J)) + in0;
    const int astride = kind == 2 ? a.c0p : 3 * a.c1p;
    int gcol[NCB];                                    // this lane's output channel per column block (clamped; columns past `on` are not stored)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) gcol[cb] = min(16 * cb + i16, on - 1);
    const float* gbase = NAA == 3 ? a.g1 : a.g0;
    const int gstride = NAA == 3 ? 3 * a.o1 : a.o0;   // floats per node

    f32x4 acc[kWgRows][NH][NCB];
    float bacc[kWgRows][NCB];
#pragma unroll
    for (int q = 0; q < kWgRows; ++q)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            bacc[q][cb] = 0.0f;
#pragma unroll
            for (int t = 0; t < NH; ++t) acc[q][t][cb] = splat4(0.0f);
        }
    struct Stage { float av[kWgRows][NH]; float mb[kWgRows]; float bv[NCB]; float okf; };   // raw loads; the mask is applied in consume()
    const int nsteps = ((ne - nb + 3) >> 2) * NAA;
    auto request = [&](int step, Stage& st) {
        const int it = NAA == 3 ? step / 3 : step, aa = NAA == 3 ? step - 3 * it : 0;
        const int n = nb + 4 * it + kq;
        const bool ok = n < ne;
        const int nk = ok ? n : ne - 1;
        const float* Mn = a.M + ((size_t)nk * a.lc + ent0 + aa * astride) * H + i16;
        const float* Mbn = a.Mb + (size_t)nk * a.lc + ent0 + aa * astride;
        const float* gn = gbase + (size_t)nk * gstride;
        st.okf = ok ? 1.0f : 0.0f;          // nodes past the range: zero B operand (output columns past `on` are never stored: no mask)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) st.bv[cb] = NAA == 3 ? gn[gcol[cb] * 3 + aa] : gn[gcol[cb]];
#pragma unroll
        for (int q = 0; q < kWgRows; ++q) {
            st.mb[q] = Mbn[q];
#pragma unroll
            for (int t = 0; t < NH; ++t) st.av[q][t] = Mn[q * H + 16 * t];
        }
    };
    auto consume = [&](const Stage& st) {
        float b[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) b[cb] = st.bv[cb] * st.okf;
#pragma unroll
        for (int q = 0; q < kWgRows; ++q) {
#pragma unroll
            for (int t = 0; t < NH; ++t)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[q][t][cb] = mfma16(st.av[q][t], b[cb], acc[q][t][cb]);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) bacc[q][cb] = fmaf(st.mb[q], b[cb], bacc[q][cb]);
        }
    };
    Stage s0, s1;
    request(0, s0);                                                       // (nsteps >= 1: the grid has no empty split)
    int step = 0;
    for (; step + 2 <= nsteps; step += 2) {
        request(step + 1, s1);
        consume(s0);
        request(min(step + 2, nsteps - 1), s0);                              // (unconditional: the last step again at the end)
        consume(s1);
    }
    if (step < nsteps) consume(s0);
    // D[h][o]: lane (o = i16, g = kq) holds h = 16 t + 4 g + r.  Parameter layout: row index (o * cin + in) (* 3 + J for R11)
    const int off = kind == 0 ? a.off00 : (kind == 1 ? a.off10 : (kind == 2 ? a.off01 : a.off11));
#pragma unroll
    for (int q = 0; q < kWgRows; ++q) {
        if (in0 + q >= cin) break;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int o = 16 * cb + i16;
            const size_t prow = (size_t)off + (kind == 3 ? ((size_t)o * cin + in0 + q) * 3 + J : (size_t)o * cin + in0 + q);
            const float bt = gsum(bacc[q][cb]);
            if (o < on) {
#pragma unroll
                for (int t = 0; t < NH; ++t) st4(slab + prow * H + 16 * t + 4 * kq, acc[q][t][cb]);
                if (kq == 0) slab[(size_t)a.n_w * H + prow] = bt;
            }
        }
    }
}

template <int NH>
__global__ __launch_bounds__(64, EQ_IR_WG_OCC) void irrep_wgrad16_kernel(const WgradArgs a) {
    constexpr int H = 16 * NH;
    const int lane = threadIdx.x, i16 = lane & 15, kq = lane >> 4;
    const int G00 = (a.c0 + kWgRows - 1) / kWgRows, G10 = (a.c1 + kWgRows - 1) / kWgRows, G01 = G00, G11 = 3 * G10;
    const int GS0 = (a.c0 + 15) >> 4, GS1 = (a.c1 + 15) >> 4;
    int grp = blockIdx.x;
    const int split = blockIdx.y;
    const int nb = split * a.nodes_per_split, ne = min(a.n_nodes, nb + a.nodes_per_split);
    float* slab = a.slab + (size_t)split * a.P;
    const int f1base = 4 * a.c0p;
    int kind;                      // 0: R00, 1: R10, 2: R01, 3: R11, 4: self0, 5: self1
    if (grp < G00) kind = 0;
    else if ((grp -= G00) < G10) kind = 1;
    else if ((grp -= G10) < G01) kind = 2;
    else if ((grp -= G01) < G11) kind = 3;
    else if ((grp -= G11) < GS0) kind = 4;
    else { grp -= GS0; kind = 5; }
    const bool deg1 = kind == 2 || kind == 3 || kind == 5;      // output gradient: g1 (three components) or g0
    const int on = deg1 ? a.o1 : a.o0;
    const int ncb = (on + 15) >> 4;
    if (kind >= 4) {
        // ---- d W_self[o][i] = sum_n g[n, o] f[n, i]: D[i][o], A lane (i = i16, k): f[n0 + k][16 grp + i16]
        const int cin = kind == 4 ? a.c0 : a.c1;
        const int i = 16 * grp + i16;
        f32x4 acc[2] = {splat4(0.0f), splat4(0.0f)};
        for (int n0 = nb; n0 < ne; n0 += 4) {
            const int nk = n0 + kq;
            const bool ok = nk < ne;
            for (int aa = 0; aa < (deg1 ? 3 : 1); ++aa) {
                const float av = (ok && i < cin) ? (deg1 ? a.f1[((size_t)nk * a.c1 + i) * 3 + aa] : a.f0[(size_t)nk * a.c0 + i]) : 0.0f;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    if (cb >= ncb) break;
                    const int o = 16 * cb + i16;
                    const float bv = (ok && o < on) ? (deg1 ? a.g1[((size_t)nk * a.o1 + o) * 3 + aa] : a.g0[(size_t)nk * a.o0 + o]) : 0.0f;
                    acc[cb] = mfma16(av, bv, acc[cb]);
                }
            }
        }
        // D[i][o]: lane (o = i16, g = kq) holds i = 16 grp + 4 g + r
        float* dst = slab + (size_t)a.n_w * (H + 1) + (kind == 4 ? 0 : (size_t)a.o0 * a.c0);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            if (cb >= ncb) break;
            const int o = 16 * cb + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ii = 16 * grp + 4 * kq + r;
                if (o < on && ii < cin) dst[(size_t)o * cin + ii] = acc[cb][r];
            }
        }
        return;
    }
    // ---- radial.2 rows: up to kWgRows (in, J) rows of one path
    if (deg1) { if (ncb == 2) wgrad_rows<NH, 2, 3>(a, kind, grp, nb, ne, slab); else wgrad_rows<NH, 1, 3>(a, kind, grp, nb, ne, slab); }
    else { if (ncb == 2) wgrad_rows<NH, 2, 1>(a, kind, grp, nb, ne, slab); else wgrad_rows<NH, 1, 1>(a, kind, grp, nb, ne, slab); }
}

// sums of the slabs in fixed order -> the parameter gradients (overwritten).  Blocks [0, nbA): one thread per radial.2 / self
// gradient entry, the slabs' values requested eight at a time; blocks nbA ..: ONE workgroup per entry of d w_r1 | d b_r1 (N / 4
// partials each: thread t sums the partials t, t + 256, ... in order, then a fixed tree over the threads).
__global__ __launch_bounds__(256) void irrep_wreduce_kernel(const float* __restrict__ slab, int splits, int P, int n_wH, int n_w, int ns0,
                                                            int ns1, float* __restrict__ d_w2, float* __restrict__ d_b2,
                                                            float* __restrict__ d_ws0, float* __restrict__ d_ws1,
                                                            const float* __restrict__ gslab, int gsplits, int H, int nbA,
                                                            float* __restrict__ d_w1, float* __restrict__ d_b1, int accumulate) {
    __shared__ float tree[256];
    if ((int)blockIdx.x < nbA) {
        const int idx = blockIdx.x * 256 + threadIdx.x;
        if (idx >= P) return;
        float s = 0.0f;
        int k = 0;
        for (; k + 8 <= splits; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(k + u) * P + idx];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < splits; ++k) s += slab[(size_t)k * P + idx];
        float* dst = idx < n_wH ? d_w2 + idx : (idx < n_wH + n_w ? d_b2 + (idx - n_wH)
                   : (idx < n_wH + n_w + ns0 ? d_ws0 + (idx - n_wH - n_w) : d_ws1 + (idx - n_wH - n_w - ns0)));
        *dst = accumulate ? *dst + s : s;
        return;
    }
    const int j = blockIdx.x - nbA;                       // 0 .. 2 H - 1
    float s = 0.0f;
    for (int k = threadIdx.x; k < gsplits; k += 256) s += gslab[(size_t)k * 2 * H + j];
    tree[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) tree[threadIdx.x] += tree[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* dst = j < H ? d_w1 + j : d_b1 + (j - H);
        *dst = accumulate ? *dst + tree[0] : tree[0];
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct Layout {
    int c0i, c1i, c0o, c1o, H;
    int off00, off01, off10, off11;        // row offsets of the four paths inside radial.2.weight / .bias (R00 | R01 | R10 | R11)
};
static Layout layout_of(const equigspr_irrep_config* c) {
    Layout L{c->c0_in, c->c1_in, c->c0_out, c->c1_out, c->hidden, 0, 0, 0, 0};
    L.off01 = L.c0o * L.c0i;
    L.off10 = L.off01 + L.c1o * L.c0i;
    L.off11 = L.off10 + L.c0o * L.c1i;
    return L;
}
static bool config_ok(const equigspr_irrep_config* c) {
    if (!c) return false;
    auto ch = [](int v) { return v >= 1 && v <= kMaxC; };
    return ch(c->c0_in) && ch(c->c1_in) && ch(c->c0_out) && ch(c->c1_out) && (c->hidden == 16 || c->hidden == 32);
}

template <int NH>
static void launch_moments(const MomArgs& a, hipStream_t s) {
    const int B0 = a.c0p / 16, units = a.c1 == 0 ? 1 : 1 + (a.c1p == 8 ? 1 : 2 * (a.c1p / 16));
    const dim3 grid(ceil_div(a.n_nodes, 4), units), block(256);
#define EQ_IR_MOM(BB, CC) \
    if (B0 == BB && a.c1p == CC) { hipLaunchKernelGGL((irrep_moments16_kernel<NH, BB, CC>), grid, block, 0, s, a); return; }
    EQ_IR_MOM(1, 8) EQ_IR_MOM(1, 16) EQ_IR_MOM(1, 32) EQ_IR_MOM(2, 8) EQ_IR_MOM(2, 16) EQ_IR_MOM(2, 32)
#undef EQ_IR_MOM
}

template <int NH>
static void launch_node(const NodeArgs& a, hipStream_t s) {
    const int cb0 = ceil_div(a.o0, 16), cb1 = ceil_div(a.o1, 16);
    const dim3 grid(ceil_div(a.n_nodes, 16)), block(64 * kNodeWaves);
#define EQ_IR_NODE(AA, BB) \
    if (cb0 == AA && cb1 == BB) { hipLaunchKernelGGL((irrep_node16_kernel<NH, AA, BB>), grid, block, 0, s, a); return; }
    EQ_IR_NODE(1, 1) EQ_IR_NODE(1, 2) EQ_IR_NODE(2, 1) EQ_IR_NODE(2, 2)
#undef EQ_IR_NODE
}

static inline int64_t r4(int64_t v) { return (v + 3) / 4 * 4; }
// per-edge geometry table in a float buffer: [E] float4 | [E] int32
static inline int64_t tab_floats(int n_edges) { return r4(5LL * (n_edges > 0 ? n_edges : 1)); }
static EdgeTab make_table(const float* x, const int* agg, const int* src, const int* perm, int n_edges, float* buf, hipStream_t s) {
    float4* geo = reinterpret_cast<float4*>(buf);
    int* srcs = reinterpret_cast<int*>(buf + 4 * (size_t)(n_edges > 0 ? n_edges : 1));
    if (n_edges > 0) hipLaunchKernelGGL(irrep_edge_table_kernel, dim3(ceil_div(n_edges, 256)), dim3(256), 0, s, x, agg, src, perm, n_edges, geo, srcs);
    return EdgeTab{geo, srcs};
}
static EdgeTab table_at(const float* buf, int n_edges) {
    return EdgeTab{reinterpret_cast<const float4*>(buf), reinterpret_cast<const int*>(buf + 4 * (size_t)(n_edges > 0 ? n_edges : 1))};
}

// forward direction: aggregate at row, gather from col, weights as stored
static void conv_forward(const equigspr_irrep_config* c, const Layout& L, const float* w_r1, const float* b_r1, const float* w_r2,
                         const float* b_r2, const float* ws0, const float* ws1, const equigspr_graph* g, const float* f0,
                         const float* f1, const EdgeTab& tab, float* out0, float* out1, float* M, float* Mb, hipStream_t s,
                         bool keep_moments = true, bool f1_zero = false) {
    // f1_zero (the first layer of IrrepEGNN: the input has no degree-1 features): the kernels run with ZERO active degree-1 input
    // channels -- same layouts and offsets (c1p), the c10 / c11 entries neither computed nor read
    const int H = L.H, c0p = pad0(L.c0i), c1p = pad1(L.c1i), lc = 4 * c0p + 10 * c1p, c1a = f1_zero ? 0 : L.c1i;
    MomArgs m{f0, f1, g->rptr, tab, w_r1, b_r1, Consts{c->k00, c->k01, c->k10, c->k11_0, c->k11_1, c->k11_2},
              g->n_nodes, L.c0i, c1a, c0p, c1p, lc, M, Mb};
    NodeArgs nd{M, Mb, f0, f1,
                PathW{w_r2 + (size_t)L.off00 * H, b_r2 + L.off00, L.c0i * H, H},
                PathW{w_r2 + (size_t)L.off10 * H, b_r2 + L.off10, L.c1i * H, H},
                PathW{w_r2 + (size_t)L.off01 * H, b_r2 + L.off01, L.c0i * H, H},
                PathW{w_r2 + (size_t)L.off11 * H, b_r2 + L.off11, L.c1i * 3 * H, 3 * H},
                SelfW{ws0, L.c0i, 1}, SelfW{ws1, L.c1i, 1},
                g->n_nodes, L.c0i, c1a, c0p, c1p, lc, L.c0o, L.c1o, out0, out1};
    if (H == 16) { launch_moments<1>(m, s); launch_node<1>(nd, s); }
    else { launch_moments<2>(m, s); launch_node<2>(nd, s); }
}


// ---- backward: scratch layout and launches
struct BwdPlan {
    int lc, lc_r, groups, splits, nodes_per_split, P, gsplits;
    int64_t o_Mr, o_Mbr, o_dM, o_dMb, o_dr, o_dxr, o_gslab, o_wslab, o_tabR, total;
};
static BwdPlan plan_bwd(const Layout& L, int n_nodes, int n_edges) {
    BwdPlan p{};
    const int H = L.H;
    p.lc = 4 * pad0(L.c0i) + 10 * pad1(L.c1i);
    p.lc_r = 4 * pad0(L.c0o) + 10 * pad1(L.c1o);
    const int n_w = L.off11 + 3 * L.c1o * L.c1i;
    p.P = n_w * (H + 1) + L.c0o * L.c0i + L.c1o * L.c1i;
    p.groups = 2 * ceil_div(L.c0i, kWgRows) + 4 * ceil_div(L.c1i, kWgRows) + ceil_div(L.c0i, 16) + ceil_div(L.c1i, 16);
#ifndef EQ_WG_WANT
#define EQ_WG_WANT 4096
#endif
    int want = ceil_div(EQ_WG_WANT, p.groups);                                 // ~4 k waves: four per SIMD (the accumulators' budget)
    want = max(1, min(want, ceil_div(max(n_nodes, 1), 16)));
    p.nodes_per_split = round_up(ceil_div(max(n_nodes, 1), want), 4);
    p.splits = ceil_div(max(n_nodes, 1), p.nodes_per_split);
    p.gsplits = ceil_div(max(n_nodes, 1), 4);
    int64_t o = 0;
    p.o_Mr = o;    o += r4((int64_t)n_nodes * p.lc_r * H);
    p.o_Mbr = o;   o += r4((int64_t)n_nodes * p.lc_r);
    p.o_dM = o;    o += r4((int64_t)n_nodes * p.lc * H);
    p.o_dMb = o;   o += r4((int64_t)n_nodes * p.lc);
    p.o_dr = o;    o += r4((int64_t)max(n_edges, 1) * 4);
    p.o_dxr = o;   o += r4((int64_t)n_nodes * 4);
    p.o_gslab = o; o += r4((int64_t)p.gsplits * 2 * H);
    p.o_wslab = o; o += r4((int64_t)p.splits * p.P);
    p.o_tabR = o;  o += tab_floats(n_edges);
    p.total = o;
    return p;
}

template <int NH>
static void launch_expand(const ExpArgs& a, dim3 grid, dim3 block, hipStream_t s) {
    const int k0 = a.o0 <= 8 ? 2 : (a.o0 <= 16 ? 4 : 8), k1 = a.o1 <= 8 ? 2 : (a.o1 <= 16 ? 4 : 8);
#define EQ_IR_EXP(A, B) if (k0 == A && k1 == B) { hipLaunchKernelGGL((irrep_expand16_kernel<NH, A, B>), grid, block, 0, s, a); return; }
    EQ_IR_EXP(2, 2) EQ_IR_EXP(2, 4) EQ_IR_EXP(2, 8) EQ_IR_EXP(4, 2) EQ_IR_EXP(4, 4) EQ_IR_EXP(4, 8) EQ_IR_EXP(8, 2) EQ_IR_EXP(8, 4) EQ_IR_EXP(8, 8)
#undef EQ_IR_EXP
}

template <int NH>
static void launch_geom(const GeomArgs& a, hipStream_t s) {
    const dim3 grid(ceil_div(a.n_nodes, 4)), block(256);
    if (a.c1p == 8) hipLaunchKernelGGL((irrep_geom_bwd_kernel<NH, 8>), grid, block, 0, s, a);
    else if (a.c1p == 16) hipLaunchKernelGGL((irrep_geom_bwd_kernel<NH, 16>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((irrep_geom_bwd_kernel<NH, 32>), grid, block, 0, s, a);
}

static void conv_backward(const equigspr_irrep_config* c, const Layout& L, const float* w_r1, const float* b_r1, const float* w_r2,
                          const float* b_r2, const float* ws0, const float* ws1, const equigspr_graph* g, const float* f0,
                          const float* f1, const EdgeTab& tabF, const EdgeTab& tabR, const float* M, const float* g0, const float* g1,
                          float* d_f0, float* d_f1, float* d_x, float* d_w1, float* d_b1, float* d_w2, float* d_b2, float* d_ws0,
                          float* d_ws1, float* work, hipStream_t s, int accumulate = 0, bool f1_zero = false) {
    const int H = L.H, N = g->n_nodes, c1a = f1_zero ? 0 : L.c1i;
    const BwdPlan p = plan_bwd(L, N, g->n_edges);
    const Consts kf{c->k00, c->k01, c->k10, c->k11_0, c->k11_1, c->k11_2};
    const int c0p = pad0(L.c0i), c1p = pad1(L.c1i);
    const float* Mb = M + (int64_t)N * p.lc * H;
    const PathW f00{w_r2 + (size_t)L.off00 * H, b_r2 + L.off00, L.c0i * H, H}, f10{w_r2 + (size_t)L.off10 * H, b_r2 + L.off10, L.c1i * H, H},
        f01{w_r2 + (size_t)L.off01 * H, b_r2 + L.off01, L.c0i * H, H}, f11{w_r2 + (size_t)L.off11 * H, b_r2 + L.off11, L.c1i * 3 * H, 3 * H};
    if (d_f0 && d_f1) {
        // the adjoint convolution: aggregate at col, gather the output gradients from row; r' = x[col] - x[row] = -r flips the two
        // kernels that are odd in u (K01 <-> K10 also swap roles); transposed weights = swapped strides
        const int c0pr = pad0(L.c0o), c1pr = pad1(L.c1o);
        MomArgs m{g0, g1, g->cptr, tabR, w_r1, b_r1, Consts{c->k00, -c->k10, -c->k01, c->k11_0, c->k11_1, c->k11_2},
                  N, L.c0o, L.c1o, c0pr, c1pr, p.lc_r, work + p.o_Mr, work + p.o_Mbr};
        NodeArgs nd{work + p.o_Mr, work + p.o_Mbr, g0, g1,
                    PathW{f00.w, f00.b, H, L.c0i * H},               // G0 -> d f0: R00^T
                    PathW{f01.w, f01.b, H, L.c0i * H},               // G1 -> d f0: R01^T
                    PathW{f10.w, f10.b, H, L.c1i * H},               // G0 -> d f1: R10^T
                    PathW{f11.w, f11.b, 3 * H, L.c1i * 3 * H},       // G1 -> d f1: R11^T
                    SelfW{ws0, 1, L.c0i}, SelfW{ws1, 1, L.c1i},
                    N, L.c0o, L.c1o, c0pr, c1pr, p.lc_r, L.c0i, L.c1i, d_f0, d_f1};
        if (H == 16) { launch_moments<1>(m, s); launch_node<1>(nd, s); }
        else { launch_moments<2>(m, s); launch_node<2>(nd, s); }
    }
    // geometry + radial.0
    ExpArgs ex{g0, g1, f00, f10, f01, f11, N, L.c0i, c1a, c0p, c1p, p.lc, L.c0o, L.c1o, work + p.o_dM, work + p.o_dMb};
    GeomArgs ge{f0, f1, g->rptr, g->rperm, tabF, w_r1, b_r1, kf, N, L.c0i, c1a, c0p, c1p, p.lc, work + p.o_dM, work + p.o_dMb,
                work + p.o_dr, work + p.o_dxr, work + p.o_gslab};
    const dim3 egrid(ceil_div(N, 16), kExpSplit), eblock(64 * kExpWaves);
    if (H == 16) { launch_expand<1>(ex, egrid, eblock, s); launch_geom<1>(ge, s); }
    else { launch_expand<2>(ex, egrid, eblock, s); launch_geom<2>(ge, s); }
    if (d_x) hipLaunchKernelGGL(irrep_dx_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, s, work + p.o_dr, work + p.o_dxr, g->cptr, g->cperm, N, d_x);
    // weight gradients
    const int n_w = L.off11 + 3 * L.c1o * L.c1i;
    WgradArgs wa{M, Mb, f0, f1, g0, g1, N, L.c0i, c1a, c0p, c1p, p.lc, L.c0o, L.c1o, H, L.off00, L.off01, L.off10, L.off11, n_w,
                 p.nodes_per_split, work + p.o_wslab, p.P};
    // (no active degree-1 input: the groups of R10 / R11 / W_s1 rows do not exist -- their slab words are cleared, their gradients are zero)
    const int groups = f1_zero ? 2 * ceil_div(L.c0i, kWgRows) + ceil_div(L.c0i, 16) : p.groups;
    if (f1_zero) (void)hipMemsetAsync(work + p.o_wslab, 0, sizeof(float) * (size_t)p.splits * p.P, s);
    const dim3 wgrid(groups, p.splits);
    if (H == 16) hipLaunchKernelGGL((irrep_wgrad16_kernel<1>), wgrid, dim3(64), 0, s, wa);
    else hipLaunchKernelGGL((irrep_wgrad16_kernel<2>), wgrid, dim3(64), 0, s, wa);
    const int nbA = ceil_div(p.P, 256);
    hipLaunchKernelGGL(irrep_wreduce_kernel, dim3(nbA + 2 * H), dim3(256), 0, s, work + p.o_wslab, p.splits, p.P, n_w * H, n_w,
                       L.c0o * L.c0i, L.c1o * L.c1i, d_w2, d_b2, d_ws0, d_ws1, work + p.o_gslab, p.gsplits, H, nbA, d_w1, d_b1, accumulate);
}


// ================================================================================================ the typed network (IrrepEGNN)
// A stack of the layer above behind the EGNN interface (h [N, in_nf], x [N, 3]) -> (h [N, out_nf], x [N, 3]):
//   f0 = h, f1 = 0;  per layer l:  (p0, p1) = conv_l(f0, f1, x);  l < L - 1:  f0 = SiLU(p0),  f1 = p1 * sigmoid(W_g p0 + b_g)  (an
//   invariant gate per degree-1 channel: keeps degree 1 covariant);  h_out = p0 of the last layer,  x_out = x + sum_i w_x[i] p1[:, i, :]
//   (degree-1 features are Cartesian vectors up to the factor of Y_1: an equivariant coordinate update, as E_GCL.coord_model :262-268).
// The coordinates stay fixed through the stack (one graph, one geometry), so d x sums over the layers.
struct GnnPlan {
    int L, H, c1, n_nodes;
    int c0[17];                                   // degree-0 widths: c0[0] = in_nf, c0[1..L-1] = hidden_nf, c0[L] = out_nf
    int64_t o_w1[16], o_b1[16], o_w2[16], o_b2[16], o_s0[16], o_s1[16], o_gw[16], o_gb[16], o_wx, n_params;
    int64_t s_f0[16], s_f1[16], s_p0[16], s_p1[16], s_M[16], s_tabF, s_tabR, saved_total;     // saved activations (floats); s_f0[0] is unused (= h_in)
    int64_t bwd_conv_work, w_g0, w_g1, w_d0, w_d1, w_slab, work_total;
    int gate_blocks;
};
static equigspr_irrep_config layer_cfg(const equigspr_irrep_gnn_config* c, const GnnPlan& P, int l) {
    return equigspr_irrep_config{P.c0[l], P.c1, P.c0[l + 1], P.c1, P.H, c->k00, c->k01, c->k10, c->k11_0, c->k11_1, c->k11_2};
}
static bool gnn_ok(const equigspr_irrep_gnn_config* c) {
    auto ch = [](int v) { return v >= 1 && v <= kMaxC; };
    return c && ch(c->in_nf) && ch(c->hidden_nf) && ch(c->out_nf) && ch(c->c1) && c->n_layers >= 1 && c->n_layers <= 16 &&
           (c->radial_hidden == 16 || c->radial_hidden == 32);
}
static GnnPlan plan_gnn(const equigspr_irrep_gnn_config* c, int n_nodes, int n_edges) {
    GnnPlan P{};
    P.L = c->n_layers; P.H = c->radial_hidden; P.c1 = c->c1; P.n_nodes = n_nodes;
    for (int l = 0; l <= P.L; ++l) P.c0[l] = l == 0 ? c->in_nf : (l == P.L ? c->out_nf : c->hidden_nf);
    // parameter block in named_parameters() order of irreps.IrrepEGNN: all layers, then all gates, then the coordinate head; every
    // tensor starts on a 16-byte boundary (the kernels read weight rows as float4)
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += r4(n); return at; };
    for (int l = 0; l < P.L; ++l) {
        const int ci = P.c0[l], co = P.c0[l + 1], c1 = P.c1;
        const int64_t n_w = (int64_t)co * ci + (int64_t)c1 * ci + (int64_t)co * c1 + 3LL * c1 * c1;
        P.o_w1[l] = take(P.H); P.o_b1[l] = take(P.H); P.o_w2[l] = take(n_w * P.H); P.o_b2[l] = take(n_w);
        P.o_s0[l] = take((int64_t)co * ci); P.o_s1[l] = take((int64_t)c1 * c1);
    }
    for (int l = 0; l + 1 < P.L; ++l) { P.o_gw[l] = take((int64_t)P.c1 * P.c0[l + 1]); P.o_gb[l] = take(P.c1); }
    P.o_wx = take(P.c1);
    P.n_params = o;
    // saved activations
    o = 0;
    int64_t conv_work = 0;
    const int64_t N = n_nodes;
    for (int l = 0; l < P.L; ++l) {
        const equigspr_irrep_config lc = layer_cfg(c, P, l);
        P.s_f0[l] = l == 0 ? -1 : take(N * P.c0[l]);
        P.s_f1[l] = take(N * P.c1 * 3);
        P.s_p0[l] = take(N * P.c0[l + 1]);
        P.s_p1[l] = take(N * P.c1 * 3);
        const int64_t lcn = 4 * pad0(lc.c0_in) + 10 * pad1(lc.c1_in);
        P.s_M[l] = take(N * lcn * (P.H + 1));
        const int64_t w = plan_bwd(layout_of(&lc), n_nodes, n_edges).total;
        conv_work = w > conv_work ? w : conv_work;
    }
    P.s_tabF = take(tab_floats(n_edges));             // the stack's coordinates are fixed: one geometry table per direction for all layers
    P.s_tabR = take(tab_floats(n_edges));
    P.saved_total = o;
    // backward scratch: the convolution's own + two pairs of gradient buffers (ping-pong through the layers) + gate / head slabs
    o = 0;
    P.bwd_conv_work = take(conv_work);
    const int cmax = max(max(c->in_nf, c->hidden_nf), c->out_nf);
    P.w_g0 = take(N * cmax); P.w_g1 = take(N * P.c1 * 3); P.w_d0 = take(N * cmax); P.w_d1 = take(N * P.c1 * 3);
    P.gate_blocks = ceil_div(max(n_nodes, 1), 64);
    P.w_slab = take((int64_t)P.gate_blocks * (P.c1 * (kMaxC + 1) + P.c1));
    P.work_total = o;
    return P;
}

// ---- gate: f0 = SiLU(p0), f1 = p1 * sigmoid(W_g p0 + b_g).  64 nodes per workgroup (256 threads: four per node); the weight
// gradients d W_g [c1, c0] | d b_g [c1] as one slab per workgroup (LDS outer product over the workgroup's nodes), summed in order later.
__global__ __launch_bounds__(256) void irrep_gate_fwd_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                             const float* __restrict__ wg, const float* __restrict__ bg, int n_nodes,
                                                             int c0, int c1, float* __restrict__ f0, float* __restrict__ f1) {
    __shared__ float w_s[kMaxC * (kMaxC + 1)];
    for (int t = threadIdx.x; t < c1 * c0; t += 256) w_s[(t / c0) * (kMaxC + 1) + t % c0] = wg[t];
    __syncthreads();
    const int n = blockIdx.x * 64 + (threadIdx.x >> 2), q = threadIdx.x & 3;
    if (n >= n_nodes) return;
    const float* pr = p0 + (size_t)n * c0;
    for (int i = q; i < c0; i += 4) f0[(size_t)n * c0 + i] = silu(pr[i]);
    for (int ch = q; ch < c1; ch += 4) {
        float z = bg[ch];
        for (int i = 0; i < c0; ++i) z = fmaf(w_s[ch * (kMaxC + 1) + i], pr[i], z);
        const float gt = sigmoid_fast(z);
#pragma unroll
        for (int a_ = 0; a_ < 3; ++a_) f1[((size_t)n * c1 + ch) * 3 + a_] = p1[((size_t)n * c1 + ch) * 3 + a_] * gt;
    }
}

// backward of the gate: inputs d f0, d f1 (gradients of the NEXT layer's inputs) -> g0 = d p0, g1 = d p1
__global__ __launch_bounds__(256) void irrep_gate_bwd_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                             const float* __restrict__ wg, const float* __restrict__ bg,
                                                             const float* __restrict__ d_f0, const float* __restrict__ d_f1, int n_nodes,
                                                             int c0, int c1, float* __restrict__ g0, float* __restrict__ g1,
                                                             float* __restrict__ slab) {
    __shared__ float w_s[kMaxC * (kMaxC + 1)];
    __shared__ float dz_s[64][kMaxC + 1];
    __shared__ float p_s[64][kMaxC + 1];
    for (int t = threadIdx.x; t < c1 * c0; t += 256) w_s[(t / c0) * (kMaxC + 1) + t % c0] = wg[t];
    const int ln = threadIdx.x >> 2, q = threadIdx.x & 3;
    const int n = blockIdx.x * 64 + ln;
    const bool ok = n < n_nodes;
    const float* pr = p0 + (size_t)(ok ? n : 0) * c0;
    for (int i = q; i < c0; i += 4) p_s[ln][i] = ok ? pr[i] : 0.0f;
    __syncthreads();
    for (int ch = q; ch < c1; ch += 4) {
        float z = bg[ch];
        for (int i = 0; i < c0; ++i) z = fmaf(w_s[ch * (kMaxC + 1) + i], p_s[ln][i], z);
        const float gt = sigmoid_fast(z);
        float dot = 0.0f;
        if (ok) {
#pragma unroll
            for (int a_ = 0; a_ < 3; ++a_) {
                const size_t at = ((size_t)n * c1 + ch) * 3 + a_;
                const float df = d_f1[at];
                g1[at] = df * gt;
                dot = fmaf(df, p1[at], dot);
            }
        }
        dz_s[ln][ch] = ok ? dot * gt * (1.0f - gt) : 0.0f;                    // d loss / d (W_g p0 + b_g)
    }
    __syncthreads();
    if (ok) {
        for (int i = q; i < c0; i += 4) {
            float v = d_f0[(size_t)n * c0 + i] * silu_grad(p_s[ln][i]);
            for (int ch = 0; ch < c1; ++ch) v = fmaf(w_s[ch * (kMaxC + 1) + i], dz_s[ln][ch], v);
            g0[(size_t)n * c0 + i] = v;
        }
    }
    // slab: d W_g[ch][i] = sum over the workgroup's nodes (ascending) of dz[n][ch] p0[n][i];  d b_g[ch] = sum dz[n][ch]
    float* sl = slab + (size_t)blockIdx.x * (c1 * (c0 + 1));
    for (int t = threadIdx.x; t < c1 * (c0 + 1); t += 256) {
        const int ch = t / (c0 + 1), i = t - ch * (c0 + 1);
        float s = 0.0f;
        for (int m = 0; m < 64; ++m) s = fmaf(dz_s[m][ch], i < c0 ? p_s[m][i] : 1.0f, s);
        sl[t] = s;
    }
}

// out[j] (+)= sum_k slab[k][j], k ascending: one workgroup per column block of 256 threads x 1 column (small slabs)
__global__ __launch_bounds__(256) void irrep_colsum_kernel(const float* __restrict__ slab, int rows, int cols, const int* __restrict__ map,
                                                           float* __restrict__ out_a, int split, float* __restrict__ out_b, int accumulate) {
    __shared__ float tree[256];
    const int j = blockIdx.x;
    float s = 0.0f;
    for (int k = threadIdx.x; k < rows; k += 256) s += slab[(size_t)k * cols + j];
    tree[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) tree[threadIdx.x] += tree[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // gate slabs are laid out [ch][c0 + 1]: column (ch, i < c0) -> out_a[ch * c0 + i], column (ch, c0) -> out_b[ch]; split = c0 + 1
        float* dst;
        if (split > 0) { const int ch = j / split, i = j - ch * split; dst = i < split - 1 ? out_a + (size_t)ch * (split - 1) + i : out_b + ch; }
        else dst = out_a + j;
        *dst = accumulate ? *dst + tree[0] : tree[0];
    }
}

// coordinate head: x_out = x + sum_i w_x[i] v[:, i, :]
__global__ void irrep_coord_fwd_kernel(const float* __restrict__ x, const float* __restrict__ v, const float* __restrict__ wx, int n_nodes,
                                       int c1, float* __restrict__ x_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_nodes * 3) return;
    const int n = t / 3, a_ = t - 3 * n;
    float s = x[t];
    for (int i = 0; i < c1; ++i) s = fmaf(wx[i], v[((size_t)n * c1 + i) * 3 + a_], s);
    x_out[t] = s;
}
// backward: g1[n, i, a] = w_x[i] g_x[n, a];  d w_x[i] = sum_{n, a} v[n, i, a] g_x[n, a] (slab per workgroup of 256 nodes);  d x += g_x
__global__ __launch_bounds__(256) void irrep_coord_bwd_kernel(const float* __restrict__ v, const float* __restrict__ wx,
                                                              const float* __restrict__ gx, int n_nodes, int c1, float* __restrict__ g1,
                                                              float* __restrict__ d_x, float* __restrict__ slab) {
    __shared__ float part[256][kMaxC + 1];
    const int n = blockIdx.x * 256 + threadIdx.x;
    const bool ok = n < n_nodes;
    float g[3] = {0.0f, 0.0f, 0.0f};
    if (ok) {
#pragma unroll
        for (int a_ = 0; a_ < 3; ++a_) {
            g[a_] = gx[(size_t)n * 3 + a_];
            if (d_x) d_x[(size_t)n * 3 + a_] += g[a_];
        }
    }
    for (int i = 0; i < c1; ++i) {
        float dot = 0.0f;
        if (ok) {
#pragma unroll
            for (int a_ = 0; a_ < 3; ++a_) {
                const size_t at = ((size_t)n * c1 + i) * 3 + a_;
                dot = fmaf(v[at], g[a_], dot);
                g1[at] = wx[i] * g[a_];
            }
        }
        part[threadIdx.x][i] = dot;
    }
    __syncthreads();
    if ((int)threadIdx.x < c1) {
        float s = 0.0f;
        for (int m = 0; m < 256; ++m) s += part[m][threadIdx.x];
        slab[(size_t)blockIdx.x * c1 + threadIdx.x] = s;
    }
}

}  // namespace ir
}  // namespace eq

extern "C" int64_t equigspr_irrep_moments_size(const equigspr_irrep_config* cfg, int32_t n_nodes, int32_t n_edges) {
    using namespace eq::ir;
    if (!config_ok(cfg) || n_nodes < 0 || n_edges < 0) return -1;
    const int64_t lc = 4 * pad0(cfg->c0_in) + 10 * pad1(cfg->c1_in);
    return r4((int64_t)n_nodes * lc * (cfg->hidden + 1)) + tab_floats(n_edges);       // M | Mb | the edge geometry table
}

extern "C" int equigspr_irrep_conv_forward(const equigspr_irrep_config* cfg, const float* w_r1, const float* b_r1, const float* w_r2,
                                           const float* b_r2, const float* w_self0, const float* w_self1, const equigspr_graph* g,
                                           const float* f0, const float* f1, const float* x, float* out0, float* out1,
                                           float* moments, int32_t keep_moments, void* stream) {
    using namespace eq;
    using namespace eq::ir;
    EQ_CHECK_ARG(config_ok(cfg), EQUIGSPR_ENOTSUP, "irrep_conv_forward: 1..%d channels per degree, hidden 16 or 32", kMaxC);
    EQ_CHECK_ARG(w_r1 && b_r1 && w_r2 && b_r2 && w_self0 && w_self1 && g && f0 && f1 && x && out0 && out1 && moments, EQUIGSPR_EINVAL,
                 "irrep_conv_forward: null argument");
    EQ_CHECK_ARG(g->n_nodes >= 1 && g->n_edges >= 0 && g->rptr && (g->n_edges == 0 || (g->col && g->rperm)), EQUIGSPR_EINVAL,
                 "irrep_conv_forward: incomplete graph (col, rptr, rperm)");
    EQ_CHECK_ARG(g->n_edges == 0 || g->row, EQUIGSPR_EINVAL, "irrep_conv_forward: incomplete graph (row)");
    const Layout L = layout_of(cfg);
    const int64_t lc = 4 * pad0(L.c0i) + 10 * pad1(L.c1i);
    float* tbuf = moments + r4((int64_t)g->n_nodes * lc * (L.H + 1));
    const EdgeTab tab = make_table(x, g->row, g->col, g->rperm, g->n_edges, tbuf, (hipStream_t)stream);
    conv_forward(cfg, L, w_r1, b_r1, w_r2, b_r2, w_self0, w_self1, g, f0, f1, tab, out0, out1, moments,
                 moments + (int64_t)g->n_nodes * lc * L.H, (hipStream_t)stream, keep_moments != 0);
    EQ_CHECK_LAUNCH("irrep_conv_forward");
    return 0;
}

extern "C" int64_t equigspr_irrep_bwd_work_size(const equigspr_irrep_config* cfg, int32_t n_nodes, int32_t n_edges) {
    using namespace eq::ir;
    if (!config_ok(cfg) || n_nodes < 0 || n_edges < 0) return -1;
    return plan_bwd(layout_of(cfg), n_nodes, n_edges).total;
}

extern "C" int equigspr_irrep_conv_backward(const equigspr_irrep_config* cfg, const float* w_r1, const float* b_r1, const float* w_r2,
                                            const float* b_r2, const float* w_self0, const float* w_self1, const equigspr_graph* g,
                                            const float* f0, const float* f1, const float* x, const float* moments, const float* g_out0,
                                            const float* g_out1, float* d_f0, float* d_f1, float* d_x, float* d_w_r1, float* d_b_r1,
                                            float* d_w_r2, float* d_b_r2, float* d_w_self0, float* d_w_self1, float* work, void* stream) {
    using namespace eq;
    using namespace eq::ir;
    EQ_CHECK_ARG(config_ok(cfg), EQUIGSPR_ENOTSUP, "irrep_conv_backward: 1..%d channels per degree, hidden 16 or 32", kMaxC);
    EQ_CHECK_ARG(w_r1 && b_r1 && w_r2 && b_r2 && w_self0 && w_self1 && g && f0 && f1 && x && moments && g_out0 && g_out1 && d_w_r1 &&
                     d_b_r1 && d_w_r2 && d_b_r2 && d_w_self0 && d_w_self1 && work, EQUIGSPR_EINVAL, "irrep_conv_backward: null argument");
    EQ_CHECK_ARG((d_f0 == nullptr) == (d_f1 == nullptr), EQUIGSPR_EINVAL, "irrep_conv_backward: d_f0 and d_f1 are computed together (both or neither)");
    EQ_CHECK_ARG(g->n_nodes >= 1 && g->n_edges >= 0 && g->rptr && g->cptr && (g->n_edges == 0 || (g->row && g->col && g->rperm && g->cperm)),
                 EQUIGSPR_EINVAL, "irrep_conv_backward: incomplete graph (row, col and both inverse indices)");
    const Layout L = layout_of(cfg);
    const int64_t lc = 4 * pad0(L.c0i) + 10 * pad1(L.c1i);
    const EdgeTab tabF = table_at(moments + r4((int64_t)g->n_nodes * lc * (L.H + 1)), g->n_edges);      // left by the forward
    const BwdPlan bp = plan_bwd(L, g->n_nodes, g->n_edges);
    const EdgeTab tabR = d_f0 ? make_table(x, g->col, g->row, g->cperm, g->n_edges, work + bp.o_tabR, (hipStream_t)stream) : EdgeTab{nullptr, nullptr};
    conv_backward(cfg, L, w_r1, b_r1, w_r2, b_r2, w_self0, w_self1, g, f0, f1, tabF, tabR, moments, g_out0, g_out1, d_f0, d_f1, d_x,
                  d_w_r1, d_b_r1, d_w_r2, d_b_r2, d_w_self0, d_w_self1, work, (hipStream_t)stream);
    EQ_CHECK_LAUNCH("irrep_conv_backward");
    return 0;
}

// ---- the typed network behind the EGNN interface (irreps.IrrepEGNN)
extern "C" int64_t equigspr_irrep_gnn_param_count(const equigspr_irrep_gnn_config* cfg) {
    using namespace eq::ir;
    if (!gnn_ok(cfg)) return -1;
    return plan_gnn(cfg, 1, 1).n_params;
}
extern "C" int64_t equigspr_irrep_gnn_param_offset(const equigspr_irrep_gnn_config* cfg, int32_t layer, int32_t which) {
    using namespace eq::ir;
    if (!gnn_ok(cfg) || layer < 0 || layer >= cfg->n_layers || which < 0 || which > 8) return -1;
    const GnnPlan P = plan_gnn(cfg, 1, 1);
    switch (which) {
        case 0: return P.o_w1[layer];
        case 1: return P.o_b1[layer];
        case 2: return P.o_w2[layer];
        case 3: return P.o_b2[layer];
        case 4: return P.o_s0[layer];
        case 5: return P.o_s1[layer];
        case 6: return layer + 1 < P.L ? P.o_gw[layer] : -1;
        case 7: return layer + 1 < P.L ? P.o_gb[layer] : -1;
        default: return P.o_wx;
    }
}
extern "C" int64_t equigspr_irrep_gnn_saved_size(const equigspr_irrep_gnn_config* cfg, int32_t n_nodes, int32_t n_edges) {
    using namespace eq::ir;
    if (!gnn_ok(cfg) || n_nodes < 0 || n_edges < 0) return -1;
    return plan_gnn(cfg, n_nodes, n_edges).saved_total;
}
extern "C" int64_t equigspr_irrep_gnn_work_size(const equigspr_irrep_gnn_config* cfg, int32_t n_nodes, int32_t n_edges) {
    using namespace eq::ir;
    if (!gnn_ok(cfg) || n_nodes < 0 || n_edges < 0) return -1;
    return plan_gnn(cfg, n_nodes, n_edges).work_total;
}

extern "C" int equigspr_irrep_gnn_forward(const equigspr_irrep_gnn_config* cfg, const float* params, const equigspr_graph* g,
                                          const float* h_in, const float* x, float* h_out, float* x_out, float* saved, int32_t training,
                                          void* stream) {
    using namespace eq;
    using namespace eq::ir;
    EQ_CHECK_ARG(gnn_ok(cfg), EQUIGSPR_ENOTSUP, "irrep_gnn_forward: 1..%d channels per degree, 1..16 layers, radial hidden 16 or 32", kMaxC);
    EQ_CHECK_ARG(params && g && h_in && x && h_out && x_out && saved, EQUIGSPR_EINVAL, "irrep_gnn_forward: null argument");
    EQ_CHECK_ARG(g->n_nodes >= 1 && g->n_edges >= 0 && g->rptr && (g->n_edges == 0 || (g->col && g->rperm)), EQUIGSPR_EINVAL,
                 "irrep_gnn_forward: incomplete graph (col, rptr, rperm)");
    EQ_CHECK_ARG(((uintptr_t)params & 15) == 0 && ((uintptr_t)saved & 15) == 0, EQUIGSPR_EINVAL, "irrep_gnn_forward: params / saved must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int N = g->n_nodes;
    const GnnPlan P = plan_gnn(cfg, N, g->n_edges);
    EQ_CHECK_ARG(hipMemsetAsync(saved + P.s_f1[0], 0, sizeof(float) * (size_t)N * P.c1 * 3, s) == hipSuccess, EQUIGSPR_EINVAL, "irrep_gnn_forward: memset failed");
    EQ_CHECK_ARG(g->n_edges == 0 || g->row, EQUIGSPR_EINVAL, "irrep_gnn_forward: incomplete graph (row)");
    const EdgeTab tabF = make_table(x, g->row, g->col, g->rperm, g->n_edges, saved + P.s_tabF, s);
    if (training) {
        EQ_CHECK_ARG(g->cptr && (g->n_edges == 0 || g->cperm), EQUIGSPR_EINVAL, "irrep_gnn_forward: training needs the col index (cptr, cperm)");
        make_table(x, g->col, g->row, g->cperm, g->n_edges, saved + P.s_tabR, s);
    }
    for (int l = 0; l < P.L; ++l) {
        const equigspr_irrep_config lc = layer_cfg(cfg, P, l);
        const Layout Ly = layout_of(&lc);
        const bool last = l + 1 == P.L;
        const float* f0 = l == 0 ? h_in : saved + P.s_f0[l];
        const float* f1 = saved + P.s_f1[l];
        float* p0 = last ? h_out : saved + P.s_p0[l];
        float* p1 = saved + P.s_p1[l];
        float* M = saved + P.s_M[l];
        const int64_t lcn = 4 * pad0(Ly.c0i) + 10 * pad1(Ly.c1i);
        conv_forward(&lc, Ly, params + P.o_w1[l], params + P.o_b1[l], params + P.o_w2[l], params + P.o_b2[l], params + P.o_s0[l],
                     params + P.o_s1[l], g, f0, f1, tabF, p0, p1, M, M + (int64_t)N * lcn * P.H, s, training != 0, l == 0);
        if (!last)
            hipLaunchKernelGGL(irrep_gate_fwd_kernel, dim3(ceil_div(N, 64)), dim3(256), 0, s, p0, p1, params + P.o_gw[l], params + P.o_gb[l], N,
                               P.c0[l + 1], P.c1, saved + P.s_f0[l + 1], saved + P.s_f1[l + 1]);
        else
            hipLaunchKernelGGL(irrep_coord_fwd_kernel, dim3(ceil_div(3 * N, 256)), dim3(256), 0, s, x, p1, params + P.o_wx, N, P.c1, x_out);
    }
    EQ_CHECK_LAUNCH("irrep_gnn_forward");
    return 0;
}

extern "C" int equigspr_irrep_gnn_backward(const equigspr_irrep_gnn_config* cfg, const float* params, const equigspr_graph* g,
                                           const float* h_in, const float* x, const float* saved, const float* g_h, const float* g_x,
                                           float* d_h_in, float* d_x, float* grads, float* work, void* stream) {
    using namespace eq;
    using namespace eq::ir;
    EQ_CHECK_ARG(gnn_ok(cfg), EQUIGSPR_ENOTSUP, "irrep_gnn_backward: unsupported config");
    EQ_CHECK_ARG(params && g && h_in && x && saved && g_h && g_x && grads && work, EQUIGSPR_EINVAL, "irrep_gnn_backward: null argument");
    EQ_CHECK_ARG(g->n_nodes >= 1 && g->n_edges >= 0 && g->rptr && g->cptr && (g->n_edges == 0 || (g->row && g->col && g->rperm && g->cperm)),
                 EQUIGSPR_EINVAL, "irrep_gnn_backward: incomplete graph (row, col and both inverse indices)");
    hipStream_t s = (hipStream_t)stream;
    const int N = g->n_nodes;
    const GnnPlan P = plan_gnn(cfg, N, g->n_edges);
    float* cw = work + P.bwd_conv_work;
    float* g0 = work + P.w_g0; float* g1 = work + P.w_g1; float* d0 = work + P.w_d0; float* d1 = work + P.w_d1;
    float* slab = work + P.w_slab;
    const EdgeTab tabF = table_at(saved + P.s_tabF, g->n_edges), tabR = table_at(saved + P.s_tabR, g->n_edges);
    // coordinate head: g1 = w_x (x) g_x, d w_x, d x += g_x
    const int cb = ceil_div(N, 256);
    hipLaunchKernelGGL(irrep_coord_bwd_kernel, dim3(cb), dim3(256), 0, s, saved + P.s_p1[P.L - 1], params + P.o_wx, g_x, N, P.c1, g1, d_x, slab);
    hipLaunchKernelGGL(irrep_colsum_kernel, dim3(P.c1), dim3(256), 0, s, slab, cb, P.c1, (const int*)nullptr, grads + P.o_wx, 0, (float*)nullptr, 1);
    const float* gout0 = g_h;
    for (int l = P.L - 1; l >= 0; --l) {
        const equigspr_irrep_config lc = layer_cfg(cfg, P, l);
        const Layout Ly = layout_of(&lc);
        const float* f0 = l == 0 ? h_in : saved + P.s_f0[l];
        const float* f1 = saved + P.s_f1[l];
        const bool want_f = l > 0 || d_h_in != nullptr;
        float* df0 = !want_f ? nullptr : (l == 0 ? d_h_in : d0);
        float* df1 = !want_f ? nullptr : d1;
        conv_backward(&lc, Ly, params + P.o_w1[l], params + P.o_b1[l], params + P.o_w2[l], params + P.o_b2[l], params + P.o_s0[l],
                      params + P.o_s1[l], g, f0, f1, tabF, tabR, saved + P.s_M[l], gout0, g1, df0, df1, d_x, grads + P.o_w1[l], grads + P.o_b1[l],
                      grads + P.o_w2[l], grads + P.o_b2[l], grads + P.o_s0[l], grads + P.o_s1[l], cw, s, 1, l == 0);
        if (l > 0) {
            // through the gate between layers l - 1 and l: (d0, d1) = d (f0_l, f1_l) -> (g0, g1) = d (p0, p1) of layer l - 1
            const int c0 = P.c0[l];
            hipLaunchKernelGGL(irrep_gate_bwd_kernel, dim3(P.gate_blocks), dim3(256), 0, s, saved + P.s_p0[l - 1], saved + P.s_p1[l - 1],
                               params + P.o_gw[l - 1], params + P.o_gb[l - 1], d0, d1, N, c0, P.c1, g0, g1, slab);
            hipLaunchKernelGGL(irrep_colsum_kernel, dim3(P.c1 * (c0 + 1)), dim3(256), 0, s, slab, P.gate_blocks, P.c1 * (c0 + 1),
                               (const int*)nullptr, grads + P.o_gw[l - 1], c0 + 1, grads + P.o_gb[l - 1], 1);
            gout0 = g0;
        }
    }
    EQ_CHECK_LAUNCH("irrep_gnn_backward");
    return 0;
}
