// 16-edge MFMA formulation of the per-edge MLP chain (hidden_nf == 32), shared by the forward and backward edge
// kernels (reference math: E_GCL.edge_model / coord_model, src/3dmatch_train_egnn_with_batch.py:231-268).
//
// A wave walks over CHUNKS of up to 64 consecutive edges:
//   front  : one lane per edge -- indices, coordinate gathers and the whole edge geometry are computed ONCE per edge
//            (the 32-edge formulation of egnn_mfma.h computes them in both lane halves) and left in wave-private LDS;
//   middle : the chunk's 16-edge SUB-TILES one after the other on v_mfma_f32_16x16x4_f32.  Lane l = (e = l & 15,
//            g = l >> 4) owns edge e of the sub-tile and 8 of its 32 features: v[b][r] = feature 16 b + 4 g + r
//            (b = 0, 1; r = 0..3).  That is the C/D layout of the instruction with D[i = feature][j = edge], and with
//            the K index of step (b, r) mapped to feature 16 b + 4 g + r it is also the B operand of the next product,
//            so the chain  t1 -> SiLU -> W2 -> LayerNorm -> W6 -> SiLU -> w7  needs no cross-lane traffic except the
//            two-step reductions over g.  A vector costs 8 registers instead of 16: the backward fits two waves per
//            SIMD with its weight-gradient accumulators, the forward four;
//   back   : one lane per edge again (coordinate scalar / geometry backward).
// The weights (A operands) are read from a per-workgroup LDS image with ds_read_b128: one read feeds four MFMAs.
// fp32 MFMA is an exact fmaf chain, so the numerics are those of the VALU formulation.
#pragma once
#include "egnn_device.h"

// The 32 x 32 mat-vec products of the edge kernels (W2, W6 and, in the backward, their transposes) run on the bf16 matrix pipe with
// 3-way split operands (x = hi + mid + lo, each bf16; the six products with i + j <= 2 accumulate in fp32): 12
// v_mfma_f32_16x16x32_bf16 (16 cycles each) instead of 16 v_mfma_f32_16x16x4_f32 (32 cycles each); differences to the fp32 products
// <= 1e-6 relative (tools/dev/bf16_split.hip: 1.33-1.35 x on the tile including the split's vector work; max error 1.9e-7 against
// fp64, the fp32 tile's is 3.3e-7).  Default since the end of round 5: 16 pairs forward 0.67 -> 0.73 / backward 0.62 -> 0.68 of the
// HBM roofline, hidden_nf 33 forward 0.60 -> 0.63, one pair backward 0.44 -> 0.46 (profiles/r05_bf16x3_edge_kernels.txt).  The hidden_nf 33
// backward keeps the fp32 products (no LDS for the fragments).  ONE path for every graph size: a cloud's outputs do not depend on
// what else is in the batch.
// RUN-TO-RUN DIFFERENT RESULTS (round 5) AND WHAT THEY WERE (round 6, profiles/r06_mfma_hazard_probe.txt): not the matrix instructions.  With
// packed fp32 instructions in the kernel (v_pk_add / mul / fma_f32) and a schedule that mixes loads or other matrix instructions into the
// products, the lane-per-edge geometry of ~0.5 % of the 64-edge chunks came out wrong in lanes 48..63 (a v_pk_add_f32 with op_sel
// returned src0 - 0 from sources that read back correct; its scalar form is exact).  The library is therefore compiled WITHOUT
// packed fp32 instructions (csrc/Makefile PACKED_FLAGS; tools/isa_lint.py checks the objects): exact and bit-identical in every schedule
// tried, fenced or not.  The forward keeps its products as closed instruction groups (EQ_BF_GUARD 1: 10 % faster at one pair), the
// backward does not (EQ_BF_GUARD 0 in egnn_bwd.hip: 2.5-4 % faster, fewer spills at hidden_nf 33).
// 0 = the fp32 products (A/B).
#ifndef EQ_EDGE_BF16X3
#define EQ_EDGE_BF16X3 1
#endif

namespace eq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ldg4(const float* __restrict__ p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// store of a kernel OUTPUT that the next launch reads (from other XCDs: through memory either way).  A plain store leaves a dirty line in
// this XCD's L2, and the release at the end of the kernel writes all of them back in one burst that the next launch waits behind;
// written through at system scope (sc0 sc1) the lines leave while the kernel still computes.  One 2 048-point pair: edge forward 11.0 ->
// 9.5 us, backward 21.9 -> 20.8 us per launch, the replayed step 0.3495 -> 0.333 ms; at 16 pairs the forward LOSES 4 % (the write-back
// burst was overlapped by the other workgroups' compute there), so the kernels do it only below EQ_STREAM_OUT_MAX_EDGES edges (the node-level forward's 2 MB of outputs: no difference either way; the gradient slabs of the small-problem node backward: st_slab, egnn_bwd.hip).
#ifndef EQ_STREAM_OUT_MAX_EDGES
#define EQ_STREAM_OUT_MAX_EDGES 500000                  // A/B knob: 0 = never, 1 << 30 = always (6 pairs of 2 048 points gain 1.6 %, 8 and the 524 288-edge eval pair are neutral, 16 lose)
#endif
__device__ __forceinline__ bool stream_out_edges(int n_edges) { return (EQ_STREAM_OUT_MAX_EDGES) > 0 && n_edges <= (EQ_STREAM_OUT_MAX_EDGES); }
// (as a raw buffer store with the sc0 | sc1 cache-policy bits, so that the compiler knows the instruction: the same store as inline
// asm fell into the hazard "VALU overwrites the data registers of a > 64-bit VMEM store in the next cycle", which the compiler only
// pads for instructions it emitted itself -- wrong rows at hidden_nf 33)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
struct OutRows {
    __amdgpu_buffer_rsrc_t rsrc;
    float* base;
    __device__ __forceinline__ explicit OutRows(float* p) : rsrc(__builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7ffffff0, 0x00020000)), base(p) {}
    // 16 bytes at float offset `off` (< 2^29 when streamed: the launchers only stream small graphs)
    __device__ __forceinline__ void st4(size_t off, f32x4 v, bool stream) const {
        if (stream) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rsrc, (int)(off * 4), 0, 17);      // aux: sc0 (1) | sc1 (16)
        else *reinterpret_cast<f32x4*>(base + off) = v;
    }
    __device__ __forceinline__ void st1(size_t off, float v, bool stream) const {
        if (stream) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), rsrc, (int)(off * 4), 0, 17);
        else base[off] = v;
    }
};
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Vector-form arithmetic: hipcc maps <2 x float> multiplies / adds / fmas to v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, which
// issue in the time of ONE v_fma_f32 for two results (tools/dev/valu_cost.hip) -- and the edge kernels are bound by the number
// of vector + matrix instructions they issue, so every per-feature expression is written on whole f32x4 values.
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
// sigmoid_fast (egnn_math.h) on four values: 2 packed multiplies, 4 v_exp_f32, 2 packed adds, 4 v_rcp_f32 (same arithmetic per element)
__device__ __forceinline__ f32x4 sigmoid4(f32x4 v) {
    const f32x4 t = v * splat4(-1.4426950408889634f);
    const f32x4 e = f32x4{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])} + splat4(1.0f);
    return f32x4{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1]), __builtin_amdgcn_rcpf(e[2]), __builtin_amdgcn_rcpf(e[3])};
}
// SiLU'(x) = s (1 + x (1 - s)) with s = sigmoid(x), as two fused steps: a = x - x s, s + s a
__device__ __forceinline__ f32x4 silu_grad4(f32x4 x, f32x4 sg) { return fma4(sg, fma4(-x, sg, x), sg); }
__device__ __forceinline__ float hsum4(f32x4 v) { const f32x2 t = v.lo + v.hi; return t[0] + t[1]; }
// sum of the eight entries of two vectors: (a + b) packed, then the halves, then the pair
__device__ __forceinline__ float hsum8(f32x4 a, f32x4 b) { return hsum4(a + b); }

// sum over the four lane groups g of an edge (lanes e, e + 16, e + 32, e + 48): two VALU lane swaps (v_permlane16_swap,
// v_permlane32_swap: no LDS round trip, unlike ds_bpermute); every lane of the edge gets the bit-identical sum
__device__ __forceinline__ float gsum(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);                // rows 0+1 | 2+3
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// gsum of TWO values for the price of three lane swaps instead of four (same summation order as gsum, bit-identical sums):
// swap16(p, q) + add -> rows [p0+p1 | q0+q1 | p2+p3 | q2+q3]; swap32 with itself + add -> [P | Q | P | Q]; swap16 with itself
// hands every row both totals.
__device__ __forceinline__ void gsum2(float& p, float& q) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(q), false, false);
    const float u = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(u), false, false);
    const float w = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    const auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(w), __float_as_uint(w), false, false);
    p = __uint_as_float(c[0]);
    q = __uint_as_float(c[1]);
}

// sum over the 16 lanes of a row (lanes 16 g .. 16 g + 15: the 16 edges of a tile), every lane gets the sum: four VALU adds whose
// second operand is swizzled in flight (DPP: quad xor 1, quad xor 2, half-row mirror, row mirror) -- the same pairing tree as
// __shfl_xor 1, 2, 4, 8 (bit-identical sums) without its four ds_bpermute round trips per value (28 values per wave at the end of
// edge_bwd16: 3.7 k -> 1.1 k cycles of every wave's tail, trace build)
__device__ __forceinline__ float row16_sum(float v) {
    auto step = [](float x, int y) { return x + __int_as_float(y); };
    v = step(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));      // quad_perm [1, 0, 3, 2]
    v = step(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));      // quad_perm [2, 3, 0, 1]
    v = step(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));     // row_half_mirror
    v = step(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));     // row_mirror
    return v;
}

// ---- 3-way bf16 split of eight floats (x = hi + mid + lo to 24 bits) and the fragment layout of a 32 x 32 matrix for
// v_mfma_f32_16x16x32_bf16: [3 levels][2 row blocks][64 lanes] x 8 bf16; K index j of lane group g <-> feature 16 (j >> 2) + 4 g + (j & 3)
#ifndef EQ_BF_SPLIT_TRUNC
#define EQ_BF_SPLIT_TRUNC 1            // 1: split by truncation with integer ops; 0: by v_cvt_pk_bf16_f32 (round to nearest)
#endif
__device__ __forceinline__ void split3_bf16(const f32x4 (&v)[2], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#if EQ_BF_SPLIT_TRUNC
    // hi = the upper 16 bits of x, mid = those of x - hi, lo = those of x - hi - mid: both differences are exact, the three pieces
    // carry 24 mantissa bits.  Only AND / SUB / V_PERM: in the edge kernels the v_cvt_pk_bf16_f32 form gave run-to-run different
    // results (a stale mid / lo operand in the matrix instruction that follows the conversion when the wave does not stall in
    // between: profiles/r05_bf16x3_edge_kernels.txt), this form does not use the conversion instruction at all.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 H, M, L;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = v[p >> 1][2 * (p & 1)], x1 = v[p >> 1][2 * (p & 1) + 1];
        const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
        const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
        const unsigned a0 = __float_as_uint(r0), a1 = __float_as_uint(r1);
        const float s0 = r0 - __uint_as_float(a0 & 0xffff0000u), s1 = r1 - __uint_as_float(a1 & 0xffff0000u);
        H[p] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);                 // [x1's upper half | x0's upper half]
        M[p] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
        L[p] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
    hi = __builtin_bit_cast(bf16x8, H); mid = __builtin_bit_cast(bf16x8, M); lo = __builtin_bit_cast(bf16x8, L);
#else
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = v[j >> 2][j & 3];
        const __bf16 h = (__bf16)x;
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        hi[j] = h; mid[j] = m; lo[j] = (__bf16)(r1 - (float)m);
    }
#endif
}
__device__ __forceinline__ void frag_store(const f32x4 (&x)[2], float* frag, int bo, int l) {
    bf16x8 hi, mid, lo;
    split3_bf16(x, hi, mid, lo);
    bf16x8* f = reinterpret_cast<bf16x8*>(frag);
    f[(0 * 2 + bo) * 64 + l] = hi; f[(1 * 2 + bo) * 64 + l] = mid; f[(2 * 2 + bo) * 64 + l] = lo;
}

// LDS image of a layer's edge-MLP weights.  Row strides chosen so that the per-lane 16-byte operand fetches
// (row = 16 bo + i, column 16 bi + 4 g) of every 16-lane ds_read_b128 group fall on distinct banks.
// hidden_nf = 33 (kB = 1): the 32 x 32 cores stay on the matrix cores; feature 32 (the "border") is a scalar per edge held by
// all four lanes of the edge -- its column of every matrix costs 8 FMAs per product, its row a per-lane partial dot product
// and one gsum; the border vectors live in a small extra block of the image.
struct W16 {
    static constexpr int WS = 40;                          // 32 x 32 matrices
    static constexpr int GS = 24;                          // Wgeo [32][16]: columns 0..11 geometry, 12 edge_attr, 13..15 zero
    static constexpr int o_w2 = 0;
    static constexpr int o_w6 = o_w2 + 32 * WS;
    static constexpr int o_wg = o_w6 + 32 * WS;
    static constexpr int o_vec = o_wg + 32 * GS;           // b2 | gamma | beta | b6 | w7
    static constexpr int fwd_size = o_vec + 5 * 32;
    static constexpr int o_w2t = fwd_size;                 // transposed copies for the backward's W^T products
    static constexpr int o_w6t = o_w2t + 32 * WS;
    static constexpr int o_wgt = o_w6t + 32 * WS;          // Wgeo^T [16][WS]
    static constexpr int bwd_size = o_wgt + 16 * WS;
    // border block (kB = 1), relative to its base: W2[:,32] | W2[32,:] | W6[:,32] | W6[32,:] | Wgeo[32,:16] | scalars
    static constexpr int b_w2col = 0, b_w2row = 32, b_w6col = 64, b_w6row = 96, b_wgrow = 128, b_sc = 144;
    static constexpr int border_size = 152;                // scalars: W2[32][32], W6[32][32], b2, gamma, beta, b6, w7 of feature 32
    static constexpr int border_base(bool bwd) { return bwd ? bwd_size : fwd_size; }
    // bf16 x 3 fragments of the 32 x 32 matrices (EQ_EDGE_BF16X3): per matrix [3 levels][2 row blocks][64 lanes] x 8 bf16 = 1536 floats;
    // forward: W2, W6; backward: + W2^T, W6^T
    static constexpr int frag_floats = 3 * 2 * 64 * 4;
#ifndef EQ_FRAG_PAD
#define EQ_FRAG_PAD 0                  // developer: floats of unused LDS in front of the fragments
#endif
    static constexpr int frag_base(bool bwd, int kB) { return (bwd ? bwd_size : fwd_size) + kB * border_size + ((bwd && kB == 1) ? 0 : EQ_FRAG_PAD); }
    static constexpr int f_w2 = 0, f_w6 = 1, f_w2t = 2, f_w6t = 3;
    // (the hidden_nf 33 backward has no LDS for images + fragments -- 187 KB -- and takes the SLIM image below)
// developer A/B: bit 0 = the forward kernel, bit 1 = the backward kernel (fragments by build_frags16 after the image's barrier)
#ifndef EQ_BF16X3_KERNELS
#define EQ_BF16X3_KERNELS 3
#endif
#ifndef EQ_BWD33_BF16
#define EQ_BWD33_BF16 1                // hidden_nf 33 backward on the split products (round 6: the SLIM image below makes the LDS room)
#endif
    static constexpr bool use_bf16(bool bwd, int kB) { return EQ_EDGE_BF16X3 != 0 && (EQ_BWD33_BF16 != 0 || !(bwd && kB == 1)) && ((EQ_BF16X3_KERNELS >> (bwd ? 1 : 0)) & 1); }
    static constexpr int image_size_fp32(bool bwd, int kB) { return frag_base(bwd, kB); }
    // SLIM image (the hidden_nf 33 backward on split products: 162.5 KB of LDS with the fp32 images AND the fragments would be 187 KB):
    // the four 32 x 32 fp32 images (W2, W6, their transposes: 20 KB) are not stored at all -- every product of them reads the bf16
    // fragments, which StageW16 builds straight from global memory -- Wgeo | vectors | Wgeo^T | border | fragments remain.
    static constexpr bool slim(bool bwd, int kB, bool bf) { return bwd && kB == 1 && bf && use_bf16(bwd, kB); }
    static constexpr int wg(bool sl) { return sl ? 0 : o_wg; }
    static constexpr int vec(bool sl) { return sl ? 32 * GS : o_vec; }
    static constexpr int wgt(bool sl) { return sl ? 32 * GS + 5 * 32 : o_wgt; }
    static constexpr int slim_size = 32 * GS + 5 * 32 + 16 * WS;
    static constexpr int border_at(bool bwd, bool sl) { return sl ? slim_size : border_base(bwd); }
    static constexpr int frag_at(bool bwd, int kB, bool sl) { return sl ? slim_size + border_size : frag_base(bwd, kB); }
    static constexpr int image_floats(bool bwd, int kB, bool bf) {
        return slim(bwd, kB, bf) ? slim_size + border_size + 4 * frag_floats : (bf && use_bf16(bwd, kB) ? image_size(bwd, kB) : image_size_fp32(bwd, kB));
    }
#ifndef EQ_FRAG_TAIL
#define EQ_FRAG_TAIL 0                 // developer: floats of unused LDS behind the fragments
#endif
    static constexpr int image_size(bool bwd, int kB) { return frag_base(bwd, kB) + (use_bf16(bwd, kB) ? (bwd ? 4 : 2) * frag_floats + EQ_FRAG_TAIL : 0); }
};

// global -> LDS in two halves, so that a kernel can put other requests between them: load() issues all of a thread's global
// loads (the edge kernels request their first indices BEFORE it and gather the coordinates between load() and store(): the
// image's latency and the index -> coordinate chain overlap instead of adding up -- at one 2 048-point pair a wave lives for
// only 2-3 tiles, so ~1 us of serial start-up is 5 % of the kernel), store() writes the image.  NT = threads of the workgroup.
template <int H, int A, bool kBwd, int NT, int kB = 0, bool kFrag = false>
struct StageW16 {
    static constexpr int F = 32 + kB;
    using LL = LP<F, H, A>;
    static constexpr int FH = F / H;
    static constexpr int PER = 1024 / NT, PG = (512 + NT - 1) / NT;
    static constexpr bool kSlim = W16::slim(kBwd, kB, kFrag);       // no fp32 images of the 32 x 32 matrices: fragments only
    static_assert(!kSlim || NT >= 256, "the slim image's transposed fragments are fetched by threads 128..255");
    float v2[PER], v6[PER], vg[PG], vv, vb;
    f32x4 f2[2], f6[2];          // kFrag: threads 0..127 also fetch the eight W2 / W6 weights of their (row block, lane) fragment (slim: 128..255 those of W2^T / W6^T)
    __device__ __forceinline__ void load(const float* __restrict__ lw) {
        if constexpr (kFrag) {
            if (threadIdx.x < 128) {
                const int bo = threadIdx.x >> 6, l = threadIdx.x & 63, ii = 16 * bo + (l & 15), gg = l >> 4;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * (j >> 2) + 4 * gg + (j & 3);
                    f2[j >> 2][j & 3] = (ii / FH == k / FH) ? lw[LL::w2_off(ii, k)] : 0.0f;
                    f6[j >> 2][j & 3] = lw[LL::w6 + ii * F + k];
                }
            } else if (kSlim && threadIdx.x < 256) {      // (W^T)[ii][k] = W[k][ii]: 16 strided 4-byte reads of L2-resident weights, once per workgroup
                const int tt = threadIdx.x - 128, bo = tt >> 6, l = tt & 63, ii = 16 * bo + (l & 15), gg = l >> 4;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * (j >> 2) + 4 * gg + (j & 3);
                    f2[j >> 2][j & 3] = (ii / FH == k / FH) ? lw[LL::w2_off(k, ii)] : 0.0f;
                    f6[j >> 2][j & 3] = lw[LL::w6 + k * F + ii];
                }
            }
        }
        if constexpr (!kSlim) {                          // the fp32 images of W2 / W6 (the slim image has fragments only)
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int t = threadIdx.x + NT * u, i = t >> 5, k = t & 31;
                v2[u] = (i / FH == k / FH) ? lw[LL::w2_off(i, k)] : 0.0f;      // block-diagonal over the heads
                v6[u] = lw[LL::w6 + i * F + k];
            }
        }
#pragma unroll
        for (int u = 0; u < PG; ++u) {
            const int t = threadIdx.x + NT * u, i = (t >> 4) & 31, k = t & 15;
            vg[u] = (t < 512 && k < kGeo + A) ? lw[LL::w1_off(i, 2 * F + k)] : 0.0f;
        }
        vv = 0.0f;
        if (threadIdx.x < 160) {
            const int w = threadIdx.x >> 5, k = threadIdx.x & 31;
            vv = lw[w == 0 ? LL::b2_off(k) : (w == 1 ? LL::ln_g + k : (w == 2 ? LL::ln_b + k : (w == 3 ? LL::b6 + k : LL::w7 + k)))];
        }
        vb = 0.0f;
        if (kB && threadIdx.x < W16::border_size) {
            const int t = threadIdx.x, k = t & 31;
            if (t < 32) vb = lw[LL::w2_off(k, 32)];
            else if (t < 64) vb = lw[LL::w2_off(32, k)];
            else if (t < 96) vb = lw[LL::w6 + k * F + 32];
            else if (t < 128) vb = lw[LL::w6 + 32 * F + k];
            else if (t < 144) vb = (t - 128 < kGeo + A) ? lw[LL::w1_off(32, 2 * F + (t - 128))] : 0.0f;
            else {
                const int q = t - 144;
                vb = q == 0 ? lw[LL::w2_off(32, 32)] : q == 1 ? lw[LL::w6 + 32 * F + 32] : q == 2 ? lw[LL::b2_off(32)] : q == 3 ? lw[LL::ln_g + 32]
                   : q == 4 ? lw[LL::ln_b + 32] : q == 5 ? lw[LL::b6 + 32] : q == 6 ? lw[LL::w7 + 32] : 0.0f;
            }
        }
    }
    __device__ __forceinline__ void store(float* sm) const {
        if constexpr (!kSlim) {
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int t = threadIdx.x + NT * u, i = t >> 5, k = t & 31;
                sm[W16::o_w2 + i * W16::WS + k] = v2[u];
                sm[W16::o_w6 + i * W16::WS + k] = v6[u];
                if (kBwd) {
                    sm[W16::o_w2t + k * W16::WS + i] = v2[u];
                    sm[W16::o_w6t + k * W16::WS + i] = v6[u];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PG; ++u) {
            const int t = threadIdx.x + NT * u, i = (t >> 4) & 31, k = t & 15;
            if (t < 512) {
                sm[W16::wg(kSlim) + i * W16::GS + k] = vg[u];
                if (kBwd) sm[W16::wgt(kSlim) + k * W16::WS + i] = vg[u];
            }
        }
        if (threadIdx.x < 160) sm[W16::vec(kSlim) + threadIdx.x] = vv;
        if (kB && threadIdx.x < W16::border_size) sm[W16::border_at(kBwd, kSlim) + threadIdx.x] = vb;
        if constexpr (kFrag) {
            float* fb = sm + W16::frag_at(kBwd, kB, kSlim);
            if (threadIdx.x < 128) {
                const int bo = threadIdx.x >> 6, l = threadIdx.x & 63;
                frag_store(f2, fb + W16::f_w2 * W16::frag_floats, bo, l);
                frag_store(f6, fb + W16::f_w6 * W16::frag_floats, bo, l);
            } else if (kSlim && threadIdx.x < 256) {
                const int tt = threadIdx.x - 128, bo = tt >> 6, l = tt & 63;
                frag_store(f2, fb + W16::f_w2t * W16::frag_floats, bo, l);
                frag_store(f6, fb + W16::f_w6t * W16::frag_floats, bo, l);
            }
        }
    }
};

template <int H, int A, bool kBwd, int NT, int kB = 0>
__device__ __forceinline__ void stage_w16(float* sm, const float* __restrict__ lw) {
    StageW16<H, A, kBwd, NT, kB> st;
    st.load(lw);
    st.store(sm);
}

// acc[bo] += W v for a 32 x 32 LDS image `mat` (row stride WS): 4 ds_read_b128, 16 MFMAs.  For y = W^T v pass the
// transposed image.  (i, g) = (lane & 15, lane >> 4).
__device__ __forceinline__ void mv16(const float* mat, int i, int g, const f32x4 (&v)[2], f32x4 (&acc)[2]) {
    f32x4 a[2][2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bo = 0; bo < 2; ++bo) a[bi][bo] = lds4(mat + (16 * bo + i) * W16::WS + 16 * bi + 4 * g);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[0] = mfma16(a[bi][0][r], v[bi][r], acc[0]);
            acc[1] = mfma16(a[bi][1][r], v[bi][r], acc[1]);
        }
}
// ---- the same product on split-bf16 operands: the lane's own eight features of the L16 layout are the B operand as they stand
// frag: [3 levels][2 row blocks][64 lanes] fragments of the matrix (build_frag16); acc[bo] += (W v)[16 bo + ...]
// EQ_BF_GUARD: 1 = the product is a closed instruction group (sched_barriers around it, the fragment reads in front of the first
// matrix instruction, EQ_BF_GUARD_NOPS x `s_nop 7` behind the last).  A SCHEDULING choice, not a hazard fence: tools/dev/mfma_war_probe.hip
// shows that a source of v_mfma_f32_16x16x32_bf16 may be overwritten by a VALU instruction, an LDS read or a global load in the very
// next slot (0 wrong results in 10^9), that interleaved and D != C chains are exact, and the unfenced build is exact once the packed
// fp32 instructions are gone (see the head of this file).  The software-managed rules the probe does confirm are the compiler's own:
// 2 wait states between a VALU write and a matrix instruction's A / B read (1 for C), 7 between the instruction and a VALU read of D, 1
// between a TRANS result and its VALU use.
#ifndef EQ_BF_GUARD
#define EQ_BF_GUARD 1
#endif
#ifndef EQ_BF_GUARD_NOPS
#define EQ_BF_GUARD_NOPS 1             // x "s_nop 7" behind the group
#endif
#ifndef EQ_BF_GUARD_LOADS_FIRST
#define EQ_BF_GUARD_LOADS_FIRST 1      // 1: no load is issued between the group's first and last matrix instruction (see mv16_bf)
#endif
#ifndef EQ_BF_GUARD_MASK
#define EQ_BF_GUARD_MASK 0             // __builtin_amdgcn_sched_barrier mask of the group's fences: which instruction classes MAY cross (0 = none; experiments only)
#endif
#ifndef EQ_BF_GUARD_HALF
#define EQ_BF_GUARD_HALF 1             // 1: one closed group per row block (three fragments alive at a time instead of six: no spills in the backward; 20.7 -> 19.1 us at one pair)
#endif
__device__ __forceinline__ void mv16_bf(const float* frag, int lane, const f32x4 (&v)[2], f32x4 (&acc)[2]) {
    const bf16x8* f = reinterpret_cast<const bf16x8*>(frag);
    bf16x8 vh, vm, vl;
    split3_bf16(v, vh, vm, vl);
#if EQ_BF_GUARD && EQ_BF_GUARD_HALF
#pragma unroll
    for (int bo = 0; bo < 2; ++bo) {
        __builtin_amdgcn_sched_barrier(EQ_BF_GUARD_MASK);
        const bf16x8 ah = f[(0 * 2 + bo) * 64 + lane], am = f[(1 * 2 + bo) * 64 + lane], al = f[(2 * 2 + bo) * 64 + lane];
#if EQ_BF_GUARD_LOADS_FIRST
        // all three fragment reads are issued before the first matrix instruction (otherwise the allocator hands the third fragment
        // the registers of the incoming accumulator and the read sits inside the chain: harmless, but the group is no longer closed)
        __builtin_amdgcn_sched_barrier(EQ_BF_GUARD_MASK);
#endif
        acc[bo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, vh, acc[bo], 0, 0, 0);        // smallest terms first
        acc[bo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, vl, acc[bo], 0, 0, 0);
        acc[bo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, vm, acc[bo], 0, 0, 0);
        acc[bo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, vh, acc[bo], 0, 0, 0);
        acc[bo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, vm, acc[bo], 0, 0, 0);
        acc[bo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, vh, acc[bo], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < EQ_BF_GUARD_NOPS; ++n)
            asm volatile("s_nop 7" : "+v"(acc[bo]) : "v"(ah), "v"(am), "v"(al), "v"(vh), "v"(vm), "v"(vl));
        __builtin_amdgcn_sched_barrier(EQ_BF_GUARD_MASK);
    }
#else
    bf16x8 ah[2], am[2], al[2];
#if EQ_BF_GUARD
    __builtin_amdgcn_sched_barrier(EQ_BF_GUARD_MASK);
#endif
#pragma unroll
    for (int bo = 0; bo < 2; ++bo) { ah[bo] = f[(0 * 2 + bo) * 64 + lane]; am[bo] = f[(1 * 2 + bo) * 64 + lane]; al[bo] = f[(2 * 2 + bo) * 64 + lane]; }
#pragma unroll
    for (int bo = 0; bo < 2; ++bo) {
        acc[bo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[bo], vh, acc[bo], 0, 0, 0);        // smallest terms first
        acc[bo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[bo], vl, acc[bo], 0, 0, 0);
        acc[bo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[bo], vm, acc[bo], 0, 0, 0);
        acc[bo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[bo], vh, acc[bo], 0, 0, 0);
        acc[bo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[bo], vm, acc[bo], 0, 0, 0);
        acc[bo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[bo], vh, acc[bo], 0, 0, 0);
    }
#if EQ_BF_GUARD
#pragma unroll
    for (int n = 0; n < EQ_BF_GUARD_NOPS; ++n)
        asm volatile("s_nop 7" : "+v"(acc[0]), "+v"(acc[1]) : "v"(ah[0]), "v"(am[0]), "v"(al[0]), "v"(ah[1]), "v"(am[1]), "v"(al[1]), "v"(vh), "v"(vm), "v"(vl));
    __builtin_amdgcn_sched_barrier(EQ_BF_GUARD_MASK);
#endif
#endif
}
// fragments of the fp32 image `mat` (row stride WS, as mv16 reads it) -> frag: threads 0..127 of the workgroup, one (row block, lane) each
__device__ __forceinline__ void build_frag16(const float* mat, float* frag, int t) {
    const int bo = t >> 6, l = t & 63, i = l & 15, g = l >> 4;
    f32x4 x[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j >> 2][j & 3] = mat[(16 * bo + i) * W16::WS + 16 * (j >> 2) + 4 * g + (j & 3)];
    frag_store(x, frag, bo, l);
}
// after the fp32 image is complete (and a workgroup barrier): the split fragments of its 32 x 32 matrices; followed by another barrier
template <bool kBwd, int kB, bool kBf = true>
__device__ __forceinline__ void build_frags16(float* sm) {
    if constexpr (kBf && W16::use_bf16(kBwd, kB) && !W16::slim(kBwd, kB, kBf)) {      // (slim image: StageW16 wrote the fragments)
    float* fb = sm + W16::frag_base(kBwd, kB);
    const int t = threadIdx.x;
    if (t < 128) {
        build_frag16(sm + W16::o_w2, fb + W16::f_w2 * W16::frag_floats, t);
        build_frag16(sm + W16::o_w6, fb + W16::f_w6 * W16::frag_floats, t);
    } else if (kBwd && t < 256) {
        build_frag16(sm + W16::o_w2t, fb + W16::f_w2t * W16::frag_floats, t - 128);
        build_frag16(sm + W16::o_w6t, fb + W16::f_w6t * W16::frag_floats, t - 128);
    }
    __syncthreads();
    }
}
// the edge kernels' product: split-bf16 fragments or the fp32 image
template <bool kBwd, int kB, bool kBf = true>
__device__ __forceinline__ void mv16_edge(const float* wsm, int which, int i, int g, const f32x4 (&v)[2], f32x4 (&acc)[2]) {
#ifndef EQ_BF16X3_WHICH
#define EQ_BF16X3_WHICH 15
#endif
    static_assert(!W16::slim(kBwd, kB, kBf) || EQ_BF16X3_WHICH == 15, "the slim image has no fp32 matrices to fall back to");
    if (kBf && W16::use_bf16(kBwd, kB) && ((EQ_BF16X3_WHICH >> which) & 1))
        mv16_bf(wsm + W16::frag_at(kBwd, kB, W16::slim(kBwd, kB, kBf)) + which * W16::frag_floats, 16 * g + i, v, acc);
    else
        mv16(wsm + (which == W16::f_w2 ? W16::o_w2 : which == W16::f_w6 ? W16::o_w6 : which == W16::f_w2t ? W16::o_w2t : W16::o_w6t), i, g, v, acc);
}

// border of y = W v (hidden_nf = 33): acc += W[:, 32] vb (column vector `col`), returns W[32, :32] . v + wcc vb (row vector `row`)
__device__ __forceinline__ float mv16_border(const float* col, const float* row, float wcc, int g, const f32x4 (&v)[2], float vb,
                                             f32x4 (&acc)[2]) {
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[b] = fma4(lds4(col + 16 * b + 4 * g), splat4(vb), acc[b]);
    const float p = hsum4(fma4(lds4(row + 16 + 4 * g), v[1], lds4(row + 4 * g) * v[0]));
    return fmaf(wcc, vb, gsum(p));
}

// forward state of one 16-edge sub-tile (what the backward needs of it); the *b members: feature 32 (hidden_nf = 33)
struct Sub16 {
    f32x4 t1[2], sg1[2];          // pre-activation of the first edge-MLP layer, sigmoid(t1)
    f32x4 xhat[2], m[2];          // LayerNorm: normalised value, output (= the message m_ij)
    f32x4 q[2], sgq[2];           // coord-MLP pre-activation, sigmoid(q)
    float rstd, s;                // LayerNorm 1/std; coord scalar s = w7 . SiLU(q)   (:264)
    float t1b, sg1b, xhatb, mb, qb, sgqb;
};

// the lane's share (features 16 b + 4 g + r, and the border feature) of the two per-node projection rows of an edge:
// W_src h[row], W_tgt h[col] + b1.  FP = padded row of a [*, hidden_nf] buffer (32 or 36 floats).
template <int kB>
struct PRowsT {
    static constexpr int FP = 32 + 4 * kB;
    f32x4 pr[2], pc[2];
    float prb, pcb;
    __device__ __forceinline__ void load_r(const float* __restrict__ P, int r, int g) {       // row side: W_src h[row], a scattered gather
        const float* a = P + (size_t)r * (2 * FP);
        pr[0] = ldg4(a + 4 * g);
        pr[1] = ldg4(a + 16 + 4 * g);
        prb = kB ? a[32] : 0.0f;
    }
    __device__ __forceinline__ void load_c(const float* __restrict__ P, int c, int g) {       // col side: W_tgt h[col] + b1; consecutive edges share their centre
        const float* b_ = P + (size_t)c * (2 * FP) + FP;
        pc[0] = ldg4(b_ + 4 * g);
        pc[1] = ldg4(b_ + 16 + 4 * g);
        pcb = kB ? b_[32] : 0.0f;
    }
    __device__ __forceinline__ void load(const float* __restrict__ P, int r, int c, int g) { load_r(P, r, g); load_c(P, c, g); }
};
using PRows = PRowsT<0>;

// wsm: the workgroup's weight image (through an opaque offset so the loop-invariant reads stay inside the loop), bord: its
// border block; geo_row: this edge's 16 geometry values in LDS; pw: its projection rows (already in registers).
template <int kB, bool kBwdImg = false, bool kBf = true>
__device__ __forceinline__ void tile_forward16(const float* wsm, const float* bord, const float* geo_row, const PRowsT<kB>& pw, int i,
                                               int g, Sub16& o) {
    constexpr float kInvF = 1.0f / (32 + kB);
    constexpr bool kSl = W16::slim(kBwdImg, kB, kBf);
    constexpr int o_wg = W16::wg(kSl), o_vec = W16::vec(kSl);
    f32x4 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[b] = pw.pr[b] + pw.pc[b];
    o.t1b = 0.0f; o.sg1b = 0.0f; o.xhatb = 0.0f; o.mb = 0.0f; o.qb = 0.0f; o.sgqb = 0.0f;
    {   // + Wgeo [geo | attr]: K = 16, step s <-> geometry entry 4 g + s
        const f32x4 gv = lds4(geo_row + 4 * g);
        const f32x4 a0 = lds4(wsm + o_wg + i * W16::GS + 4 * g), a1 = lds4(wsm + o_wg + (16 + i) * W16::GS + 4 * g);
        EQ_STAMP(4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0] = mfma16(a0[s], gv[s], acc[0]);
            acc[1] = mfma16(a1[s], gv[s], acc[1]);
        }
        if constexpr (kB == 1) o.t1b = pw.prb + pw.pcb + gsum(hsum4(lds4(bord + W16::b_wgrow + 4 * g) * gv));
    }
    f32x4 a1v[2], u[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        o.t1[b] = acc[b];
        o.sg1[b] = sigmoid4(acc[b]);
        a1v[b] = acc[b] * o.sg1[b];                                    // SiLU (:245)
        u[b] = lds4(wsm + o_vec + 0 * 32 + 16 * b + 4 * g);       // b2
    }
    EQ_STAMP(5);
    mv16_edge<kBwdImg, kB, kBf>(wsm, W16::f_w2, i, g, a1v, u);
    float ub = 0.0f;
    if constexpr (kB == 1) {
        o.sg1b = sigmoid_fast(o.t1b);
        ub = bord[W16::b_sc + 2] + mv16_border(bord + W16::b_w2col, bord + W16::b_w2row, bord[W16::b_sc + 0], g, a1v, o.t1b * o.sg1b, u);
    }
    EQ_STAMP(6);
    // LayerNorm over the hidden features (biased variance, eps 1e-5; :249)
    const float nmean = (kB ? gsum(hsum8(u[0], u[1])) + ub : gsum(hsum8(u[0], u[1]))) * -kInvF;
    const f32x4 c0 = u[0] + splat4(nmean), c1 = u[1] + splat4(nmean);
    const float cb = ub + nmean;
    const float sq = gsum(hsum4(fma4(c1, c1, c0 * c0)));
    const float var = (kB ? fmaf(cb, cb, sq) : sq) * kInvF;
    o.rstd = __builtin_amdgcn_rsqf(var + 1e-5f);
    o.xhat[0] = c0 * splat4(o.rstd);
    o.xhat[1] = c1 * splat4(o.rstd);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const f32x4 gam = lds4(wsm + o_vec + 1 * 32 + 16 * b + 4 * g), bet = lds4(wsm + o_vec + 2 * 32 + 16 * b + 4 * g);
        o.m[b] = fma4(o.xhat[b], gam, bet);
        o.q[b] = lds4(wsm + o_vec + 3 * 32 + 16 * b + 4 * g);     // b6
    }
    EQ_STAMP(7);
    mv16_edge<kBwdImg, kB, kBf>(wsm, W16::f_w6, i, g, o.m, o.q);
    if constexpr (kB == 1) {
        o.xhatb = cb * o.rstd;
        o.mb = fmaf(o.xhatb, bord[W16::b_sc + 3], bord[W16::b_sc + 4]);
        o.qb = bord[W16::b_sc + 5] + mv16_border(bord + W16::b_w6col, bord + W16::b_w6row, bord[W16::b_sc + 1], g, o.m, o.mb, o.q);
    }
    EQ_STAMP(8);
    {
        o.sgq[0] = sigmoid4(o.q[0]);
        o.sgq[1] = sigmoid4(o.q[1]);
        const f32x4 w70 = lds4(wsm + o_vec + 4 * 32 + 4 * g), w71 = lds4(wsm + o_vec + 4 * 32 + 16 + 4 * g);
        o.s = gsum(hsum4(fma4(w71, o.q[1] * o.sgq[1], w70 * (o.q[0] * o.sgq[0]))));
    }
    if constexpr (kB == 1) {
        o.sgqb = sigmoid_fast(o.qb);
        o.s = fmaf(bord[W16::b_sc + 6], o.qb * o.sgqb, o.s);
    }
    EQ_STAMP(9);
}

// wave-private LDS scratch of a chunk (floats)
struct Chunk16 {
    static constexpr int GS = 16;                       // per-edge geometry row [radial, dist, dot, so3(9), attr, 0, 0, 0]
    static constexpr int o_geo = 0;                     // [64][GS]
    static constexpr int o_r = 64 * GS, o_c = o_r + 64, o_a = o_c + 64, o_b = o_a + 64;    // row idx, col idx, two words per edge
    static constexpr int fwd_size = o_b + 64;
    static constexpr int FS = 12;                       // backward: the edge's x_r, x_c, G_x[row] kept for the back phase
    static constexpr int o_front = fwd_size;            // [64][FS]
    static constexpr int LD = 36;                       // transposing tiles X, Y [16][LD] of the backward
    static constexpr int o_x = o_front + 64 * FS, o_y = o_x + 16 * LD;
    static constexpr int bwd_size = o_y + 16 * LD;
};

// front of a chunk: the lane's own edge -> geometry row [radial, dist, dot, so3(9), attr, 0, 0, 0] into LDS
__device__ __forceinline__ void store_geo_row(float* row, const EdgeGeo& eg, float attr) {
    st4(row + 0, f32x4{eg.g[0], eg.g[1], eg.g[2], eg.g[3]});
    st4(row + 4, f32x4{eg.g[4], eg.g[5], eg.g[6], eg.g[7]});
    st4(row + 8, f32x4{eg.g[8], eg.g[9], eg.g[10], eg.g[11]});
    st4(row + 12, f32x4{attr, 0.0f, 0.0f, 0.0f});
}

}  // namespace eq
