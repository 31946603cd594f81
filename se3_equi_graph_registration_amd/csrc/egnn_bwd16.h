// Shared pieces of the backward kernels (egnn_bwd.hip: edge kernels and the per-feature node kernels; egnn_node_bwd.hip: the
// 16-node MFMA node kernels): the per-workgroup gradient slab layouts and the outer-product pass over a 16-row operand tile.
#pragma once
#include "egnn_mfma16.h"

namespace eq {

#ifndef EQ_NODE_BWD_BLOCKS
#define EQ_NODE_BWD_BLOCKS 1024
#endif
constexpr int kNodeBwdBlocks = EQ_NODE_BWD_BLOCKS;   // max slab count of the node-level kernels

// ---- slab sizes ---------------------------------------------------------------------------
template <int F> struct Slabs {
    static constexpr int node = F * 2 * F + F + F * F + F;          // dW4 | db4 | dW5 | db5
    static constexpr int gather = 2 * F * F + F;                    // dW1a | dW1b | db1
    static constexpr int e_geo = 0;                                  // F x 16 (12 geo + attr + pad)
    static constexpr int e_w2 = F * 16;
    static constexpr int e_w6 = e_w2 + F * F;
    static constexpr int e_b2 = e_w6 + F * F;
    static constexpr int e_b6 = e_b2 + F;
    static constexpr int e_gam = e_b6 + F;
    static constexpr int e_bet = e_gam + F;
    static constexpr int e_w7 = e_bet + F;
    static constexpr int edge = e_w7 + F;
};


// acc[bx][by] += X^T Y over the 16 edges of the tiles X [16][LD], Y [16][ldy] (16 columns per block);
// K step s, lane group g <-> edge s + 4 g (rows 4 apart: the two groups of a 32-lane LDS access hit disjoint banks).
// Border (hidden_nf = 33, kB): column 32 of a tile row holds the edge's 33rd feature (a scalar xb / yb per edge):
//   bcol[bx] += X[:, 16 bx + i] * yb (column 32 of the gradient), brow[by] += xb * Y[:, 16 by + i] (row 32),
//   corner += xb * yb, xbs += xb; all per lane group g, reduced over g after the chunk loop.
struct PassBorder {
    float bcol[2], brow[2], corner, xbs;
    __device__ __forceinline__ void init() { bcol[0] = bcol[1] = brow[0] = brow[1] = corner = xbs = 0.0f; }
};
template <int NBY, bool kColSum, int kB = 0>
__device__ __forceinline__ void pass16(const float* X, const float* Y, int ldy, int i, int g, f32x4 (&acc)[2][NBY], f32x2& cs,
                                       PassBorder& pb) {
    float a0[4], a1[4], b0[4], b1[4], xb[4], yb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int ed = s + 4 * g;
        a0[s] = X[ed * Chunk16::LD + i];
        a1[s] = X[ed * Chunk16::LD + 16 + i];
        b0[s] = Y[ed * ldy + i];
        b1[s] = (NBY == 2) ? Y[ed * ldy + 16 + i] : 0.0f;
        xb[s] = kB ? X[ed * Chunk16::LD + 32] : 0.0f;
        yb[s] = (kB && NBY == 2) ? Y[ed * ldy + 32] : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (kColSum) cs += f32x2{a0[s], a1[s]};          // the two column sums as one packed add
        acc[0][0] = mfma16(a0[s], b0[s], acc[0][0]);
        acc[1][0] = mfma16(a1[s], b0[s], acc[1][0]);
        if constexpr (NBY == 2) {
            acc[0][1] = mfma16(a0[s], b1[s], acc[0][1]);
            acc[1][1] = mfma16(a1[s], b1[s], acc[1][1]);
        }
        if constexpr (kB == 1) {
            pb.brow[0] = fmaf(xb[s], b0[s], pb.brow[0]);
            if constexpr (NBY == 2) {
                pb.brow[1] = fmaf(xb[s], b1[s], pb.brow[1]);
                pb.bcol[0] = fmaf(a0[s], yb[s], pb.bcol[0]);
                pb.bcol[1] = fmaf(a1[s], yb[s], pb.bcol[1]);
                pb.corner = fmaf(xb[s], yb[s], pb.corner);
                if (kColSum) pb.xbs += xb[s];
            }
        }
    }
}


// arguments of the 16-node MFMA node backward (egnn_node_bwd.hip).  mode bits: kGather = gather backward of layer l (lw_g, h_g,
// dt1 / gxr / gxc of its edge backward), kNode = node-MLP backward of layer l - 1 (or of layer L - 1 when alone: G_h from Gh_in)
struct NodeBwd16Args {
    const float* lw_g; const float* lw_n;
    const int* rptr; const int* rperm; const int* cptr; const int* cperm;
    const float* dt1; const float* gxr; const float* gxc; const float* Gx;
    const float* h_g; const float* Gh_in; const float* h_n; const float* agg_n;
    int n_nodes, n_edges;
    float* dh_part; float* dagg; float* P; float* Gx_new; float* gx_out3; float* Gh_out;
    float* gat_slab; float* node_slab;
    // kNodeBwdEmbed (hidden_nf == in_nf == out_nf): with the node-only launch the embedding_out backward (w_out, gh_out [N][F], gx_out
    // [N][3], hL = the last layer's output rows; G_x goes to Gx_new), with the gather-only launch the embedding_in backward (w_in,
    // h_in [N][F], grad_h_in [N][F] or NULL); emb_slab = that embedding's slab set
    const float* w_out; const float* gh_out; const float* gx_out; const float* hL;
    const float* w_in; const float* h_in; float* grad_h_in;
    float* emb_slab;
};
enum { kNodeBwdGather = 1, kNodeBwdNode = 2, kNodeBwdEmbed = 4 };
// launches the kernel for (hidden_nf F, heads H, edge attributes A); *n_slabs = slab copies written (per slab kind)
int launch_node_bwd16(int F, int H, int A, int mode, const NodeBwd16Args& args, hipStream_t s, int* n_slabs);
int node_bwd16_max_slabs(int n_nodes);

}  // namespace eq
