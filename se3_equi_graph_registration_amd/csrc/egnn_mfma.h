// Small shared pieces of the 32 x 32 x 2 fp32 MFMA idiom (v_mfma_f32_32x32x2_f32) used by the node-level outer products
// (egnn_bwd.hip), the SO(3) tensor-product layer (so3_tp.hip) and the descriptor matcher (match.hip), plus the layout of the
// activations the EGNN forward keeps for its backward.  The edge kernels themselves use the 16-edge formulation of
// egnn_mfma16.h (v_mfma_f32_16x16x4_f32).
//
// D tile of the 32x32x2 instruction: lane l = (column j = l & 31, half h = l >> 5) holds rows fidx(r, h), r = 0..15.
#pragma once
#include "egnn_device.h"

namespace eq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ constexpr int fidx(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }   // partner lane of the other half

// 16 of the 32 floats of a 128-byte row: features fidx(r,h), r = 0..15 (4 x 16-byte accesses)
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

static inline int64_t saved_node_floats(int FP, int L, int64_t N) {      // h_0..h_L, x_0..x_L, agg_0..agg_{L-1}; padded to 16 bytes
    return ((int64_t)(L + 1) * N * FP + (int64_t)(L + 1) * N * 3 + (int64_t)L * N * FP + 3) / 4 * 4;
}

}  // namespace eq
