// SO3TensorProductLayer.forward (src/clean_so3_model.py:34-52) for gfx950 -- SURVEY §8(a) row A15, the one
// genuinely GEMM-shaped op of the reference:  y = W2 relu(W1 vec(D * x x^T) + b1) + b2  with D = input_dim
// (the bmm of the two broadcast views at :48-50 sums D identical products).
// The [E, D*D] tensor product is never materialised: a wave owns a 32-edge tile, keeps x[e][0..31] in
// registers and forms the B operand of every k-step (two entries of D*x_i*x_k) on the fly, streaming
// W1^T (256 KB, L2-resident) as the A operand of 2 x 512 v_mfma_f32_32x32x2_f32 per tile.
// Supported shape: input_dim = 32, output_dim = 32 (the reference's only instantiation, :85).
#include "egnn_mfma.h"

namespace eq {

constexpr int kSoD = 32, kSoH = 64, kSoO = 32;      // input_dim, hidden (2*output_dim), output_dim

// W1 [2*out][D*D] -> the per-lane A-operand stream: chunk i (= first tensor index), quad q, lane, 4 floats
//   w1p[((i*8 + q)*64 + lane)*4 + j] = W1[o = (j&1)*32 + (lane&31)][kk = i*32 + 2*(2q + (j>>1)) + (lane>>5)]
// so that every dwordx4 load of the main loop is one fully coalesced 1 KB wave access.
__global__ void so3_pack_w1_kernel(const float* __restrict__ w1, float* __restrict__ w1p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < kSoH * kSoD * kSoD) {
        const int j = t & 3, lane = (t >> 2) & 63, q = (t >> 8) & 7, i = t >> 11;
        const int kp = 2 * q + (j >> 1), o = (j & 1) * 32 + (lane & 31), kk = i * 32 + 2 * kp + (lane >> 5);
        w1p[t] = w1[o * (kSoD * kSoD) + kk];
    }
}

constexpr int kSoTiles = 2;            // 32-edge tiles per wave: halves the L2 operand stream per flop
constexpr int kSoXs = kSoD + 1;        // odd LDS row stride
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void so3_tp_fwd_kernel(const float* __restrict__ x, int n_rows, const float4* __restrict__ w1p,
                                                         const float* __restrict__ b1, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, float* __restrict__ y) {
    __shared__ float xs_all[4][kSoTiles][32][kSoXs];
    const int lane = threadIdx.x & 63, el = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wv, n_waves = gridDim.x * 4;
    float (*xs)[32][kSoXs] = xs_all[wv];
    const int groups = (n_rows + 32 * kSoTiles - 1) / (32 * kSoTiles);
    for (int grp = wave; grp < groups; grp += n_waves) {
        int e[kSoTiles];
        bool valid[kSoTiles];
        float xh[kSoTiles][16];                        // this half's k operands x[2kp + h]
#pragma unroll
        for (int t = 0; t < kSoTiles; ++t) {
            const int e_raw = (grp * kSoTiles + t) * 32 + el;
            valid[t] = e_raw < n_rows;
            e[t] = valid[t] ? e_raw : n_rows - 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {              // half h stages features 16h .. 16h+15 of its edge, pre-scaled by D
                const float4 a = *reinterpret_cast<const float4*>(x + (size_t)e[t] * kSoD + 16 * h + 4 * q);
                float* d = &xs[t][el][16 * h + 4 * q];
                d[0] = a.x * (float)kSoD; d[1] = a.y * (float)kSoD; d[2] = a.z * (float)kSoD; d[3] = a.w * (float)kSoD;
            }
#pragma unroll
            for (int kp = 0; kp < 16; ++kp) xh[t][kp] = x[(size_t)e[t] * kSoD + 2 * kp + h];
        }
        f32x16 acc[kSoTiles][2];
#pragma unroll
        for (int t = 0; t < kSoTiles; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[t][0][r] = b1[fidx(r, h)]; acc[t][1][r] = b1[32 + fidx(r, h)]; }
        // Operand stream, software-pipelined by hand: the compiler sinks ordinary prefetch loads to their first use
        // (exposing the L2 latency once per 16 MFMAs), so the 8 loads of chunk i+1 are issued as asm before the
        // 128 MFMAs of chunk i and collected by an explicit s_waitcnt that the consumers depend on.
        const f32x4* wp = reinterpret_cast<const f32x4*>(w1p) + lane;
        f32x4 bufA[8], bufB[8];
        auto issue = [&](f32x4 (&dst)[8], int chunk) {
            const f32x4* p = wp + (size_t)chunk * 8 * 64;
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst[q]) : "v"(p + q * 64) : "memory");
        };
        auto collect = [&](f32x4 (&d)[8]) {
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : : "memory");
        };
        auto multiply = [&](const f32x4 (&cur)[8], int i) {
            float xi[kSoTiles];
#pragma unroll
            for (int t = 0; t < kSoTiles; ++t) xi[t] = xs[t][el][i];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int t = 0; t < kSoTiles; ++t) {
                    const float bA = xi[t] * xh[t][2 * q], bB = xi[t] * xh[t][2 * q + 1];
                    acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[q][0], bA, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[q][1], bA, acc[t][1], 0, 0, 0);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[q][2], bB, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[q][3], bB, acc[t][1], 0, 0, 0);
                }
            }
        };
        // drain the compiler-tracked loads (x, biases) here: otherwise their pending waits are re-issued inside the
        // loop as vmcnt(0) and serialise the hand-issued prefetch as well
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
        issue(bufA, 0);
        collect(bufA);
#pragma unroll 1
        for (int i = 0; i < kSoD; i += 2) {
            issue(bufB, i + 1);
            __builtin_amdgcn_sched_barrier(0);         // nothing may cross: keeps the MFMAs between issue and collect
            multiply(bufA, i);
            __builtin_amdgcn_sched_barrier(0);
            collect(bufB);
            issue(bufA, i + 2 < kSoD ? i + 2 : i);
            __builtin_amdgcn_sched_barrier(0);
            multiply(bufB, i + 1);
            __builtin_amdgcn_sched_barrier(0);
            collect(bufA);
        }
        // second Linear: hidden (ReLU) sits in the D layout = B operand of the next product
#pragma unroll
        for (int t = 0; t < kSoTiles; ++t) {
            f32x16 out;
#pragma unroll
            for (int r = 0; r < 16; ++r) out[r] = b2[fidx(r, h)];
#pragma unroll
            for (int s = 0; s < 16; ++s)
                out = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[el * kSoH + fidx(s, h)], fmaxf(acc[t][0][s], 0.0f), out, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 16; ++s)
                out = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[el * kSoH + 32 + fidx(s, h)], fmaxf(acc[t][1][s], 0.0f), out, 0, 0, 0);
            if (valid[t]) {
                float o16[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) o16[r] = out[r];
                store_half_row(y + (size_t)e[t] * kSoO, h, o16);
            }
        }
    }
}

}  // namespace eq

extern "C" int equigspr_so3_tensor_product_forward(const float* x, int32_t n_rows, int32_t input_dim, int32_t output_dim,
                                                   const float* w1, const float* b1, const float* w2, const float* b2,
                                                   float* w1t_scratch, float* y, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(x && w1 && b1 && w2 && b2 && w1t_scratch && y && n_rows >= 1, EQUIGSPR_EINVAL, "so3_tensor_product: bad arguments");
    EQ_CHECK_ARG(input_dim == kSoD && output_dim == kSoO, EQUIGSPR_ENOTSUP,
                 "so3_tensor_product: only input_dim=32, output_dim=32 (src/clean_so3_model.py:85) is compiled");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(so3_pack_w1_kernel, dim3(ceil_div(kSoH * kSoD * kSoD, 256)), dim3(256), 0, s, w1, w1t_scratch);
    const int groups = ceil_div(n_rows, 32 * kSoTiles);
    int blocks = ceil_div(groups, 4);
    blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
    hipLaunchKernelGGL(so3_tp_fwd_kernel, dim3(blocks), dim3(256), 0, s, x, n_rows, (const float4*)w1t_scratch, b1, w2, b2, y);
    EQ_CHECK_LAUNCH("so3_tensor_product_forward");
    return 0;
}
