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

// kHid (backward, round 6): the same first Linear, the pre-ReLU hidden layer h1 [n_rows, 64] is stored to `y` and the second Linear skipped
template <bool kHid = false>
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
        if constexpr (kHid) {
#pragma unroll
            for (int t = 0; t < kSoTiles; ++t) {
                if (valid[t]) {
                    float lo16[16], hi16[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) { lo16[r] = acc[t][0][r]; hi16[r] = acc[t][1][r]; }
                    store_half_row(y + (size_t)e[t] * kSoH, h, lo16);
                    store_half_row(y + (size_t)e[t] * kSoH + 32, h, hi16);
                }
            }
            continue;
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

// ---------------------------------------------------------------------------------------
// backward (round 6; clean_so3_model.py:34-52 as a trainable module -- no script of the reference imports that module).
//   h1 = W1 z + b1 (z = D x (x) x, never materialised)          so3_tp_fwd_kernel<true>
//   dh = (W2^T dy) . [h1 > 0], a = ReLU(h1)                      so3_dh_kernel (in place over h1; a to its own buffer)
//   dW2 = dy^T a, db2 = sum dy, db1 = sum dh                     so3_small_grads_kernel: per-workgroup slabs, summed in fixed order
//   dW1[o][i 32 + j] = sum_e dh[e][o] D x_i x_j                  so3_w1_grad_kernel: 32 x 32 x 2 MFMA, K = edges, B operand formed on the fly
//   dx[i] = D sum_j (G[i][j] + G[j][i]) x_j, G = W1^T dh         so3_dx_kernel: G block by block on the MFMA, contracted in registers
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void so3_dh_kernel(const float* __restrict__ dy, const float* __restrict__ w2, int n_rows,
                                                     float* __restrict__ hid /* in: h1, out: dh */, float* __restrict__ act) {
    __shared__ float w2s[kSoO * kSoH];                 // W2 [32][64]
    __shared__ float dys[4][kSoO];
    for (int t = threadIdx.x; t < kSoO * kSoH; t += 256) w2s[t] = w2[t];
    const int sl = threadIdx.x >> 6, o = threadIdx.x & 63;
    __syncthreads();
    for (int e0 = blockIdx.x * 4; e0 < n_rows; e0 += gridDim.x * 4) {
        const int e = e0 + sl;
        if (e < n_rows && o < kSoO) dys[sl][o] = dy[(size_t)e * kSoO + o];
        __syncthreads();
        if (e < n_rows) {
            float da = 0.0f;
#pragma unroll
            for (int q = 0; q < kSoO; ++q) da = fmaf(w2s[q * kSoH + o], dys[sl][q], da);
            const float h1 = hid[(size_t)e * kSoH + o];
            act[(size_t)e * kSoH + o] = fmaxf(h1, 0.0f);
            hid[(size_t)e * kSoH + o] = h1 > 0.0f ? da : 0.0f;
        }
        __syncthreads();
    }
}

// slab per workgroup: dW2 [32][64] | db2 [32] | db1 [64] over the workgroup's rows
constexpr int kSoSmall = kSoO * kSoH + kSoO + kSoH;
__global__ __launch_bounds__(256) void so3_small_grads_kernel(const float* __restrict__ dy, const float* __restrict__ act,
                                                              const float* __restrict__ dh, int n_rows, int rows_per_block,
                                                              float* __restrict__ slab) {
    __shared__ float dys[32][kSoO + 1];
    __shared__ float as[32][kSoH + 1];
    __shared__ float dhs[32][kSoH + 1];
    const int q = threadIdx.x & 31, ob = threadIdx.x >> 5;        // thread: dW2[q][8 ob .. 8 ob + 7]
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0f;
    float sb2 = 0.0f, sb1 = 0.0f;                                  // threads 0..31: db2[q]; threads 0..63: db1[o]
    const int lo = blockIdx.x * rows_per_block, hi = min(n_rows, lo + rows_per_block);
    for (int e0 = lo; e0 < hi; e0 += 32) {
        const int cnt = min(32, hi - e0);
        __syncthreads();
        for (int t = threadIdx.x; t < 32 * kSoO; t += 256) { const int r = t >> 5, c = t & 31; dys[r][c] = r < cnt ? dy[(size_t)(e0 + r) * kSoO + c] : 0.0f; }
        for (int t = threadIdx.x; t < 32 * kSoH; t += 256) {
            const int r = t >> 6, c = t & 63;
            as[r][c] = r < cnt ? act[(size_t)(e0 + r) * kSoH + c] : 0.0f;
            dhs[r][c] = r < cnt ? dh[(size_t)(e0 + r) * kSoH + c] : 0.0f;
        }
        __syncthreads();
        for (int r = 0; r < 32; ++r) {
            const float d = dys[r][q];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = fmaf(d, as[r][8 * ob + u], acc[u]);
        }
        if (threadIdx.x < 32) for (int r = 0; r < 32; ++r) sb2 += dys[r][threadIdx.x];
        if (threadIdx.x < 64) for (int r = 0; r < 32; ++r) sb1 += dhs[r][threadIdx.x];
    }
    float* my = slab + (size_t)blockIdx.x * kSoSmall;
#pragma unroll
    for (int u = 0; u < 8; ++u) my[q * kSoH + 8 * ob + u] = acc[u];
    if (threadIdx.x < 32) my[kSoO * kSoH + threadIdx.x] = sb2;
    if (threadIdx.x < 64) my[kSoO * kSoH + kSoO + threadIdx.x] = sb1;
}

// dst[idx] = sum_s slab[s * stride + idx] for idx < count: a 256-thread workgroup = 32 entries x 8 partitions; partition p sums the slabs
// p, p + 8, ... (coalesced 128-byte rows, eight independent chains instead of one), the eight partials are added in order: fixed order
__global__ __launch_bounds__(256) void so3_slab_sum_kernel(const float* __restrict__ slab, int n_slabs, int stride, int count,
                                                           float* __restrict__ dst) {
    __shared__ float red[8][33];
    const int lane = threadIdx.x & 31, part = threadIdx.x >> 5, idx = blockIdx.x * 32 + lane;
    float a = 0.0f;
    if (idx < count)
        for (int s = part; s < n_slabs; s += 8) a += slab[(size_t)s * stride + idx];
    red[part][lane] = a;
    __syncthreads();
    if (part == 0 && idx < count) {
        float v = red[0][lane];
#pragma unroll
        for (int p = 1; p < 8; ++p) v += red[p][lane];
        dst[idx] = v;
    }
}

// dW1: grid (32 first indices i, kSoKs K-splits), 4 waves per workgroup each over a quarter of the split's rows:
//   D[o][j] += dh[e][o] * (D x_i[e] x_j[e]) on v_mfma_f32_32x32x2_f32 (two rows per instruction), two o-blocks;
//   the waves' tiles are added in LDS in fixed order -> slab[ks][o][i 32 + j]
constexpr int kSoKs = 32;          // 32 x 32 workgroups of 4 waves: four waves per SIMD hide the row loads (8 splits, one wave per SIMD: 508 us at 81 920 rows)
__global__ __launch_bounds__(256) void so3_w1_grad_kernel(const float* __restrict__ x, const float* __restrict__ dh, int n_rows,
                                                          float* __restrict__ slab) {
    __shared__ float part[4][2][32 * 33];
    const int i = blockIdx.x, ks = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, el = lane & 31, hh = lane >> 5;
    const int per_split = (n_rows + kSoKs - 1) / kSoKs, per_wave = ((per_split + 3) / 4 + 1) & ~1;      // even: two rows per instruction
    const int lo = ks * per_split + wv * per_wave;
    const int hi = min(min(n_rows, (ks + 1) * per_split), lo + per_wave);
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; }
    for (int e0 = lo; e0 < hi; e0 += 16) {
        float d0[8], d1[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {                  // eight instructions' operands in flight
            const int e = e0 + 2 * u + hh;
            const int ec = e < hi ? e : hi - 1;        // clamped address, the mask applied to the value
            const float keep = e < hi ? 1.0f : 0.0f;
            d0[u] = dh[(size_t)ec * kSoH + el] * keep;
            d1[u] = dh[(size_t)ec * kSoH + 32 + el] * keep;
            bv[u] = (float)kSoD * x[(size_t)ec * kSoD + i] * x[(size_t)ec * kSoD + el];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(d0[u], bv[u], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(d1[u], bv[u], a1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {                     // D[row o = fidx(r, hh)][col j = el]
        part[wv][0][fidx(r, hh) * 33 + el] = a0[r];
        part[wv][1][fidx(r, hh) * 33 + el] = a1[r];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * 32 * 32; t += 256) {
        const int blk = t >> 10, o = (t >> 5) & 31, j = t & 31;
        const float v = part[0][blk][o * 33 + j] + part[1][blk][o * 33 + j] + part[2][blk][o * 33 + j] + part[3][blk][o * 33 + j];
        slab[((size_t)ks * kSoH + 32 * blk + o) * (kSoD * kSoD) + i * kSoD + j] = v;
    }
}

// dx: a wave per 32-row tile.  Block i of G = W1^T dh (rows i 32 + j, j = 0..31): D[j][e] = sum_o W1[o][i 32 + j] dh[e][o] (32 instructions);
// s1[e][i] = sum_j G[i][j] x_j (lane partial + the other half), t[j] += G[i][j] x_i;  dx[e][j] = D (t[j] + s1[e][j])
__global__ __launch_bounds__(256) void so3_dx_kernel(const float* __restrict__ x, const float* __restrict__ dh, const float* __restrict__ w1,
                                                     int n_rows, float* __restrict__ dx) {
    __shared__ float xs_all[4][32][kSoD + 1];
    __shared__ float s1_all[4][32][kSoD + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, el = lane & 31, hh = lane >> 5;
    float (*xs)[kSoD + 1] = xs_all[wv];
    float (*s1)[kSoD + 1] = s1_all[wv];
    const int tiles = (n_rows + 31) / 32;
    for (int tile = blockIdx.x * 4 + wv; tile < tiles; tile += gridDim.x * 4) {
        const int e_raw = tile * 32 + el;
        const bool valid = e_raw < n_rows;
        const int e = valid ? e_raw : n_rows - 1;
        float bop[32];                                 // B operands: dh[e][2 s + hh]
#pragma unroll
        for (int s_ = 0; s_ < 32; ++s_) bop[s_] = valid ? dh[(size_t)e * kSoH + 2 * s_ + hh] : 0.0f;
        float xj[16], tacc[16];
        load_half_row(x + (size_t)e * kSoD, hh, xj);
#pragma unroll
        for (int r = 0; r < 16; ++r) { tacc[r] = 0.0f; xs[el][fidx(r, hh)] = xj[r]; }
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < kSoD; ++i) {
            f32x16 g;
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = 0.0f;
            const float* wrow = w1 + (size_t)i * kSoD + el;          // W1[o][i 32 + el], o-stride D*D
#pragma unroll
            for (int s_ = 0; s_ < 32; ++s_)                          // fully unrolled: bop[] stays in registers, the 32 weight loads go out together
                g = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[(size_t)(2 * s_ + hh) * (kSoD * kSoD)], bop[s_], g, 0, 0, 0);
            const float xi = xs[el][i];
            float p = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { p = fmaf(g[r], xj[r], p); tacc[r] = fmaf(g[r], xi, tacc[r]); }
            p += xhalf(p);
            if (hh == 0) s1[el][i] = p;
        }
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            float o16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) o16[r] = (float)kSoD * (tacc[r] + s1[el][fidx(r, hh)]);
            store_half_row(dx + (size_t)e * kSoD, hh, o16);
        }
        __builtin_amdgcn_wave_barrier();
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
    hipLaunchKernelGGL(so3_tp_fwd_kernel<false>, dim3(blocks), dim3(256), 0, s, x, n_rows, (const float4*)w1t_scratch, b1, w2, b2, y);
    EQ_CHECK_LAUNCH("so3_tensor_product_forward");
    return 0;
}

namespace eq {
static int so3_small_blocks(int n_rows) { int b = ceil_div(n_rows, 128); return b < 1 ? 1 : (b > 256 ? 256 : b); }      // slab copies of dW2 | db2 | db1
}
/* floats of `work` for equigspr_so3_tensor_product_backward */
extern "C" int64_t equigspr_so3_tensor_product_backward_work_size(int32_t n_rows) {
    using namespace eq;
    if (n_rows < 1) return -1;
    return (int64_t)kSoH * kSoD * kSoD                              /* packed W1 */
         + 2 * (int64_t)n_rows * kSoH                              /* h1 -> dh, ReLU(h1) */
         + (int64_t)so3_small_blocks(n_rows) * kSoSmall            /* slabs of dW2 | db2 | db1 */
         + (int64_t)kSoKs * kSoH * kSoD * kSoD;                    /* K-split slabs of dW1 */
}

/* Backward of equigspr_so3_tensor_product_forward (ABI 14; input_dim = output_dim = 32).  grad_y [n_rows, 32]; grad_x [n_rows, 32],
 * grad_w1 [64, 1024], grad_b1 [64], grad_w2 [32, 64], grad_b2 [32]: each may be NULL; written, not accumulated.  The [n_rows, D*D] tensor
 * product is not materialised in the backward either.  Deterministic (fixed-order slab sums). */
extern "C" int equigspr_so3_tensor_product_backward(const float* x, const float* grad_y, int32_t n_rows, int32_t input_dim, int32_t output_dim,
                                                    const float* w1, const float* b1, const float* w2, float* work, float* grad_x,
                                                    float* grad_w1, float* grad_b1, float* grad_w2, float* grad_b2, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(x && grad_y && w1 && b1 && w2 && work && n_rows >= 1, EQUIGSPR_EINVAL, "so3_tensor_product_backward: bad arguments");
    EQ_CHECK_ARG(input_dim == kSoD && output_dim == kSoO, EQUIGSPR_ENOTSUP,
                 "so3_tensor_product_backward: only input_dim=32, output_dim=32 (src/clean_so3_model.py:85) is compiled");
    hipStream_t s = (hipStream_t)stream;
    float* w1p = work;
    float* hid = w1p + (size_t)kSoH * kSoD * kSoD;
    float* act = hid + (size_t)n_rows * kSoH;
    const int sb = so3_small_blocks(n_rows);
    float* slab_s = act + (size_t)n_rows * kSoH;
    float* slab_w1 = slab_s + (size_t)sb * kSoSmall;
    hipLaunchKernelGGL(so3_pack_w1_kernel, dim3(ceil_div(kSoH * kSoD * kSoD, 256)), dim3(256), 0, s, w1, w1p);
    const int groups = ceil_div(n_rows, 32 * kSoTiles);
    int blocks = ceil_div(groups, 4);
    blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
    hipLaunchKernelGGL(so3_tp_fwd_kernel<true>, dim3(blocks), dim3(256), 0, s, x, n_rows, (const float4*)w1p, b1, w2, (const float*)nullptr, hid);
    int dblocks = ceil_div(n_rows, 4);
    dblocks = dblocks > 1024 ? 1024 : dblocks;
    hipLaunchKernelGGL(so3_dh_kernel, dim3(dblocks), dim3(256), 0, s, grad_y, w2, n_rows, hid, act);
    if (grad_w2 || grad_b2 || grad_b1) {
        const int rpb = ceil_div(ceil_div(n_rows, sb), 32) * 32;
        hipLaunchKernelGGL(so3_small_grads_kernel, dim3(sb), dim3(256), 0, s, grad_y, act, hid, n_rows, rpb, slab_s);
        // the slab is dW2 | db2 | db1: three destination ranges of one fixed-order sum
        if (grad_w2) hipLaunchKernelGGL(so3_slab_sum_kernel, dim3(ceil_div(kSoO * kSoH, 32)), dim3(256), 0, s, slab_s, sb, kSoSmall, kSoO * kSoH, grad_w2);
        if (grad_b2) hipLaunchKernelGGL(so3_slab_sum_kernel, dim3(ceil_div(kSoO, 32)), dim3(256), 0, s, slab_s + kSoO * kSoH, sb, kSoSmall, kSoO, grad_b2);
        if (grad_b1) hipLaunchKernelGGL(so3_slab_sum_kernel, dim3(ceil_div(kSoH, 32)), dim3(256), 0, s, slab_s + kSoO * kSoH + kSoO, sb, kSoSmall, kSoH, grad_b1);
    }
    if (grad_w1) {
        hipLaunchKernelGGL(so3_w1_grad_kernel, dim3(kSoD, kSoKs), dim3(256), 0, s, x, hid, n_rows, slab_w1);
        hipLaunchKernelGGL(so3_slab_sum_kernel, dim3(ceil_div(kSoH * kSoD * kSoD, 32)), dim3(256), 0, s, slab_w1, kSoKs, kSoH * kSoD * kSoD, kSoH * kSoD * kSoD, grad_w1);
    }
    if (grad_x) {
        int xb = ceil_div(ceil_div(n_rows, 32), 4);
        xb = xb < 1 ? 1 : (xb > 1024 ? 1024 : xb);
        hipLaunchKernelGGL(so3_dx_kernel, dim3(xb), dim3(256), 0, s, x, hid, w1, n_rows, grad_x);
    }
    EQ_CHECK_LAUNCH("so3_tensor_product_backward");
    return 0;
}
