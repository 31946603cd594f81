// Device side of the batched weighted Kabsch forward (moments + 3x3 SVD solve), shared by csrc/kabsch.hip (two launches) and
// the training head's tail kernel in csrc/head.hip (one launch: the last moments block of a pair solves it).
#pragma once
#include "common.h"

namespace eq {

__device__ inline double det3(const double m[9]) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// 1/x and 1/sqrt(x) for normal positive x: hardware seed (~2^-26) + two Newton steps (full fp64 precision to a few ulp)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = fma(fma(-0.5 * x * r, r, 0.5), r, r);
    r = fma(fma(-0.5 * x * r, r, 0.5), r, r);
    return r;
}

// One-sided Jacobi: A V = B with orthogonal columns; sigma_i = |B_i|, U_i = B_i / sigma_i.
__device__ inline void svd3_jacobi(const double A[9], double U[9], double sig[3], double V[9]) {
    double B[9];
    for (int i = 0; i < 9; ++i) { B[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool rotated = false;
        for (int pair = 0; pair < 3; ++pair) {
            const int p = pair == 2 ? 1 : 0, q = pair == 0 ? 1 : 2;
            double al = 0, be = 0, ga = 0;
            for (int r = 0; r < 3; ++r) {
                al += B[r * 3 + p] * B[r * 3 + p];
                be += B[r * 3 + q] * B[r * 3 + q];
                ga += B[r * 3 + p] * B[r * 3 + q];
            }
            if (fabs(ga) > 1e-300 && ga * ga > 1e-30 * (al * be)) {   // |ga| > 1e-15 sqrt(al be)  (1e-17 is below fp64 round-off: it never converged)
                rotated = true;
                // tan of the rotation angle, the smaller root of t^2 + 2 zeta t - 1 = 0 with zeta = (be - al) / (2 ga):
                //   t = sign(de) 2 ga / (|de| + sqrt(de^2 + 4 ga^2)),  c = 1 / sqrt(1 + t^2),  s = c t
                // with hardware reciprocal / reciprocal-square-root seeds + Newton steps (two rsqrt, one rcp) instead of three IEEE
                // divisions and two square roots on one lane -- any (c, s) with c^2 + s^2 = 1 to round-off keeps the iteration orthogonal
                const double de = be - al;
                const double w = de * de + 4.0 * ga * ga;
                const double tt = copysign(2.0 * ga, de >= 0.0 ? ga : -ga) * fast_rcp(fabs(de) + w * fast_rsqrt(w));
                const double c = fast_rsqrt(1.0 + tt * tt), s = c * tt;
                for (int r = 0; r < 3; ++r) {
                    const double bp = B[r * 3 + p], bq = B[r * 3 + q];
                    B[r * 3 + p] = c * bp - s * bq;
                    B[r * 3 + q] = s * bp + c * bq;
                    const double vp = V[r * 3 + p], vq = V[r * 3 + q];
                    V[r * 3 + p] = c * vp - s * vq;
                    V[r * 3 + q] = s * vp + c * vq;
                }
            }
        }
        if (!rotated) break;
    }
    for (int i = 0; i < 3; ++i) sig[i] = sqrt(B[0 * 3 + i] * B[0 * 3 + i] + B[1 * 3 + i] * B[1 * 3 + i] + B[2 * 3 + i] * B[2 * 3 + i]);
    // smallest direction may be numerically null: complete it from the other two
    int lo = 0;
    if (sig[1] < sig[lo]) lo = 1;
    if (sig[2] < sig[lo]) lo = 2;
    const double smax = fmax(sig[0], fmax(sig[1], sig[2]));
    for (int i = 0; i < 3; ++i) {
        if (i == lo && sig[i] <= 1e-280 + 1e-18 * smax) continue;
        for (int r = 0; r < 3; ++r) U[r * 3 + i] = B[r * 3 + i] / sig[i];
    }
    if (sig[lo] <= 1e-280 + 1e-18 * smax) {
        const int a = (lo + 1) % 3, b = (lo + 2) % 3;
        U[0 * 3 + lo] = U[1 * 3 + a] * U[2 * 3 + b] - U[2 * 3 + a] * U[1 * 3 + b];
        U[1 * 3 + lo] = U[2 * 3 + a] * U[0 * 3 + b] - U[0 * 3 + a] * U[2 * 3 + b];
        U[2 * 3 + lo] = U[0 * 3 + a] * U[1 * 3 + b] - U[1 * 3 + a] * U[0 * 3 + b];
    }
}

// aux layout (floats): cs[0:3] ct[3:6] U[6:15] s~[15:18] R[18:27] sum w(s-cs)[27:30] sum w(t-ct)[30:33]
//
// Forward in two short launches so that a single pair (B = 1) is not serialised on one CU:
//   kabsch_moments_kernel  grid (nb, B): fp64 partial raw moments  W = sum w, S = sum w s, T = sum w t, M = sum w s t^T
//   kabsch_solve_kernel    grid B: cs = fl32(S), ct = fl32(T) (the reference does not divide by W),
//                          H = sum w (s-cs)(t-ct)^T = M - S ct^T - cs T^T + W cs ct^T, then the 3x3 SVD on one lane.
constexpr int kKabschParts = 16;   // doubles per partial record: W, S[3], T[3], M[9]
constexpr int kKabschMaxBlocks = 8;   // partial records live in the pair's aux record behind the first 48 floats
static_assert(48 + kKabschMaxBlocks * kKabschParts * 2 <= EQUIGSPR_KABSCH_AUX, "aux record too small");
__device__ __forceinline__ double* kabsch_part(float* aux, int b) { return reinterpret_cast<double*>(aux + (size_t)b * EQUIGSPR_KABSCH_AUX + 48); }

__device__ __forceinline__ double block_sum_d256(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// one block's share of pair b's moments: block xb of nblocks (256 threads)
__device__ __forceinline__ void kabsch_moments_block(const float* __restrict__ src, const float* __restrict__ tgt,
                                                     const float* __restrict__ w, int n, float* __restrict__ aux, int b, int xb,
                                                     int nblocks) {
    double* part = kabsch_part(aux, b);
    const float* s = src + (size_t)b * n * 3;
    const float* t = tgt + (size_t)b * n * 3;
    const float* wb = w + (size_t)b * n;
    double m[kKabschParts];
#pragma unroll
    for (int q = 0; q < kKabschParts; ++q) m[q] = 0.0;
    for (int i = xb * 256 + threadIdx.x; i < n; i += nblocks * 256) {
        const double wi = wb[i];
        const double sv[3] = {s[i * 3], s[i * 3 + 1], s[i * 3 + 2]}, tv[3] = {t[i * 3], t[i * 3 + 1], t[i * 3 + 2]};
        m[0] += wi;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            m[1 + a] += wi * sv[a];
            m[4 + a] += wi * tv[a];
#pragma unroll
            for (int c = 0; c < 3; ++c) m[7 + a * 3 + c] += wi * sv[a] * tv[c];
        }
    }
    // all 16 sums together.  Per wave a reduce-scatter butterfly: at distance o = 32, 16, 8, 4 a lane keeps one half of its values
    // (the half selected by lane & o) and adds the partner's copy of it -- 8 + 4 + 2 + 1 exchanges instead of 16 x 6 -- then two
    // more steps over lane & 3: lane l with (l & 3) == 0 ends up with the wave total of quantity q(l).  ONE barrier, then threads
    // 0..15 add the four wave totals in wave order.  (16 separate block sums, 32 barriers, were 6 us of the tail launch.)
    __shared__ double wsum[4][kKabschParts];
    const int lane = threadIdx.x & 63;
    int q_of_lane = 0;
    {
        double v[kKabschParts];
#pragma unroll
        for (int q = 0; q < kKabschParts; ++q) v[q] = m[q];
        int n = kKabschParts;
#pragma unroll
        for (int o = 32; o >= 4; o >>= 1) {
            n >>= 1;
            const bool up = (lane & o) != 0;
#pragma unroll
            for (int i2 = 0; i2 < 8; ++i2) {
                if (i2 < n) {
                    const double keep = up ? v[i2 + n] : v[i2], send = up ? v[i2] : v[i2 + n];
                    v[i2] = keep + __shfl_xor(send, o, 64);
                }
            }
            q_of_lane += up ? n : 0;
        }
        v[0] += __shfl_xor(v[0], 2, 64);
        v[0] += __shfl_xor(v[0], 1, 64);
        m[0] = v[0];
    }
    __syncthreads();                                     // (a previous use of wsum by this block is complete)
    if ((lane & 3) == 0) wsum[threadIdx.x >> 6][q_of_lane] = m[0];
    __syncthreads();
    if (threadIdx.x < kKabschParts)
        part[(size_t)xb * kKabschParts + threadIdx.x] = wsum[0][threadIdx.x] + wsum[1][threadIdx.x] + wsum[2][threadIdx.x] + wsum[3][threadIdx.x];
}

// pair b's solve from its nb partial records; called by ONE full wave (lane = 0..63)
__device__ __forceinline__ void kabsch_solve_wave(int nb, float* __restrict__ Rout, float* __restrict__ tout, float* __restrict__ aux,
                                                  int b, int lane) {
    const double* part = kabsch_part(aux, b);
    // lane q < 16 adds quantity q over the nb (<= 8) partial records in block order; lane 0 then collects the 16 sums through
    // v_readlane (the 16 six-step fp64 shuffle trees this replaces were a quarter of the solve)
    double mine = 0.0;
    if (lane < kKabschParts) {
        double pv[kKabschMaxBlocks];
#pragma unroll
        for (int k = 0; k < kKabschMaxBlocks; ++k) pv[k] = k < nb ? part[(size_t)k * kKabschParts + lane] : 0.0;
#pragma unroll
        for (int k = 0; k < kKabschMaxBlocks; ++k) mine += pv[k];
    }
    double m[kKabschParts];
#pragma unroll
    for (int q = 0; q < kKabschParts; ++q) {
        const unsigned long long bits = __double_as_longlong(mine);
        const unsigned lo_ = __builtin_amdgcn_readlane((int)(unsigned)(bits & 0xffffffffull), q);
        const unsigned hi_ = __builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), q);
        m[q] = __longlong_as_double(((unsigned long long)hi_ << 32) | lo_);
    }
    if (lane == 0) {
        const double W = m[0];
        const float csf[3] = {(float)m[1], (float)m[2], (float)m[3]}, ctf[3] = {(float)m[4], (float)m[5], (float)m[6]};
        double H[9], U[9], V[9], sg[3];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {
                // H about the fp32-ROUNDED centroids the backward also uses (consistency matters: with near
                // one-hot weights s_j - cs is ~1e-5 |s| and the gradient terms cancel to that accuracy)
                const float h = (float)(m[7 + a * 3 + c] - m[1 + a] * (double)ctf[c] - (double)csf[a] * m[4 + c] +
                                        W * (double)csf[a] * (double)ctf[c]);
                H[a * 3 + c] = (double)(h + ((a == c) ? 1e-6f : 0.0f));                        // :738 (fp32 add)
            }
        svd3_jacobi(H, U, sg, V);
        int lo = 0;
        if (sg[1] < sg[lo]) lo = 1;
        if (sg[2] < sg[lo]) lo = 2;
        double sgn[3] = {1.0, 1.0, 1.0};
        if (det3(V) * det3(U) < 0.0) sgn[lo] = -1.0;                                             // :749-751
        double R[9];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {
                double acc = 0;
                for (int i = 0; i < 3; ++i) acc += sgn[i] * V[a * 3 + i] * U[c * 3 + i];
                R[a * 3 + c] = acc;
            }
        float* ax = aux + (size_t)b * EQUIGSPR_KABSCH_AUX;
        for (int a = 0; a < 3; ++a) {
            const double ta = (double)ctf[a] - (R[a * 3 + 0] * csf[0] + R[a * 3 + 1] * csf[1] + R[a * 3 + 2] * csf[2]);   // :754
            tout[b * 3 + a] = (float)ta;
            ax[a] = csf[a];
            ax[3 + a] = ctf[a];
            ax[15 + a] = (float)(sgn[a] * sg[a]);
            ax[27 + a] = (float)(m[1 + a] - W * (double)csf[a]);       // sum w (s - cs)
            ax[30 + a] = (float)(m[4 + a] - W * (double)ctf[a]);       // sum w (t - ct)
        }
        for (int i = 0; i < 9; ++i) {
            Rout[b * 9 + i] = (float)R[i];
            ax[6 + i] = (float)U[i];
            ax[18 + i] = (float)R[i];
        }
    }
}

// Kabsch backward of pair b, node share xb of nxb (256 threads; every block redoes the 3x3 part on its thread 0):
// closed form through the polar factor (finite for repeated singular values); grads w.r.t. the clouds and the weights.
__device__ __forceinline__ void kabsch_bwd_block(const float* __restrict__ src, const float* __restrict__ tgt,
                                                 const float* __restrict__ w, const float* __restrict__ aux,
                                                 const float* gRb /* [9] of pair b */, const float* gtb /* [3] */, int n,
                                                 float* __restrict__ gsrc, float* __restrict__ gtgt, float* __restrict__ gw,
                                                 int b, int xb, int nxb) {
    __shared__ float sh[9 + 3 + 3];   // gH, g_cs, g_ct
    const float* ax = aux + (size_t)b * EQUIGSPR_KABSCH_AUX;
    if (threadIdx.x == 0) {
        double cs[3], U[9], st[3], R[9], G[9], g_cs[3], g_ct[3], gtv[3];
        for (int a = 0; a < 3; ++a) { cs[a] = ax[a]; st[a] = ax[15 + a]; gtv[a] = gtb[a]; }
        for (int i = 0; i < 9; ++i) { U[i] = ax[6 + i]; R[i] = ax[18 + i]; }
        for (int a = 0; a < 3; ++a) {
            g_ct[a] = gtv[a];
            g_cs[a] = -(R[0 * 3 + a] * gtv[0] + R[1 * 3 + a] * gtv[1] + R[2 * 3 + a] * gtv[2]);
            for (int c = 0; c < 3; ++c) G[a * 3 + c] = (double)gRb[a * 3 + c] - gtv[a] * cs[c];
        }
        // M = U^T R^T G U
        double T1[9], T2[9], M[9];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {                 // T1 = R^T G
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += R[k * 3 + a] * G[k * 3 + c];
                T1[a * 3 + c] = acc;
            }
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {                 // T2 = T1 U
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += T1[a * 3 + k] * U[k * 3 + c];
                T2[a * 3 + c] = acc;
            }
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {                 // M = U^T T2
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += U[k * 3 + a] * T2[k * 3 + c];
                M[a * 3 + c] = acc;
            }
        double Nm[9];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {
                const double den = st[a] + st[c];
                Nm[a * 3 + c] = (a != c && fabs(den) > 1e-30) ? M[a * 3 + c] / den : 0.0;
            }
        double Sk[9];                                       // N - N^T
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) Sk[a * 3 + c] = Nm[a * 3 + c] - Nm[c * 3 + a];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {                 // T1 = U Sk
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += U[a * 3 + k] * Sk[k * 3 + c];
                T1[a * 3 + c] = acc;
            }
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {                 // T2 = T1 U^T  (= K - K^T)
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += T1[a * 3 + k] * U[c * 3 + k];
                T2[a * 3 + c] = acc;
            }
        double gH[9];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) {                 // gA = R T2 ; gH = gA^T
                double acc = 0;
                for (int k = 0; k < 3; ++k) acc += R[a * 3 + k] * T2[k * 3 + c];
                gH[c * 3 + a] = acc;
            }
        for (int a = 0; a < 3; ++a) {
            double a1 = 0, a2 = 0;
            for (int c = 0; c < 3; ++c) {
                a1 += gH[a * 3 + c] * ax[30 + c];            // gH * sum w (t-ct)
                a2 += gH[c * 3 + a] * ax[27 + c];            // gH^T * sum w (s-cs)
            }
            g_cs[a] -= a1;
            g_ct[a] -= a2;
        }
        for (int i = 0; i < 9; ++i) sh[i] = (float)gH[i];
        for (int a = 0; a < 3; ++a) { sh[9 + a] = (float)g_cs[a]; sh[12 + a] = (float)g_ct[a]; }
    }
    __syncthreads();
    float gH[9], g_cs[3], g_ct[3], cs[3], ct[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) gH[i] = sh[i];
#pragma unroll
    for (int a = 0; a < 3; ++a) { g_cs[a] = sh[9 + a]; g_ct[a] = sh[12 + a]; cs[a] = ax[a]; ct[a] = ax[3 + a]; }
    const float* s = src + (size_t)b * n * 3;
    const float* t = tgt + (size_t)b * n * 3;
    for (int i = xb * 256 + threadIdx.x; i < n; i += nxb * 256) {
        const float wi = w[(size_t)b * n + i];
        float sv[3], tv[3], ds[3], dt[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { sv[a] = s[i * 3 + a]; tv[a] = t[i * 3 + a]; ds[a] = sv[a] - cs[a]; dt[a] = tv[a] - ct[a]; }
        float gwi = 0.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float hs = gH[a * 3 + 0] * dt[0] + gH[a * 3 + 1] * dt[1] + gH[a * 3 + 2] * dt[2];      // (gH dt)_a
            const float ht = gH[0 * 3 + a] * ds[0] + gH[1 * 3 + a] * ds[1] + gH[2 * 3 + a] * ds[2];      // (gH^T ds)_a
            if (gsrc) gsrc[((size_t)b * n + i) * 3 + a] = wi * (hs + g_cs[a]);
            if (gtgt) gtgt[((size_t)b * n + i) * 3 + a] = wi * (ht + g_ct[a]);
            gwi += ds[a] * hs + sv[a] * g_cs[a] + tv[a] * g_ct[a];
        }
        if (gw) gw[(size_t)b * n + i] = gwi;
    }
}

__host__ __device__ inline int kabsch_bwd_blocks(int n) { const int nb = (n + 255) / 256; return nb < 1 ? 1 : (nb > 8 ? 8 : nb); }

}  // namespace eq
