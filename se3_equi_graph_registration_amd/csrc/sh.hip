// Real spherical harmonics and the equivariant kernel basis for gfx950.
//   tools/spherical_harmonics.py:35-123  lpmv / get_spherical_harmonics (tesseral, Condon-Shortley phase)
//   tools/basis.py:57-95                 get_spherical_from_cartesian (axes (x, y, z) = components (2, 0, 1))
//   tools/basis.py:140-151,:153-205      precompute_sh, get_basis:  K_J = Y_J Q_J^T per (d_in, d_out)
// All degrees 0..max_l come from ONE pass per point: the Legendre recursion runs in fp64 registers and the
// trigonometric factors come from the Cartesian components (or the given angles) by a Chebyshev recurrence, so
// no per-degree re-evaluation and no memo dict.  HBM-bound: 12 B in, 4 (max_l+1)^2 B out per point.
#include "common.h"

namespace eq {

constexpr int kShMaxL = 8;
struct ShNorm { double n[(kShMaxL + 1) * (kShMaxL + 1)]; };    // N_l^m at [l*(kShMaxL+1) + m], m >= 0

// mode 0: in = cartesian [n,3] (also writes sph = [radius, alpha, beta] when non-null)
// mode 1: in = [n,2] (theta, phi)            mode 2: in = spherical [n,3] (radius, alpha, beta) -> theta = pi - beta, phi = alpha
template <typename TIn>
__global__ void sh_kernel(const TIn* __restrict__ in, int n, int mode, int max_l, ShNorm norm, float* __restrict__ sph,
                          float* __restrict__ y, double* __restrict__ y64) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double ct, st, cphi, sphi;
    if (mode == 0) {
        const double c0 = (double)in[3 * i], c1 = (double)in[3 * i + 1], c2 = (double)in[3 * i + 2];
        const double rxy2 = c2 * c2 + c0 * c0, rxy = sqrt(rxy2), r = sqrt(rxy2 + c1 * c1);
        // theta = pi - beta with beta = atan2(rxy, c1):  cos(theta) = -c1/r, sin(theta) = rxy/r;  atan2(0, 0) = 0
        ct = r > 0.0 ? -c1 / r : -1.0;
        st = r > 0.0 ? rxy / r : 0.0;
        cphi = rxy > 0.0 ? c2 / rxy : 1.0;               // phi = alpha = atan2(c0, c2)
        sphi = rxy > 0.0 ? c0 / rxy : 0.0;
        if (sph) {
            sph[3 * i] = (float)r;
            sph[3 * i + 1] = (float)atan2(c0, c2);
            sph[3 * i + 2] = (float)atan2(rxy, c1);
        }
    } else {
        double theta, phi;
        if (mode == 1) { theta = (double)in[2 * i]; phi = (double)in[2 * i + 1]; }
        else { theta = 3.14159265358979323846 - (double)in[3 * i + 2]; phi = (double)in[3 * i + 1]; }
        ct = cos(theta);
        st = sqrt(fmax(0.0, 1.0 - ct * ct));             // pow(1 - x^2, m/2) of the reference: |sin|^m
        cphi = cos(phi); sphi = sin(phi);
    }
    const int stride = (max_l + 1) * (max_l + 1);
    double cm = 1.0, sm = 0.0;                           // cos(m phi), sin(m phi)
    double pmm = 1.0;                                    // P_m^m = (-1)^m (2m-1)!! sin^m
    for (int m = 0; m <= max_l; ++m) {
        if (m > 0) {
            const double c_next = cm * cphi - sm * sphi;
            sm = sm * cphi + cm * sphi;
            cm = c_next;
            pmm *= -(double)(2 * m - 1) * st;
        }
        double p2 = 0.0, p1 = pmm;
        for (int l = m; l <= max_l; ++l) {
            double p;
            if (l == m) p = pmm;
            else {
                p = ((double)(2 * l - 1) * ct * p1 - (double)(l + m - 1) * p2) / (double)(l - m);
                p2 = p1; p1 = p;
            }
            const double nl = norm.n[l * (kShMaxL + 1) + m] * p;
            const size_t base = (size_t)i * stride + l * l + l;
            if (m == 0) {
                if (y) y[base] = (float)nl;
                if (y64) y64[base] = nl;
            } else {
                if (y) { y[base + m] = (float)(nl * cm); y[base - m] = (float)(nl * sm); }
                if (y64) { y64[base + m] = nl * cm; y64[base - m] = nl * sm; }
            }
        }
    }
}

// out[e][oi][jx] = sum_m Y_J[e][m] * Q_J[oi][m],  J = |d_in - d_out| + jx   (basis.py:186-198)
// q: [nJ][OI][QM] (QM = 2 (d_in + d_out) + 1, rows zero-padded)
__global__ void sh_basis_kernel(const float* __restrict__ y, long long n, int y_stride, int j0, int n_j, int oi_count, int qm,
                                const float* __restrict__ q, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = oi_count * n_j;
    if (t >= n * per) return;
    const long long e = t / per;
    const int r = (int)(t - e * per), oi = r / n_j, jx = r - oi * n_j, J = j0 + jx;
    const float* yr = y + e * y_stride + J * J;
    const float* qr = q + ((size_t)jx * oi_count + oi) * qm;
    float s = 0.0f;
    for (int m = 0; m < 2 * J + 1; ++m) s = fmaf(yr[m], qr[m], s);
    out[t] = s;
}

}  // namespace eq

namespace {
eq::ShNorm make_norm() {
    eq::ShNorm t;
    for (int l = 0; l <= eq::kShMaxL; ++l)
        for (int m = 0; m <= eq::kShMaxL; ++m) {
            double v = 0.0;
            if (m <= l) {
                v = (2.0 * l + 1.0) / (4.0 * 3.14159265358979323846);
                if (m > 0) {
                    double poch = 1.0;                   // (l-m+1)(l-m+2)...(l+m)   (spherical_harmonics.py:25-27,:103)
                    for (int x = l - m + 1; x <= l + m; ++x) poch *= (double)x;
                    v *= 2.0 / poch;
                }
                v = sqrt(v);
            }
            t.n[l * (eq::kShMaxL + 1) + m] = v;
        }
    return t;
}
}  // namespace

extern "C" int equigspr_sh_forward(const void* in, int32_t in_is_f64, int64_t n, int32_t mode, int32_t max_l, float* sph, float* y,
                                   double* y64, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(in && (y || y64) && n >= 1 && n < (1ll << 31), EQUIGSPR_EINVAL, "sh_forward: bad arguments");
    EQ_CHECK_ARG(mode >= 0 && mode <= 2 && (mode == 0 || !sph), EQUIGSPR_EINVAL, "sh_forward: mode in {0 cartesian, 1 (theta,phi), 2 spherical}");
    EQ_CHECK_ARG(max_l >= 0 && max_l <= kShMaxL, EQUIGSPR_ENOTSUP, "sh_forward: max_l <= 8 (max_degree <= 4)");
    static const ShNorm norm = make_norm();
    const dim3 grid(ceil_div(n, 128)), block(128);
    if (in_is_f64)
        hipLaunchKernelGGL(sh_kernel<double>, grid, block, 0, (hipStream_t)stream, (const double*)in, (int)n, mode, max_l, norm, sph, y, y64);
    else
        hipLaunchKernelGGL(sh_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)in, (int)n, mode, max_l, norm, sph, y, y64);
    EQ_CHECK_LAUNCH("sh_forward");
    return 0;
}

extern "C" int equigspr_sh_basis_forward(const float* y, int64_t n, int32_t max_l, int32_t d_in, int32_t d_out, const float* q,
                                         float* out, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(y && q && out && n >= 1 && d_in >= 0 && d_out >= 0, EQUIGSPR_EINVAL, "sh_basis_forward: bad arguments");
    EQ_CHECK_ARG(d_in + d_out <= max_l && max_l <= kShMaxL, EQUIGSPR_EINVAL, "sh_basis_forward: Y must hold degrees up to d_in + d_out");
    const int j0 = d_in > d_out ? d_in - d_out : d_out - d_in, n_j = 2 * (d_in < d_out ? d_in : d_out) + 1;
    const int oi = (2 * d_out + 1) * (2 * d_in + 1), qm = 2 * (d_in + d_out) + 1;
    const long long total = (long long)n * oi * n_j;
    hipLaunchKernelGGL(sh_basis_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, y, (long long)n,
                       (max_l + 1) * (max_l + 1), j0, n_j, oi, qm, q, out);
    EQ_CHECK_LAUNCH("sh_basis_forward");
    return 0;
}
