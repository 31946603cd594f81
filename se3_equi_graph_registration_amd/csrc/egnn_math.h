// Per-edge math of one E_GCL layer, shared by the HIP kernels and by the host-side unit
// test (tests/test_host_math.py compiles this header with g++ to check the formulas on CPU
// before they ever reach a GPU).  Everything here is `EQ_HD` (host + device) and scalar.
//
// Reference math: src/3dmatch_train_egnn_with_batch.py:
//   compute_so3_matrix :128-173, compute_edge_features :176-181, E_GCL.coord2radial :271-278.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define EQ_HD __host__ __device__ __forceinline__
#else
#define EQ_HD inline
#endif

namespace eq {

// z-vector geometry block, in the reference's concatenation order (:238):
//   [radial, dist, dot, a0,b0,c0, a1,b1,c1, a2,b2,c2]
constexpr int kGeo = 12;

struct EdgeGeo {
    float d[3];       // coord_diff = x[row] - x[col]                      (:273)
    float g[kGeo];    // radial, dist, dot, so3_flat[9]
    float inv_na;     // 1/(|d|+1e-8)    (kept for backward)
    float inv_nb;     // 1/(|p|+1e-8)
    float p[3];       // cross(x_r, x_c)
    float pn;         // |p|
    bool  degenerate; // frame replaced by identity (:152-163)
};

// sigmoid with the hardware reciprocal (v_rcp_f32, <= 1 ulp) instead of the ~10-instruction IEEE division
// sequence; used by the MFMA edge kernels where SiLU / SiLU' are a third of the vector work.
EQ_HD float sigmoid_fast(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    // raw v_exp_f32 (2^x) and v_rcp_f32: 4 instructions; expf() expands to a ~10-instruction range-reduced sequence whose
    // extra accuracy (the exponent product rounds to |v| * 2^-24) is irrelevant once 1 is added
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
#else
    return 1.0f / (1.0f + expf(-v));
#endif
}
EQ_HD float silu(float v) { return v * sigmoid_fast(v); }
// d/dv silu(v) = s(v) * (1 + v * (1 - s(v)))
EQ_HD float silu_grad(float v) {
    const float s = sigmoid_fast(v);
    return s * (1.0f + v * (1.0f - s));
}

// Square root / reciprocal of the per-edge geometry.  On the device: the hardware instructions (v_sqrt_f32, v_rcp_f32,
// <= 1 ulp) instead of the correctly rounded ~15-instruction sequences hipcc emits for sqrtf() and 1.0f / x -- the edge
// geometry was a sixth of the edge kernels' vector instructions.  The host twin keeps libm.
// CONTRACT: the identity-frame decision of edge_geometry (norm < 1e-6, train :152-163) is evaluated on these <= 1-ulp values.
// Device and host twin can take different branches only when a norm lies within an ulp of 1e-6, i.e. |x_r - x_c| or
// |x_r x x_c| of order 1e-14 (coordinates within ~1e-7 of each other and of the origin); exact zeros -- self loops, duplicated
// points, a point at the origin -- give exactly 0 on both sides.  Outside that band the results agree to the anchored
// tolerances of the parity tests, not bit for bit ("Frame decision contract").
#if defined(__HIP_DEVICE_COMPILE__)
#define EQ_SQRT(v) __builtin_amdgcn_sqrtf(v)
#define EQ_RCP(v) __builtin_amdgcn_rcpf(v)
#else
#define EQ_SQRT(v) sqrtf(v)
#define EQ_RCP(v) (1.0f / (v))
#endif

EQ_HD void cross3(const float* a, const float* b, float* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
EQ_HD float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

EQ_HD void edge_geometry(const float xr[3], const float xc[3], EdgeGeo& e) {
    for (int i = 0; i < 3; ++i) e.d[i] = xr[i] - xc[i];
    const float radial = dot3(e.d, e.d);                    // :274
    const float dist = EQ_SQRT(radial);                     // :179
    e.g[0] = radial;
    e.g[1] = dist;
    e.g[2] = dot3(xr, xc);                                  // :180
    e.inv_na = EQ_RCP(dist + 1e-8f);                        // :140
    float a[3], b[3], c[3];
    for (int i = 0; i < 3; ++i) a[i] = e.d[i] * e.inv_na;
    cross3(xr, xc, e.p);                                    // :143
    e.pn = EQ_SQRT(dot3(e.p, e.p));
    e.inv_nb = EQ_RCP(e.pn + 1e-8f);                        // :144
    for (int i = 0; i < 3; ++i) b[i] = e.p[i] * e.inv_nb;
    cross3(a, b, c);                                        // :149
    const float thr2 = 1e-6f * 1e-6f;                       // :130, norms compared as squares
    e.degenerate = (dot3(a, a) < thr2) || (dot3(b, b) < thr2) || (dot3(c, c) < thr2);
    if (e.degenerate) {                                     // :163 identity frame
        for (int i = 0; i < 9; ++i) e.g[3 + i] = (i % 4 == 0) ? 1.0f : 0.0f;
    } else {                                                // stack(dim=2).view(-1,9): [a_i, b_i, c_i] per row i
        for (int i = 0; i < 3; ++i) {
            e.g[3 + 3 * i + 0] = a[i];
            e.g[3 + 3 * i + 1] = b[i];
            e.g[3 + 3 * i + 2] = c[i];
        }
    }
}

// Backward of edge_geometry (+ the direct coord_diff path of coord_model, :264).
//   gg[12]  = dL/d g,   gd[3] = dL/d coord_diff (from trans = coord_diff * s)
// Outputs dL/dx_r, dL/dx_c (overwritten).
EQ_HD void edge_geometry_backward(const float xr[3], const float xc[3], const EdgeGeo& e,
                                  const float gg[kGeo], const float gd[3], float gxr[3], float gxc[3]) {
    float gdt[3];                      // total gradient w.r.t. d = x_r - x_c
    const float dist = e.g[1];
    const float inv_dist = dist > 0.0f ? EQ_RCP(dist) : 0.0f;    // torch.norm backward: 0 at 0
    for (int i = 0; i < 3; ++i) {
        gdt[i] = gd[i] + 2.0f * e.d[i] * gg[0] + e.d[i] * inv_dist * gg[1];
        gxr[i] = xc[i] * gg[2];
        gxc[i] = xr[i] * gg[2];
    }
    if (!e.degenerate) {
        float a[3], b[3], ga[3], gb[3], gc[3];
        for (int i = 0; i < 3; ++i) {
            a[i] = e.d[i] * e.inv_na;
            b[i] = e.p[i] * e.inv_nb;
            ga[i] = gg[3 + 3 * i + 0];
            gb[i] = gg[3 + 3 * i + 1];
            gc[i] = gg[3 + 3 * i + 2];
        }
        // c = a x b  ->  ga += b x gc ; gb += gc x a
        float t0[3], t1[3];
        cross3(b, gc, t0);
        cross3(gc, a, t1);
        for (int i = 0; i < 3; ++i) { ga[i] += t0[i]; gb[i] += t1[i]; }
        // a = d / (n + eps):  gd += ga/(n+eps) - d (d.ga) / (n (n+eps)^2)
        const float dga = dot3(e.d, ga);
        const float ka = dga * inv_dist * e.inv_na * e.inv_na;
        for (int i = 0; i < 3; ++i) gdt[i] += ga[i] * e.inv_na - e.d[i] * ka;
        // b = p / (|p| + eps)
        const float inv_pn = e.pn > 0.0f ? EQ_RCP(e.pn) : 0.0f;
        const float pgb = dot3(e.p, gb);
        const float kb = pgb * inv_pn * e.inv_nb * e.inv_nb;
        float gp[3];
        for (int i = 0; i < 3; ++i) gp[i] = gb[i] * e.inv_nb - e.p[i] * kb;
        // p = x_r x x_c  ->  gxr += x_c x gp ; gxc += gp x x_r
        cross3(xc, gp, t0);
        cross3(gp, xr, t1);
        for (int i = 0; i < 3; ++i) { gxr[i] += t0[i]; gxc[i] += t1[i]; }
    }
    for (int i = 0; i < 3; ++i) { gxr[i] += gdt[i]; gxc[i] -= gdt[i]; }
}

// ------------------------------------------------------------------------------------
// edge_basis = EQUIGSPR_BASIS_IRREPS: the 12 geometry entries of z are rotation INVARIANTS obtained by
// contracting the degree <= 1 equivariant basis kernels of tools/basis.py::get_basis (K_J = Y_J Q_J^T, :179-198) with the
// type-1 features an edge has, its endpoint coordinates -- instead of the raw frame-vector components of compute_so3_matrix,
// which are not rotation covariant.  With d = x_r - x_c, dh = d / (|d| + 1e-8), the kernels have the closed forms (checked
// against sh.get_basis / oracle.sh_oracle.get_basis in the tests, signs = this package's deterministic Q_J signs)
//   K^(1->0)   = kC10 dh^T            K_0^(1->1) = kC0 I          K_2^(1->1) = kC2 (dh dh^T - I / 3)
//   K_1^(1->1) = kC1 [dh]_x  (its contraction x_r^T K_1 x_c = kC1 dh . (x_r x x_c) vanishes identically: entry 10 is 0)
// and z_geo = [ |d|^2, |d|, x_r^T K_0 x_c, K^(1->0) x_r, K^(1->0) x_c, x_r^T K_2 x_c, x_r^T K_2 x_r, x_c^T K_2 x_c,
//               x_r^T K_0 x_r, x_c^T K_0 x_c, 0, 0 ].
// Every entry is invariant under x -> R x, so h is invariant and the coordinate update (x_r - x_c) s covariant: the layer
// is SO(3)-equivariant about the origin (the reference's frame features are not).  Zero-length edges (self loops): dh = 0
// with zero gradient.
// ------------------------------------------------------------------------------------
constexpr float kC0 = 0.16286750396763996f;      // 1 / sqrt(12 pi)
constexpr float kC10 = -0.28209479177387814f;    // -1 / (2 sqrt(pi))
constexpr float kC2 = 0.3454941494713355f;       // sqrt(3 / (8 pi))
EQ_HD void edge_geometry_irreps(const float xr[3], const float xc[3], EdgeGeo& e) {
    for (int i = 0; i < 3; ++i) e.d[i] = xr[i] - xc[i];
    const float radial = dot3(e.d, e.d);
    const float dist = EQ_SQRT(radial);
    e.inv_na = dist > 0.0f ? EQ_RCP(dist + 1e-8f) : 0.0f;
    float dh[3];
    for (int i = 0; i < 3; ++i) dh[i] = e.d[i] * e.inv_na;
    const float u = dot3(dh, xr), v = dot3(dh, xc), p = dot3(xr, xc), a2 = dot3(xr, xr), b2 = dot3(xc, xc);
    e.g[0] = radial;
    e.g[1] = dist;
    e.g[2] = kC0 * p;
    e.g[3] = kC10 * u;
    e.g[4] = kC10 * v;
    e.g[5] = kC2 * (u * v - p * (1.0f / 3.0f));
    e.g[6] = kC2 * (u * u - a2 * (1.0f / 3.0f));
    e.g[7] = kC2 * (v * v - b2 * (1.0f / 3.0f));
    e.g[8] = kC0 * a2;
    e.g[9] = kC0 * b2;
    e.g[10] = 0.0f;
    e.g[11] = 0.0f;
    e.inv_nb = 0.0f; e.pn = 0.0f; e.p[0] = e.p[1] = e.p[2] = 0.0f;
    e.degenerate = false;
}
EQ_HD void edge_geometry_irreps_backward(const float xr[3], const float xc[3], const EdgeGeo& e, const float gg[kGeo],
                                         const float gd[3], float gxr[3], float gxc[3]) {
    const float dist = e.g[1];
    const float inv_dist = dist > 0.0f ? EQ_RCP(dist) : 0.0f;
    float dh[3];
    for (int i = 0; i < 3; ++i) dh[i] = e.d[i] * e.inv_na;
    const float u = dot3(dh, xr), v = dot3(dh, xc);
    const float Lu = kC10 * gg[3] + kC2 * (gg[5] * v + 2.0f * gg[6] * u);
    const float Lv = kC10 * gg[4] + kC2 * (gg[5] * u + 2.0f * gg[7] * v);
    const float Lp = kC0 * gg[2] - kC2 * gg[5] * (1.0f / 3.0f);
    const float La = kC0 * gg[8] - kC2 * gg[6] * (1.0f / 3.0f);
    const float Lb = kC0 * gg[9] - kC2 * gg[7] * (1.0f / 3.0f);
    float Gh[3];                                   // dL / d dh
    for (int i = 0; i < 3; ++i) Gh[i] = Lu * xr[i] + Lv * xc[i];
    // dh = d / (n + eps):  G_d = G_dh / (n + eps) - d (d . G_dh) / (n (n + eps)^2);  n = 0: dh is the constant 0
    const float dgh = dot3(e.d, Gh);
    const float kd = dgh * inv_dist * e.inv_na * e.inv_na;
    for (int i = 0; i < 3; ++i) {
        const float gdt = gd[i] + 2.0f * e.d[i] * gg[0] + e.d[i] * inv_dist * gg[1] + Gh[i] * e.inv_na - e.d[i] * kd;
        gxr[i] = Lu * dh[i] + Lp * xc[i] + 2.0f * La * xr[i] + gdt;
        gxc[i] = Lv * dh[i] + Lp * xr[i] + 2.0f * Lb * xc[i] - gdt;
    }
}
// the two geometry modes behind one call (wave-uniform branch in the kernels)
EQ_HD void edge_geometry_mode(int basis, const float xr[3], const float xc[3], EdgeGeo& e) {
    if (basis) edge_geometry_irreps(xr, xc, e);
    else edge_geometry(xr, xc, e);
}
EQ_HD void edge_geometry_backward_mode(int basis, const float xr[3], const float xc[3], const EdgeGeo& e, const float gg[kGeo],
                                       const float gd[3], float gxr[3], float gxc[3]) {
    if (basis) edge_geometry_irreps_backward(xr, xc, e, gg, gd, gxr, gxc);
    else edge_geometry_backward(xr, xc, e, gg, gd, gxr, gxc);
}

// ------------------------------------------------------------------------------------
// Flat parameter-block layout (include/equigspr.h "parameter block").
// ------------------------------------------------------------------------------------
struct LayerLayout {
    int F, H, A, FH, D;
    // offsets (floats) relative to the layer block
    int head_stride, w1, b1, w2, b2;      // per head: + j*head_stride
    int ln_g, ln_b, w4, b4, w5, b5, w6, b6, w7, size;
};

EQ_HD LayerLayout make_layer_layout(int F, int H, int A) {
    LayerLayout L;
    L.F = F; L.H = H; L.A = A; L.FH = F / H; L.D = 2 * F + kGeo + A;
    L.w1 = 0;
    L.b1 = L.w1 + L.FH * L.D;
    L.w2 = L.b1 + L.FH;
    L.b2 = L.w2 + L.FH * L.FH;
    L.head_stride = L.b2 + L.FH;
    int o = L.H * L.head_stride;
    L.ln_g = o; o += F;
    L.ln_b = o; o += F;
    L.w4 = o; o += F * 2 * F;
    L.b4 = o; o += F;
    L.w5 = o; o += F * F;
    L.b5 = o; o += F;
    L.w6 = o; o += F * F;
    L.b6 = o; o += F;
    L.w7 = o; o += F;
    L.size = o;
    return L;
}

// Compile-time twin of LayerLayout for the kernels (every offset folds to an immediate).
template <int F_, int H_, int A_>
struct LP {
    static constexpr int F = F_, H = H_, A = A_;
    static constexpr int FH = F / H;
    static constexpr int D = 2 * F + kGeo + A;
    static constexpr int w1 = 0;
    static constexpr int b1 = w1 + FH * D;
    static constexpr int w2 = b1 + FH;
    static constexpr int b2 = w2 + FH * FH;
    static constexpr int head_stride = b2 + FH;
    static constexpr int ln_g = H * head_stride;
    static constexpr int ln_b = ln_g + F;
    static constexpr int w4 = ln_b + F;
    static constexpr int b4 = w4 + F * 2 * F;
    static constexpr int w5 = b4 + F;
    static constexpr int b5 = w5 + F * F;
    static constexpr int w6 = b5 + F;
    static constexpr int b6 = w6 + F * F;
    static constexpr int w7 = b6 + F;
    static constexpr int size = w7 + F;
    // W1[o][c], stacked over heads
    static constexpr EQ_HD int w1_off(int o, int c) { return (o / FH) * head_stride + w1 + (o % FH) * D + c; }
    static constexpr EQ_HD int b1_off(int o) { return (o / FH) * head_stride + b1 + (o % FH); }
    // W2 is block diagonal: valid only when o and i are in the same head
    static constexpr EQ_HD int w2_off(int o, int i) { return (o / FH) * head_stride + w2 + (o % FH) * FH + (i % FH); }
    static constexpr EQ_HD int b2_off(int o) { return (o / FH) * head_stride + b2 + (o % FH); }
};

struct ModelLayout {
    int in_nf, F, out_nf, A, L, H;
    int w_in, b_in, w_out, b_out, layer0, layer_size, total;
};
EQ_HD ModelLayout make_model_layout(int in_nf, int F, int out_nf, int A, int L, int H) {
    ModelLayout M;
    M.in_nf = in_nf; M.F = F; M.out_nf = out_nf; M.A = A; M.L = L; M.H = H;
    int o = 0;
    M.w_in = o; o += F * in_nf;
    M.b_in = o; o += F;
    M.w_out = o; o += out_nf * F;
    M.b_out = o; o += out_nf;
    M.layer0 = o;
    M.layer_size = make_layer_layout(F, H, A).size;
    M.total = o + L * M.layer_size;
    return M;
}

}  // namespace eq
