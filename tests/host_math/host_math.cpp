// Host build of se3_equi_graph_registration_amd/csrc/egnn_math.h for the CPU unit tests
// (tests/test_host_logic.py): the same scalar code the HIP kernels inline.
#include "../../se3_equi_graph_registration_amd/csrc/egnn_math.h"

extern "C" {
// out: g[12], d[3], degenerate flag
void hm_edge_geometry(const float* xr, const float* xc, float* g, float* d, int* degenerate) {
    eq::EdgeGeo e;
    eq::edge_geometry(xr, xc, e);
    for (int i = 0; i < eq::kGeo; ++i) g[i] = e.g[i];
    for (int i = 0; i < 3; ++i) d[i] = e.d[i];
    *degenerate = e.degenerate ? 1 : 0;
}
void hm_edge_geometry_backward(const float* xr, const float* xc, const float* gg, const float* gd, float* gxr, float* gxc) {
    eq::EdgeGeo e;
    eq::edge_geometry(xr, xc, e);
    eq::edge_geometry_backward(xr, xc, e, gg, gd, gxr, gxc);
}
// edge_basis = irreps (N4): forward values and backward of the irrep-basis invariants
void hm_edge_geometry_irreps(const float* xr, const float* xc, float* g, float* d) {
    eq::EdgeGeo e;
    eq::edge_geometry_mode(1, xr, xc, e);
    for (int i = 0; i < eq::kGeo; ++i) g[i] = e.g[i];
    for (int i = 0; i < 3; ++i) d[i] = e.d[i];
}
void hm_edge_geometry_irreps_backward(const float* xr, const float* xc, const float* gg, const float* gd, float* gxr, float* gxc) {
    eq::EdgeGeo e;
    eq::edge_geometry_mode(1, xr, xc, e);
    eq::edge_geometry_backward_mode(1, xr, xc, e, gg, gd, gxr, gxc);
}
float hm_silu(float v) { return eq::silu(v); }
float hm_silu_grad(float v) { return eq::silu_grad(v); }
void hm_layer_layout(int F, int H, int A, int* out /*16*/) {
    eq::LayerLayout L = eq::make_layer_layout(F, H, A);
    int v[16] = {L.FH, L.D, L.head_stride, L.w1, L.b1, L.w2, L.b2, L.ln_g, L.ln_b, L.w4, L.b4, L.w5, L.b5, L.w6, L.b6, L.w7};
    for (int i = 0; i < 16; ++i) out[i] = v[i];
}
int hm_layer_size(int F, int H, int A) { return eq::make_layer_layout(F, H, A).size; }
int hm_lp_check() {   // compile-time twin == runtime layout
    using A = eq::LP<32, 4, 1>;
    eq::LayerLayout L = eq::make_layer_layout(32, 4, 1);
    return A::size == L.size && A::w7 == L.w7 && A::head_stride == L.head_stride && A::ln_g == L.ln_g &&
           A::w1_off(9, 70) == (9 / L.FH) * L.head_stride + (9 % L.FH) * L.D + 70;
}
}
