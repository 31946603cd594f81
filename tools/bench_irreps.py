"""HIP-event timing of the typed irrep-basis layer (irreps.IrrepConv, csrc/irreps.hip) at the headline graph size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import se3_equi_graph_registration_amd as pk
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b = pk.synthetic_batch(1234, B, n=2048, device="cuda")
graph = pk.build_pair_graph(b["src"], b["tgt"], 20, need_col_index=False)
x = torch.cat([b["src"].reshape(-1, 3), b["tgt"].reshape(-1, 3)]).contiguous()
n = x.shape[0]
f0, f1 = torch.randn(n, 32, device="cuda"), torch.randn(n, 8, 3, device="cuda")
layer = pk.IrrepConv(32, 8, 32, 8)


def t(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


with torch.no_grad():
    dist, k00, k01, k10, k11 = layer._geometry(x, graph)
    radial = layer._radial(dist)
    args = (f0, f1, radial, layer.self0.weight, layer.self1.weight, dist, k00, k01, k10, k11, graph)
    print(f"B={B}: N={n} E={graph.n_edges} n_w={layer.n_w}")
    print(f"   basis (sh_forward + 4 x sh_basis): {t(lambda: layer._geometry(x, graph)):.1f} us")
    print(f"   radial profile (Linear-SiLU-Linear over E, library GEMM): {t(lambda: layer._radial(dist)):.1f} us")
    print(f"   irrep_conv_kernel (contraction + aggregation): {t(lambda: layer._hip(*args)):.1f} us   = {graph.n_edges * layer.n_w * 4 / 1e6:.0f} MB of radial weights read")
    fast = (f0, f1, layer.self0.weight, layer.self1.weight, dist, k00, k01, k10, k11, graph)
    print(f"   aggregate-then-transform: moments launch + 4 node-level GEMMs: {t(lambda: layer._hip_fast(*fast)):.1f} us")
    print(f"   whole layer forward: {t(lambda: layer(f0, f1, x, graph)):.1f} us;  the same in device ops (einsum / index_add): {t(lambda: layer._aten(*args)):.1f} us")
    # pieces of the fast path
    import ctypes
    from se3_equi_graph_registration_amd import _lib
    from se3_equi_graph_registration_amd.ops import _p, _stream
    H, HA = layer.hidden, layer.hidden + 1
    lin1 = layer.radial[0]
    hid = torch.zeros(graph.n_edges, 32, device="cuda"); hid[:, :H] = torch.nn.functional.silu(torch.addmm(lin1.bias, dist[:, None], lin1.weight.t()))
    Lc = 4 * layer.c0_in + 10 * layer.c1_in
    M = torch.empty(n * Lc * HA, device="cuda")
    def mom():
        _lib.check(_lib.lib().equigspr_irrep_moments_forward(_p(f0), _p(f1), _p(dist), _p(hid), _p(k00), _p(k01), _p(k10), _p(k11), _p(graph.col),
                                                             _p(graph.rptr), _p(graph.rperm), n, graph.n_edges, layer.c0_in, layer.c1_in, H, _p(M), _stream()), "m")
    print(f"   moments kernel alone: {t(mom):.1f} us  (M = {M.numel() * 4 / 1e6:.0f} MB);  hid (Linear + SiLU over E): {t(lambda: torch.nn.functional.silu(torch.addmm(lin1.bias, dist[:, None], lin1.weight.t()))):.1f} us")
