"""HIP-event timing of the typed irrep-basis layer (irreps.IrrepConv, csrc/irreps.hip) at the headline graph size:
   python tools/bench_irreps.py [pairs] [c0 c1]
forward (moments + node launch) and forward + backward (adjoint convolution, expand, geometry, dx, weight gradients, reduce) per layer.
Under `rocprofv3 --kernel-trace --stats` the per-kernel split."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import se3_equi_graph_registration_amd as pk

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
c0 = int(sys.argv[2]) if len(sys.argv) > 2 else 32
c1 = int(sys.argv[3]) if len(sys.argv) > 3 else 8
b = pk.synthetic_batch(1234, B, n=2048, device="cuda")
graph = pk.build_pair_graph(b["src"], b["tgt"], 20)
x = torch.cat([b["src"].reshape(-1, 3), b["tgt"].reshape(-1, 3)]).contiguous()
n = x.shape[0]
f0, f1 = torch.randn(n, c0, device="cuda"), torch.randn(n, c1, 3, device="cuda")
layer = pk.IrrepConv(c0, c1, c0, c1)


def t(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def fwd():
    with torch.no_grad():
        return layer(f0, f1, x, graph)


a0, a1, ax = f0.clone().requires_grad_(True), f1.clone().requires_grad_(True), x.clone().requires_grad_(True)


def fwd_bwd():
    o0, o1 = layer(a0, a1, ax, graph)
    torch.autograd.backward([o0, o1], [o0.detach(), o1.detach()])


lc = 4 * ((c0 + 15) // 16 * 16) + 10 * (8 if c1 <= 8 else 16 if c1 <= 16 else 32)
print(f"pairs={B}: N={n} E={graph.n_edges} channels {c0}+{c1} hidden {layer.hidden} n_w={layer.n_w}  moments {n * lc * 33 * 4 / 1e6:.0f} MB")
def fwd_train():
    o0, o1 = layer(a0, a1, ax, graph)


print(f"   forward, inference (edge table + moments + node):  {t(fwd):9.1f} us")
print(f"   forward, training (moments kept: 2 launches): {t(fwd_train):9.1f} us")
print(f"   forward + backward (eager, autograd): {t(fwd_bwd):9.1f} us")
