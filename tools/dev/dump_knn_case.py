"""developer: EGNN forward + backward on the k-NN pair graph (self loops included) of synthetic clouds; dump to npz.
python tools/dev/dump_knn_case.py out.npz pairs n k"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import se3_equi_graph_registration_amd as pk
out, B, n, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
b = pk.synthetic_batch(100, B, n=n, device="cuda")
graph = pk.build_pair_graph(b["src"], b["tgt"], k, loop=True)
x = torch.cat([b["src"].reshape(-1, 3), b["tgt"].reshape(-1, 3)]).contiguous()
torch.manual_seed(3)
m = pk.EGNN(32, 32, 32, in_edge_nf=1, device="cuda", n_layers=2)
g = torch.Generator().manual_seed(1)
h = torch.randn(x.shape[0], 32, generator=g).cuda()
hc, xc = h.clone().requires_grad_(True), x.clone().requires_grad_(True)
from se3_equi_graph_registration_amd.egnn import EGNN_ONES
ho, xo = m(hc, xc, graph, EGNN_ONES if len(sys.argv) < 6 else torch.rand(graph.n_edges, 1, device="cuda"))
m.zero_grad()
(ho.square().sum() + xo.square().sum()).backward()
d = {"ho": ho, "xo": xo, "grad_h": hc.grad, "grad_x": xc.grad, "g.flat": m.flat_grads}
print("finite:", {k_: bool(torch.isfinite(v).all()) for k_, v in d.items()})
np.savez(out, **{k_: v.detach().cpu().numpy() for k_, v in d.items()})
