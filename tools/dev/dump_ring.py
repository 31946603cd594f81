"""developer: one EGNN layer stack on a ring graph whose node n owns the 16-edge tile n (row = n, col = n + 1): a wrong tile of the edge
backward shows up at nodes n / n + 1 only.  python tools/dev/dump_ring.py out.npz n_nodes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import se3_equi_graph_registration_amd as pk
out, n, feat = sys.argv[1], int(sys.argv[2]), 32
g = torch.Generator().manual_seed(5)
torch.manual_seed(2)
m = pk.EGNN(feat, feat, feat, in_edge_nf=1, device="cuda", n_layers=1, num_heads=1)
h = torch.randn(n, feat, generator=g); x = torch.rand(n, 3, generator=g) * 2
row = torch.arange(n).repeat_interleave(16); col = (row + 1) % n
ea = torch.rand(row.numel(), 1, generator=g)
gh = torch.randn(n, feat, generator=g); gx = torch.randn(n, 3, generator=g)
hc = h.cuda().requires_grad_(True); xc = x.cuda().requires_grad_(True)
ho, xo = m(hc, xc, [row.cuda(), col.cuda()], ea.cuda())
m.zero_grad()
((ho * gh.cuda()).sum() + (xo * gx.cuda()).sum()).backward()
np.savez(out, grad_h=hc.grad.cpu().numpy(), grad_x=xc.grad.cpu().numpy())
