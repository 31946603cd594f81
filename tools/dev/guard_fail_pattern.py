"""Which tiles does an unfenced split-bf16 build get wrong, and what do the wrong values look like?
   EQUIGSPR_LIB=<variant .so> python tools/dev/guard_fail_pattern.py   (ring graph of tools/guard_margin.py: node n owns the 16-edge tile n)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import se3_equi_graph_registration_amd as pk
from se3_equi_graph_registration_amd import _lib
L = _lib.lib()
def run(n, feat, fp32, cap):
    _lib.check(L.equigspr_set_edge_products(1 if fp32 else 0), "p"); _lib.check(L.equigspr_set_edge_grid_cap(cap), "c")
    torch.manual_seed(2)
    m = pk.EGNN(feat, feat, feat, in_edge_nf=1, device="cuda", n_layers=1, num_heads=1)
    g = torch.Generator().manual_seed(5)
    h = torch.randn(n, feat, generator=g).cuda(); x = (torch.rand(n, 3, generator=g) * 2).cuda()
    row = torch.arange(n).repeat_interleave(16); col = (row + 1) % n
    ea = torch.rand(row.numel(), 1, generator=g).cuda()
    gh = torch.randn(n, feat, generator=g).cuda(); gx = torch.randn(n, 3, generator=g).cuda()
    hc, xc = h.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ho, xo = m(hc, xc, [row.cuda(), col.cuda()], ea)
    m.zero_grad(); ((ho * gh).sum() + (xo * gx).sum()).backward(); torch.cuda.synchronize()
    return hc.grad.double().cpu(), xc.grad.double().cpu()
n, feat = 40000, 32
rh, rx = run(n, feat, True, 0)
torch.set_printoptions(precision=4, linewidth=250, sci_mode=True)
for trial in range(3):
    gh, gx = run(n, feat, False, 0)
    dh = (gh - rh); dx = gx - rx
    badh = ((dh.abs().max(dim=1).values / rh.abs().max()) > 2e-5).nonzero().flatten().tolist()
    badx = ((dx.abs().max(dim=1).values / rx.abs().max()) > 2e-5).nonzero().flatten().tolist()
    print(f"trial {trial}: wrong grad_h nodes {badh}\n          wrong grad_x nodes {badx}")
    for i in badh[:4]:
        print("  node", i, "tile/64 =", i // 64, "tile%64 =", i % 64, " grad_h diff / ref row max:", (dh[i] / rh[i].abs().max()))
    for i in badx[:6]:
        print("  node", i, "tile%64 =", i % 64, " grad_x diff:", dx[i], " ref:", rx[i])
