"""With a -DEQ_DEBUG_BACK -DEQ_BF_GUARD=0 build (EQUIGSPR_LIB): which input of the back phase differs on the tiles that come out wrong?
   Compares the per-edge dump of a split-bf16 run with the fp32-product run of the same library."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import se3_equi_graph_registration_amd as pk
from se3_equi_graph_registration_amd import _lib
L = _lib.lib()
raw = ctypes.CDLL(os.environ["EQUIGSPR_LIB"])
raw.equigspr_debug_back.argtypes = [ctypes.c_void_p, ctypes.c_int]
NAMES = [f"gg{k}" for k in range(12)] + ["xr0", "xr1", "xr2", "xc0", "xc1", "xc2", "gx0", "gx1", "gx2", "gxr0", "gxr1", "gxr2", "gxc0", "gxc1", "gxc2", "d0", "d1", "d2", "s", "ds"]
def run(n, feat, fp32):
    _lib.check(L.equigspr_set_edge_products(1 if fp32 else 0), "p"); _lib.check(L.equigspr_set_edge_grid_cap(0), "c")
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
    dbg = np.zeros((n * 16, 32), np.float32)
    rc = raw.equigspr_debug_back(dbg.ctypes.data, n * 16)
    assert rc == 0, rc
    return dbg, xc.grad.double().cpu().numpy()
n = 40000
ref, rx = run(n, 32, True)
np.set_printoptions(precision=5, linewidth=220, suppress=False)
for trial in range(3):
    d, gx = run(n, 32, False)
    bad_nodes = np.nonzero(np.abs(gx - rx).max(axis=1) / np.abs(rx).max() > 2e-5)[0]
    rel = np.abs(d - ref) / (np.abs(ref).max(axis=0, keepdims=True) + 1e-30)
    bad_edges = np.nonzero(rel.max(axis=1) > 1e-4)[0]
    print(f"trial {trial}: wrong grad_x nodes {bad_nodes.tolist()[:40]}")
    print(f"   edges whose dump differs by > 1e-4 of the column's range: {len(bad_edges)}; tiles {sorted(set((bad_edges // 16).tolist()))[:40]}")
    cols = (rel[bad_edges] > 1e-4).sum(axis=0)
    print("   columns that differ:", {NAMES[k]: int(cols[k]) for k in range(32) if cols[k]})
    for e in bad_edges[::16][:12]:
        print(f"   edge {e} (tile {e // 16}, lane {e % 64}):")
        k_ = np.nonzero(rel[e] > 1e-4)[0]
        print("      differing:", {NAMES[k]: (float(d[e][k]), float(ref[e][k])) for k in k_})
