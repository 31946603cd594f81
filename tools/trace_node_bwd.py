"""Per-wave phase timing of the 16-node MFMA node backward (csrc/egnn_node_bwd.hip) from the developer trace build (make -C csrc trace).
   python tools/trace_node_bwd.py [batch]
Prints, per phase, the median/max wave time in shader cycles, and the distribution of wave start/end times."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("EQUIGSPR_LIB", os.path.join(ROOT, "se3_equi_graph_registration_amd", "libequigspr_trace.so"))
import numpy as np
import torch
import se3_equi_graph_registration_amd as pk
from se3_equi_graph_registration_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N, K, F = 2048, 20, 32
dev = torch.device("cuda:0")
L = _lib.lib()
raw = ctypes.CDLL(os.environ["EQUIGSPR_LIB"])
batch = pk.synthetic_batch(1234, B, n=N, feat=F, device=dev)
egnn = pk.EGNN(F, F, F, in_edge_nf=1, device=dev, n_layers=4)
egnn._ensure_flat()
graph = pk.build_pair_graph(batch["src"], batch["tgt"], K, loop=True)
h = torch.cat([batch["feat_src"].reshape(B * N, -1), batch["feat_tgt"].reshape(B * N, -1)]).contiguous()
x = torch.cat([batch["src"].reshape(B * N, 3), batch["tgt"].reshape(B * N, 3)]).contiguous()
cfg = egnn.cfg
E = graph.n_edges
SLOTS = 16
WAVES = 16384                       # EQ_TRACE_WAVES: waves per kernel id
buf = torch.zeros(4 * WAVES * SLOTS, dtype=torch.int64, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ms = ctypes.c_float(0.0)


def report(name, stamps, labels, kid=0):
    st = stamps.cpu().numpy().reshape(-1, SLOTS)[kid * WAVES:(kid + 1) * WAVES]
    xcd = (np.arange(st.shape[0]) // 4) % 8                 # workgroups round-robin over the 8 XCDs; s_memtime is per XCD
    live = st[:, 0] > 0
    st, xcd = st[live], xcd[live]
    last = labels[-1][0]
    rel0, rel1 = np.zeros(len(st)), np.zeros(len(st))
    for q in range(8):
        sel = xcd == q
        if sel.any():
            t0 = st[sel, 0].min()
            rel0[sel], rel1[sel] = st[sel, 0] - t0, st[sel, last] - t0
    print(f"== {name}: {live.sum()} waves; per-XCD span (first start -> last end) {int(rel1.max())} cycles")
    print(f"   wave start after the XCD's first wave: median {int(np.median(rel0))}  p90 {int(np.percentile(rel0, 90))}  max {int(rel0.max())}")
    prev = 0
    for k, lab in labels:
        ok = st[:, k] > 0
        d = (st[ok, k] - st[ok, prev])
        print(f"   {lab:44s} median {int(np.median(d)):7d}  p90 {int(np.percentile(d, 90)):7d}  max {int(d.max()):7d}")
        prev = k
    life = st[:, last] - st[:, 0]
    print(f"   wave life (0 -> {last}): median {int(np.median(life))}  max {int(life.max())};  wave end: median {int(np.median(rel1))} max {int(rel1.max())}")


work = torch.empty(int(L.equigspr_egnn_work_size(ctypes.byref(cfg), graph.n_nodes, E)), device=dev)

buf = torch.zeros(5 * WAVES * SLOTS, dtype=torch.int64, device=dev)
_, _, saved = ops.egnn_forward(cfg, egnn.flat_params, graph, h, x, None, True)
gh = torch.randn(graph.n_nodes, F, device=dev); gx = torch.randn(graph.n_nodes, 3, device=dev)
egnn.flat_grads.zero_()
for it in range(2):
    buf.zero_()
    raw.equigspr_trace_set_node_bwd(ctypes.c_void_p(buf.data_ptr()))
    ops.egnn_backward(cfg, egnn.flat_params, graph, h, None, saved, gh, gx, egnn.flat_grads, (False, False))
    torch.cuda.synchronize()
    raw.equigspr_trace_set_node_bwd(None)
print(f"B={B}: stamps of the LAST tile of every wave, last fused launch; cycles of the 100 MHz s_memtime counter x 1 (gfx950: constant 100 MHz)")
report("node_bwd16 (gather + node)", buf, [(1, "hoisted weight loads, earlier tiles"), (2, "rptr / cptr, inputs requested"),
                                           (3, "segment loop"), (4, "part reduce, G_x, W1^T products"), (5, "gather-side passes"),
                                           (6, "node MLP backward products, P, stores"), (7, "node-side passes"),
                                           (8, "gsum of partials, barrier"), (9, "slab copy to LDS, barrier"), (10, "copies added, slab store")], kid=4)
