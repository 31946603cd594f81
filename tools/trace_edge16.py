"""Per-wave phase timing of the 16-edge edge kernels from the developer trace build (make -C csrc trace).
   EQUIGSPR_EDGE_SAVE_MB=0 python tools/trace_edge16.py [batch]
Stamps are those of the LAST chunk / sub-tile of every wave; each stamp drains the wave's counters first."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("EQUIGSPR_LIB", os.path.join(ROOT, "se3_equi_graph_registration_amd", "libequigspr_trace.so"))
import numpy as np
import torch
import se3_equi_graph_registration_amd as pk
from se3_equi_graph_registration_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N, K, F = 2048, (int(sys.argv[2]) if len(sys.argv) > 2 else 20), 32
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
SLOTS, WAVES = 16, 16384
buf = torch.zeros(4 * WAVES * SLOTS, dtype=torch.int64, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ms = ctypes.c_float(0.0)


def report(name, stamps, labels, wpb):
    st = stamps.cpu().numpy().reshape(-1, SLOTS)[:WAVES]
    xcd = (np.arange(st.shape[0]) // wpb) % 8
    live = st[:, 0] > 0
    st, xcd = st[live], xcd[live]
    last = labels[-1][0]
    rel0, rel1 = np.zeros(len(st)), np.zeros(len(st))
    for q in range(8):
        sel = xcd == q
        if sel.any():
            t0 = st[sel, 0].min()
            rel0[sel], rel1[sel] = st[sel, 0] - t0, st[sel, last] - t0
    print(f"== {name}: {live.sum()} waves; per-XCD span {int(rel1.max())} cycles")
    prev = 0
    for k, lab in labels:
        ok = (st[:, k] > 0) & (st[:, prev] > 0)
        d = (st[ok, k] - st[ok, prev])
        print(f"   {lab:52s} median {int(np.median(d)):7d}  p90 {int(np.percentile(d, 90)):7d}")
        prev = k
    life = st[:, last] - st[:, 0]
    print(f"   wave life: median {int(np.median(life))}  max {int(life.max())}")


work = torch.empty(int(L.equigspr_egnn_work_size(ctypes.byref(cfg), graph.n_nodes, E)), device=dev)
args_f = lambda it: (ctypes.byref(cfg), ctypes.c_void_p(egnn.flat_params.data_ptr()), ctypes.byref(graph.struct), ctypes.c_void_p(h.data_ptr()),
                     ctypes.c_void_p(x.data_ptr()), None, ctypes.c_void_p(work.data_ptr()), it, ctypes.byref(ms), stream)
_lib.check(L.equigspr_time_edge_fwd(*args_f(5)), "time_edge_fwd")
raw.equigspr_trace_set_fwd(ctypes.c_void_p(buf.data_ptr()))
_lib.check(L.equigspr_time_edge_fwd(*args_f(1)), "time_edge_fwd")
torch.cuda.synchronize()
print(f"edge_fwd16 (trace build): {ms.value * 1e3:.1f} us/launch, E={E}")
report("edge_fwd16 (last chunk / sub-tile of each wave)", buf,
       [(1, "stage weights"), (2, "row/col/rinv load"), (3, "x gathers + geometry + LDS rows"), (4, "idx LDS read, P rows, geo operands"),
        (5, "geo product (8 MFMA), sigmoid"), (6, "W2 (16 MFMA)"), (7, "LayerNorm"), (8, "W6 (16 MFMA)"), (9, "sigmoid, w7 dot, gsum"),
        (10, "stores"), (11, "back: trans store")], 4)
raw.equigspr_trace_set_fwd(None)

buf.zero_()
_, _, saved = ops.egnn_forward(cfg, egnn.flat_params, graph, h, x, None, True)
bwork = torch.empty(int(L.equigspr_egnn_bwd_work_size(ctypes.byref(cfg), graph.n_nodes, E)), device=dev)
args_b = lambda it: (ctypes.byref(cfg), ctypes.c_void_p(egnn.flat_params.data_ptr()), ctypes.byref(graph.struct), None,
                     ctypes.c_void_p(saved.data_ptr()), ctypes.c_void_p(bwork.data_ptr()), it, ctypes.byref(ms), stream)
_lib.check(L.equigspr_time_edge_bwd(*args_b(5)), "time_edge_bwd")
raw.equigspr_trace_set_bwd(ctypes.c_void_p(buf.data_ptr()))
_lib.check(L.equigspr_time_edge_bwd(*args_b(1)), "time_edge_bwd")
torch.cuda.synchronize()
print(f"edge_bwd16 (trace build): {ms.value * 1e3:.1f} us/launch")
report("edge_bwd16 (last chunk / sub-tile of each wave)", buf,
       [(1, "stage weights"), (2, "row/col load"), (3, "x/Gx gathers + geometry + LDS rows"), (4, "idx LDS read, P rows, geo operands"),
        (5, "geo product, sigmoid"), (6, "W2"), (7, "LayerNorm"), (8, "W6"), (9, "sigmoid, w7 dot"), (10, "dq, stage, dW6 pass"),
        (11, "dagg rows, W6^T dq"), (12, "LN bwd, stage, dW2 pass"), (13, "W2^T du, dt1 store, dWgeo pass, Wgeo^T dt1"),
        (14, "back: geometry bwd + stores"), (15, "other chunks' wait + block reduce + slab")], 8)
if os.environ.get("EQ_TRACE_MERGE"):       # trace library built with -DEQ_TRACE_MERGE: slots 4, 5, 6 are re-used by the merge
    st = buf.cpu().numpy().reshape(-1, SLOTS)[:WAVES]
    st = st[st[:, 0] > 0]
    for a, b, lab in [(14, 4, "lane sums"), (4, 5, "slab copy -> LDS"), (5, 6, "workgroup barrier"), (6, 15, "block sum + slab store (drained)")]:
        d = st[:, b] - st[:, a]
        print(f"   merge: {lab:40s} median {int(np.median(d)):7d}  p90 {int(np.percentile(d, 90)):7d}")
