"""Per-wave phase timing of the edge kernels from the developer trace build (make -C csrc trace).
   EQUIGSPR_LIB=.../libequigspr_trace.so python tools/trace_edge.py [batch]
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
_lib.check(L.equigspr_time_edge_fwd(ctypes.byref(cfg), ctypes.c_void_p(egnn.flat_params.data_ptr()), ctypes.byref(graph.struct),
                                    ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(x.data_ptr()), None, ctypes.c_void_p(work.data_ptr()),
                                    5, ctypes.byref(ms), stream), "time_edge_fwd")
raw.equigspr_trace_set_fwd(ctypes.c_void_p(buf.data_ptr()))
_lib.check(L.equigspr_time_edge_fwd(ctypes.byref(cfg), ctypes.c_void_p(egnn.flat_params.data_ptr()), ctypes.byref(graph.struct),
                                    ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(x.data_ptr()), None, ctypes.c_void_p(work.data_ptr()),
                                    1, ctypes.byref(ms), stream), "time_edge_fwd")
torch.cuda.synchronize()
print(f"edge_fwd (trace build, with drains): {ms.value * 1e3:.1f} us/launch, E={E}")
report("edge_fwd", buf, [(1, "stage weights -> LDS, barrier, W.load"), (2, "row/col index load"), (3, "x gathers + geometry"),
                          (4, "P row gathers (+acc init)"), (5, "t1 (7 MFMA), SiLU, W2 (16 MFMA), sum"), (6, "LN, W6 (16 MFMA), SiLU, w7"),
                          (7, "stores")])
raw.equigspr_trace_set_fwd(None)

buf.zero_()
_, _, saved = ops.egnn_forward(cfg, egnn.flat_params, graph, h, x, None, True)
bwork = torch.empty(int(L.equigspr_egnn_bwd_work_size(ctypes.byref(cfg), graph.n_nodes, E)), device=dev)
_lib.check(L.equigspr_time_edge_bwd(ctypes.byref(cfg), ctypes.c_void_p(egnn.flat_params.data_ptr()), ctypes.byref(graph.struct), None,
                                    ctypes.c_void_p(saved.data_ptr()), ctypes.c_void_p(bwork.data_ptr()), 5, ctypes.byref(ms), stream),
           "time_edge_bwd")
raw.equigspr_trace_set_bwd(ctypes.c_void_p(buf.data_ptr()))
_lib.check(L.equigspr_time_edge_bwd(ctypes.byref(cfg), ctypes.c_void_p(egnn.flat_params.data_ptr()), ctypes.byref(graph.struct), None,
                                    ctypes.c_void_p(saved.data_ptr()), ctypes.c_void_p(bwork.data_ptr()), 1, ctypes.byref(ms), stream),
           "time_edge_bwd")
torch.cuda.synchronize()
print(f"edge_bwd (trace build, with drains): {ms.value * 1e3:.1f} us/launch")
report("edge_bwd (last tile of each wave)", buf,
       [(1, "stage weights -> LDS, barrier, W.load"), (2, "row/col index load"), (3, "x/Gx gathers + geometry"), (4, "P row gathers"),
        (5, "t1, SiLU, W2, sum"), (6, "LN, W6, SiLU, w7"), (7, "coord head bwd + dW6 pass"), (8, "dagg gather + W6^T dq"),
        (9, "dw7/dgamma/dbeta passes"), (10, "LN bwd + dW2 pass"), (11, "W2^T du"), (12, "dWgeo pass"),
        (13, "Wgeo^T dt1, geometry bwd, stores"), (14, "remaining tiles' wait + block reduce + slab flush")])

# ---- node-level kernels: one full forward / backward with the trace pointer set (kernel ids 1, 2)
raw.equigspr_trace_set_bwd(None)
buf.zero_()
raw.equigspr_trace_set_fwd(ctypes.c_void_p(buf.data_ptr()))
_, _, saved = ops.egnn_forward(cfg, egnn.flat_params, graph, h, x, None, True)
torch.cuda.synchronize()
raw.equigspr_trace_set_fwd(None)
report("node_fwd (last layer launched, last group of each wave)", buf,
       [(1, "stage W4^T, W5^T, next-layer W_src/W_tgt^T"), (2, "segment sums over incoming edges (m, trans)"),
        (3, "node MLP (3 barriers)"), (4, "project P for the next layer / embedding_out")], kid=1)
buf.zero_()
raw.equigspr_trace_set_bwd(ctypes.c_void_p(buf.data_ptr()))
gh = torch.randn(graph.n_nodes, F, device=dev); gx = torch.randn(graph.n_nodes, 3, device=dev)
egnn.flat_grads.zero_()
ops.egnn_backward(cfg, egnn.flat_params, graph, h, None, saved, gh, gx, egnn.flat_grads, (False, False))
torch.cuda.synchronize()
raw.equigspr_trace_set_bwd(None)
report("node_bwd", buf, [(1, "stage W4^T, W_src^T, W_tgt^T, W4, W5 (29 KB)"), (2, "load Gh, h, agg"), (3, "W4 fwd + W5^T"), (4, "project P"),
                          (5, "W4^T products, stores"), (6, "outer-product accumulation"), (7, "slab store")], kid=1)
report("gather_bwd", buf, [(1, "stage W1[:, :2F]"), (2, "segment sums of dt1 over in/out edges"), (3, "W_src^T, W_tgt^T products"),
                            (4, "outer-product accumulation"), (5, "slab store")], kid=2)

# ---- head select kernel (kernel id 3): one full pair-model forward with the head trace pointer set
raw.equigspr_trace_set_bwd(None)
buf.zero_()
raw.equigspr_trace_set_head(ctypes.c_void_p(buf.data_ptr()))
model = pk.CrossAttentionPoseRegression(egnn, num_nodes=N, hidden_nf=F, device=dev).to(dev)
with torch.no_grad():
    model(batch["feat_src"], batch["src"], graph, None, batch["feat_tgt"], batch["tgt"], graph, None, batch["corr"], batch["labels"], batch["gt_pose"])
torch.cuda.synchronize()
raw.equigspr_trace_set_head(None)
report("head_pair_kernel", buf, [(1, "batch statistics from partials (fp64)"), (2, "masked-softmax scalars"), (3, "keys + 4 radix-select passes"),
                                 (4, "candidate compaction"), (5, "rank by counting + writes")], kid=3)
