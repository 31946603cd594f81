"""Per-wave phase timing of the head's tail launch (Kabsch moments, ticket, 3x3 SVD solve) from the developer trace build."""


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

buf = torch.zeros(8 * WAVES * SLOTS, dtype=torch.int64, device=dev)
model = pk.CrossAttentionPoseRegression(egnn, num_nodes=N, hidden_nf=F, device=dev).to(dev)
for it in range(2):
    buf.zero_()
    raw.equigspr_trace_set_head(ctypes.c_void_p(buf.data_ptr()))
    with torch.no_grad():
        model(batch["feat_src"], batch["src"], graph, None, batch["feat_tgt"], batch["tgt"], graph, None, batch["corr"], batch["labels"], batch["gt_pose"])
    torch.cuda.synchronize()
    raw.equigspr_trace_set_head(None)
st = buf.cpu().numpy().reshape(-1, SLOTS)[5 * WAVES:6 * WAVES]
live = st[:, 0] > 0
st = st[live]
print("head_tail waves:", live.sum())
for k, lab in ((1, "moments block (from start)"), (2, "fence + ticket"), (3, "solve (last block only)")):
    ok = (st[:, k] > 0) & (st[:, k - 1] > 0)
    d = st[ok, k] - st[ok, k - 1]
    if ok.sum() == 0:
        print(f"  {lab:32s} (no stamps: this phase runs inside another launch now)")
        continue
    print(f"  {lab:32s} n={ok.sum():3d}  cycles: median {int(np.median(d)):7d}  max {int(d.max()):7d}")
st = buf.cpu().numpy().reshape(-1, SLOTS)[3 * WAVES:4 * WAVES]
live = st[:, 0] > 0
st = st[live]
print("head_pair waves:", live.sum())
for k, lab in ((1, "batch statistics from partials (fp64)"), (2, "masked-softmax scalars"), (3, "keys + 4 radix-select passes"), (4, "candidate compaction"), (5, "rank by counting + writes")):
    ok = (st[:, k] > 0) & (st[:, k - 1] > 0)
    d = st[ok, k] - st[ok, k - 1]
    if ok.sum() == 0:
        print(f"  {lab:40s} (no stamps)")
        continue
    print(f"  {lab:40s} n={ok.sum():3d}  cycles: median {int(np.median(d)):7d}  max {int(d.max()):7d}")
