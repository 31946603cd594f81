"""Where does the time of train.GraphedEval.run go?  (developer probe)"""
import itertools, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import se3_equi_graph_registration_amd as pkg
from se3_equi_graph_registration_amd import train
dev = torch.device("cuda:0")
torch.manual_seed(0)
N, K, L = 8192, 32, 8
egnn = pkg.EGNN(32, 32, 32, in_edge_nf=1, device=dev, n_layers=L)
model = pkg.CrossAttentionPoseRegression(egnn, num_nodes=N, hidden_nf=32, device=dev, variant="eval").to(dev).eval()
ring = []
for j in range(8):
    b = pkg.synthetic_batch(1234 + j, 1, n=N)
    ring.append((b["corr"], b["labels"], b["src"], b["tgt"], b["feat_src"], b["feat_tgt"], b["gt_pose"]))
ws = torch.cuda.Stream(); torch.cuda.set_stream(ws)
ge = train.GraphedEval(model, k=K)
ge.run(itertools.islice(itertools.cycle(ring), 10))
for n in (100, 100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ge.run(itertools.islice(itertools.cycle(ring), n))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"run(): {1e3 * dt / n:.4f} ms/pair")


def raw(label, fn, n=200):
    with torch.cuda.stream(ge.stream):
        for _ in range(10):
            fn(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"   {label}: {1e3 * dt / n:.4f} ms/pair")


raw("replay only", lambda i: ge.graphs[i % 3].replay())
raw("replay + clone", lambda i: (ge.graphs[i % 3].replay(), ge.outs[i % 3].clone()))
pin = train._Slot(1, N, 32, dev, pinned=True); pin.fill(ring[0])


def h2d(i):
    with torch.cuda.stream(ge.copy_stream):
        ge.slots[(i + 1) % 3].packet.copy_(pin.packet, non_blocking=True)
    ge.graphs[i % 3].replay()


raw("replay + H2D (no throttle)", h2d)
t0 = time.perf_counter()
for _ in range(50):
    pin.fill(ring[1])
print(f"   pack one pair into a pinned packet: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
