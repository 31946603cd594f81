"""GraphedStep.run alone (for a rocprofv3 kernel trace of the product loop)."""
import itertools, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import se3_equi_graph_registration_amd as pkg
from se3_equi_graph_registration_amd import train
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
egnn = pkg.EGNN(32, 32, 32, in_edge_nf=1, device=dev, n_layers=4)
model = pkg.CrossAttentionPoseRegression(egnn, num_nodes=2048, hidden_nf=32, device=dev).to(dev)
opt = train.make_optimizer(model, lr=1e-4)
ring = []
for j in range(max(2, 64 // B)):
    b = pkg.synthetic_batch(1234 + j * B, B)
    ring.append((b["corr"], b["labels"], b["src"], b["tgt"], b["feat_src"], b["feat_tgt"], b["gt_pose"]))
ws = torch.cuda.Stream(); torch.cuda.set_stream(ws)
if os.environ.get('SWITCH'):
    sys.setswitchinterval(float(os.environ['SWITCH']))
st = train.GraphedStep(model, opt, k=20)
st.run(itertools.islice(itertools.cycle(ring), 20), log_interval=0, skip_nonfinite=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
st.run(itertools.islice(itertools.cycle(ring), n_steps), log_interval=0, skip_nonfinite=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"B={B}: {1e3 * dt / n_steps:.4f} ms/step  mode={st.graph_mode}  {getattr(st, 'timing', None)}")
