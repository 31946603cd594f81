"""developer probe (round 6): the fixed cost of ONE train.GraphedStep.run() call (a bench block of K steps pays it once: the driver's
--steps 20 reads 8 % below --steps 200).  Times blocks of 1..200 steps -> per-block overhead = intercept of time(K); cProfile of a
20-step block for the host side."""
import cProfile, itertools, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import se3_equi_graph_registration_amd as pkg
from se3_equi_graph_registration_amd import train
dev = torch.device("cuda:0")
torch.manual_seed(0)
egnn = pkg.EGNN(32, 32, 32, in_edge_nf=1, device=dev, n_layers=4)
model = pkg.CrossAttentionPoseRegression(egnn, num_nodes=2048, hidden_nf=32, device=dev).to(dev)
opt = train.make_optimizer(model, lr=1e-4)
ring = []
for j in range(64):
    b = pkg.synthetic_batch(1234 + j, 1)
    ring.append((b["corr"], b["labels"], b["src"], b["tgt"], b["feat_src"], b["feat_tgt"], b["gt_pose"]))
ws = torch.cuda.Stream(); torch.cuda.set_stream(ws)
st = train.GraphedStep(model, opt, k=20)
st.run(itertools.islice(itertools.cycle(ring), 20), log_interval=0, skip_nonfinite=True)
def block(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.run(itertools.islice(itertools.cycle(ring), n), log_interval=0, skip_nonfinite=True)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
res = {}
for n in (1, 2, 5, 20, 50, 200):
    ts = sorted(block(n) for _ in range(9))
    res[n] = ts[4]
    print(f"K = {n:4d}: {1e3 * ts[4]:8.3f} ms per block  = {1e3 * ts[4] / n:.4f} ms/step   (min {1e3 * ts[0]:.3f})")
slope = (res[200] - res[50]) / 150
print(f"slope {1e3 * slope:.4f} ms/step, intercept at K = 20: {1e3 * (res[20] - 20 * slope):.3f} ms")
os.environ["EQUIGSPR_STEP_TIMING"] = "1"
for n in (1, 20):
    for _ in range(3):
        t = block(n)
        tm = st.timing
        print(f"K = {n}: block {1e3 * t:.3f} ms; phases (ms since entry): setup {1e3 * tm['phase_setup']:.3f}  last launch {1e3 * tm['phase_loop']:.3f}  drained {1e3 * tm['phase_drain']:.3f}  "
              f"stats read {1e3 * tm['phase_stats']:.3f};  get {1e3 * tm['get']:.3f} wait_calls {1e3 * tm['wait_calls']:.3f} ev_sync {1e3 * tm['ev_sync']:.3f} launch {1e3 * tm['launch']:.3f} h2d {1e3 * tm['h2d']:.3f}")
