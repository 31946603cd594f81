import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import se3_equi_graph_registration_amd as pkg
dev = torch.device("cuda:0")
N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 32
for seed in range(1234, 1242):
    b = pkg.synthetic_batch(seed, 1, n=N, device=dev)
    for _ in range(3):
        g = pkg.build_pair_graph(b["src"], b["tgt"], K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g = pkg.build_pair_graph(b["src"], b["tgt"], K)
    e1.record(); torch.cuda.synchronize()
    print(f"seed {seed}: build_pair_graph {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
