"""dev: the k-NN launch alone (HIP events around back-to-back launches) for the c5 and KITTI-shard shapes, over a few seeds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import se3_equi_graph_registration_amd as pkg
from se3_equi_graph_registration_amd import ops
dev = torch.device("cuda:0")
for pairs, n, k in ((1, 8192, 32), (8, 4096, 20), (1, 4096, 20), (4, 8192, 32), (1, 3000, 16), (1, 2048, 20), (16, 2048, 20)):
    ts = []
    for seed in range(1234, 1238):
        b = pkg.synthetic_batch(seed, pairs, n=n, device=dev)
        x = torch.cat([b["src"], b["tgt"]]).contiguous()
        for _ in range(3):
            ops.knn_graph_batched(x, k, True, want_i64=False, want_i32=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.knn_graph_batched(x, k, True, want_i64=False, want_i32=True)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"clouds {2 * pairs} x {n}, k={k}: " + " ".join(f"{t:.1f}" for t in ts) + " us (launch + output allocation)")
