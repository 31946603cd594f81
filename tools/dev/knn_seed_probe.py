import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import se3_equi_graph_registration_amd as pkg
from se3_equi_graph_registration_amd import ops
dev = torch.device("cuda:0")


def t(x, k, reps=10):
    for _ in range(2):
        ops.knn_graph_batched(x, k, True, want_i64=False, want_i32=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.knn_graph_batched(x, k, True, want_i64=False, want_i32=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for seed in (1234, 1239):
    for n, k in ((4096, 20), (8192, 32)):
        b = pkg.synthetic_batch(seed, 1, n=n, device=dev)
        src, tgt = b["src"].contiguous(), b["tgt"].contiguous()
        print(seed, n, k, "src %.1f us  tgt %.1f us  both %.1f us" % (t(src, k), t(tgt, k), t(torch.cat([src, tgt]), k)),
              "src finite", bool(torch.isfinite(src).all()), "tgt finite", bool(torch.isfinite(tgt).all()),
              "src range", float(src.min()), float(src.max()), "tgt range", float(tgt.min()), float(tgt.max()))
        # which queries are slow?  time halves of the tgt cloud's queries is not possible; instead count exact duplicates
        u = torch.unique(tgt[0], dim=0).shape[0]
        print("   distinct tgt points", u, "of", n)
