"""developer: which tensors of the FIRST training step (fresh process, fresh allocations) are non-finite -- python tools/dev/nf_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import se3_equi_graph_registration_amd as pk
from se3_equi_graph_registration_amd import train
import test_gpu_graphed_step as T
m = T._model(pk, seed=3)
o = train.make_optimizer(m, lr=1e-3)
b = T._batches(pk, 1)[0]
batch = tuple(t.cuda() if torch.is_tensor(t) else t for t in b)
for it in range(3):
    m.zero_grad()
    o.dp.zero_grad()
    out, loss, rot, trans = train._forward(m, batch, 8)
    loss.backward()
    ge = m.egnn.flat_grads
    print("iter", it, "loss", float(loss), "finite: loss", bool(torch.isfinite(loss)), "egnn grads", bool(torch.isfinite(ge).all()),
          "n_bad", int((~torch.isfinite(ge)).sum()), "head grads", all(bool(torch.isfinite(p.grad).all()) for p in m.parameters() if p.grad is not None))
    if not torch.isfinite(ge).all():
        bad = (~torch.isfinite(ge)).nonzero().flatten()
        print("   first bad flat indices", bad[:12].tolist(), "last", bad[-3:].tolist(), "of", ge.numel())
