import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import se3_equi_graph_registration_amd as pkg
from se3_equi_graph_registration_amd import train
dev = torch.device("cuda:0")
egnn = pkg.EGNN(32, 32, 32, in_edge_nf=1, device=dev, n_layers=4)
model = pkg.CrossAttentionPoseRegression(egnn, num_nodes=2048, hidden_nf=32, device=dev).to(dev)
opt = train.make_optimizer(model, lr=1e-4)
adam = opt.adam
adam.push_hyper(1.0)
loss = torch.ones((), device=dev)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)


def t(label, fn, n=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"   {label}: {e0.elapsed_time(e1) / n * 1e3:.2f} us/launch")


t("adam.step(zero_grad=True)", lambda: adam.step(zero_grad=True))
t("adam.step_guarded(loss)", lambda: adam.step_guarded(loss=loss))
t("adam.step_guarded(no loss)", lambda: adam.step_guarded())
