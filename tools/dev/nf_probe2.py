"""developer: the pair model's first step -- where does a non-finite gradient first appear?  python tools/dev/nf_probe2.py [save|load] file"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import se3_equi_graph_registration_amd as pk
from se3_equi_graph_registration_amd import train
import test_gpu_graphed_step as T
mode, path = sys.argv[1], sys.argv[2]
m = T._model(pk, seed=3)
b = T._batches(pk, 1)[0]
batch = tuple(t.cuda() if torch.is_tensor(t) else t for t in b)
keep = {}
orig = m.egnn.forward
def wrapped(h, x, edges, ea=None):
    ho, xo = orig(h, x, edges, ea)
    if mode == "load":
        d = torch.load(path)
        print("   egnn out vs saved: h", float((ho - d["ho"]).abs().max()), "x", float((xo - d["xo"]).abs().max()))
        ho = ho + (d["ho"] - ho).detach(); xo = xo + (d["xo"] - xo).detach()        # the saved (fp32-build) values, gradients still flow
    ho.retain_grad(); xo.retain_grad()
    keep["ho"], keep["xo"] = ho, xo
    return ho, xo
m.egnn.forward = wrapped
m.zero_grad()
out, loss, rot, trans = train._forward(m, batch, 8)
loss.backward()
if mode == "save":
    torch.save({"ho": keep["ho"].detach(), "xo": keep["xo"].detach()}, path)
print(mode, "loss", float(loss), "rot", [round(float(v), 6) for v in rot.reshape(-1)], "trans", [round(float(v), 6) for v in trans.reshape(-1)], "| d loss/d egnn out finite:", bool(torch.isfinite(keep["ho"].grad).all()),
      bool(torch.isfinite(keep["xo"].grad).all()), "| egnn param grads finite:", bool(torch.isfinite(m.egnn.flat_grads).all()))
