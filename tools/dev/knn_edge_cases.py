"""developer: edge cases of the round-6 k-NN additions against the oracle (n = 1, k = n, k > 64, n > 8 192, one-point clouds, ragged batches)."""
import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import se3_equi_graph_registration_amd as pk
from oracle import egnn_oracle as orc
g = torch.Generator().manual_seed(1)
def chk(n, k, loop, batch=None):
    x = torch.rand(n, 3, generator=g)
    got = pk.knn_graph(x.cuda(), k, loop=loop).cpu()
    ref = orc.knn_graph(x, k, loop=loop)
    assert torch.equal(got, ref), (n, k, loop)
for n, k, loop in [(1, 1, True), (2, 1, False), (65, 65, True), (66, 65, False), (200, 128, True), (8193, 1, True), (8193, 3, False), (10000, 65, True), (64, 64, True), (1025, 20, True), (2048, 64, True)]:
    chk(n, k, loop)
    print("ok", n, k, loop)
# batch of clouds with one point each
x = torch.rand(5, 3, generator=g)
e = pk.knn_graph(x.cuda(), 1, batch=torch.arange(5).cuda(), loop=True).cpu()
assert torch.equal(e, torch.stack([torch.arange(5), torch.arange(5)])), e
# ragged with k too large for the small cloud raises
try:
    pk.knn_graph(torch.rand(7, 3).cuda(), 4, batch=torch.tensor([0, 0, 0, 0, 0, 1, 1]).cuda(), loop=True)
    raise SystemExit("expected an error")
except RuntimeError as ex:
    print("raises:", str(ex)[:80])
# empty input with batch
e = pk.knn_graph(torch.zeros(0, 3).cuda(), 1, batch=torch.zeros(0, dtype=torch.long).cuda(), loop=True)
assert tuple(e.shape) == (2, 0)
print("edge cases ok")
