"""``gcnLayer`` with the reference interface (src/gcnLayer.py), HIP forward.

``build_graph`` keeps the normalised kNN adjacency sparse on the device (``KnnAdjacency``: CSR by node) rather
than as the reference's scipy ``[N, N]`` matrix; ``GraphConvolution`` consumes it in one fused gather + 1x1-conv
kernel (a dense ``[B, N, N]`` adjacency tensor, as the reference's ``forward`` signature allows, is also accepted:
the product with it is one library GEMM, the conv stays in HIP); ``GlobalPooling`` is a block reduction.
Inference only -- the reference imports these layers but never calls them."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from . import _lib
from .ops import _need, _p, _stream, knn_graph_batched


class KnnAdjacency:
    """Symmetric normalised adjacency of ``n_clouds`` clouds of ``n`` points: ``ptr [N+1]``, ``idx`` (local column
    ids) and ``val`` (fp32)."""

    def __init__(self, ptr, idx, val, n_clouds, n, squeeze):
        self.ptr, self.idx, self.val, self.n_clouds, self.n, self._squeeze = ptr, idx, val, n_clouds, n, squeeze

    @property
    def nnz(self):
        return int(self.ptr[-1].item())

    def to_dense(self):
        """``[n, n]`` for a single ``[n, 3]`` input (what ``build_graph(...).toarray()`` gives in the reference), else ``[B, n, n]``."""
        dense = torch.zeros(self.n_clouds, self.n, self.n, dtype=torch.float32, device=self.val.device)
        _lib.check(_lib.lib().equigspr_gcn_adjacency_to_dense(_p(self.ptr), _p(self.idx), _p(self.val), self.n_clouds, self.n,
                                                              _p(dense), _stream()), "gcn_adjacency_to_dense")
        return dense[0] if self._squeeze else dense


def build_graph(coordinates, k=25, device="cuda"):
    """``coordinates``: ``[n, 3]`` (numpy or tensor, as in the reference) or a batch ``[B, n, 3]``; ``k`` counts the
    point itself, as ``cKDTree.query(coordinates, k)`` does (gcnLayer.py:65-66)."""
    x = torch.as_tensor(np.asarray(coordinates) if not torch.is_tensor(coordinates) else coordinates)
    squeeze = x.dim() == 2
    x = (x.unsqueeze(0) if squeeze else x).to(device=device if not x.is_cuda else x.device, dtype=torch.float32).contiguous()
    _need(x, "coordinates", torch.float32, 3)
    nc, n, _ = x.shape
    _, nbr, col = knn_graph_batched(x, int(k), loop=True, want_i64=False, want_i32=True)
    nt, dev = nc * n, x.device
    L = _lib.lib()
    rptr = torch.empty(nt + 1, dtype=torch.int32, device=dev)
    rperm = torch.empty(nt * k, dtype=torch.int32, device=dev)
    scratch = torch.empty(nt + 1, dtype=torch.int32, device=dev)
    _lib.check(L.equigspr_build_csr(_p(nbr), nt * k, nt, _p(rptr), _p(rperm), None, _p(scratch), 0, _stream()), "build_csr")
    sigma2 = torch.empty(nc, dtype=torch.float64, device=dev)
    dinv = torch.empty(nt, dtype=torch.float32, device=dev)
    ptr = torch.empty(nt + 1, dtype=torch.int32, device=dev)
    cap = 2 * nt * (k - 1)
    idx = torch.empty(cap, dtype=torch.int32, device=dev)
    val = torch.empty(cap, dtype=torch.float32, device=dev)
    _lib.check(L.equigspr_gcn_build_graph(_p(x), nc, n, int(k), _p(nbr), _p(col), _p(rptr), _p(rperm), _p(sigma2), _p(scratch),
                                          _p(dinv), _p(ptr), _p(idx), _p(val), _stream()), "gcn_build_graph")
    return KnnAdjacency(ptr, idx, val, nc, n, squeeze)


def _no_grad_only(name, *tensors):
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        raise NotImplementedError(f"{name}: the HIP op is forward-only; call it under torch.no_grad()")


class GraphConvolution(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.conv = nn.Conv1d(in_features, out_features, kernel_size=1, bias=bias)

    def forward(self, adj, x):
        """``x [B, C_in, n]``; ``adj``: ``KnnAdjacency`` (1 or B clouds) or dense ``[B, n, n]`` -> ``[B, C_out, n]``."""
        _no_grad_only("GraphConvolution", x, *self.parameters())
        x = _need(x.contiguous().float(), "x", torch.float32, 3)
        b, c_in, n = x.shape
        w = self.conv.weight.detach().reshape(self.conv.out_channels, c_in).contiguous()
        bias = self.conv.bias.detach().contiguous() if self.conv.bias is not None else None
        y = torch.empty(b, self.conv.out_channels, n, dtype=torch.float32, device=x.device)
        L = _lib.lib()
        if isinstance(adj, KnnAdjacency):
            if adj.n != n or adj.n_clouds not in (1, b):
                raise RuntimeError(f"GraphConvolution: adjacency ({adj.n_clouds} x {adj.n}) does not match x {tuple(x.shape)}")
            _lib.check(L.equigspr_gcn_conv_forward(_p(x), b, c_in, self.conv.out_channels, n, _p(adj.ptr), _p(adj.idx), _p(adj.val),
                                                   adj.n_clouds, _p(w), _p(bias), _p(y), _stream()), "gcn_conv_forward")
        else:
            agg = torch.bmm(x, _need(adj, "adj", torch.float32, 3)).contiguous()       # plain library GEMM
            _lib.check(L.equigspr_gcn_conv_forward(_p(agg), b, c_in, self.conv.out_channels, n, None, None, None, 0, _p(w), _p(bias),
                                                   _p(y), _stream()), "gcn_conv_forward")
        return y


class GlobalPooling(nn.Module):
    def forward(self, x):
        """``x [B, C, n]`` -> ``[B, 2C]`` = max over points || mean over points (gcnLayer.py:90-94)."""
        _no_grad_only("GlobalPooling", x)
        x = _need(x.contiguous().float(), "x", torch.float32, 3)
        b, c, n = x.shape
        out = torch.empty(b, 2 * c, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().equigspr_gcn_global_pool(_p(x), b, c, n, _p(out), _stream()), "gcn_global_pool")
        return out
