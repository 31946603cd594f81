"""``gcnLayer`` with the reference interface (src/gcnLayer.py), HIP forward.

``build_graph`` keeps the normalised kNN adjacency sparse on the device (``KnnAdjacency``: CSR by node) rather
than as the reference's scipy ``[N, N]`` matrix; ``GraphConvolution`` consumes it in one fused gather + 1x1-conv
kernel (a dense ``[B, N, N]`` adjacency tensor, as the reference's ``forward`` signature allows, is also accepted:
the product with it is one library GEMM, the conv stays in HIP); ``GlobalPooling`` is a block reduction.
The reference imports these layers but never calls them; here they are trainable, forward AND backward in HIP (round 6:
``_GcnConvFn`` / ``_GcnPoolFn`` over ``equigspr_gcn_conv_backward_weights`` / ``equigspr_gcn_global_pool_backward``; the input
gradient of the convolution is the forward kernel itself on the output gradient with the transposed weight)."""
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
    _lib.check(L.equigspr_build_csr(_p(nbr), nt * k, nt, _p(rptr), _p(rperm), None, _p(scratch), 0, None, _stream()), "build_csr")
    sigma2 = torch.empty(nc, dtype=torch.float64, device=dev)
    dinv = torch.empty(nt, dtype=torch.float32, device=dev)
    ptr = torch.empty(nt + 1, dtype=torch.int32, device=dev)
    cap = 2 * nt * (k - 1)
    idx = torch.empty(cap, dtype=torch.int32, device=dev)
    val = torch.empty(cap, dtype=torch.float32, device=dev)
    _lib.check(L.equigspr_gcn_build_graph(_p(x), nc, n, int(k), _p(nbr), _p(col), _p(rptr), _p(rperm), _p(sigma2), _p(scratch),
                                          _p(dinv), _p(ptr), _p(idx), _p(val), _stream()), "gcn_build_graph")
    return KnnAdjacency(ptr, idx, val, nc, n, squeeze)


def _gcn_conv(x, w2, bias, adj, sparse, dense_adj):
    """y = W (x aggregated over adj) + bias through the HIP kernel; ``adj``: KnnAdjacency (sparse) or ignored (dense: ``dense_adj``)."""
    b, c_in, n = x.shape
    c_out = w2.shape[0]
    L = _lib.lib()
    y = torch.empty(b, c_out, n, dtype=torch.float32, device=x.device)
    if sparse:
        _lib.check(L.equigspr_gcn_conv_forward(_p(x), b, c_in, c_out, n, _p(adj.ptr), _p(adj.idx), _p(adj.val), adj.n_clouds,
                                               _p(w2), _p(bias), _p(y), _stream()), "gcn_conv_forward")
    else:
        agg = torch.bmm(x, dense_adj).contiguous()       # plain library GEMM
        _lib.check(L.equigspr_gcn_conv_forward(_p(agg), b, c_in, c_out, n, None, None, None, 0, _p(w2), _p(bias), _p(y), _stream()),
                   "gcn_conv_forward")
    return y


class _GcnConvFn(torch.autograd.Function):
    """GraphConvolution.forward (gcnLayer.py:70-80) with a HIP backward: grad_x = the forward kernel on grad_y with W^T (the normalised
    adjacency is symmetric; a dense adjacency is transposed explicitly), grad_W / grad_b from ``equigspr_gcn_conv_backward_weights``."""

    @staticmethod
    def forward(ctx, x, weight, bias, adj, sparse):
        c_out, c_in = weight.shape[0], x.shape[1]
        w2 = weight.reshape(c_out, c_in).contiguous()
        dense = None if sparse else (adj if adj.dim() == 3 else adj[None]).expand(x.shape[0], -1, -1).contiguous()
        y = _gcn_conv(x, w2, None if bias is None else bias.contiguous(), adj, sparse, dense)
        ctx.save_for_backward(x, w2, dense if dense is not None else x.new_empty(0))
        ctx.adj, ctx.sparse, ctx.has_bias, ctx.wshape = adj, sparse, bias is not None, weight.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w2, dense = ctx.saved_tensors
        gy = gy.contiguous().float()
        b, c_in, n = x.shape
        c_out = w2.shape[0]
        L = _lib.lib()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            dense_t = None if ctx.sparse else dense.transpose(1, 2).contiguous()
            if c_out <= 64:
                gx = _gcn_conv(gy, w2.t().contiguous(), None, ctx.adj, ctx.sparse, dense_t)
            else:       # the kernel aggregates <= 64 channels per thread: mix the channels first (one library GEMM), aggregate with W = I
                mixed = torch.matmul(w2.t(), gy).contiguous()
                gx = _gcn_conv(mixed, torch.eye(c_in, dtype=torch.float32, device=x.device), None, ctx.adj, ctx.sparse, dense_t)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw = torch.empty(c_out, c_in, dtype=torch.float32, device=x.device)
            gb = torch.empty(c_out, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            if ctx.sparse:
                agg = torch.empty(b, c_in, n, dtype=torch.float32, device=x.device)
                _lib.check(L.equigspr_gcn_conv_backward_weights(_p(x), _p(gy), b, c_in, c_out, n, _p(ctx.adj.ptr), _p(ctx.adj.idx),
                                                                _p(ctx.adj.val), ctx.adj.n_clouds, _p(agg), _p(gw), _p(gb), _stream()),
                           "gcn_conv_backward_weights")
            else:
                agg = torch.bmm(x, dense).contiguous()
                _lib.check(L.equigspr_gcn_conv_backward_weights(None, _p(gy), b, c_in, c_out, n, None, None, None, 0, _p(agg), _p(gw), _p(gb),
                                                                _stream()), "gcn_conv_backward_weights")
            gw = gw.reshape(ctx.wshape)
        return gx, gw, gb, None, None


class _GcnPoolFn(torch.autograd.Function):
    """GlobalPooling.forward (gcnLayer.py:90-94): max || mean over the points; HIP backward (ties of the max share its gradient evenly)."""

    @staticmethod
    def forward(ctx, x):
        b, c, n = x.shape
        out = torch.empty(b, 2 * c, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().equigspr_gcn_global_pool(_p(x), b, c, n, _p(out), _stream()), "gcn_global_pool")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        b, c, n = x.shape
        gx = torch.empty_like(x)
        _lib.check(_lib.lib().equigspr_gcn_global_pool_backward(_p(x), _p(g.contiguous().float()), b, c, n, _p(gx), _stream()),
                   "gcn_global_pool_backward")
        return gx


def _wants_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class GraphConvolution(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.conv = nn.Conv1d(in_features, out_features, kernel_size=1, bias=bias)

    def forward(self, adj, x):
        """``x [B, C_in, n]``; ``adj``: ``KnnAdjacency`` (1 or B clouds) or dense ``[B, n, n]`` -> ``[B, C_out, n]``.
        Forward and backward are HIP launches (``_GcnConvFn``); the adjacency is a constant."""
        x = _need(x.contiguous().float(), "x", torch.float32, 3)
        b, c_in, n = x.shape
        sparse = isinstance(adj, KnnAdjacency)
        if sparse:
            if adj.n != n or adj.n_clouds not in (1, b):
                raise RuntimeError(f"GraphConvolution: adjacency ({adj.n_clouds} x {adj.n}) does not match x {tuple(x.shape)}")
        else:
            adj = _need(adj, "adj", torch.float32, 3)
        w, bias = self.conv.weight, self.conv.bias
        if _wants_grad(x, w, bias):
            return _GcnConvFn.apply(x, w, bias, adj, sparse)
        with torch.no_grad():
            return _GcnConvFn.apply(x, w.detach(), None if bias is None else bias.detach(), adj, sparse)


class GlobalPooling(nn.Module):
    def forward(self, x):
        """``x [B, C, n]`` -> ``[B, 2C]`` = max over points || mean over points (gcnLayer.py:90-94); HIP forward and backward."""
        x = _need(x.contiguous().float(), "x", torch.float32, 3)
        return _GcnPoolFn.apply(x)
