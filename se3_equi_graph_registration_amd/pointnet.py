"""PointNet encoder with the reference interface (src/3dmatch_train_egnn_with_batch.py:52-96), HIP forward
(SURVEY §8(f) N1).  ``PointNetLayer`` is a max-aggregation message-passing layer; the per-edge first Linear is
folded into per-node projections (csrc/pointnet.hip).  The reference instantiates a fresh, untrained ``PointNet`` per batch
(train :1034) and never optimises it; the modules here are trainable all the same, forward AND backward in HIP (round 6:
``equigspr_pointnet_layer_backward`` -- the maximum's gradient goes to one incoming edge per channel, as torch_geometric's max
aggregation routes it -- and ``equigspr_global_max_pool_backward``)."""
from __future__ import annotations

import torch
from torch import nn

from . import _lib
from .ops import DeviceGraph, _need, _p, _stream


def _target_graph(edge_index, n_nodes):
    """``edge_index [2,E]`` (int64, [0] = source j, [1] = target i, PyG flow source_to_target) or a ``DeviceGraph``
    whose ``row`` = source and ``col`` = target with the by-col index built."""
    if isinstance(edge_index, DeviceGraph):
        if edge_index.cptr is None:
            raise RuntimeError("PointNetLayer: the DeviceGraph needs its col (target) index")
        return edge_index
    return DeviceGraph.from_edge_lists([edge_index], n_nodes, need_col_index=True)


class _PointNetLayerFn(torch.autograd.Function):
    """PointNetLayer.forward (train :61-68) with a HIP backward (csrc/pointnet.hip)."""

    @staticmethod
    def forward(ctx, h, pos, w1, b1, w2, b2, g, relu_out):
        n, c_in = h.shape
        c_out = w2.shape[0]
        w1, b1, w2, b2 = (t.contiguous() for t in (w1, b1, w2, b2))
        work = torch.empty(2 * n * c_out, dtype=torch.float32, device=h.device)
        out = torch.empty(n, c_out, dtype=torch.float32, device=h.device)
        _lib.check(_lib.lib().equigspr_pointnet_layer_forward(_p(h), _p(pos), n, c_in, c_out, _p(g.row), _p(g.cptr), _p(g.cperm), _p(w1), _p(b1),
                                                              _p(w2), _p(b2), int(bool(relu_out)), _p(work), _p(out), _stream()),
                   "pointnet_layer_forward")
        ctx.save_for_backward(h, pos, w1, b1, w2, b2)
        ctx.g, ctx.relu_out = g, bool(relu_out)
        return out

    @staticmethod
    def backward(ctx, gout):
        h, pos, w1, b1, w2, b2 = ctx.saved_tensors
        g = ctx.g
        n, c_in = h.shape
        c_out = w2.shape[0]
        L = _lib.lib()
        dev = h.device
        need = ctx.needs_input_grad
        gh = torch.empty_like(h) if need[0] else None
        gp = torch.empty_like(pos) if need[1] else None
        gw1 = torch.empty_like(w1) if need[2] else None
        gb1 = torch.empty_like(b1) if need[3] else None
        gw2 = torch.empty_like(w2) if need[4] else None
        gb2 = torch.empty_like(b2) if need[5] else None
        work = torch.empty(int(L.equigspr_pointnet_layer_backward_work_size(n, g.n_edges, c_in, c_out)), dtype=torch.float32, device=dev)
        _lib.check(L.equigspr_pointnet_layer_backward(_p(h), _p(pos), n, g.n_edges, c_in, c_out, _p(g.row), _p(g.cptr), _p(g.cperm), _p(g.rptr),
                                                      _p(g.rperm), _p(w1), _p(b1), _p(w2), _p(b2), int(ctx.relu_out),
                                                      _p(gout.contiguous().float()), _p(work), _p(gh), _p(gp), _p(gw1), _p(gb1), _p(gw2), _p(gb2),
                                                      _stream()), "pointnet_layer_backward")
        return gh, gp, gw1, gb1, gw2, gb2, None, None


class _GlobalMaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, batch, nb):
        out = torch.empty(nb, x.shape[1], dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().equigspr_global_max_pool(_p(x), _p(batch), x.shape[0], x.shape[1], nb, _p(out), _stream()), "global_max_pool")
        ctx.save_for_backward(x, batch)
        ctx.nb = nb
        return out

    @staticmethod
    def backward(ctx, g):
        x, batch = ctx.saved_tensors
        gx = torch.empty_like(x)
        _lib.check(_lib.lib().equigspr_global_max_pool_backward(_p(x), _p(batch), x.shape[0], x.shape[1], ctx.nb, _p(g.contiguous().float()),
                                                                _p(gx), _stream()), "global_max_pool_backward")
        return gx, None, None


class PointNetLayer(nn.Module):
    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(in_channels + 3, out_channels), nn.ReLU(), nn.Linear(out_channels, out_channels))

    def forward(self, h, pos, edge_index, _relu_out=False):
        h = _need(h.contiguous().float(), "h", torch.float32, 2)
        pos = _need(pos.contiguous().float(), "pos", torch.float32, 2)
        n, c_in = h.shape
        c_out = self.mlp[2].out_features
        if self.mlp[0].in_features != c_in + 3 or pos.shape != (n, 3):
            raise RuntimeError(f"PointNetLayer: h {tuple(h.shape)} / pos {tuple(pos.shape)} do not match in_channels={self.mlp[0].in_features - 3}")
        g = _target_graph(edge_index, n)
        if g.n_nodes != n:
            raise RuntimeError("PointNetLayer: graph and features disagree on the node count")
        params = (self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias)
        if torch.is_grad_enabled() and (h.requires_grad or pos.requires_grad or any(p.requires_grad for p in params)):
            return _PointNetLayerFn.apply(h, pos, *params, g, bool(_relu_out))
        with torch.no_grad():
            return _PointNetLayerFn.apply(h, pos, *(t.detach() for t in params), g, bool(_relu_out))


def global_max_pool(x, batch, size=None):
    """``torch_geometric.nn.global_max_pool`` for an ascending ``batch`` vector: ``[num_graphs, C]``."""
    x = _need(x.contiguous().float(), "x", torch.float32, 2)
    batch = _need(batch.contiguous(), "batch", torch.int64, 1)
    nb = int(size) if size is not None else int(batch[-1].item()) + 1
    return _GlobalMaxPoolFn.apply(x, batch, nb)


class PointNet(nn.Module):
    def __init__(self, in_num_feature=32, hidden_num_feature=64, output_num_feature=32, device="cuda:0"):
        super().__init__()
        self.hidden_num_feature = hidden_num_feature
        self.conv1 = PointNetLayer(in_num_feature, hidden_num_feature)
        self.conv2 = PointNetLayer(hidden_num_feature, output_num_feature)
        self.device = device

    def encode(self, pos, feats, edge_index):
        """Per-point features after conv1/ReLU/conv2/ReLU (``[N, output_num_feature]``) -- what the call site at
        train :1037 reshapes to ``[batch, 2048, 32]``."""
        g = _target_graph(edge_index, pos.shape[0])
        h = self.conv1(feats, pos, g, _relu_out=True)
        return self.conv2(h, pos, g, _relu_out=True)

    def forward(self, pos, feats, edge_index, batch):
        """``pos [N,3]``, ``feats [N,in]``, ``edge_index [2,E]``, ``batch [N]`` (ascending) -> ``[num_graphs, out]`` (:79-96)."""
        return global_max_pool(self.encode(pos, feats, edge_index), batch)
