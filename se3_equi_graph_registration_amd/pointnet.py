"""PointNet encoder with the reference interface (src/3dmatch_train_egnn_with_batch.py:52-96), HIP forward
(SURVEY §8(f) N1).  ``PointNetLayer`` is a max-aggregation message-passing layer; the per-edge first Linear is
folded into per-node projections (csrc/pointnet.hip).  The reference instantiates a fresh, untrained ``PointNet`` per batch
(train :1034) and never optimises it; the modules here are trainable all the same: under grad the backward re-runs the layer in
device ops on the saved inputs (ops.hip_forward_aten_backward)."""
from __future__ import annotations

import torch
from torch import nn

from . import _lib
from .ops import DeviceGraph, _need, _p, _stream, hip_forward_aten_backward


def _target_graph(edge_index, n_nodes):
    """``edge_index [2,E]`` (int64, [0] = source j, [1] = target i, PyG flow source_to_target) or a ``DeviceGraph``
    whose ``row`` = source and ``col`` = target with the by-col index built."""
    if isinstance(edge_index, DeviceGraph):
        if edge_index.cptr is None:
            raise RuntimeError("PointNetLayer: the DeviceGraph needs its col (target) index")
        return edge_index
    return DeviceGraph.from_edge_lists([edge_index], n_nodes, need_col_index=True)


class PointNetLayer(nn.Module):
    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(in_channels + 3, out_channels), nn.ReLU(), nn.Linear(out_channels, out_channels))

    @staticmethod
    def _aten(h, pos, w1, b1, w2, b2, row, col, relu_out):
        """PointNetLayer.forward :61-68 in device ops (backward recompute): max over incoming edges of mlp([h_j, pos_j - pos_i]),
        0 for nodes without incoming edges."""
        j, i = row.long(), col.long()
        msg = torch.cat([h[j], pos[j] - pos[i]], dim=-1)
        msg = torch.addmm(b2, torch.relu(torch.addmm(b1, msg, w1.t())), w2.t())
        out = torch.zeros(h.shape[0], msg.shape[1], dtype=msg.dtype, device=msg.device)
        out = out.scatter_reduce(0, i[:, None].expand_as(msg), msg, reduce="amax", include_self=False)
        return torch.relu(out) if relu_out else out

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

        def hip(h_, pos_, w1, b1, w2, b2):
            work = torch.empty(2 * n * c_out, dtype=torch.float32, device=h_.device)
            out = torch.empty(n, c_out, dtype=torch.float32, device=h_.device)
            _lib.check(_lib.lib().equigspr_pointnet_layer_forward(_p(h_), _p(pos_), n, c_in, c_out, _p(g.row), _p(g.cptr), _p(g.cperm),
                                                                  _p(w1.contiguous()), _p(b1.contiguous()), _p(w2.contiguous()),
                                                                  _p(b2.contiguous()), int(bool(_relu_out)), _p(work), _p(out), _stream()),
                       "pointnet_layer_forward")
            return out

        if torch.is_grad_enabled() and (h.requires_grad or pos.requires_grad or any(p.requires_grad for p in params)):
            row, col, relu = g.row[:g.n_edges], g.col[:g.n_edges], bool(_relu_out)
            return hip_forward_aten_backward(hip, lambda h_, p_, w1, b1, w2, b2: self._aten(h_, p_, w1, b1, w2, b2, row, col, relu),
                                             (h, pos, *params))
        return hip(h, pos, *(t.detach() for t in params))


def global_max_pool(x, batch, size=None):
    """``torch_geometric.nn.global_max_pool`` for an ascending ``batch`` vector: ``[num_graphs, C]``."""
    x = _need(x.contiguous().float(), "x", torch.float32, 2)
    batch = _need(batch.contiguous(), "batch", torch.int64, 1)
    nb = int(size) if size is not None else int(batch[-1].item()) + 1

    def hip(x_):
        out = torch.empty(nb, x_.shape[1], dtype=torch.float32, device=x_.device)
        _lib.check(_lib.lib().equigspr_global_max_pool(_p(x_), _p(batch), x_.shape[0], x_.shape[1], nb, _p(out), _stream()), "global_max_pool")
        return out

    def aten(x_):
        out = torch.full((nb, x_.shape[1]), float("-inf"), dtype=x_.dtype, device=x_.device)
        return out.scatter_reduce(0, batch[:, None].expand_as(x_), x_, reduce="amax", include_self=True)

    if torch.is_grad_enabled() and x.requires_grad:
        return hip_forward_aten_backward(hip, aten, (x,))
    return hip(x)


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
