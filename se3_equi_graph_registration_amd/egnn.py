"""Host-side mirror of the reference EGNN interface, backed by the HIP kernels.

``EGNN`` / ``E_GCL`` keep the constructor signatures, parameter names (``state_dict`` keys)
and the ``forward(h, x, edges, edge_attr)`` contract of the reference
(src/3dmatch_train_egnn_with_batch.py:185-340; signature donor src/clean_so3_model.py:162),
plus ``num_heads`` so the shipped checkpoints load.

MI355X-first layout: all parameters of an ``EGNN`` are views into ONE flat fp32 device
buffer (``EGNN.flat_params``), all their gradients views into ONE flat buffer
(``EGNN.flat_grads``) -- the kernels read the block directly and the data-parallel
all-reduce sends it as a single bucket over xGMI.  There is no CPU fallback: calling
``forward`` with CPU tensors raises.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch import nn

from . import ops
from .flat import FlatParamsMixin
from .ops import DeviceGraph


EGNN_ONES = "ones"      # sentinel for edge_attr: the constant all-ones attribute (kernel-side, no tensor)


class E_GCL(nn.Module):
    """Parameter container of one layer; same submodule names as the reference (train :185-229).
    The layer math runs inside ``EGNN.forward`` (fused HIP path); a stand-alone, differentiable
    ``E_GCL.forward`` is provided through a one-layer EGNN with identity embeddings that shares this module's parameters."""

    def __init__(self, input_nf, output_nf, hidden_nf, edges_in_d=0, num_heads=1, act_fn=nn.SiLU(), residual=True,
                 attention=False, normalize=False, tanh=False, device="cuda:0"):
        super().__init__()
        if not (input_nf == output_nf == hidden_nf):
            raise NotImplementedError("HIP E_GCL needs input_nf == hidden_nf == output_nf (as every reference call site)")
        if not isinstance(act_fn, nn.SiLU):
            raise NotImplementedError("HIP E_GCL implements act_fn=nn.SiLU() only (the reference default)")
        if not residual or normalize or tanh:
            raise NotImplementedError("HIP E_GCL implements residual=True, normalize=False, tanh=False "
                                      "(the only combination the reference scripts use; normalize=True would hit "
                                      "the undefined self.epsilon at train :276)")
        if hidden_nf % num_heads:
            raise ValueError("hidden_nf must be divisible by num_heads")
        self.residual, self.attention, self.normalize, self.tanh = residual, attention, normalize, tanh
        self.num_heads, self.device = num_heads, device
        self.hidden_nf, self.edges_in_d = hidden_nf, edges_in_d
        feature_dim = 2 * input_nf + edges_in_d + 1 + 9 + 2          # train :196-199
        hh = hidden_nf // num_heads
        self.edge_mlps = nn.ModuleList([
            nn.Sequential(nn.Linear(feature_dim, hh), act_fn, nn.Linear(hh, hh)) for _ in range(num_heads)])
        self.layer_norm = nn.LayerNorm(hidden_nf)
        self.node_mlp = nn.Sequential(nn.Linear(hidden_nf + input_nf, hidden_nf), act_fn, nn.Linear(hidden_nf, output_nf))
        last = nn.Linear(hidden_nf, 1, bias=False)
        nn.init.xavier_uniform_(last.weight, gain=1e-3)             # train :219-220
        self.coord_mlp = nn.Sequential(nn.Linear(hidden_nf, hidden_nf), act_fn, last)

    def forward(self, h, edge_index, coord, edge_attr=None):
        """Stand-alone layer call (train :280-289) -> (h, coord, edge_attr).  Runs as a one-layer EGNN with identity embeddings
        that holds a COPY of this layer's weights (refreshed at every call: a few KB on the device); the gradients of the copy are
        handed back through autograd to this module's own parameters.  The layer's ``nn.Parameter``s are never re-homed, so a layer
        that belongs to a model (``egnn.gcl_0(h, edges, x)``) keeps living in its EGNN's flat block -- optimizers and
        data-parallel buckets built on that block stay valid."""
        wrap = self.__dict__.get("_standalone")
        dev = h.device
        if wrap is None or wrap.flat_params is None or wrap.flat_params.device != dev:
            f = self.hidden_nf
            wrap = EGNN(f, f, f, in_edge_nf=self.edges_in_d, device=dev, n_layers=1, num_heads=self.num_heads)
            with torch.no_grad():
                wrap.embedding_in.weight.copy_(torch.eye(f)); wrap.embedding_in.bias.zero_()
                wrap.embedding_out.weight.copy_(torch.eye(f)); wrap.embedding_out.bias.zero_()
            wrap._ensure_flat()
            self.__dict__["_standalone"] = wrap        # not a registered submodule: no parameter cycle, not in state_dict()
        if h.dim() != 2 or coord.dim() != 2:
            raise RuntimeError("E_GCL.forward expects h [N,F] and coord [N,3]")
        graph = wrap._graph_for(edge_index, h.shape[0])
        if self.edges_in_d == 0:
            ea = None
        elif edge_attr is None:
            raise RuntimeError("E_GCL was built with edges_in_d=1 but edge_attr is None")
        else:
            ea = edge_attr.reshape(-1).contiguous().float()
        params = list(self.parameters())
        training = torch.is_grad_enabled() and (any(p.requires_grad for p in params) or h.requires_grad or coord.requires_grad)
        if training and graph.cptr is None:
            raise RuntimeError("E_GCL.forward: training needs a DeviceGraph built with need_col_index=True")
        h2, x2 = _LayerFn.apply(h.contiguous().float(), coord.contiguous().float(), wrap, graph, ea, training, *params)
        return h2, x2, edge_attr

    def __deepcopy__(self, memo):                       # the cached one-layer wrapper is per instance and rebuilt on demand
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != "_standalone":
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k != "_standalone"}


class _LayerFn(torch.autograd.Function):
    """One E_GCL through the one-layer wrapper EGNN ``wrap`` (identity embeddings).  The layer's parameters are autograd inputs:
    forward copies them into the wrapper's block, backward runs the HIP backward into the wrapper's gradient block and returns
    its slices -- autograd then accumulates them into the layer's own ``.grad`` wherever those live."""

    @staticmethod
    def forward(ctx, h, x, wrap, graph, edge_attr, training, *params):
        layer = wrap._modules["gcl_0"]
        with torch.no_grad():
            for dst, src in zip(layer.parameters(), params):
                dst.copy_(src.detach())
        h_out, x_out, saved = ops.egnn_forward(wrap.cfg, wrap.flat_params, graph, h, x, edge_attr, training)
        ctx.wrap, ctx.graph, ctx.edge_attr, ctx.n_params = wrap, graph, edge_attr, len(params)
        if training:
            ctx.save_for_backward(h, saved, wrap.flat_params.clone())      # the weights THIS call ran with
        return h_out, x_out

    @staticmethod
    def backward(ctx, gh, gx):
        wrap = ctx.wrap
        h, saved, flat = ctx.saved_tensors
        n = ctx.graph.n_nodes
        gh = torch.zeros(n, wrap.cfg.out_nf, dtype=torch.float32, device=h.device) if gh is None else gh
        gx = torch.zeros(n, 3, dtype=torch.float32, device=h.device) if gx is None else gx
        wrap.flat_grads.zero_()
        wrap._attach_grads()
        ghi, gxi = ops.egnn_backward(wrap.cfg, flat, ctx.graph, h, ctx.edge_attr, saved, gh.contiguous(), gx.contiguous(),
                                     wrap.flat_grads, (ctx.needs_input_grad[0], ctx.needs_input_grad[1]))
        layer = wrap._modules["gcl_0"]
        pg = [p.grad.clone() if need else None for p, need in zip(layer.parameters(), ctx.needs_input_grad[6:])]
        return (ghi, gxi, None, None, None, None, *pg)


class _EgnnFn(torch.autograd.Function):
    """Autograd node around equigspr_egnn_forward / equigspr_egnn_backward.  Parameter
    gradients are accumulated by the kernel straight into ``model.flat_grads`` (whose views
    are the parameters' ``.grad``); only the input gradients travel through autograd."""

    @staticmethod
    def forward(ctx, h, x, anchor, model, graph, edge_attr, training):
        h_out, x_out, saved = ops.egnn_forward(model.cfg, model.flat_params, graph, h, x, edge_attr, training)
        ctx.model, ctx.graph, ctx.edge_attr = model, graph, edge_attr
        if training:
            ctx.save_for_backward(h, saved)
        return h_out, x_out

    @staticmethod
    def backward(ctx, gh, gx):
        model = ctx.model
        h, saved = ctx.saved_tensors
        n = ctx.graph.n_nodes
        gh = torch.zeros(n, model.cfg.out_nf, dtype=torch.float32, device=h.device) if gh is None else gh
        gx = torch.zeros(n, 3, dtype=torch.float32, device=h.device) if gx is None else gx
        model._attach_grads()
        ghi, gxi = ops.egnn_backward(model.cfg, model.flat_params, ctx.graph, h, ctx.edge_attr, saved, gh, gx,
                                     model.flat_grads, (ctx.needs_input_grad[0], ctx.needs_input_grad[1]))
        return ghi, gxi, None, None, None, None, None


class EGNN(FlatParamsMixin, nn.Module):
    """Drop-in for the reference ``EGNN`` (train :293-340)."""

    def __init__(self, in_node_nf, hidden_nf, out_node_nf, in_edge_nf=0, device="cuda:0", act_fn=nn.SiLU(), n_layers=5,
                 residual=True, attention=True, normalize=False, tanh=False, num_heads=1, edge_basis="frames"):
        """``edge_basis="irreps"`` (extension): the 12 geometric inputs of every edge MLP are rotation
        invariants from the degree <= 1 irrep basis kernels of ``tools/basis.py::get_basis`` contracted with the endpoint
        coordinates, instead of the reference's frame-vector components -- same parameter shapes, an SO(3)-equivariant layer
        (h invariant, x covariant under rotations about the origin)."""
        super().__init__()
        self.edge_basis = edge_basis
        self.hidden_nf, self.device, self.n_layers = hidden_nf, device, n_layers
        self.in_node_nf, self.out_node_nf, self.in_edge_nf, self.num_heads = in_node_nf, out_node_nf, in_edge_nf, num_heads
        self.embedding_in = nn.Linear(in_node_nf, hidden_nf)
        self.embedding_out = nn.Linear(hidden_nf, out_node_nf)
        for i in range(n_layers):
            self.add_module("gcl_%d" % i, E_GCL(hidden_nf, hidden_nf, hidden_nf, edges_in_d=in_edge_nf, num_heads=num_heads,
                                                act_fn=act_fn, residual=residual, attention=attention,
                                                normalize=normalize, tanh=tanh))
        self.cfg = ops.make_config(in_node_nf, hidden_nf, out_node_nf, in_edge_nf, n_layers, num_heads, edge_basis)
        self.flat_params: Optional[torch.Tensor] = None
        self.flat_grads: Optional[torch.Tensor] = None
        self._graph_cache = {}
        self.to(device)                                             # train :326

    # -- flat parameter block: flat.FlatParamsMixin (named_parameters() order == the C-ABI block order, include/equigspr.h)
    def _flat_layout(self):
        offs, total = FlatParamsMixin._flat_layout(self)
        if total != ops.param_count(self.cfg):
            raise RuntimeError("internal: parameter count differs from the C-ABI block layout")
        return offs, total

    # -- graph handling ----------------------------------------------------------------
    def _graph_for(self, edges, n_nodes: int) -> DeviceGraph:
        if isinstance(edges, DeviceGraph):
            return edges
        if isinstance(edges, torch.Tensor):
            row, col = edges[0], edges[1]
        else:
            row, col = edges[0], edges[1]
        key = (row.data_ptr(), col.data_ptr(), row._version, col._version, row.numel(), n_nodes)
        g = self._graph_cache.get(key)
        if g is None:
            if len(self._graph_cache) > 8:
                self._graph_cache.clear()
            e = torch.stack([row, col]).to(torch.int64)
            g = DeviceGraph.from_edge_lists([e], n_nodes)
            g._keepalive = (row, col)
            self._graph_cache[key] = g
        return g

    # -- reference signature -----------------------------------------------------------
    def forward(self, h, x, edges, edge_attr):
        """``h [N,in_node_nf]``, ``x [N,3]``, ``edges=[row,col]`` (LongTensors, list or ``[2,E]``) or a
        prebuilt ``DeviceGraph``, ``edge_attr [E,in_edge_nf]`` -> ``(h [N,out_node_nf], x [N,3])``."""
        self._ensure_flat()
        if h.dim() != 2 or x.dim() != 2:
            raise RuntimeError("EGNN.forward expects h [N,F] and x [N,3] (one cloud or a pre-batched graph)")
        graph = self._graph_for(edges, h.shape[0])
        if self.in_edge_nf == 0 or (isinstance(edge_attr, str) and edge_attr == EGNN_ONES):
            edge_attr = None
        elif edge_attr is None:
            raise RuntimeError("EGNN was built with in_edge_nf=1 but edge_attr is None")
        else:
            edge_attr = edge_attr.reshape(-1).contiguous().float()
        anchor = self.embedding_in.weight
        training = torch.is_grad_enabled() and (anchor.requires_grad or h.requires_grad or x.requires_grad)
        if training and graph.cptr is None:
            raise RuntimeError("EGNN.forward: training needs a DeviceGraph built with need_col_index=True")
        return _EgnnFn.apply(h.contiguous().float(), x.contiguous().float(), anchor, self, graph, edge_attr, training)
