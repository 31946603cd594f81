"""Typed irrep-basis message passing ("irrep-basis message passing proper").

The reference ships the equivariant weight basis (``tools/basis.py::get_basis`` :153-205, ``tools/irr_repr.py:103-104``,
``tools/spherical_harmonics.py:75-123``; here ``sh.get_basis``) and no layer that uses it -- its scripts run the frame-vector EGNN.
``IrrepConv`` is that layer for degrees 0 and 1: degree-0 features ``f0 [N, C0]`` and degree-1 features ``f1 [N, C1, 3]`` are BOTH
carried through, every (d_in, d_out) path is a pairwise kernel ``sum_J R_J(|r_ij|) K_J(r_ij / |r_ij|)`` built from the basis tensors,
messages are computed from the features at ``col`` and aggregated at ``row`` (the reference's conventions, train :252-256,:270-278),
and a per-degree channel mixing acts as self-interaction.  The definition is written out in DESIGN.md section 2 and in the test
infrastructure (parity unpinned by construction -- there is no reference layer -- pinned by SO(3) covariance).

One kernel path (``csrc/irreps.hip``, round 5), forward and backward, no library call and no basis tensor in HBM:
  * the degree <= 1 basis in closed form inside the kernels (``sh.degree1_closed_form``: six constants fitted to ``get_basis``);
  * aggregate-then-transform: the radial profile is LINEAR in the hidden radial activations, so per-node moments of
    [basis x gathered features] (x) [hid, 1] are accumulated first (outer products on the matrix cores, ascending edge order) and the
    radial weights act on them per NODE in a second launch that reads the weights in place;
  * backward for every input: features (the adjoint convolution -- the same two kernels on the reversed edges with transposed
    weights), coordinates ``x``, radial and self-interaction weights.

Degree-1 components are in the basis' own real-SH order (m = -1, 0, 1, polar axis on input component 1): ``Y_1(x) = A x / |x|`` with
``A = sh1_matrix()``; in this convention ``A = -sqrt(3 / 4 pi) * I``, so a Cartesian vector ``v`` enters as ``A @ v`` and rotates
with the plain rotation matrix.  No CPU path.
"""
from __future__ import annotations

import ctypes

import torch
from torch import nn

from . import _lib, sh
from .flat import FlatParamsMixin
from .ops import DeviceGraph, _need, _p, _stream


def sh1_matrix(device="cuda", dtype=torch.float32) -> torch.Tensor:
    """``A [3, 3]`` with ``Y_1(x) = A x / |x|`` in the basis' component order (evaluated by the HIP SH kernel, not hard-coded)."""
    eye = torch.eye(3, dtype=torch.float32, device=device)
    y = sh.precompute_sh(sh.get_spherical_from_cartesian(eye), 1)[1].reshape(3, 3)       # rows: Y_1(e_k)
    return y.T.contiguous().to(dtype)


def _config(c0_in, c1_in, c0_out, c1_out, hidden) -> _lib.IrrepConfig:
    return _lib.IrrepConfig(c0_in, c1_in, c0_out, c1_out, hidden, *sh.degree1_closed_form())


class _IrrepConvFn(torch.autograd.Function):
    """``equigspr_irrep_conv_forward`` / ``_backward``: gradients for f0, f1, x and all six parameter tensors from HIP kernels."""

    @staticmethod
    def forward(ctx, f0, f1, x, w1, b1, w2, b2, ws0, ws1, cfg, graph, training):
        L = _lib.lib()
        n = x.shape[0]
        dev = x.device
        out0 = torch.empty(n, cfg.c0_out, dtype=torch.float32, device=dev)
        out1 = torch.empty(n, cfg.c1_out, 3, dtype=torch.float32, device=dev)
        moments = torch.empty(L.equigspr_irrep_moments_size(ctypes.byref(cfg), n, graph.n_edges), dtype=torch.float32, device=dev)
        _lib.check(L.equigspr_irrep_conv_forward(ctypes.byref(cfg), _p(w1), _p(b1), _p(w2), _p(b2), _p(ws0), _p(ws1),
                                                 ctypes.byref(graph.struct), _p(f0), _p(f1), _p(x), _p(out0), _p(out1), _p(moments),
                                                 int(bool(training)), _stream()), "irrep_conv_forward")
        if training:
            ctx.save_for_backward(f0, f1, x, w1, b1, w2, b2, ws0, ws1, moments)
            ctx.cfg, ctx.graph = cfg, graph
        return out0, out1

    @staticmethod
    def backward(ctx, g0, g1):
        f0, f1, x, w1, b1, w2, b2, ws0, ws1, moments = ctx.saved_tensors
        cfg, graph = ctx.cfg, ctx.graph
        L = _lib.lib()
        n, dev = x.shape[0], x.device
        g0 = torch.zeros(n, cfg.c0_out, dtype=torch.float32, device=dev) if g0 is None else g0.contiguous().float()
        g1 = torch.zeros(n, cfg.c1_out, 3, dtype=torch.float32, device=dev) if g1 is None else g1.contiguous().float()
        need_f = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        d_f0 = torch.empty_like(f0) if need_f else None
        d_f1 = torch.empty_like(f1) if need_f else None
        d_x = torch.zeros_like(x) if ctx.needs_input_grad[2] else None             # the kernel ACCUMULATES into d_x
        dw1, db1, dw2, db2, dws0, dws1 = (torch.empty_like(t) for t in (w1, b1, w2, b2, ws0, ws1))
        work = torch.empty(L.equigspr_irrep_bwd_work_size(ctypes.byref(cfg), n, graph.n_edges), dtype=torch.float32, device=dev)
        _lib.check(L.equigspr_irrep_conv_backward(ctypes.byref(cfg), _p(w1), _p(b1), _p(w2), _p(b2), _p(ws0), _p(ws1),
                                                  ctypes.byref(graph.struct), _p(f0), _p(f1), _p(x), _p(moments), _p(g0), _p(g1),
                                                  _p(d_f0), _p(d_f1), _p(d_x), _p(dw1), _p(db1), _p(dw2), _p(db2), _p(dws0), _p(dws1),
                                                  _p(work), _stream()), "irrep_conv_backward")
        return d_f0, d_f1, d_x, dw1, db1, dw2, db2, dws0, dws1, None, None, None


class IrrepConv(nn.Module):
    """``IrrepConv(c0_in, c1_in, c0_out, c1_out, hidden=32)``; ``forward(f0, f1, x, edges) -> (out0 [N, c0_out], out1 [N, c1_out, 3])``.

    ``edges``: ``[row, col]`` (LongTensors / ``[2, E]``) or a ``DeviceGraph``.  1..32 channels per degree; ``hidden`` (the width of
    the radial MLP) 16 or 32.  Differentiable in f0, f1, x and every parameter (HIP backward)."""

    def __init__(self, c0_in, c1_in, c0_out, c1_out, hidden=32, device="cuda"):
        super().__init__()
        for c in (c0_in, c1_in, c0_out, c1_out):
            if not 1 <= c <= 32:
                raise NotImplementedError("IrrepConv: 1..32 channels per degree")
        if hidden not in (16, 32):
            raise NotImplementedError("IrrepConv: the radial MLP's hidden width must be 16 or 32 (16-column MFMA blocks)")
        self.c0_in, self.c1_in, self.c0_out, self.c1_out, self.hidden = c0_in, c1_in, c0_out, c1_out, hidden
        self.n_w = c0_out * c0_in + c1_out * c0_in + c0_out * c1_in + c1_out * c1_in * 3
        self.radial = nn.Sequential(nn.Linear(1, hidden), nn.SiLU(), nn.Linear(hidden, self.n_w))
        self.self0 = nn.Linear(c0_in, c0_out, bias=False)
        self.self1 = nn.Linear(c1_in, c1_out, bias=False)
        self.cfg = _config(c0_in, c1_in, c0_out, c1_out, hidden)
        self.to(device)

    @staticmethod
    def _graph(edges, n_nodes, need_col_index=True):
        if isinstance(edges, DeviceGraph):
            return edges
        row, col = edges[0], edges[1]
        return DeviceGraph.from_edge_lists([torch.stack([row, col]).to(torch.int64)], n_nodes, need_col_index=need_col_index)

    def forward(self, f0, f1, x, edges):
        f0 = _need(f0.contiguous().float(), "f0", torch.float32, 2)
        f1 = _need(f1.contiguous().float(), "f1", torch.float32, 3)
        x = _need(x.contiguous().float(), "x", torch.float32, 2)
        n = x.shape[0]
        if f0.shape != (n, self.c0_in) or f1.shape != (n, self.c1_in, 3) or x.shape[1] != 3:
            raise RuntimeError(f"IrrepConv: need f0 [N, {self.c0_in}], f1 [N, {self.c1_in}, 3], x [N, 3]")
        lin1, lin2 = self.radial[0], self.radial[2]
        params = (lin1.weight, lin1.bias, lin2.weight, lin2.bias, self.self0.weight, self.self1.weight)
        training = torch.is_grad_enabled() and (f0.requires_grad or f1.requires_grad or x.requires_grad or
                                                any(p.requires_grad for p in params))
        graph = self._graph(edges, n, need_col_index=training)
        if training and graph.cptr is None:
            raise RuntimeError("IrrepConv.forward: the backward needs a DeviceGraph built with need_col_index=True")
        return _IrrepConvFn.apply(f0, f1, x, *params, self.cfg, graph, training)


class IrrepGNN(nn.Module):
    """A small stack of ``IrrepConv`` layers with gated non-linearities (norm-gated degree-1 channels, SiLU on degree 0) -- the typed
    counterpart of ``EGNN``: ``forward(h [N, c0], v [N, c1, 3] or None, x [N, 3], edges) -> (h', v')``.  ``v = None`` starts the
    degree-1 channels at zero (they are then created from the geometry by the 0 -> 1 path)."""

    def __init__(self, c0_in, c0_hidden, c0_out, c1=8, n_layers=2, hidden=32, device="cuda"):
        super().__init__()
        dims0 = [c0_in] + [c0_hidden] * (n_layers - 1) + [c0_out]
        self.c1 = c1
        self.layers = nn.ModuleList([IrrepConv(dims0[i], c1, dims0[i + 1], c1, hidden, device) for i in range(n_layers)])
        self.gates = nn.ModuleList([nn.Linear(dims0[i + 1], c1) for i in range(n_layers - 1)]).to(device)

    def forward(self, h, v, x, edges):
        n = x.shape[0]
        if v is None:
            v = torch.zeros(n, self.c1, 3, dtype=torch.float32, device=x.device)
        graph = IrrepConv._graph(edges, n)
        xc = _need(x.contiguous().float(), "x", torch.float32, 2)
        for i, layer in enumerate(self.layers):
            h, v = layer(h, v, xc, graph)
            if i + 1 < len(self.layers):
                v = v * torch.sigmoid(self.gates[i](h))[:, :, None]       # invariant gate: keeps degree 1 covariant
                h = torch.nn.functional.silu(h)
        return h, v


# ------------------------------------------------------------------------------------------------ the typed network behind the EGNN interface
class _IrrepGnnFn(torch.autograd.Function):
    """``equigspr_irrep_gnn_forward`` / ``_backward``: the whole stack in HIP launches; parameter gradients are accumulated by the
    kernels straight into ``model.flat_grads`` (whose views are the parameters' ``.grad``), only the input gradients travel through
    autograd (``anchor`` = one parameter, so that the node exists when only the weights require gradients)."""

    @staticmethod
    def forward(ctx, h, x, anchor, model, graph, training):
        L = _lib.lib()
        cfg, n, dev = model.cfg, x.shape[0], x.device
        h_out = torch.empty(n, cfg.out_nf, dtype=torch.float32, device=dev)
        x_out = torch.empty(n, 3, dtype=torch.float32, device=dev)
        saved = torch.empty(L.equigspr_irrep_gnn_saved_size(ctypes.byref(cfg), n, graph.n_edges), dtype=torch.float32, device=dev)
        _lib.check(L.equigspr_irrep_gnn_forward(ctypes.byref(cfg), _p(model.flat_params), ctypes.byref(graph.struct), _p(h), _p(x),
                                                _p(h_out), _p(x_out), _p(saved), int(bool(training)), _stream()), "irrep_gnn_forward")
        if training:
            ctx.save_for_backward(h, x, saved)
            ctx.model, ctx.graph = model, graph
        return h_out, x_out

    @staticmethod
    def backward(ctx, gh, gx):
        h, x, saved = ctx.saved_tensors
        model, graph = ctx.model, ctx.graph
        L = _lib.lib()
        cfg, n, dev = model.cfg, x.shape[0], x.device
        gh = torch.zeros(n, cfg.out_nf, dtype=torch.float32, device=dev) if gh is None else gh.contiguous().float()
        gx = torch.zeros(n, 3, dtype=torch.float32, device=dev) if gx is None else gx.contiguous().float()
        model._attach_grads()
        d_h = torch.empty_like(h) if ctx.needs_input_grad[0] else None
        d_x = torch.zeros_like(x) if ctx.needs_input_grad[1] else None            # accumulated over the layers and the head
        work = torch.empty(L.equigspr_irrep_gnn_work_size(ctypes.byref(cfg), n, graph.n_edges), dtype=torch.float32, device=dev)
        _lib.check(L.equigspr_irrep_gnn_backward(ctypes.byref(cfg), _p(model.flat_params), ctypes.byref(graph.struct), _p(h), _p(x), _p(saved),
                                                 _p(gh), _p(gx), _p(d_h), _p(d_x), _p(model.flat_grads), _p(work), _stream()),
                   "irrep_gnn_backward")
        return d_h, d_x, None, None, None, None


class IrrepEGNN(FlatParamsMixin, nn.Module):
    """The typed irrep-basis network behind the reference's ``EGNN`` interface (train :293-340; signature donor
    ``src/clean_so3_model.py:162``): ``forward(h [N, in_node_nf], x [N, 3], edges, edge_attr) -> (h [N, out_node_nf], x [N, 3])``, so it
    drops into ``CrossAttentionPoseRegression(egnn=IrrepEGNN(...))``, ``train.make_optimizer`` and ``train.GraphedStep`` unchanged.

    ``n_layers`` x ``IrrepConv`` with degree-0 widths ``in -> hidden -> ... -> out`` and ``c1`` degree-1 channels (created from the
    geometry by the 0 -> 1 path: the input has none), SiLU on degree 0 and an invariant sigmoid gate per degree-1 channel between
    the layers, and the equivariant coordinate update ``x + sum_i w_x[i] v_i`` at the end: h is rotation / translation INVARIANT, x
    covariant.  ``edge_attr`` is accepted and ignored (``in_edge_nf`` must be 0: the kernels of this layer are functions of the
    geometry alone).  All parameters live in one flat block (``flat.FlatParamsMixin``); every launch is a hand-written HIP kernel
    (``equigspr_irrep_gnn_forward`` / ``_backward``): 3 per layer forward, 8-10 per layer backward, hipGraph-capturable."""

    def __init__(self, in_node_nf, hidden_nf, out_node_nf, in_edge_nf=0, device="cuda:0", n_layers=4, c1=8, radial_hidden=32):
        super().__init__()
        if in_edge_nf != 0:
            raise NotImplementedError("IrrepEGNN: in_edge_nf must be 0 (the typed layer has no edge attribute input)")
        self.in_node_nf, self.hidden_nf, self.out_node_nf, self.in_edge_nf = in_node_nf, hidden_nf, out_node_nf, 0
        self.n_layers, self.c1, self.device = n_layers, c1, device
        dims0 = [in_node_nf] + [hidden_nf] * (n_layers - 1) + [out_node_nf]
        self.layers = nn.ModuleList([IrrepConv(dims0[i], c1, dims0[i + 1], c1, radial_hidden, device) for i in range(n_layers)])
        self.gates = nn.ModuleList([nn.Linear(dims0[i + 1], c1) for i in range(n_layers - 1)])
        self.coord = nn.Linear(c1, 1, bias=False)
        nn.init.xavier_uniform_(self.coord.weight, gain=1e-3)           # a small coordinate step at initialisation, as train :219-220
        self.cfg = _lib.IrrepGnnConfig(in_node_nf, hidden_nf, out_node_nf, c1, n_layers, radial_hidden, *sh.degree1_closed_form())
        if _lib.lib().equigspr_irrep_gnn_param_count(ctypes.byref(self.cfg)) < 0:
            raise NotImplementedError("IrrepEGNN: 1..32 channels per degree, 1..16 layers, radial_hidden 16 or 32")
        self.to(device)

    def _flat_layout(self):
        if getattr(self, "_layout_cache", None) is not None:          # the configuration is fixed at construction: 8 L ctypes calls once, not per forward
            return self._layout_cache
        L = _lib.lib()
        cfg = ctypes.byref(self.cfg)
        offs = []
        for l in range(self.n_layers):
            offs += [int(L.equigspr_irrep_gnn_param_offset(cfg, l, w)) for w in range(6)]
        for l in range(self.n_layers - 1):
            offs += [int(L.equigspr_irrep_gnn_param_offset(cfg, l, 6)), int(L.equigspr_irrep_gnn_param_offset(cfg, l, 7))]
        offs.append(int(L.equigspr_irrep_gnn_param_offset(cfg, 0, 8)))
        ps = self._ordered_params()
        if len(offs) != len(ps) or min(offs) < 0:
            raise RuntimeError("internal: IrrepEGNN parameters do not match the C-ABI block layout")
        self._layout_cache = (offs, int(L.equigspr_irrep_gnn_param_count(cfg)))
        return self._layout_cache

    def forward(self, h, x, edges, edge_attr=None):
        self._ensure_flat()
        if h.dim() != 2 or x.dim() != 2 or h.shape[1] != self.in_node_nf or x.shape[1] != 3 or h.shape[0] != x.shape[0]:
            raise RuntimeError(f"IrrepEGNN.forward expects h [N, {self.in_node_nf}] and x [N, 3] (one cloud or a pre-batched graph)")
        h = _need(h.contiguous().float(), "h", torch.float32, 2)
        x = _need(x.contiguous().float(), "x", torch.float32, 2)
        # training = any parameter or input wants a gradient (a frozen `coord.weight` beside trainable layers used to
        # switch the whole forward to inference); the autograd anchor is a parameter that requires grad, so that backward is called
        wants = [p for p in self.parameters() if p.requires_grad]
        anchor = wants[0] if wants else self.coord.weight
        training = torch.is_grad_enabled() and (bool(wants) or h.requires_grad or x.requires_grad)
        graph = IrrepConv._graph(edges, h.shape[0], need_col_index=training)
        if training and graph.cptr is None:
            raise RuntimeError("IrrepEGNN.forward: training needs a DeviceGraph built with need_col_index=True")
        return _IrrepGnnFn.apply(h, x, anchor, self, graph, training)
