"""Typed irrep-basis message passing ("irrep-basis message passing proper").

The reference ships the equivariant weight basis (``tools/basis.py::get_basis`` :153-205, ``tools/irr_repr.py:103-104``,
``tools/spherical_harmonics.py:75-123``; here ``sh.get_basis``) and no layer that uses it -- its scripts run the frame-vector EGNN.
``IrrepConv`` is that layer for degrees 0 and 1: degree-0 features ``f0 [N, C0]`` and degree-1 features ``f1 [N, C1, 3]`` are BOTH
carried through, every (d_in, d_out) path is a pairwise kernel ``sum_J R_J(|r_ij|) K_J(r_ij / |r_ij|)`` built from the basis tensors,
messages are computed from the features at ``col`` and aggregated at ``row`` (the reference's conventions, train :252-256,:270-278),
and a per-degree channel mixing acts as self-interaction.  The definition is written out in ``oracle/irreps_oracle.py`` (test
infrastructure; parity unpinned by construction -- there is no reference layer -- pinned by SO(3) covariance).

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
        moments = torch.empty(L.equigspr_irrep_moments_size(ctypes.byref(cfg), n), dtype=torch.float32, device=dev)
        _lib.check(L.equigspr_irrep_conv_forward(ctypes.byref(cfg), _p(w1), _p(b1), _p(w2), _p(b2), _p(ws0), _p(ws1),
                                                 ctypes.byref(graph.struct), _p(f0), _p(f1), _p(x), _p(out0), _p(out1), _p(moments),
                                                 _stream()), "irrep_conv_forward")
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
