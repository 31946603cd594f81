"""ORACLE (test infrastructure -- never imported by the product): the DEFINITION of the typed irrep-basis message-passing layer
("irrep-basis message passing proper").

PARITY UNPINNED by construction: the reference ships the basis (``tools/basis.py::get_basis`` :153-205, restated and pinned in
``oracle/sh_oracle.py``) but no layer that consumes it -- its ``README`` / paper title promise one, the scripts run the frame-vector
EGNN.  This file therefore states the layer the package implements (``se3_equi_graph_registration_amd/irreps.py``, ``csrc/irreps.hip``)
as plain PyTorch CPU code on top of the pinned basis, and the tests pin it by the property it exists for: degree-0 outputs are
invariant and degree-1 outputs transform with D^1(R) when the cloud and the degree-1 inputs are rotated.

The layer is the pairwise tensor-field convolution the basis was written for (the "kernel" K = sum_J R_J(|r|) K_J(r / |r|) of
``tools/basis.py``'s docstring), restricted to degrees 0 and 1, on the reference's graph conventions (``edges = [row, col]``, messages
computed from the features at ``col`` and aggregated at ``row`` like ``unsorted_segment_sum(edge_feat, row)``, train :252-256;
``r_ij = x[row] - x[col]`` like ``coord2radial``, train :270-278):

    d            = |r_ij|
    hid          = SiLU(w_r1 * d + b_r1)                                   [H]        (radial profile, shared by all paths)
    R            = W_r2 hid + b_r2  ->  R00 [C0o, C0i], R01 [C1o, C0i], R10 [C0o, C1i], R11 [C1o, C1i, 3 (J = 0, 1, 2)]
    K            = get_basis(r_ij, max_degree = 1)                         keys 'd_in,d_out'
    m0[e, o]     = sum_i R00[o, i] K['0,0'] f0[col, i]  +  sum_i R10[o, i] (K['1,0'] . f1[col, i, :])
    m1[e, o, :]  = sum_i R01[o, i] K['0,1'][:] f0[col, i]  +  sum_i sum_J R11[o, i, J] K['1,1'][:, :, J] f1[col, i, :]
    out0[n]      = W_s0 f0[n] + sum_{e: row = n} m0[e]          out1[n] = W_s1 f1[n] + sum_{e: row = n} m1[e]   (self-interaction: channel mixing)
On an edge with r_ij = 0 (self loops) the kernels with J >= 1 are zero: K['0,1'], K['1,0'] and the J = 1, 2 parts of K['1,1'].

Degree-1 features live in the basis' own real-SH component order (m = -1, 0, 1 with the polar axis on input component 1,
``tools/basis.py:76``): Y_1(x) = A x / |x| for the constant matrix ``sh1_matrix()``; a Cartesian vector v enters as ``A v``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import sh_oracle as so


def sh1_matrix(dtype=torch.float64) -> torch.Tensor:
    """A [3, 3] with Y_1(x) = A x / |x| (the l = 1 block of the pinned real spherical harmonics)."""
    eye = torch.eye(3, dtype=torch.float64)
    y = so.precompute_sh(so.get_spherical_from_cartesian(eye), 1)[1]          # rows: Y_1(e_k)
    return y.T.contiguous().to(dtype)                                         # column k = Y_1(e_k)


def param_shapes(c0_in, c1_in, c0_out, c1_out, hidden):
    n_w = c0_out * c0_in + c1_out * c0_in + c0_out * c1_in + c1_out * c1_in * 3
    return {"radial.0.weight": (hidden, 1), "radial.0.bias": (hidden,), "radial.2.weight": (n_w, hidden), "radial.2.bias": (n_w,),
            "self0.weight": (c0_out, c0_in), "self1.weight": (c1_out, c1_in)}


def irrep_conv(w, f0, f1, x, edges):
    """w: dict of the tensors of ``param_shapes``; f0 [N, C0i], f1 [N, C1i, 3] (SH component order), x [N, 3], edges = [row, col].
    Returns (out0 [N, C0o], out1 [N, C1o, 3])."""
    row, col = edges[0].long(), edges[1].long()
    c0o, c0i = w["self0.weight"].shape
    c1o, c1i = w["self1.weight"].shape
    r = x[row] - x[col]
    d = r.norm(dim=-1, keepdim=True)
    hid = F.silu(d @ w["radial.0.weight"].T + w["radial.0.bias"])
    R = hid @ w["radial.2.weight"].T + w["radial.2.bias"]
    o = 0
    R00 = R[:, o:o + c0o * c0i].view(-1, c0o, c0i); o += c0o * c0i
    R01 = R[:, o:o + c1o * c0i].view(-1, c1o, c0i); o += c1o * c0i
    R10 = R[:, o:o + c0o * c1i].view(-1, c0o, c1i); o += c0o * c1i
    R11 = R[:, o:o + c1o * c1i * 3].view(-1, c1o, c1i, 3)
    # (differentiable in x: on a self loop the direction handed to the basis is a constant -- the J >= 1 kernels are masked below and
    # the J = 0 kernels do not depend on it -- so that atan2(0, 0) never enters the autograd graph)
    safe = torch.where(d > 0, r, torch.tensor([0.3, 0.4, 0.5], dtype=r.dtype).expand_as(r))
    K = so.get_basis(safe.to(torch.float64), 1)
    k00 = K["0,0"].to(x.dtype).reshape(-1)                      # [E]
    k01 = K["0,1"].to(x.dtype).reshape(-1, 3)                   # [E, 3 (out component)]
    k10 = K["1,0"].to(x.dtype).reshape(-1, 3)                   # [E, 3 (in component)]
    k11 = K["1,1"].to(x.dtype).reshape(-1, 3, 3, 3)             # [E, out, in, J]
    # a self loop (the reference's graphs have them: knn_graph(..., loop=True), train :1005) has r = 0, where Y_J, J >= 1, is whatever
    # atan2(0, 0) makes of it -- a fixed direction, not covariant.  Only the J = 0 kernels (rotation invariant) act on such an edge.
    live = (d > 0).to(x.dtype)                                  # [E, 1]
    k01, k10 = k01 * live, k10 * live
    k11 = torch.cat([k11[..., :1], k11[..., 1:] * live[:, :, None, None]], dim=-1)
    g0, g1 = f0[col], f1[col]                                   # [E, C0i], [E, C1i, 3]
    s1 = torch.einsum("eb,eib->ei", k10, g1)                    # K^{1,0} . f1
    m0 = torch.einsum("eoi,ei->eo", R00, g0) * k00[:, None] + torch.einsum("eoi,ei->eo", R10, s1)
    t1 = torch.einsum("eabj,eib->eiaj", k11, g1)                # [E, C1i, out a, J]
    m1 = torch.einsum("eoi,ei,ea->eoa", R01, g0, k01) + torch.einsum("eoij,eiaj->eoa", R11, t1)
    n = x.shape[0]
    out0 = f0 @ w["self0.weight"].T
    out1 = torch.einsum("oi,nia->noa", w["self1.weight"], f1)
    out0 = out0.index_add(0, row, m0)
    out1 = out1.index_add(0, row, m1)
    return out0, out1


def irrep_egnn(w, h, x, edges, n_layers):
    """DEFINITION of ``irreps.IrrepEGNN`` (the typed network behind the EGNN interface): ``w`` = its ``state_dict`` (fp64):
    ``layers.l.*`` (``param_shapes``), ``gates.l.{weight, bias}``, ``coord.weight [1, c1]``.  f1 starts at zero; SiLU on degree 0 and an
    invariant sigmoid gate per degree-1 channel between the layers; ``x_out = x + sum_i w_x[i] f1[:, i, :]`` after the last."""
    c1 = w["coord.weight"].shape[1]
    f0, f1 = h, torch.zeros(x.shape[0], c1, 3, dtype=h.dtype)
    for l in range(n_layers):
        wl = {k[len(f"layers.{l}."):]: v for k, v in w.items() if k.startswith(f"layers.{l}.")}
        f0, f1 = irrep_conv(wl, f0, f1, x, edges)
        if l + 1 < n_layers:
            f1 = f1 * torch.sigmoid(f0 @ w[f"gates.{l}.weight"].T + w[f"gates.{l}.bias"])[:, :, None]
            f0 = F.silu(f0)
    return f0, x + torch.einsum("i,nia->na", w["coord.weight"][0], f1)
