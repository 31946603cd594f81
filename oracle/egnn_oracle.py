"""CPU oracle for the Equi-GSPR registration hot path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement of the reference algorithm (PyTorch CPU ops, fp32 or
fp64).  It exists to *check* the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; the product package
(``se3_equi_graph_registration_amd``) never does and fails loudly without its HIP
library.

Pinning status: PINNED.  ``tests/golden/gen_golden.py`` imports the real reference
scripts from /root/reference (stubbed PyG / torch_cluster), runs them on seeded
inputs, and stores inputs + reference outputs in ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function below against those vectors.
The one exception is ``knn_graph``: the reference calls the third-party
``torch_cluster.knn_graph`` (torch-cluster==1.6.3+pt20cu118, environment.yml:158),
whose source is not under /root/reference -> "parity unpinned" for tie ordering at
that boundary; the restatement below is pinned by known-answer tests instead.

Every function cites the reference lines it follows (paths relative to
/root/reference, "train" = src/3dmatch_train_egnn_with_batch.py, "eval" =
src/eval_egnn_metrics.py).

Weights are passed as a plain ``dict`` keyed by the reference ``state_dict`` names
(``embedding_in.weight``, ``gcl_0.edge_mlps.0.0.weight`` ...).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# A1  k-NN graph  (train :1005-1006 call sites; torch_cluster.knn_graph semantics)
# --------------------------------------------------------------------------------------
def knn_graph(x: Tensor, k: int, loop: bool = True) -> Tensor:
    """k nearest neighbours of every point, torch_cluster ``[neighbour; centre]`` layout.

    Exact arithmetic contract shared with the HIP kernel (so index output is bit-exact):
    ``d2 = fl(fl(fl(dx*dx) + fl(dy*dy)) + fl(dz*dz))`` in fp32 with ``dx = fl(xq - xc)``,
    candidates ordered by ``(d2, index)`` lexicographically, ``loop=False`` removes the
    query's own index.  Output: int64 ``[2, N*k]``; row 0 = neighbour, row 1 = centre,
    centre-major, ascending ``(d2, index)``.
    """
    xn = x.detach().cpu().numpy().astype(np.float32)
    n = xn.shape[0]
    if k > (n if loop else n - 1):
        raise ValueError("k exceeds the number of candidate points")
    nbr = np.empty((n, k), dtype=np.int64)
    ar = np.arange(n)
    for q0 in range(0, n, 256):
        q = xn[q0:q0 + 256]
        dx = (q[:, None, 0] - xn[None, :, 0]).astype(np.float32)
        dy = (q[:, None, 1] - xn[None, :, 1]).astype(np.float32)
        dz = (q[:, None, 2] - xn[None, :, 2]).astype(np.float32)
        d2 = ((dx * dx).astype(np.float32) + (dy * dy).astype(np.float32)).astype(np.float32)
        d2 = (d2 + (dz * dz).astype(np.float32)).astype(np.float32)
        if not loop:
            d2[np.arange(q.shape[0]), q0 + np.arange(q.shape[0])] = np.inf
        # stable argsort == lexicographic (d2, index)
        order = np.argsort(d2, axis=1, kind="stable")[:, :k]
        nbr[q0:q0 + 256] = order
    row = torch.from_numpy(nbr.reshape(-1))
    col = torch.from_numpy(np.repeat(ar, k))
    return torch.stack([row, col], dim=0)


def knn_graph_cdist(x: Tensor, k: int, loop: bool = True) -> Tensor:
    """Same neighbour lists through ``cdist`` + ``topk`` (what BASELINE.md timed in place of the
    absent torch_cluster wheel).  Used by bench.py's ``cpu_baseline`` leg: it is the fair CPU cost of the neighbour search,
    whereas ``knn_graph`` above spells out the fp32 arithmetic contract for bit-exact index parity.  The two agree except on
    the order of exactly tied distances (tests/test_oracle_golden.py checks the neighbour SETS)."""
    xf = x.detach().cpu().float()
    n = xf.shape[0]
    d = torch.cdist(xf, xf)
    if not loop:
        d.fill_diagonal_(float("inf"))
    idx = torch.topk(d, k, dim=1, largest=False, sorted=True).indices
    return torch.stack([idx.reshape(-1), torch.arange(n).repeat_interleave(k)], dim=0)


# --------------------------------------------------------------------------------------
# A2  edge helpers (train :372-403)
# --------------------------------------------------------------------------------------
def get_edges_batch(graph_idx: Tensor, n_nodes: int, batch_size: int) -> Tuple[List[Tensor], Tensor]:
    """train :380-403 -- edges [row, col] replicated with node offsets, all-ones edge_attr."""
    row, col = graph_idx[0], graph_idx[1]
    attr = torch.ones(row.numel() * batch_size, 1)
    if batch_size == 1:
        return [row, col], attr
    rows = [row + n_nodes * i for i in range(batch_size)]
    cols = [col + n_nodes * i for i in range(batch_size)]
    return [torch.cat(rows), torch.cat(cols)], attr


def unsorted_segment_sum(data: Tensor, segment_ids: Tensor, num_segments: int) -> Tensor:
    """train :343-348 -- scatter_add_ over expanded index (CPU: ascending source-row order)."""
    out = data.new_zeros((num_segments, data.size(1)))
    out.scatter_add_(0, segment_ids.unsqueeze(-1).expand(-1, data.size(1)), data)
    return out


def unsorted_segment_mean(data: Tensor, segment_ids: Tensor, num_segments: int) -> Tensor:
    """train :351-358."""
    idx = segment_ids.unsqueeze(-1).expand(-1, data.size(1))
    tot = data.new_zeros((num_segments, data.size(1)))
    cnt = data.new_zeros((num_segments, data.size(1)))
    tot.scatter_add_(0, idx, data)
    cnt.scatter_add_(0, idx, torch.ones_like(data))
    return tot / cnt.clamp(min=1)


# --------------------------------------------------------------------------------------
# A4-A6  per-edge geometry (train :128-181, :271-278)
# --------------------------------------------------------------------------------------
def edge_geometry(x: Tensor, row: Tensor, col: Tensor):
    """Returns (coord_diff[E,3], radial[E,1], dist[E,1], dot[E,1], so3_flat[E,9])."""
    xr, xc = x[row], x[col]
    diff = xr - xc                                            # :273 / :139 / :178
    radial = torch.sum(diff ** 2, -1).unsqueeze(-1)           # :274
    dist = torch.norm(diff, dim=1, keepdim=True)              # :179
    dot = (xr * xc).sum(dim=1, keepdim=True)                  # :180
    a = diff / (diff.norm(dim=1, keepdim=True) + 1e-8)        # :140
    p = torch.cross(xr, xc, dim=1)                            # :143
    b = p / (p.norm(dim=1, keepdim=True) + 1e-8)              # :144
    c = torch.cross(a, b, dim=1)                              # :149
    bad = (a.norm(dim=1) < 1e-6) | (b.norm(dim=1) < 1e-6) | (c.norm(dim=1) < 1e-6)   # :152-156
    frame = torch.stack([a, b, c], dim=2)                     # :159  (columns a,b,c)
    eye = torch.eye(3, dtype=x.dtype).unsqueeze(0).expand(frame.shape[0], -1, -1)
    frame = torch.where(bad[:, None, None], eye, frame)       # :163 (masked assignment)
    return diff, radial, dist, dot, frame.reshape(-1, 9)      # :165


# --------------------------------------------------------------------------------------
# N4: irrep-basis edge features.  No reference layer uses tools/basis.py, so there is nothing to import: this is
# the DEFINITION the HIP kernels (csrc/egnn_math.h: edge_geometry_irreps) are checked against, written through the A16
# restatement (oracle/sh_oracle.py::get_basis, itself pinned by the reference's SH vectors and basis properties).
# --------------------------------------------------------------------------------------
_BASIS_CACHE: Dict[str, object] = {}


def edge_geometry_irreps(x: Tensor, row: Tensor, col: Tensor):
    """Returns (coord_diff [E,3], geo [E,12]) with geo = [ |d|^2, |d|, x_r^T K_0 x_c, K^(1->0) x_r, K^(1->0) x_c,
    x_r^T K_2 x_c, x_r^T K_2 x_r, x_c^T K_2 x_c, x_r^T K_0 x_r, x_c^T K_0 x_c, 0, 0 ], where K^(d_in->d_out)_J are the degree <= 1
    kernels of ``get_basis(d, max_degree=1)`` (tools/basis.py:153-205) at the edge direction d = x_r - x_c.  All rotation
    invariants.  Zero-length edges (self loops) have no direction: the direction-dependent part of K is dropped (dh = 0)."""
    from . import sh_oracle as so
    xr, xc = x[row], x[col]
    diff = xr - xc
    radial = torch.sum(diff ** 2, -1)
    dist = torch.norm(diff, dim=1)
    nz = dist > 0
    cols = [radial, dist]
    u, v = xr[nz], xc[nz]
    B = so.get_basis(diff[nz].detach() if not diff.requires_grad else diff[nz], 1)
    k10 = B["1,0"][:, 0, 0, 0, :, 0]                           # [n, 3]: d_in = 1 -> d_out = 0
    k11 = B["1,1"][:, 0, :, 0, :, :]                           # [n, out 3, in 3, J 3]
    con = lambda a, j, b: torch.einsum("ni,nij,nj->n", a, k11[..., j], b)  # noqa: E731
    gen = [con(u, 0, v), (k10 * u).sum(-1), (k10 * v).sum(-1), con(u, 2, v), con(u, 2, u), con(v, 2, v), con(u, 0, u), con(v, 0, v)]
    # constants of the direction-free part, read off the basis itself (K_0 = c0 I, K_2 = c2 (dh dh^T - I/3))
    if "c" not in _BASIS_CACHE:
        e = torch.tensor([[0.0, 0.0, 1.0]], dtype=torch.float64)
        Be = so.get_basis(e, 1)["1,1"][0, 0, :, 0, :, :]
        _BASIS_CACHE["c"] = (float(Be[0, 0, 0]), float(Be[2, 2, 2]) * 1.5)      # c0 ; c2 (K_2[z][z] = c2 * 2/3)
    c0, c2 = _BASIS_CACHE["c"]
    uz, vz = xr[~nz], xc[~nz]
    p, a2, b2 = (uz * vz).sum(-1), (uz * uz).sum(-1), (vz * vz).sum(-1)
    zero = torch.zeros_like(p)
    deg = [c0 * p, zero, zero, -c2 * p / 3, -c2 * a2 / 3, -c2 * b2 / 3, c0 * a2, c0 * b2]
    geo = x.new_zeros(row.shape[0], 12)
    geo = geo.index_put((torch.arange(row.shape[0]),), torch.stack(cols + [zero.new_zeros(row.shape[0])] * 10, dim=1))
    body = x.new_zeros(row.shape[0], 8)
    body = body.index_put((torch.nonzero(nz).squeeze(1),), torch.stack(gen, dim=1))
    body = body.index_put((torch.nonzero(~nz).squeeze(1),), torch.stack(deg, dim=1))
    geo = torch.cat([geo[:, :2], body, geo[:, 10:]], dim=1)
    return diff, geo


# --------------------------------------------------------------------------------------
# A7-A9  one E_GCL layer (train :185-289)
# --------------------------------------------------------------------------------------
def _num_heads(w: Dict[str, Tensor], prefix: str) -> int:
    h = 0
    while f"{prefix}edge_mlps.{h}.0.weight" in w:
        h += 1
    return h


def egcl_layer(w: Dict[str, Tensor], prefix: str, h: Tensor, x: Tensor, row: Tensor, col: Tensor,
               edge_attr: Optional[Tensor], return_intermediates: bool = False, edge_basis: str = "frames"):
    """One layer; ``prefix`` is e.g. ``"gcl_0."``.  Returns (h', x').  ``edge_basis="irreps"``: the 12 geometry inputs are the
    irrep-basis invariants of ``edge_geometry_irreps`` (extension N4) instead of the reference's."""
    n = h.shape[0]
    if edge_basis == "irreps":
        diff, geo = edge_geometry_irreps(x, row, col)
        radial, dist, dot, so3 = geo[:, 0:1], geo[:, 1:2], geo[:, 2:3], geo[:, 3:12]
    else:
        diff, radial, dist, dot, so3 = edge_geometry(x, row, col)
    feats = [h[row], h[col], radial, dist, dot, so3]          # :238
    if edge_attr is not None:
        feats.append(edge_attr)                               # :240-241
    z = torch.cat(feats, dim=1)                               # :242
    heads = []
    for j in range(_num_heads(w, prefix)):                    # :245
        t = F.linear(z, w[f"{prefix}edge_mlps.{j}.0.weight"], w[f"{prefix}edge_mlps.{j}.0.bias"])
        t = F.silu(t)
        t = F.linear(t, w[f"{prefix}edge_mlps.{j}.2.weight"], w[f"{prefix}edge_mlps.{j}.2.bias"])
        heads.append(t)
    u = torch.cat(heads, dim=1)                               # :246
    m = F.layer_norm(u, (u.shape[1],), w[f"{prefix}layer_norm.weight"], w[f"{prefix}layer_norm.bias"], 1e-5)  # :249
    # coord_model :262-268
    q = F.silu(F.linear(m, w[f"{prefix}coord_mlp.0.weight"], w[f"{prefix}coord_mlp.0.bias"]))
    s = F.linear(q, w[f"{prefix}coord_mlp.2.weight"])
    trans = diff * s
    x_new = x + unsorted_segment_sum(trans, row, n)
    # node_model :252-260
    agg = unsorted_segment_sum(m, row, n)
    v = F.linear(torch.cat([h, agg], dim=1), w[f"{prefix}node_mlp.0.weight"], w[f"{prefix}node_mlp.0.bias"])
    h_new = h + F.linear(F.silu(v), w[f"{prefix}node_mlp.2.weight"], w[f"{prefix}node_mlp.2.bias"])
    if return_intermediates:
        return h_new, x_new, dict(so3=so3, radial=radial, dist=dist, dot=dot, edge_feat=m, trans=trans, agg=agg)
    return h_new, x_new


def n_layers_of(w: Dict[str, Tensor]) -> int:
    n = 0
    while f"gcl_{n}.layer_norm.weight" in w:
        n += 1
    return n


def egnn_forward(w: Dict[str, Tensor], h: Tensor, x: Tensor, edges: Sequence[Tensor],
                 edge_attr: Optional[Tensor], return_layers: bool = False, edge_basis: str = "frames"):
    """EGNN.forward (train :328-340).  ``edges`` = [row, col]."""
    row, col = edges[0], edges[1]
    h = F.linear(h, w["embedding_in.weight"], w["embedding_in.bias"])         # :332
    per_layer = []
    for i in range(n_layers_of(w)):                                             # :333-334
        h, x = egcl_layer(w, f"gcl_{i}.", h, x, row, col, edge_attr, edge_basis=edge_basis)
        if return_layers:
            per_layer.append((h, x))
    h = F.linear(h, w["embedding_out.weight"], w["embedding_out.bias"])       # :337
    if return_layers:
        return h, x, per_layer
    return h, x


# --------------------------------------------------------------------------------------
# A12  weighted Kabsch (train :726-758 / eval :786-818)
# --------------------------------------------------------------------------------------
def weighted_kabsch(src: Tensor, tgt: Tensor, wts: Tensor) -> Tuple[Tensor, Tensor]:
    """``src,tgt [n,3]``, ``wts [n]`` (already normalised by the caller).  Returns (R[3,3], t[3])."""
    cs = (wts[:, None] * src).sum(dim=0, keepdim=True)                          # :726
    ct = (wts[:, None] * tgt).sum(dim=0, keepdim=True)                          # :727
    sc, tc = src - cs, tgt - ct                                                 # :729-730
    H = (wts[:, None, None] * sc[:, :, None] @ tc[:, None, :]).sum(dim=0)       # :732
    H = H + 1e-6 * torch.eye(3, dtype=H.dtype)                                  # :738
    U, S, Vt = torch.linalg.svd(H)                                              # :741
    R = Vt.T @ U.T                                                              # :746
    if torch.det(R) < 0:                                                        # :749-751
        Vt = Vt.clone()
        Vt[-1, :] = Vt[-1, :] * -1
        R = Vt.T @ U.T
    t = ct.squeeze() - R @ cs.squeeze()                                         # :754
    return R, t


# --------------------------------------------------------------------------------------
# A13  losses (train :799-962)
# --------------------------------------------------------------------------------------
def egnn_equi_loss(h_src, x_src, h_tgt, x_tgt, R_gt, t_gt, labels):
    """train :860-893."""
    xs = torch.einsum("bij,bnj->bni", R_gt, x_src) + t_gt[:, None, :]
    cham = F.mse_loss(xs, x_tgt, reduction="none").sum(dim=-1)
    rot = (cham * labels).mean()
    feat = F.mse_loss(F.cosine_similarity(h_src, h_tgt, dim=-1), labels.to(h_src.dtype))
    return rot + feat


def pose_loss(pred_rot, pred_t, gt_pose):
    """train :896-962 -- returns (geodesic rotation loss [B], translation-direction loss [B])."""
    gt_t, gt_R = gt_pose[:, :3, 3], gt_pose[:, :3, :3]
    Rd = torch.matmul(pred_rot.transpose(-1, -2), gt_R)
    tr = Rd.diagonal(dim1=-2, dim2=-1).sum(-1)
    rot_loss = torch.arccos(torch.clamp((tr - 1) / 2, min=-1, max=1))           # :952
    cosv = torch.sum(pred_t * gt_t, dim=-1) / (torch.norm(pred_t, dim=-1) * torch.norm(gt_t, dim=-1))
    return rot_loss, torch.arccos(torch.clamp(cosv, min=-1, max=1))             # :960


def compute_losses(rot, translation, h_src, x_src, h_tgt, x_tgt, gt_labels):
    """train :799-858 -- (mean inlier point error, mean inlier feature distance)."""
    xs = torch.matmul(rot, x_src.transpose(1, 2)).transpose(1, 2) + translation.unsqueeze(1)
    pd = torch.norm(xs - x_tgt, dim=-1) * gt_labels
    cnt = torch.clamp(torch.sum(gt_labels, dim=1), min=1)
    point_error = torch.mean(torch.sum(pd, dim=1) / cnt)
    sel = gt_labels == 1
    feature_loss = torch.mean(torch.norm(h_src[sel] - h_tgt[sel], dim=-1))
    return point_error, feature_loss


# --------------------------------------------------------------------------------------
# A10  training-variant pair model (train :634-796)
# --------------------------------------------------------------------------------------
def _corr_mlp(w: Dict[str, Tensor], z: Tensor) -> Tensor:
    """CrossAttentionPoseRegression.mlp (train :594-600): Linear-ReLU-Linear-ReLU-Linear."""
    z = F.relu(F.linear(z, w["mlp.0.weight"], w["mlp.0.bias"]))
    z = F.relu(F.linear(z, w["mlp.2.weight"], w["mlp.2.bias"]))
    return F.linear(z, w["mlp.4.weight"], w["mlp.4.bias"])


def _run_egnn_pairs(we, h_src, x_src, edges_src, ea_src, h_tgt, x_tgt, edges_tgt, ea_tgt):
    hs, xs, ht, xt = [], [], [], []
    for i in range(h_src.shape[0]):                                             # :645-669
        a, b = egnn_forward(we, h_src[i], x_src[i], list(edges_src[i]), ea_src[i])
        c, d = egnn_forward(we, h_tgt[i], x_tgt[i], list(edges_tgt[i]), ea_tgt[i])
        hs.append(a); xs.append(b); ht.append(c); xt.append(d)
    return torch.stack(hs), torch.stack(xs), torch.stack(ht), torch.stack(xt)


def pair_forward_train(we: Dict[str, Tensor], wc: Dict[str, Tensor],
                       h_src, x_src, edges_src, ea_src, h_tgt, x_tgt, edges_tgt, ea_tgt,
                       corr, labels, gt_pose, topk: int = 128, kitti: bool = False):
    """Training head.  Returns the reference 9-tuple plus a dict of extras (logits, top-k idx).
    ``kitti=True``: the forward of src/kitti_train_egnn_with_batch.py:619-766 -- identical except ``top_k = 2048`` (:661, pass
    ``topk=2048``) and no similarity-MSE term: it returns ``corr_loss`` alone (:766)."""
    raw_hs, raw_ht = h_src, h_tgt
    B, N, Fd = h_src.shape
    hs, xs, ht, xt = _run_egnn_pairs(we, h_src, x_src, edges_src, ea_src, h_tgt, x_tgt, edges_tgt, ea_tgt)
    equi = egnn_equi_loss(hs, xs, ht, xt, gt_pose[:, :3, :3], gt_pose[:, :3, -1], labels)     # :677
    sim = torch.sum(hs * ht, dim=-1, keepdim=True)                              # :681
    raw_sim = torch.sum(raw_hs * raw_ht, dim=-1, keepdim=True)                  # :682
    _, top_idx = torch.topk(sim.squeeze(-1), k=topk, dim=-1)                    # :684
    gi = top_idx.unsqueeze(-1).expand(-1, -1, Fd)
    chs, cht = torch.gather(hs, 1, gi), torch.gather(ht, 1, gi)                 # :688-689
    clab = torch.gather(labels, 1, top_idx)                                     # :694
    valid = labels.squeeze(-1).bool()                                           # :696
    R = torch.zeros(B, 3, 3, dtype=xs.dtype)
    t = torch.zeros(B, 3, dtype=xs.dtype)
    for b in range(B):                                                          # :702-758
        vs, vt = xs[b][valid[b]], xt[b][valid[b]]
        if vs.shape[0] == 0:
            R[b] = torch.eye(3, dtype=xs.dtype)
            continue
        sc = torch.sum(hs[b][valid[b]] * ht[b][valid[b]], dim=-1)               # :717
        wts = F.softmax(sc, dim=-1)                                             # :718
        wts = wts / (wts.sum() + 1e-6)                                          # :724
        R[b], t[b] = weighted_kabsch(vs, vt, wts)
    logits = _corr_mlp(wc, torch.cat([chs, cht], dim=-1).view(-1, 2 * Fd)).view(B, topk)     # :760-770
    corr_loss = F.binary_cross_entropy_with_logits(logits, clab)                # :772-773
    sim_n = (sim - sim.mean()) / (sim.std() + 1e-6)                             # :777
    raw_n = (raw_sim - raw_sim.mean()) / (raw_sim.std() + 1e-6)                 # :778
    sim_loss = F.mse_loss(sim_n, raw_n)                                         # :781
    extras = dict(logits=logits, top_idx=top_idx, corr_loss=corr_loss, sim_loss=sim_loss)
    if kitti:
        return (R, t, corr_loss, equi, hs, xs, ht, xt, labels), extras                 # kitti :766
    return (R, t, corr_loss + sim_loss, equi, hs, xs, ht, xt, labels), extras


# --------------------------------------------------------------------------------------
# A11  eval-variant pair model (eval :643-827), B = 1 only
# --------------------------------------------------------------------------------------
def pair_forward_eval(we, wc, h_src, x_src, edges_src, ea_src, h_tgt, x_tgt, edges_tgt, ea_tgt,
                      corr, labels, gt_pose, topk: int = 128):
    """Eval head incl. the broadcasting quirk of eval :742,:758-768 (only ``p[0]`` is used)."""
    assert h_src.shape[0] == 1, "the reference eval head only works for B=1 (eval :719,:773)"
    B, N, Fd = h_src.shape
    hs, xs, ht, xt = _run_egnn_pairs(we, h_src, x_src, edges_src, ea_src, h_tgt, x_tgt, edges_tgt, ea_tgt)
    equi = egnn_equi_loss(hs, xs, ht, xt, gt_pose[:, :3, :3], gt_pose[:, :3, -1], labels)     # eval :687
    sim0 = torch.sum(h_src * h_tgt, dim=-1, keepdim=True)                       # eval :691  [1,N,1]
    _, top_idx = torch.topk(sim0.squeeze(-1), k=topk, dim=-1)                   # eval :694
    gi = top_idx.unsqueeze(-1).expand(-1, -1, Fd)
    p = _corr_mlp(wc, torch.cat([torch.gather(hs, 1, gi), torch.gather(ht, 1, gi)], dim=-1).view(-1, 2 * Fd)).squeeze(-1)  # :742
    s_top = torch.gather(sim0, 1, top_idx.unsqueeze(-1))                        # :758   [1,K,1]
    cond1 = (p > 0.5) & (torch.abs(p - 1) < s_top)                              # :761   -> [1,K,K]
    cond2 = (p > 0.5) & (p < s_top)                                             # :762
    w_top = torch.where(cond1 | cond2, p, s_top)                                # :764   [1,K,K]
    W = sim0.clone()
    W.scatter_(dim=1, index=top_idx.unsqueeze(-1), src=w_top)                   # :768 (consumes column 0)
    W = W / (W.sum(dim=1, keepdim=True) + 1e-6)                                 # :771
    wts = F.softmax(W.squeeze(0).squeeze(-1), dim=-1)                           # :773-774
    wts = wts / (wts.sum() + 1e-6)                                              # :783
    R, t = weighted_kabsch(x_src[0], x_tgt[0], wts)                             # :786-818 raw coords, all N
    extras = dict(weights=wts, top_idx=top_idx, p=p)
    return (R.unsqueeze(0), t.unsqueeze(0), None, equi, hs, xs, ht, xt, labels), extras


# --------------------------------------------------------------------------------------
# N1: PointNet encoder (train :52-96)
# --------------------------------------------------------------------------------------
def pointnet_layer(w: Dict[str, Tensor], h: Tensor, pos: Tensor, edge_index: Tensor) -> Tensor:
    """``PointNetLayer.forward`` :61-68: ``propagate`` with aggr='max', flow source_to_target (j = edge_index[0] sends to
    i = edge_index[1]); message = mlp([h_j, pos_j - pos_i]); nodes without incoming edges get 0 (PyG's scatter-max fill).
    ``w``: ``mlp.0.weight [out,in+3]``, ``mlp.0.bias``, ``mlp.2.weight``, ``mlp.2.bias``."""
    j, i = edge_index[0], edge_index[1]
    msg = torch.cat([h[j], pos[j] - pos[i]], dim=-1)
    msg = torch.relu(msg @ w["mlp.0.weight"].t() + w["mlp.0.bias"]) @ w["mlp.2.weight"].t() + w["mlp.2.bias"]
    out = torch.zeros(h.shape[0], msg.shape[1], dtype=msg.dtype)
    return out.scatter_reduce(0, i[:, None].expand_as(msg), msg, reduce="amax", include_self=False)


def pointnet_forward(w: Dict[str, Tensor], pos: Tensor, feats: Tensor, edge_index: Tensor, batch: Tensor):
    """``PointNet.forward`` :79-96 -> (per-point features before pooling, pooled ``[num_graphs, out]``)."""
    sub = lambda pre: {k[len(pre):]: v for k, v in w.items() if k.startswith(pre)}
    h = torch.relu(pointnet_layer(sub("conv1."), feats, pos, edge_index))
    h = torch.relu(pointnet_layer(sub("conv2."), h, pos, edge_index))
    nb = int(batch.max()) + 1
    pooled = torch.full((nb, h.shape[1]), float("-inf"), dtype=h.dtype).scatter_reduce(0, batch[:, None].expand_as(h), h, reduce="amax")
    return h, pooled


# --------------------------------------------------------------------------------------
# N3: correspondence construction (data_preprocess/3DMatch_Feature.py:157-196; test split :331-351)
# PARITY UNPINNED: the reference code is script-level numpy inside a function that needs open3d and the raw dataset, so it
# cannot be imported for golden vectors; the lines are restated verbatim here.
# --------------------------------------------------------------------------------------
def build_correspondences(src_desc, tgt_desc, src_keypts, tgt_keypts, gt_pose, use_mutual=False, inlier_threshold=0.10):
    import numpy as np
    distance = np.sqrt(2 - 2 * (src_desc @ tgt_desc.T) + 1e-6)                                   # :157
    source_idx = np.argmin(distance, axis=1)
    source_dis = np.min(distance, axis=1)
    if use_mutual:                                                                               # :160-163
        target_idx = np.argmin(distance, axis=0)
        mutual_nearest = (target_idx[source_idx] == np.arange(source_idx.shape[0]))
        corr = np.concatenate([np.where(mutual_nearest == 1)[0][:, None], source_idx[mutual_nearest][:, None]], axis=-1)
    else:
        corr = np.concatenate([np.arange(source_idx.shape[0])[:, None], source_idx[:, None]], axis=-1)
    frag1 = src_keypts[corr[:, 0]]                                                               # :186-190
    frag2 = tgt_keypts[corr[:, 1]]
    frag1_warp = frag1 @ gt_pose[:3, :3].T + gt_pose[:3, 3]                                      # transform(): R @ pts + t
    dist = np.sqrt(np.sum(np.power(frag1_warp - frag2, 2), axis=1))
    labels = (dist < inlier_threshold).astype(np.int32)
    return corr, labels, source_dis, distance


# --------------------------------------------------------------------------------------
# N2: evaluation metrics (tools/evaluation_metrics.py:14-43, eval loop eval :1245-1296)
# --------------------------------------------------------------------------------------
def eval_metrics(gt_pose, pred_pose, src_pts, tgt_pts, tau: float = 0.09):
    """numpy fp64: (rotation error [deg], translation error [cm], recall, precision, f1)."""
    import numpy as np
    gt_pose, pred_pose = np.asarray(gt_pose, dtype=np.float64), np.asarray(pred_pose, dtype=np.float64)
    src_pts, tgt_pts = np.asarray(src_pts, dtype=np.float64), np.asarray(tgt_pts, dtype=np.float64)
    trans_err = np.linalg.norm(gt_pose[:3, 3] - pred_pose[:3, 3]) * 100
    rel = gt_pose[:3, :3].T @ pred_pose[:3, :3]
    rot_err = np.degrees(np.arccos(np.clip((np.trace(rel) - 1) / 2, -1.0, 1.0)))
    moved = (pred_pose[:3, :3] @ src_pts.T).T + pred_pose[:3, 3]
    tp = np.sum(np.linalg.norm(moved - tgt_pts, axis=1) < tau)
    recall = np.sqrt(tp / len(src_pts))
    precision = tp / len(moved) if len(moved) > 0 else 0.0
    return rot_err, trans_err, recall, precision, 2 * (precision * recall) / (precision + recall + 1e-6)


# --------------------------------------------------------------------------------------
# A17: gcnLayer (src/gcnLayer.py)
# --------------------------------------------------------------------------------------
def gcn_build_graph(coordinates, k: int = 25):
    """Dense fp64 restatement of ``build_graph`` :58-67 (``_get_weights`` :28-46, ``get_normalize_adj`` :49-56):
    k nearest points INCLUDING the point itself (cKDTree.query), Gaussian weights with sigma2 = mean(k-th distance)^2,
    zero diagonal, symmetrise by max, D^-1/2 A D^-1/2.  Brute-force distances (N <= a few thousand)."""
    import numpy as np
    x = np.asarray(coordinates, dtype=np.float64)
    n = x.shape[0]
    d = np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(-1))
    order = np.argsort(d, axis=1, kind="stable")[:, :k]
    dist = np.take_along_axis(d, order, axis=1)
    sigma2 = np.mean(dist[:, -1]) ** 2
    w = np.zeros((n, n))
    np.put_along_axis(w, order, np.exp(-dist ** 2 / sigma2), axis=1)
    np.fill_diagonal(w, 0.0)
    w = np.maximum(w, w.T)
    with np.errstate(divide="ignore"):
        d_inv = np.power(w.sum(1), -0.5)
    d_inv[np.isinf(d_inv)] = 0.0
    return (w * d_inv[None, :]).T * d_inv[None, :]


def gcn_graph_convolution(w: Dict[str, Tensor], adj: Tensor, x: Tensor) -> Tensor:
    """``GraphConvolution.forward`` :77-80: ``conv1x1(bmm(x, adj))``; ``w``: ``conv.weight [out,in,1]``, ``conv.bias``."""
    y = torch.einsum("oc,bcn->bon", w["conv.weight"][:, :, 0], torch.bmm(x, adj))
    return y + w["conv.bias"][None, :, None] if "conv.bias" in w else y


def gcn_global_pooling(x: Tensor) -> Tensor:
    """``GlobalPooling.forward`` :90-94."""
    return torch.cat((x.max(dim=2).values, x.mean(dim=2)), dim=-1)


# --------------------------------------------------------------------------------------
# A15: SO3TensorProductLayer (src/clean_so3_model.py:34-52)
# --------------------------------------------------------------------------------------
def so3_tensor_product(w: Dict[str, Tensor], x: Tensor) -> Tensor:
    """``mlp(vec(bmm(x[:, :, None].expand(D), x[:, None, :].expand(D))))`` -- the bmm of the two broadcast
    views sums D identical products, so entry (i, k) is ``D * x_i * x_k`` (clean_so3_model.py:45-52).
    ``w``: state_dict of the layer (``mlp.0.weight [2*out, D*D]``, ``mlp.0.bias``, ``mlp.2.weight``, ``mlp.2.bias``)."""
    d = x.shape[1]
    tp = torch.bmm(x.unsqueeze(2).expand(-1, -1, d), x.unsqueeze(1).expand(-1, d, -1)).reshape(x.shape[0], -1)
    hid = torch.relu(tp @ w["mlp.0.weight"].t() + w["mlp.0.bias"])
    return hid @ w["mlp.2.weight"].t() + w["mlp.2.bias"]


# --------------------------------------------------------------------------------------
# §8(d) synthetic pair generator (mirrors datasets/ThreeDMatch.py:18-43 augmentation shape)
# --------------------------------------------------------------------------------------
def synthetic_pair(seed: int, n: int = 2048, feat: int = 32, scale=(3.0, 3.0, 3.0), inlier: float = 0.5):
    """Deterministic synthetic correspondence cloud.  All fp32 CPU tensors."""
    g = torch.Generator().manual_seed(seed)
    sc = torch.tensor(scale, dtype=torch.float32)
    src = torch.rand(n, 3, generator=g) * sc
    axis = int(torch.randint(0, 3, (1,), generator=g))
    ang = float(torch.rand(1, generator=g)) * 2 * math.pi
    c, s = math.cos(ang), math.sin(ang)
    R = torch.eye(3)
    i, j = [(1, 2), (0, 2), (0, 1)][axis]
    R[i, i], R[i, j], R[j, i], R[j, j] = c, -s, s, c
    tr = torch.rand(3, generator=g) * 0.5
    tgt = src @ R.T + tr + torch.rand(n, 3, generator=g) * 0.005
    labels = (torch.rand(n, generator=g) < inlier).float()
    out_pts = torch.rand(n, 3, generator=g) * sc
    tgt = torch.where(labels[:, None] > 0, tgt, out_pts)
    fs = F.normalize(torch.randn(n, feat, generator=g), dim=1)
    ft_in = F.normalize(fs + 0.05 * torch.randn(n, feat, generator=g), dim=1)
    ft_out = F.normalize(torch.randn(n, feat, generator=g), dim=1)
    ft = torch.where(labels[:, None] > 0, ft_in, ft_out)
    pose = torch.eye(4)
    pose[:3, :3], pose[:3, 3] = R, tr
    corr = torch.stack([torch.arange(n), torch.arange(n)], dim=1).float()
    return dict(src=src, tgt=tgt, feat_src=fs, feat_tgt=ft, labels=labels, gt_pose=pose, corr=corr)
