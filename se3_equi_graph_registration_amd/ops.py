"""Thin wrappers: torch device tensors -> raw pointers -> the C-ABI of libequigspr.so.

PyTorch is plumbing here (device memory, the current HIP stream, autograd bookkeeping);
all arithmetic happens in the HIP kernels.  Every wrapper validates device / dtype /
contiguity and raises ``RuntimeError`` -- the error behaviour the reference gets from ATen.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import EgnnConfig, Graph, KABSCH_AUX


def _need(t: torch.Tensor, name: str, dtype, ndim: Optional[int] = None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the MI355X (cuda/HIP) device, got {t.device}; "
                           "this package has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")
    return t


def _p(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


EDGE_BASIS = {"frames": 0, "irreps": 1}      # EQUIGSPR_BASIS_FRAMES / EQUIGSPR_BASIS_IRREPS


def make_config(in_nf, hidden_nf, out_nf, in_edge_nf, n_layers, num_heads, edge_basis="frames") -> EgnnConfig:
    if edge_basis not in EDGE_BASIS:
        raise ValueError("edge_basis: 'frames' (the reference's frame-vector inputs) or 'irreps' (irrep-basis invariants)")
    cfg = EgnnConfig(int(in_nf), int(hidden_nf), int(out_nf), int(in_edge_nf), int(n_layers), int(num_heads), EDGE_BASIS[edge_basis])
    if _lib.lib().equigspr_egnn_param_count(ctypes.byref(cfg)) < 0:
        raise NotImplementedError(
            f"EGNN config outside the compiled HIP kernel set: hidden_nf={hidden_nf} (16|32|33|48|64), num_heads={num_heads} "
            f"(1|2|4|8; 1 at hidden_nf 33), in/out_nf<=64, in_edge_nf in (0,1), n_layers<=16")
    return cfg


def param_count(cfg: EgnnConfig) -> int:
    return int(_lib.lib().equigspr_egnn_param_count(ctypes.byref(cfg)))


def cat_or_view(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``torch.cat([a, b], dim=0)`` -- without the copy when ``b`` already follows ``a`` in the same storage (e.g. the two
    halves of one stacked ``[2, B, N, C]`` tensor): then the result is a view."""
    if (a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.shape[1:] == b.shape[1:] and not a.requires_grad
            and not b.requires_grad and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel()):
        shape = (a.shape[0] + b.shape[0],) + tuple(a.shape[1:])
        return a.as_strided(shape, a.stride())
    return torch.cat([a, b], dim=0)


# ----------------------------------------------------------------------------------- graph
_KNN_COL_INDEX = {}


_CSR_SCRATCH = {}
_CSR_GRAPH_OWNED = []      # counter arrays adopted by captured hipGraphs (see _csr_scratch)


def reserve_csr_scratch(dev, n_words):
    """Put one more zeroed counter array of ``n_words`` words into the cache for the next hipGraph capture to adopt (a capture
    takes a cached array rather than allocating one it would have to clear at every replay): ``train.GraphedStep`` calls
    this before capturing each further copy of the step."""
    key = (str(torch.device(dev)), -1 - len(_CSR_GRAPH_OWNED) - len(_CSR_SCRATCH), int(n_words))
    _CSR_SCRATCH[key] = torch.zeros(int(n_words), dtype=torch.int32, device=dev)


def _csr_scratch(dev, n_words):
    """The counter array of ``equigspr_build_csr``: the call leaves it all-zero, so one zero-initialised buffer per
    (device, stream, size) is handed from build to build and the per-call clear (two fill launches) disappears.
    Returns ``(tensor, is_zero)``."""
    key = (str(dev), _stream().value, int(n_words))
    buf = _CSR_SCRATCH.get(key)
    if buf is not None:
        return buf, 1
    if torch.cuda.is_current_stream_capturing():
        # The capture stream is a fresh one: take the zeroed buffer an eager warm-up run left behind for this device and size.
        # The captured graph OWNS it from here on (its replays read and re-zero it through the raw pointer): it leaves the
        # eager cache -- a later eager build of the same size, on whatever stream, gets a buffer of its own instead of
        # racing with a replay -- and is kept alive for the life of the process.
        for other_key, other in list(_CSR_SCRATCH.items()):
            if other_key[0] == key[0] and other_key[2] == key[2]:
                del _CSR_SCRATCH[other_key]
                _CSR_GRAPH_OWNED.append(other)
                return other, 1
        return torch.empty(n_words, dtype=torch.int32, device=dev), 0
    if len(_CSR_SCRATCH) > 32:
        _CSR_SCRATCH.clear()
    buf = _CSR_SCRATCH[key] = torch.zeros(n_words, dtype=torch.int32, device=dev)
    return buf, 1


class DeviceGraph:
    """Batched graph in the device format (int32 global ids + inverse indices)."""

    def __init__(self, row32, col32, n_nodes: int, knn_k: Optional[int] = None, need_col_index: bool = True,
                 row_hist: Optional[torch.Tensor] = None, arrival: Optional[torch.Tensor] = None):
        """``row_hist``: the zero-on-exit counter array already holding the row histogram (``knn_graph_batched``);
        ``arrival`` (with it): every edge's slot inside its row segment, as the search's histogram atomics returned it --
        the index fill is then a plain scatter."""
        L = _lib.lib()
        self.row, self.col = _need(row32, "row32", torch.int32, 1), _need(col32, "col32", torch.int32, 1)
        self.n_nodes, self.n_edges = int(n_nodes), int(row32.numel())
        dev = row32.device
        scratch, zeroed = (row_hist, 2) if row_hist is not None else _csr_scratch(dev, self.n_nodes + 1)
        self.rptr = torch.empty(self.n_nodes + 1, dtype=torch.int32, device=dev)
        self.rperm = torch.empty(max(self.n_edges, 1), dtype=torch.int32, device=dev)
        self.rinv = torch.empty(max(self.n_edges, 1), dtype=torch.int32, device=dev)
        try:      # one call: counting sort by row, the inverse permutation from the same pass, scratch handed back zeroed
            _lib.check(L.equigspr_build_csr(_p(self.row), self.n_edges, self.n_nodes, _p(self.rptr), _p(self.rperm),
                                            _p(self.rinv), _p(scratch), zeroed, _p(arrival) if row_hist is not None else None, _stream()),
                       "build_csr(row)")
        except Exception:
            _CSR_SCRATCH.clear()
            raise
        self.cptr = self.cperm = None
        if need_col_index:
            if knn_k is not None:      # centre-major kNN list: the col index is the identity permutation (cached)
                key = (self.n_edges, int(knn_k), str(dev))
                if key not in _KNN_COL_INDEX:
                    if len(_KNN_COL_INDEX) > 16:
                        _KNN_COL_INDEX.clear()
                    _KNN_COL_INDEX[key] = (torch.arange(0, self.n_edges + 1, knn_k, dtype=torch.int32, device=dev),
                                           torch.arange(self.n_edges, dtype=torch.int32, device=dev))
                self.cptr, self.cperm = _KNN_COL_INDEX[key]
            else:
                self.cptr = torch.empty(self.n_nodes + 1, dtype=torch.int32, device=dev)
                self.cperm = torch.empty(max(self.n_edges, 1), dtype=torch.int32, device=dev)
                try:
                    _lib.check(L.equigspr_build_csr(_p(self.col), self.n_edges, self.n_nodes, _p(self.cptr),
                                                    _p(self.cperm), None, _p(scratch), 1, None, _stream()), "build_csr(col)")
                except Exception:
                    _CSR_SCRATCH.clear()
                    raise
        self.struct = Graph(self.n_nodes, self.n_edges, self.row.data_ptr(), self.col.data_ptr(),
                            self.rptr.data_ptr(), self.rperm.data_ptr(),
                            self.cptr.data_ptr() if self.cptr is not None else None,
                            self.cperm.data_ptr() if self.cperm is not None else None, self.rinv.data_ptr())

    @staticmethod
    def from_edge_lists(edge_lists, n_per_cloud: int, need_col_index: bool = True) -> "DeviceGraph":
        """``edge_lists``: iterable of int64 ``[2,E_i]`` tensors with LOCAL ids, one per cloud
        (the reference's per-sample ``edges`` argument); clouds get node offsets i*n_per_cloud."""
        L = _lib.lib()
        edge_lists = list(edge_lists)
        tot = sum(int(e.shape[1]) for e in edge_lists)
        dev = edge_lists[0].device
        row = torch.empty(max(tot, 1), dtype=torch.int32, device=dev)[:tot]
        col = torch.empty(max(tot, 1), dtype=torch.int32, device=dev)[:tot]
        off = 0
        for i, e in enumerate(edge_lists):
            e = _need(e if e.is_contiguous() else e.contiguous(), "edges", torch.int64, 2)
            ne = int(e.shape[1])
            _lib.check(L.equigspr_edges_i64_to_i32(_p(e), ne, i * n_per_cloud, ctypes.c_void_p(row.data_ptr() + 4 * off),
                                                   ctypes.c_void_p(col.data_ptr() + 4 * off), _stream()),
                       "edges_i64_to_i32")
            off += ne
        return DeviceGraph(row, col, len(edge_lists) * n_per_cloud, need_col_index=need_col_index)


def knn_graph_batched(x: torch.Tensor, k: int, loop: bool = True, want_i64: bool = True, want_i32: bool = False,
                      force_slow_path: bool = False, row_hist: Optional[torch.Tensor] = None, arrival: Optional[torch.Tensor] = None):
    """``x [n_clouds, n, 3]`` -> (edges int64 ``[n_clouds,2,n*k]`` | None, row32 | None, col32 | None).
    ``row_hist`` (int32, >= n_clouds*n words, zero on entry): also counts the edges per row node, which is where
    ``equigspr_build_csr`` would otherwise start; ``arrival`` (int32 ``[n_clouds*n*k]``, with ``row_hist``): receives every
    edge's slot inside its row segment (``DeviceGraph(..., row_hist=, arrival=)``)."""
    _need(x, "x", torch.float32, 3)
    nc, n, three = x.shape
    if three != 3:
        raise RuntimeError("knn_graph: x must be [n_clouds, n, 3]")
    dev = x.device
    e64 = torch.empty(nc, 2, n * k, dtype=torch.int64, device=dev) if want_i64 else None
    r32 = torch.empty(nc * n * k, dtype=torch.int32, device=dev) if want_i32 else None
    c32 = torch.empty(nc * n * k, dtype=torch.int32, device=dev) if want_i32 else None
    if arrival is not None and row_hist is None:
        raise RuntimeError("knn_graph: arrival needs row_hist")
    _lib.check(_lib.lib().equigspr_knn_graph(_p(x), nc, n, int(k), int(bool(loop)), _p(e64), _p(r32), _p(c32), _p(row_hist),
                                             _p(arrival), int(bool(force_slow_path)), _stream()), "knn_graph")
    return e64, r32, c32


# ----------------------------------------------------------------------------------- EGNN
def egnn_forward(cfg: EgnnConfig, params: torch.Tensor, graph: DeviceGraph, h: torch.Tensor, x: torch.Tensor,
                 edge_attr: Optional[torch.Tensor], training: bool):
    L = _lib.lib()
    _need(params, "params", torch.float32, 1)
    _need(h, "h", torch.float32, 2)
    _need(x, "x", torch.float32, 2)
    n = graph.n_nodes
    if h.shape != (n, cfg.in_nf) or x.shape != (n, 3):
        raise RuntimeError(f"egnn_forward: h {tuple(h.shape)} / x {tuple(x.shape)} do not match the graph "
                           f"({n} nodes) and in_node_nf={cfg.in_nf}")
    if params.numel() != param_count(cfg):
        raise RuntimeError("egnn_forward: parameter block has the wrong size")
    if edge_attr is not None:
        _need(edge_attr, "edge_attr", torch.float32)
        if cfg.in_edge_nf != 1 or edge_attr.numel() != graph.n_edges:
            raise RuntimeError(f"egnn_forward: edge_attr must be [E,{cfg.in_edge_nf}] with E={graph.n_edges}")
    dev = h.device
    h_out = torch.empty(n, cfg.out_nf, dtype=torch.float32, device=dev)
    x_out = torch.empty(n, 3, dtype=torch.float32, device=dev)
    work = torch.empty(int(L.equigspr_egnn_work_size(ctypes.byref(cfg), n, graph.n_edges)), dtype=torch.float32, device=dev)
    saved = torch.empty(int(L.equigspr_egnn_saved_size(ctypes.byref(cfg), n, graph.n_edges)), dtype=torch.float32, device=dev) if training else None
    _lib.check(L.equigspr_egnn_forward(ctypes.byref(cfg), _p(params), ctypes.byref(graph.struct), _p(h), _p(x),
                                       _p(edge_attr), _p(h_out), _p(x_out), _p(saved), _p(work), _stream()),
               "egnn_forward")
    return h_out, x_out, saved


def egnn_backward(cfg: EgnnConfig, params, graph: DeviceGraph, h_in, edge_attr, saved, grad_h_out, grad_x_out,
                  grad_params, need_input_grads: Tuple[bool, bool]):
    L = _lib.lib()
    n = graph.n_nodes
    if graph.cptr is None:
        raise RuntimeError("egnn_backward: the graph was built without its col index")
    dev = h_in.device
    gh = _need(grad_h_out.contiguous(), "grad_h_out", torch.float32, 2)
    gx = _need(grad_x_out.contiguous(), "grad_x_out", torch.float32, 2)
    _need(grad_params, "grad_params", torch.float32, 1)
    grad_h_in = torch.empty(n, cfg.in_nf, dtype=torch.float32, device=dev) if need_input_grads[0] else None
    grad_x_in = torch.empty(n, 3, dtype=torch.float32, device=dev) if need_input_grads[1] else None
    work = torch.empty(int(L.equigspr_egnn_bwd_work_size(ctypes.byref(cfg), n, graph.n_edges)), dtype=torch.float32, device=dev)
    _lib.check(L.equigspr_egnn_backward(ctypes.byref(cfg), _p(params), ctypes.byref(graph.struct), _p(h_in),
                                        _p(edge_attr), _p(saved), _p(gh), _p(gx), _p(grad_params), _p(grad_h_in),
                                        _p(grad_x_in), _p(work), _stream()), "egnn_backward")
    return grad_h_in, grad_x_in


# ----------------------------------------------------------------------------------- Kabsch
class _KabschFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, tgt, w):
        src = _need(src.contiguous(), "src", torch.float32, 3)
        tgt = _need(tgt.contiguous(), "tgt", torch.float32, 3)
        w = _need(w.contiguous(), "weights", torch.float32, 2)
        b, n, _ = src.shape
        if tgt.shape != src.shape or w.shape != (b, n) or src.shape[2] != 3:
            raise RuntimeError("weighted_kabsch: src/tgt [B,n,3], weights [B,n]")
        R = torch.empty(b, 3, 3, dtype=torch.float32, device=src.device)
        t = torch.empty(b, 3, dtype=torch.float32, device=src.device)
        aux = torch.empty(b, KABSCH_AUX, dtype=torch.float32, device=src.device)
        _lib.check(_lib.lib().equigspr_kabsch_forward(_p(src), _p(tgt), _p(w), b, n, _p(R), _p(t), _p(aux), _stream()),
                   "kabsch_forward")
        ctx.save_for_backward(src, tgt, w, aux)
        return R, t

    @staticmethod
    def backward(ctx, gR, gt):
        src, tgt, w, aux = ctx.saved_tensors
        b, n, _ = src.shape
        gR = torch.zeros(b, 3, 3, dtype=torch.float32, device=src.device) if gR is None else gR.contiguous()
        gt = torch.zeros(b, 3, dtype=torch.float32, device=src.device) if gt is None else gt.contiguous()
        gs, gtg, gw = torch.empty_like(src), torch.empty_like(tgt), torch.empty_like(w)
        _lib.check(_lib.lib().equigspr_kabsch_backward(_p(src), _p(tgt), _p(w), _p(aux), _p(gR), _p(gt), b, n,
                                                       _p(gs), _p(gtg), _p(gw), _stream()), "kabsch_backward")
        return gs, gtg, gw


def weighted_kabsch(src: torch.Tensor, tgt: torch.Tensor, weights: torch.Tensor):
    """Batched weighted Kabsch (reference train :726-758).  ``src,tgt [B,n,3]``, ``weights [B,n]``
    (zero weight = masked out) -> ``R [B,3,3]``, ``t [B,3]``; differentiable."""
    return _KabschFn.apply(src, tgt, weights)


# ----------------------------------------------------------------------------------- fused training head
class _HeadFn(torch.autograd.Function):
    """equigspr_head_forward + equigspr_kabsch_forward, and their backwards, as one autograd node.
    Outputs: R [B,3,3], t [B,3], corr_loss+sim_loss (scalar), equi loss (scalar), logits [B,K], top_idx [B,K], and the training
    loop's step loss (train :1117-1120) evaluated inside the head's launches: loss (scalar, differentiable), rot_loss [B],
    trans_loss [B] (logging values)."""

    @staticmethod
    def forward(ctx, h_all, x_all, raw_hs, raw_ht, labels, gt_pose, topk, flags, *mlp_params):
        # h_all [2*B*N, F], x_all [2*B*N, 3]: the EGNN outputs for [src clouds..., tgt clouds...]; taking the
        # un-sliced tensors keeps four slice-backward (zeros + copy) launches out of the autograd graph
        L = _lib.lib()
        h_all = _need(h_all.contiguous(), "h_all", torch.float32, 2)
        x_all = _need(x_all.contiguous(), "x_all", torch.float32, 2)
        B0, N0 = raw_hs.shape[0], raw_hs.shape[1]
        hs, ht = h_all[:B0 * N0].view(B0, N0, -1), h_all[B0 * N0:].view(B0, N0, -1)
        xs, xt = x_all[:B0 * N0].view(B0, N0, 3), x_all[B0 * N0:].view(B0, N0, 3)
        raw_hs = _need(raw_hs.contiguous().float(), "raw_h_src", torch.float32, 3)
        raw_ht = _need(raw_ht.contiguous().float(), "raw_h_tgt", torch.float32, 3)
        B, N, Fd = hs.shape
        labels = _need(labels.reshape(B, N).contiguous().float(), "labels", torch.float32, 2)
        gt = _need(gt_pose.reshape(B, 16).contiguous().float(), "gt_pose", torch.float32, 2)
        K = int(topk)
        dev = hs.device
        msz = int(L.equigspr_head_mlp_size(Fd))
        # the six MLP tensors are adjacent views of one bucket when dp.FlatBucket owns them; else pack them
        flat = None
        p0 = mlp_params[0]
        off = 0
        adjacent = True
        for q in mlp_params:
            if q.data_ptr() != p0.data_ptr() + 4 * off or not q.is_contiguous():
                adjacent = False
                break
            off += q.numel()
        if adjacent and off == msz:
            flat = p0.detach().reshape(-1).as_strided((msz,), (1,))
        else:
            flat = torch.cat([q.detach().reshape(-1) for q in mlp_params])
        if flat.numel() != msz:
            raise RuntimeError("pose head MLP must be Linear(2F,F)-ReLU-Linear(F,F//2)-ReLU-Linear(F//2,1)")
        f32 = dict(dtype=torch.float32, device=dev)
        sim, rsim, cosv, wts = (torch.empty(B, N, **f32) for _ in range(4))
        pos = torch.empty(B, N, dtype=torch.int32, device=dev)
        top_idx = torch.empty(B, K, dtype=torch.int64, device=dev)
        logits = torch.empty(B, K, **f32)
        a1 = torch.empty(B, K, Fd, **f32)
        a2 = torch.empty(B, K, Fd // 2, **f32)
        work = torch.empty(int(L.equigspr_head_work_bytes(B, N)) // 8 + 1, dtype=torch.float64, device=dev)
        fin = torch.empty(16, **f32)
        R = torch.empty(B, 3, 3, **f32)
        t = torch.empty(B, 3, **f32)
        aux = torch.empty(B, KABSCH_AUX, **f32)
        step = torch.empty(1 + 2 * B, **f32)
        # the weighted Kabsch of (xs, xt, wts) rides in the head's second launch (weights -> ticket -> moments, solve); the step loss in the third
        _lib.check(L.equigspr_head_forward(_p(hs), _p(ht), _p(raw_hs), _p(raw_ht), _p(xs), _p(xt), _p(labels), _p(gt),
                                           _p(flat), B, N, Fd, raw_hs.shape[2], K, int(flags), _p(sim), _p(rsim), _p(cosv), _p(pos),
                                           _p(wts), _p(top_idx), _p(logits), _p(a1), _p(a2), _p(work), _p(fin),
                                           _p(R), _p(t), _p(aux), _p(step), _stream()), "head_forward")
        ctx.save_for_backward(hs, ht, xs, xt, labels, gt, flat, sim, rsim, cosv, pos, wts, top_idx, logits, a1, a2, fin, aux, work, t)
        ctx.dims = (B, N, Fd, K, msz)
        ctx.flags = int(flags)
        ctx.mlp_shapes = [tuple(q.shape) for q in mlp_params]
        ctx.mlp_params = list(mlp_params)
        rot_l, trans_l = step[1:1 + B], step[1 + B:]
        ctx.mark_non_differentiable(top_idx, logits, rot_l, trans_l)      # the logits' only consumer (BCE) is inside the head
        ctx.set_materialize_grads(False)
        return R, t, fin[0], fin[1], logits, top_idx, step[0], rot_l, trans_l

    @staticmethod
    def backward(ctx, gR, gt_, g_cs, g_eq, g_logits, _g_idx, g_step, _g_rot, _g_trans):
        L = _lib.lib()
        (hs, ht, xs, xt, labels, gt, flat, sim, rsim, cosv, pos, wts, top_idx, logits, a1, a2, fin, aux, work, t_sv) = ctx.saved_tensors
        B, N, Fd, K, msz = ctx.dims
        dev = hs.device
        f32 = dict(dtype=torch.float32, device=dev)
        if (gR is None) != (gt_ is None):               # the C entry takes both or neither (neither = zero)
            gR = torch.zeros(B, 3, 3, **f32) if gR is None else gR
            gt_ = torch.zeros(B, 3, **f32) if gt_ is None else gt_
        gR = None if gR is None else gR.contiguous()
        gt_ = None if gt_ is None else gt_.contiguous()
        g_step = None if g_step is None else g_step.contiguous().float()
        gx_all = torch.empty(2 * B * N, 3, **f32)
        gxs, gxt, gw = gx_all[:B * N], gx_all[B * N:], torch.empty_like(wts)
        # the Kabsch backward (grad_R, grad_t -> coordinate / weight gradients) runs inside head_backward's first launch, beside the MLP
        # backward blocks (ABI 8); equigspr_kabsch_backward stays the stand-alone op's backward
        gh_all = torch.empty(2 * B * N, Fd, **f32)
        ghs, ght = gh_all[:B * N], gh_all[B * N:]
        dinp = torch.empty(B, K, 2 * Fd, **f32)
        slab = torch.empty(int(L.equigspr_head_slab_count(B, K)), msz, **f32)
        # The kernel ACCUMULATES the MLP gradient.  When the parameters' .grad tensors are adjacent views of one flat
        # bucket (dp.FlatBucket) it accumulates straight into that bucket -- no per-tensor autograd adds.
        in_place = _adjacent_grad_slice(ctx.mlp_params, msz)
        gmlp = in_place if in_place is not None else torch.zeros(msz, **f32)
        g_cs = None if g_cs is None else g_cs.contiguous().float()
        g_eq = None if g_eq is None else g_eq.contiguous().float()
        _lib.check(L.equigspr_head_backward(_p(hs), _p(ht), _p(xs), _p(xt), _p(labels), _p(gt), _p(flat), _p(sim), _p(rsim),
                                            _p(cosv), _p(pos), _p(wts), _p(top_idx), _p(logits), _p(a1), _p(a2), _p(fin),
                                            _p(g_cs), _p(g_eq), _p(gw), _p(aux), _p(gR), _p(gt_), _p(t_sv), _p(g_step), B, N, Fd, K, ctx.flags, _p(work), _p(dinp), _p(slab), _p(gmlp),
                                            _p(ghs), _p(ght), _p(gxs), _p(gxt), _stream()), "head_backward")
        if in_place is not None:
            return (gh_all, gx_all, None, None, None, None, None, None, *([None] * len(ctx.mlp_shapes)))
        grads, off = [], 0
        for shp in ctx.mlp_shapes:
            n = 1
            for d in shp:
                n *= d
            grads.append(gmlp[off:off + n].view(shp))
            off += n
        return (gh_all, gx_all, None, None, None, None, None, None, *grads)


def _adjacent_grad_slice(params, total):
    """The flat fp32 tensor covering ``[p.grad for p in params]`` if those are adjacent contiguous views, else None."""
    g0 = params[0].grad
    if g0 is None:
        return None
    off = 0
    for q in params:
        g = q.grad
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() != g0.data_ptr() + 4 * off:
            return None
        off += q.numel()
    if off != total:
        return None
    return g0.reshape(-1).as_strided((total,), (1,))


HEAD_NO_SIM_LOSS = 1      # EQUIGSPR_HEAD_NO_SIM_LOSS


def head_fuses_mlp(B: int, N: int, F: int, K: int) -> bool:
    """Whether the training head runs the MLP blocks of this shape inside its pair launch (whole fused grid resident on the
    current device) or as a launch of their own."""
    return bool(_lib.lib().equigspr_head_fuses_mlp(int(B), int(N), int(F), int(K)))


def head_wait_failures() -> int:
    """Bounded in-launch waits of the fused head that ever ran out on the current device (synchronises; 0 on a healthy device).
    The training loops check it when an epoch ends: such an event poisons the loss with NaN and must not pass for a genuinely
    non-finite step."""
    n = int(_lib.lib().equigspr_head_wait_failures())
    if n < 0:
        raise RuntimeError("equigspr_head_wait_failures: could not read the device counter")
    return n


def fused_pose_head(h_all, x_all, raw_hs, raw_ht, labels, gt_pose, topk, mlp_params, flags=0):
    """``h_all [2BN,F]``, ``x_all [2BN,3]`` = EGNN outputs for [src clouds..., tgt clouds...].  ``flags``: HEAD_NO_SIM_LOSS for
    the KITTI script's forward (corr_loss alone)."""
    return _HeadFn.apply(h_all, x_all, raw_hs, raw_ht, labels, gt_pose, topk, flags, *mlp_params)


def _flat_mlp(mlp_params, msz):
    """The pose-head MLP parameters as ONE flat fp32 tensor (a view when they are adjacent slices of a flat bucket)."""
    p0 = mlp_params[0]
    off, adjacent = 0, True
    for q in mlp_params:
        if q.data_ptr() != p0.data_ptr() + 4 * off or not q.is_contiguous():
            adjacent = False
            break
        off += q.numel()
    if adjacent and off == msz:
        return p0.detach().reshape(-1).as_strided((msz,), (1,))
    flat = torch.cat([q.detach().reshape(-1) for q in mlp_params])
    if flat.numel() != msz:
        raise RuntimeError("pose head MLP must be Linear(2F,F)-ReLU-Linear(F,F//2)-ReLU-Linear(F//2,1)")
    return flat


@torch.no_grad()
def fused_eval_head(h_all, x_all, raw_hs, raw_ht, raw_xs, raw_xt, labels, gt_pose, topk, mlp_params):
    """The evaluation pair model after the EGNN (eval :687-818) in three HIP launches (csrc/head.hip, equigspr_head_eval_forward).
    ``h_all [2BN,F]``, ``x_all [2BN,3]`` = EGNN outputs for [src clouds..., tgt clouds...]; ``raw_*`` = the model inputs.
    Returns ``(R [B,3,3], t [B,3], equi_loss, weights [B,N], top_idx [B,K], p0)``.  Inference only, like the reference's use."""
    L = _lib.lib()
    B, N, Fr = raw_hs.shape
    Fd = h_all.shape[1]
    dev = h_all.device
    hs, ht = h_all[:B * N], h_all[B * N:]
    xs, xt = x_all[:B * N], x_all[B * N:]
    for name, t_ in (("h", h_all), ("x", x_all)):
        _need(t_, name, torch.float32, 2)
    raw_hs, raw_ht = _need(raw_hs.contiguous(), "h_src", torch.float32, 3), _need(raw_ht.contiguous(), "h_tgt", torch.float32, 3)
    raw_xs, raw_xt = _need(raw_xs.contiguous(), "x_src", torch.float32, 3), _need(raw_xt.contiguous(), "x_tgt", torch.float32, 3)
    labels = _need(labels.reshape(B, N).contiguous().float(), "labels", torch.float32, 2)
    gt = _need(gt_pose.reshape(B, 16).contiguous().float(), "gt_pose", torch.float32, 2)
    K = int(topk)
    msz = int(L.equigspr_head_mlp_size(Fd))
    flat = _flat_mlp(mlp_params, msz)
    f32 = dict(dtype=torch.float32, device=dev)
    sim, sim0, cosv, wts = (torch.empty(B, N, **f32) for _ in range(4))
    pos = torch.empty(B, N, dtype=torch.int32, device=dev)
    top_idx = torch.empty(B, K, dtype=torch.int64, device=dev)
    work = torch.empty(int(L.equigspr_head_work_bytes(B, N)) // 8 + 1, dtype=torch.float64, device=dev)
    fin = torch.empty(16, **f32)
    R, t = torch.empty(B, 3, 3, **f32), torch.empty(B, 3, **f32)
    aux = torch.empty(B, KABSCH_AUX, **f32)
    _lib.check(L.equigspr_head_eval_forward(_p(hs), _p(ht), _p(raw_hs), _p(raw_ht), _p(xs), _p(xt), _p(raw_xs), _p(raw_xt), _p(labels),
                                            _p(gt), _p(flat), B, N, Fd, Fr, K, _p(sim), _p(sim0), _p(cosv), _p(pos), _p(wts), _p(top_idx),
                                            _p(work), _p(fin), _p(R), _p(t), _p(aux), _stream()), "head_eval_forward")
    return R, t, fin[1], wts, top_idx, fin[10]


# ----------------------------------------------------------------------------------- pose loss / Adam
class _PoseLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R, t, gt_pose):
        R = _need(R.contiguous(), "pred_rot", torch.float32, 3)
        t = _need(t.contiguous(), "pred_translation", torch.float32, 2)
        B = R.shape[0]
        gt = _need(gt_pose.reshape(B, 16).contiguous().float(), "gt_pose", torch.float32, 2)
        rot = torch.empty(B, dtype=torch.float32, device=R.device)
        tr = torch.empty(B, dtype=torch.float32, device=R.device)
        _lib.check(_lib.lib().equigspr_pose_loss_forward(_p(R), _p(t), _p(gt), B, _p(rot), _p(tr), _stream()), "pose_loss_forward")
        ctx.save_for_backward(R, t, gt)
        ctx.set_materialize_grads(False)
        return rot, tr

    @staticmethod
    def backward(ctx, g_rot, g_tr):
        R, t, gt = ctx.saved_tensors
        B = R.shape[0]
        gR, gt_ = torch.empty_like(R), torch.empty_like(t)
        g_rot = None if g_rot is None else g_rot.contiguous().float()
        g_tr = None if g_tr is None else g_tr.contiguous().float()
        _lib.check(_lib.lib().equigspr_pose_loss_backward(_p(R), _p(t), _p(gt), _p(g_rot), _p(g_tr), B, _p(gR), _p(gt_),
                                                          _stream()), "pose_loss_backward")
        return gR, gt_, None


def fused_pose_loss(pred_rot, pred_translation, gt_pose):
    return _PoseLossFn.apply(pred_rot, pred_translation, gt_pose)


class _StepLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R, t, corr_loss, gt_pose):
        R = _need(R.contiguous(), "pred_rot", torch.float32, 3)
        t = _need(t.contiguous(), "pred_translation", torch.float32, 2)
        B = R.shape[0]
        gt = _need(gt_pose.reshape(B, 16).contiguous().float(), "gt_pose", torch.float32, 2)
        if corr_loss is not None:
            corr_loss = _need(corr_loss.reshape(1).float(), "corr_loss", torch.float32, 1)
        out = torch.empty(1 + 2 * B, dtype=torch.float32, device=R.device)
        _lib.check(_lib.lib().equigspr_step_loss_forward(_p(R), _p(t), _p(gt), _p(corr_loss), B, _p(out), _p(out[1:]), _p(out[1 + B:]),
                                                         _stream()), "step_loss_forward")
        ctx.save_for_backward(R, t, gt)
        ctx.has_corr = corr_loss is not None
        ctx.set_materialize_grads(False)              # no zero-filled grads for the two logging outputs
        rot, tr = out[1:1 + B], out[1 + B:]
        ctx.mark_non_differentiable(rot, tr)
        return out[0], rot, tr

    @staticmethod
    def backward(ctx, g, _g_rot, _g_tr):
        R, t, gt = ctx.saved_tensors
        B = R.shape[0]
        gR, gt_ = torch.empty_like(R), torch.empty_like(t)
        g = None if g is None else g.reshape(1).contiguous().float()
        _lib.check(_lib.lib().equigspr_step_loss_backward(_p(R), _p(t), _p(gt), _p(g), B, _p(gR), _p(gt_), _stream()), "step_loss_backward")
        return gR, gt_, (g.reshape(()) if (ctx.has_corr and g is not None) else None), None


def fused_step_loss(pred_rot, pred_translation, corr_loss, gt_pose):
    """``corr_loss.mean() + rot_loss.mean() + trans_loss.mean()`` of the training loop (train :1120) in one launch.
    Returns ``(loss, rot_loss [B], trans_loss [B])``; the per-pair losses are for logging (not differentiable)."""
    return _StepLossFn.apply(pred_rot, pred_translation, corr_loss, gt_pose)


class FlatAdam:
    """torch.optim.Adam(lr, betas, eps) semantics over flat fp32 buckets, one HIP launch per step
    (reference optimizer: train :1412).  ``buckets`` = list of (param_flat, grad_flat) device tensors."""

    def __init__(self, buckets, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.buckets = [(p, g) for p, g in buckets if p is not None]
        if not 1 <= len(self.buckets) <= 4:
            raise ValueError("FlatAdam takes 1..4 buckets")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = [torch.zeros_like(p) for p, _ in self.buckets]
        self.exp_avg_sq = [torch.zeros_like(p) for p, _ in self.buckets]
        self.step_dev = torch.zeros(2, dtype=torch.float32, device=self.buckets[0][0].device)
        n = len(self.buckets)
        arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        self._p, self._g = arr([p for p, _ in self.buckets]), arr([g for _, g in self.buckets])
        self._m, self._v = arr(self.exp_avg), arr(self.exp_avg_sq)
        self._n = (ctypes.c_int64 * n)(*[p.numel() for p, _ in self.buckets])

    def step(self, zero_grad: bool = False):
        """One Adam update; ``zero_grad=True`` also clears the gradient buckets in the same launch (the fused form of
        ``optimizer.step(); optimizer.zero_grad()``)."""
        _lib.check(_lib.lib().equigspr_adam_step(len(self.buckets), self._p, self._g, self._m, self._v, self._n,
                                                 _p(self.step_dev), self.lr, self.betas[0], self.betas[1], self.eps,
                                                 int(bool(zero_grad)), _stream()), "adam_step")

    # ---- the guarded step of the graphed training loop (train.GraphedStep)
    def _guard_buffers(self):
        if getattr(self, "hyper_dev", None) is None:
            dev = self.buckets[0][0].device
            self.hyper_dev = torch.zeros(8, dtype=torch.float32, device=dev)
            self.stats_dev = torch.zeros(8, dtype=torch.float32, device=dev)
            self.progress_host = torch.zeros(4, dtype=torch.float32).pin_memory()      # [0] = stats_dev[5], written by the kernel
            self.progress = self.progress_host.numpy()
            self._hyper_host = None

    def push_hyper(self, loss_weight: float = 1.0):
        """Device copy of (lr, beta1, beta2, eps, loss weight) for ``step_guarded``; re-sent only when a value changed, so a
        scheduler's ``param_groups`` edit reaches a captured graph at its next replay (stream-ordered copy)."""
        self._guard_buffers()
        vals = (self.lr, self.betas[0], self.betas[1], self.eps, float(loss_weight))
        if vals != self._hyper_host:
            self.hyper_dev[:5].copy_(torch.tensor(vals, dtype=torch.float32), non_blocking=False)
            self._hyper_host = vals

    def step_guarded(self, loss: Optional[torch.Tensor] = None, extra: Optional[torch.Tensor] = None):
        """``step(zero_grad=True)`` unless the loss, a gradient or a word of ``extra`` is not finite -- then nothing is updated and
        the gradients are cleared; the verdict is taken inside the launch (``equigspr_adam_step_guarded``) and counted in
        ``stats_dev``: no host round trip, hipGraph-capturable.  Call ``push_hyper`` first (outside any capture)."""
        self._guard_buffers()
        if self._hyper_host is None:
            raise RuntimeError("FlatAdam.step_guarded: call push_hyper() first")
        if loss is not None:
            _need(loss.reshape(1), "loss", torch.float32, 1)
        n_extra = 0 if extra is None else int(_need(extra, "extra", torch.float32, 1).numel())
        _lib.check(_lib.lib().equigspr_adam_step_guarded(len(self.buckets), self._p, self._g, self._m, self._v, self._n,
                                                         _p(self.step_dev), _p(self.hyper_dev), _p(loss), _p(self.stats_dev),
                                                         _p(extra), n_extra, ctypes.c_void_p(self.progress_host.data_ptr()), _stream()),
                   "adam_step_guarded")

    def guard_stats(self):
        """Host copy (synchronises) of the guarded steps' running totals: ``loss_sum`` (weighted), ``applied``, ``skipped``,
        ``first_skipped`` (0-based index among the guarded calls, ``None`` if none), ``last_loss``, ``calls``."""
        self._guard_buffers()
        v = self.stats_dev.tolist()
        return {"loss_sum": v[0], "applied": int(v[1]), "skipped": int(v[2]), "first_skipped": int(v[3]) - 1 if v[3] > 0 else None,
                "last_loss": v[4], "calls": int(v[5]), "void": int(v[6])}

    def reset_state(self):
        for t in self.exp_avg + self.exp_avg_sq + [self.step_dev]:
            t.zero_()

    # ---- torch.optim.Adam-compatible state (the reference's checkpoints store ``optimizer.state_dict()``, train :1330-1336)
    def _locate(self, p):
        """(bucket index, offset) of parameter ``p``, which must be a view into one of the flat parameter buckets."""
        for k, (flat, _) in enumerate(self.buckets):
            off = p.data_ptr() - flat.data_ptr()
            if p.device == flat.device and 0 <= off and off + p.numel() * 4 <= flat.numel() * 4 and off % 4 == 0:
                return k, off // 4
        raise ValueError("parameter is not a view into this optimizer's flat buckets")

    def state_dict(self, params):
        """``torch.optim.Adam(params).state_dict()`` layout for ``params`` (an ordered list of views into the buckets):
        per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` and one param group.  A parameter whose moments are still
        exactly zero after at least one step never received a gradient and gets no entry -- as in torch."""
        params = list(params)
        step = float(self.step_dev[0].item())
        state = {}
        for i, p in enumerate(params):
            k, off = self._locate(p)
            m = self.exp_avg[k][off:off + p.numel()].view(p.shape)
            v = self.exp_avg_sq[k][off:off + p.numel()].view(p.shape)
            if step > 0 and bool((m != 0).any() | (v != 0).any()):
                state[i] = {"step": torch.tensor(step), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd, params):
        """Inverse of ``state_dict``; also accepts what ``torch.optim.Adam.state_dict()`` of the reference run holds.  All
        stepped parameters must share one step count (one flat update has one bias correction); weight decay and amsgrad
        are not supported, as the reference does not use them (train :1412)."""
        params = list(params)
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(params):
            raise ValueError(f"optimizer state has {sum(len(g['params']) for g in groups)} parameters in {len(groups)} groups; "
                             f"expected one group of {len(params)}")
        g = groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
            raise ValueError("FlatAdam: weight_decay / amsgrad / maximize are not supported")
        steps = {float(st["step"]) for st in sd["state"].values()}
        if len(steps) > 1:
            raise ValueError(f"FlatAdam keeps one step count; the state holds {sorted(steps)}")
        self.reset_state()
        for i, st in sd["state"].items():
            p = params[g["params"].index(i)] if g["params"] != list(range(len(params))) else params[int(i)]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} vs parameter {tuple(p.shape)}")
            k, off = self._locate(p)
            self.exp_avg[k][off:off + p.numel()].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[k][off:off + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
        self.step_dev[0] = steps.pop() if steps else 0.0
        self.lr, self.betas, self.eps = float(g["lr"]), (float(g["betas"][0]), float(g["betas"][1])), float(g["eps"])
