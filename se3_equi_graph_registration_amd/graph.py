"""Graph construction with the reference call signatures, on the HIP path.

``knn_graph(x, k, loop=True)`` mirrors ``torch_cluster.knn_graph`` as the scripts call it
(src/3dmatch_train_egnn_with_batch.py:1005-1006); ``get_edges_batch`` mirrors train :380-403.
``build_pair_graph`` is the batched MI355X entry the training harness uses: one kNN launch for
all 2B clouds, emitting the device graph directly (no int64 edge lists, no per-sample loop).
"""
from __future__ import annotations

import torch

from . import ops
from .ops import DeviceGraph


def knn_graph(x: torch.Tensor, k: int, batch=None, loop: bool = False, flow: str = "source_to_target",
              force_slow_path: bool = False) -> torch.Tensor:
    """``x [N,3]`` -> LongTensor ``[2, N*k]``: row 0 = neighbour, row 1 = centre, centre-major,
    ascending (distance, index).  NOTE torch_cluster's default is ``loop=False``; the reference
    always passes ``loop=True``.

    ``batch`` (``src/gcnn.py:31`` passes it; train :1473-1474 calls the function on a whole batch): a sorted LongTensor ``[N]``
    assigning every point to a cloud, neighbours are searched inside a point's own cloud only.  Clouds of EQUAL size are one
    launch of the batched kernel (``ops.knn_graph_batched`` over ``[B, n, 3]``); a ragged batch takes one launch per cloud."""
    if flow != "source_to_target":
        raise NotImplementedError("knn_graph: only flow='source_to_target' (the reference default)")
    if x.dim() != 2 or x.shape[1] != 3:
        raise RuntimeError("knn_graph: x must be [N,3]")
    if batch is None:
        e64, _, _ = ops.knn_graph_batched(x.contiguous().float().unsqueeze(0), k, loop, want_i64=True,
                                          force_slow_path=force_slow_path)
        return e64[0]
    if batch.dim() != 1 or batch.numel() != x.shape[0]:
        raise RuntimeError("knn_graph: batch must be a vector with one entry per point")
    if x.shape[0] == 0:
        return torch.empty(2, 0, dtype=torch.int64, device=x.device)
    counts = torch.unique_consecutive(batch, return_counts=True)[1]
    n = int(counts[0])
    if bool((batch[1:] < batch[:-1]).any()):
        raise RuntimeError("knn_graph: batch must be sorted (torch_cluster's contract)")
    if bool((counts != n).any()):
        # ragged batch: one launch per cloud (the batched kernel's clouds share one size); same edges, same order as torch_cluster's
        out, start = [], 0
        for c in counts.tolist():
            if k > (c if loop else c - 1):
                raise RuntimeError(f"knn_graph: k={k} exceeds the {c} candidates of a cloud of this batch")
            e64, _, _ = ops.knn_graph_batched(x[start:start + c].contiguous().float().unsqueeze(0), k, loop, want_i64=True,
                                              force_slow_path=force_slow_path)
            out.append(e64[0] + start)
            start += c
        return torch.cat(out, dim=1)
    nb = int(counts.numel())
    e64, _, _ = ops.knn_graph_batched(x.contiguous().float().view(nb, n, 3), k, loop, want_i64=True, force_slow_path=force_slow_path)
    off = (torch.arange(nb, device=x.device, dtype=torch.int64) * n).view(nb, 1, 1)
    return (e64 + off).permute(1, 0, 2).reshape(2, nb * n * k)


def get_edges_from_idx(graph_idx):
    """train :372-378."""
    return [graph_idx[0], graph_idx[1]]


def get_edges_batch(graph_idx, n_nodes: int, batch_size: int):
    """train :380-403 -> ([row, col], ones[E*batch_size, 1])."""
    row, col = graph_idx[0], graph_idx[1]
    edge_attr = torch.ones(row.numel() * batch_size, 1, device=row.device)
    if batch_size == 1:
        return [row, col], edge_attr
    rows = [row + n_nodes * i for i in range(batch_size)]
    cols = [col + n_nodes * i for i in range(batch_size)]
    return [torch.cat(rows), torch.cat(cols)], edge_attr


def build_pair_graph(xyz_src: torch.Tensor, xyz_tgt: torch.Tensor, k: int, loop: bool = True,
                     need_col_index: bool = True) -> DeviceGraph:
    """``xyz_* [B,N,3]`` -> one DeviceGraph over the 2B clouds ``[src_0..src_B-1, tgt_0..tgt_B-1]``."""
    x = ops.cat_or_view(xyz_src.float(), xyz_tgt.float()).contiguous()     # a view when src/tgt are halves of one tensor
    nc, n, _ = x.shape
    hist, zeroed = ops._csr_scratch(x.device, nc * n + 1)      # the search leaves the row histogram of the index build in it
    if not zeroed:
        hist = None
    # the histogram atomics of the search also hand every edge its slot inside its row segment: the index fill needs no atomics
    arrival = torch.empty(nc * n * k, dtype=torch.int32, device=x.device) if hist is not None else None
    try:
        _, r32, c32 = ops.knn_graph_batched(x, k, loop, want_i64=False, want_i32=True, row_hist=hist, arrival=arrival)
    except Exception:
        ops._CSR_SCRATCH.clear()
        raise
    return DeviceGraph(r32, c32, nc * n, knn_k=k, need_col_index=need_col_index, row_hist=hist, arrival=arrival)
