"""``TORCH_LIBRARY(equigspr, ...)``: the C-ABI entry points as dispatcher ops.

``load()`` loads ``libequigspr_torch.so`` (built by ``csrc/Makefile`` from ``csrc/torch_ops.cpp``: host-only C++ that allocates
outputs through ATen and calls the same ``equigspr_*`` functions of ``libequigspr.so`` as the ctypes binding, on the current
HIP stream) and returns ``torch.ops.equigspr``:

    ops = torch_library.load()
    edges = ops.knn_graph(x, 20, True)                                   # torch_cluster.knn_graph (train :1005)
    ptr, perm, inv = ops.build_inverse_index(row32, n_nodes)
    h_out, x_out = ops.egnn_forward(flat_params, h, x, row32, col32, None, 32, 32, 4, 1, 0)
    R, t, aux = ops.weighted_kabsch_forward(src, tgt, w)                 # train :726-758
    g_src, g_tgt, g_w = ops.weighted_kabsch_backward(src, tgt, w, aux, grad_R, grad_t)
    h2, x2 = ops.egcl_forward(layer_block, h, x, row32, col32, edge_attr, num_heads, 0)     # one E_GCL (train :280-289), differentiable
    g_layer, g_h, g_x = ops.egcl_backward(layer_block, h, x, row32, col32, edge_attr, grad_h, grad_x, num_heads, 0)
    R, t, equi, weights, top_idx = ops.head_eval_forward(h_all, x_all, raw_hs, raw_ht, raw_xs, raw_xt, labels, gt_pose, mlp_block, 128)  # eval :687-818
    rot_loss, trans_loss = ops.pose_loss_forward(R, t, gt_pose)          # train :896-962, differentiable; pose_loss_backward is its explicit VJP

No CPU kernels are registered: calling an op with CPU tensors raises the dispatcher's "no backend" error."""
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libequigspr_torch.so")
_loaded = False


def load():
    global _loaded
    if not _loaded:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C se3_equi_graph_registration_amd/csrc` "
                               "(or __graft_entry__.build()); there is no fallback")
        torch.ops.load_library(LIB_PATH)
        _loaded = True
    return torch.ops.equigspr
