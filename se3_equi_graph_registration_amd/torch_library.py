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
