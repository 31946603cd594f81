"""ctypes binding of libequigspr.so (the C-ABI declared in include/equigspr.h).

The product path has NO fallback: if the HIP library is missing or fails to load, every
op raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C se3_equi_graph_registration_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import POINTER, Structure, c_char_p, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EQUIGSPR_LIB", os.path.join(_HERE, "libequigspr.so"))   # override: kernel experiments only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "equigspr.h")
KABSCH_AUX = 320


class EgnnConfig(Structure):
    _fields_ = [("in_nf", c_int32), ("hidden_nf", c_int32), ("out_nf", c_int32),
                ("in_edge_nf", c_int32), ("n_layers", c_int32), ("num_heads", c_int32), ("edge_basis", c_int32)]


class Graph(Structure):
    _fields_ = [("n_nodes", c_int32), ("n_edges", c_int32), ("row", c_void_p), ("col", c_void_p),
                ("rptr", c_void_p), ("rperm", c_void_p), ("cptr", c_void_p), ("cperm", c_void_p), ("rinv", c_void_p)]


class IrrepConfig(Structure):
    _fields_ = [("c0_in", c_int32), ("c1_in", c_int32), ("c0_out", c_int32), ("c1_out", c_int32), ("hidden", c_int32),
                ("k00", ctypes.c_float), ("k01", ctypes.c_float), ("k10", ctypes.c_float), ("k11_0", ctypes.c_float),
                ("k11_1", ctypes.c_float), ("k11_2", ctypes.c_float)]


class IrrepGnnConfig(Structure):
    _fields_ = [("in_nf", c_int32), ("hidden_nf", c_int32), ("out_nf", c_int32), ("c1", c_int32), ("n_layers", c_int32),
                ("radial_hidden", c_int32), ("k00", ctypes.c_float), ("k01", ctypes.c_float), ("k10", ctypes.c_float),
                ("k11_0", ctypes.c_float), ("k11_1", ctypes.c_float), ("k11_2", ctypes.c_float)]


_SIGNATURES = {
    "equigspr_abi_version": (ctypes.c_int, []),
    "equigspr_last_error": (c_char_p, []),
    "equigspr_clear_device_error": (ctypes.c_int, []),
    "equigspr_egnn_param_count": (c_int64, [POINTER(EgnnConfig)]),
    "equigspr_egnn_layer_offset": (c_int64, [POINTER(EgnnConfig), c_int32]),
    "equigspr_knn_graph": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "equigspr_build_csr": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "equigspr_invert_perm": (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p]),
    "equigspr_edges_i64_to_i32": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "equigspr_egnn_saved_size": (c_int64, [POINTER(EgnnConfig), c_int32, c_int32]),
    "equigspr_egnn_work_size": (c_int64, [POINTER(EgnnConfig), c_int32, c_int32]),
    "equigspr_egnn_forward": (ctypes.c_int, [POINTER(EgnnConfig), c_void_p, POINTER(Graph), c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "equigspr_egnn_bwd_work_size": (c_int64, [POINTER(EgnnConfig), c_int32, c_int32]),
    "equigspr_egnn_backward": (ctypes.c_int, [POINTER(EgnnConfig), c_void_p, POINTER(Graph), c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p]),
    "equigspr_set_edge_chunk": (ctypes.c_int, [c_int32]),
    "equigspr_set_edge_products": (ctypes.c_int, [c_int32]),
    "equigspr_set_edge_grid_cap": (ctypes.c_int, [c_int32]),
    "equigspr_time_edge_fwd": (ctypes.c_int, [POINTER(EgnnConfig), c_void_p, POINTER(Graph), c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_int32, POINTER(ctypes.c_float), c_void_p]),
    "equigspr_time_edge_bwd": (ctypes.c_int, [POINTER(EgnnConfig), c_void_p, POINTER(Graph), c_void_p, c_void_p, c_void_p,
                                              c_int32, POINTER(ctypes.c_float), c_void_p]),
    "equigspr_head_mlp_size": (c_int64, [c_int32]),
    "equigspr_head_work_bytes": (c_int64, [c_int32, c_int32]),
    "equigspr_head_slab_count": (c_int64, [c_int32, c_int32]),
    "equigspr_head_forward": (ctypes.c_int, [c_void_p] * 9 + [c_int32] * 6 + [c_void_p] * 16),
    "equigspr_head_backward": (ctypes.c_int, [c_void_p] * 25 + [c_int32] * 5 + [c_void_p] * 9),
    "equigspr_head_eval_forward": (ctypes.c_int, [c_void_p] * 11 + [c_int32] * 5 + [c_void_p] * 12),
    "equigspr_head_fuses_mlp": (ctypes.c_int, [c_int32] * 4),
    "equigspr_head_wait_failures": (c_int64, []),
    "equigspr_pose_loss_forward": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "equigspr_pose_loss_backward": (ctypes.c_int, [c_void_p] * 5 + [c_int32] + [c_void_p] * 3),
    "equigspr_adam_step": (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                          POINTER(c_int64), c_void_p] + [ctypes.c_float] * 4 + [c_int32, c_void_p]),
    "equigspr_adam_step_guarded": (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                                  POINTER(c_int64), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "equigspr_so3_tensor_product_forward": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32] + [c_void_p] * 7),
    "equigspr_gcn_build_graph": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32] + [c_void_p] * 11),
    "equigspr_gcn_adjacency_to_dense": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "equigspr_so3_tensor_product_backward_work_size": (c_int64, [c_int32]),
    "equigspr_so3_tensor_product_backward": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "equigspr_gcn_conv_forward": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                                 c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "equigspr_gcn_global_pool": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "equigspr_gcn_conv_backward_weights": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                                          c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "equigspr_gcn_global_pool_backward": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "equigspr_sh_forward": (ctypes.c_int, [c_void_p, c_int32, ctypes.c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "equigspr_sh_basis_forward": (ctypes.c_int, [c_void_p, ctypes.c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "equigspr_irrep_moments_size": (c_int64, [POINTER(IrrepConfig), c_int32, c_int32]),
    "equigspr_irrep_bwd_work_size": (c_int64, [POINTER(IrrepConfig), c_int32, c_int32]),
    "equigspr_irrep_conv_forward": (ctypes.c_int, [POINTER(IrrepConfig)] + [c_void_p] * 6 + [POINTER(Graph)] + [c_void_p] * 6 + [c_int32, c_void_p]),
    "equigspr_irrep_gnn_param_count": (c_int64, [POINTER(IrrepGnnConfig)]),
    "equigspr_irrep_gnn_param_offset": (c_int64, [POINTER(IrrepGnnConfig), c_int32, c_int32]),
    "equigspr_irrep_gnn_saved_size": (c_int64, [POINTER(IrrepGnnConfig), c_int32, c_int32]),
    "equigspr_irrep_gnn_work_size": (c_int64, [POINTER(IrrepGnnConfig), c_int32, c_int32]),
    "equigspr_irrep_gnn_forward": (ctypes.c_int, [POINTER(IrrepGnnConfig), c_void_p, POINTER(Graph)] + [c_void_p] * 5 + [c_int32, c_void_p]),
    "equigspr_irrep_gnn_backward": (ctypes.c_int, [POINTER(IrrepGnnConfig), c_void_p, POINTER(Graph)] + [c_void_p] * 10),
    "equigspr_irrep_conv_backward": (ctypes.c_int, [POINTER(IrrepConfig)] + [c_void_p] * 6 + [POINTER(Graph)] + [c_void_p] * 17),
    "equigspr_eval_metrics": (ctypes.c_int, [c_void_p] * 5 + [c_int32, c_int32, ctypes.c_double, c_void_p, c_void_p]),
    "equigspr_pointnet_layer_forward": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32] + [c_void_p] * 7 + [c_int32, c_void_p, c_void_p, c_void_p]),
    "equigspr_global_max_pool": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "equigspr_pointnet_layer_backward_work_size": (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    "equigspr_pointnet_layer_backward": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "equigspr_global_max_pool_backward": (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "equigspr_step_loss_forward": (ctypes.c_int, [c_void_p] * 4 + [c_int32] + [c_void_p] * 4),
    "equigspr_step_loss_backward": (ctypes.c_int, [c_void_p] * 4 + [c_int32] + [c_void_p] * 3),
    "equigspr_match_descriptors": (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "equigspr_correspondence_labels": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, ctypes.c_float, c_void_p, c_void_p, c_void_p]),
    "equigspr_kabsch_forward": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                               c_void_p, c_void_p]),
    "equigspr_kabsch_backward": (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
}

_lib = None


def declared_symbols():
    """Function names declared in include/equigspr.h (used by the export test)."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(equigspr_[a-z0-9_]+)\s*\(", text)))


def lib():
    """Load (once) and return the shared library; raises loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. "
                "Run `python -c \"import __graft_entry__ as g; g.build()\"` "
                "(there is no CPU fallback in this package).")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if L.equigspr_abi_version() != 14:
            raise RuntimeError("libequigspr.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().equigspr_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
