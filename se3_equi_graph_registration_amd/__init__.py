"""MI355X-native Equi-GSPR registration hot path (k-NN graph -> EGNN message passing ->
weighted-SVD pose head) behind the reference's own Python interface.

Everything computes in hand-written HIP kernels for gfx950 reached through the C-ABI of
``include/equigspr.h`` (``libequigspr.so``); importing the ops without the built library
raises -- there is no CPU fallback.
"""
from ._lib import lib, LIB_PATH, declared_symbols  # noqa: F401
from .ops import DeviceGraph, weighted_kabsch, knn_graph_batched, FlatAdam  # noqa: F401
from .graph import knn_graph, get_edges_batch, get_edges_from_idx, build_pair_graph  # noqa: F401
from .egnn import EGNN, E_GCL  # noqa: F401
from .pose import (CrossAttentionPoseRegression, pose_loss, registration_loss, compute_losses, egnn_equi_loss,  # noqa: F401
                   unsorted_segment_sum, unsorted_segment_mean)
from .synth import synthetic_pair, synthetic_batch  # noqa: F401
from . import pose  # noqa: F401
from .so3 import SO3TensorProductLayer  # noqa: F401
from .correspondences import match_descriptors, build_correspondences  # noqa: F401
from .pointnet import PointNetLayer, PointNet, global_max_pool  # noqa: F401
from .metrics import evaluate_batch, calculate_pose_error, registration_recall  # noqa: F401
from . import sh  # noqa: F401  (A16: spherical harmonics / irrep basis)
from .sh import (get_basis, get_spherical_harmonics, precompute_sh, basis_transformation_Q_J, irr_repr,  # noqa: F401
                 wigner_d_matrix, compose, J_dense)
from .gcn import build_graph, GraphConvolution, GlobalPooling, KnnAdjacency  # noqa: F401
from .irreps import IrrepConv, IrrepEGNN, IrrepGNN, sh1_matrix  # noqa: F401  (N4: typed irrep-basis message passing)
from . import datasets  # noqa: F401  (N3: pair-file formats, dataset classes, HBM-resident batches)
from . import train  # noqa: F401  (train_one_epoch / validate / checkpoints in the reference layout)
from .datasets import ThreeDMatchTrainVal, ThreeDMatchTest, KITTItrainVal, KITTItest, DeviceBatches  # noqa: F401

__all__ = ["EGNN", "E_GCL", "CrossAttentionPoseRegression", "knn_graph", "get_edges_batch", "get_edges_from_idx",
           "build_pair_graph", "DeviceGraph", "weighted_kabsch", "pose_loss", "registration_loss", "compute_losses", "egnn_equi_loss",
           "unsorted_segment_sum", "unsorted_segment_mean", "synthetic_pair", "synthetic_batch", "SO3TensorProductLayer", "FlatAdam", "build_graph", "GraphConvolution",
           "GlobalPooling", "KnnAdjacency", "IrrepConv", "IrrepGNN", "ThreeDMatchTrainVal", "ThreeDMatchTest", "KITTItrainVal", "KITTItest", "DeviceBatches"]
