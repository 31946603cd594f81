"""Correspondence construction on the device: nearest-neighbour matching of descriptors, optional
mutual check, and ground-truth inlier labels -- what ``data_preprocess/3DMatch_Feature.py:157-196`` (train split) and
``:331-351`` (test split) do in numpy per pair, producing the ``corr`` / ``labels`` entries of the pair format
(SURVEY Appendix C).  The N x M distance matrix is never materialised (csrc/match.hip)."""
from __future__ import annotations

import torch

from . import _lib
from .ops import _need, _p, _stream


def match_descriptors(src_desc, tgt_desc):
    """``argmin_j sqrt(2 - 2 <s_i, t_j> + 1e-6)`` for every source row: ``(idx int64 [N], dist float32 [N])``."""
    s = _need(src_desc.contiguous().float(), "src_desc", torch.float32, 2)
    t = _need(tgt_desc.contiguous().float(), "tgt_desc", torch.float32, 2)
    if s.shape[1] != t.shape[1]:
        raise RuntimeError("match_descriptors: descriptor widths differ")
    idx = torch.empty(s.shape[0], dtype=torch.int64, device=s.device)
    dist = torch.empty(s.shape[0], dtype=torch.float32, device=s.device)
    _lib.check(_lib.lib().equigspr_match_descriptors(_p(s), s.shape[0], _p(t), t.shape[0], s.shape[1], _p(idx), _p(dist), _stream()),
               "match_descriptors")
    return idx, dist


def build_correspondences(src_desc, tgt_desc, src_keypts, tgt_keypts, gt_pose, use_mutual=False, inlier_threshold=0.10):
    """-> ``(corr int64 [C,2], labels int32 [C], source_dis float32 [N])`` with the reference's semantics: one-directional
    nearest neighbours in feature space, or only the mutual ones; a correspondence is an inlier when the ground-truth
    pose maps its source key point within ``inlier_threshold`` of its target key point."""
    source_idx, source_dis = match_descriptors(src_desc, tgt_desc)
    n = source_idx.shape[0]
    ar = torch.arange(n, device=source_idx.device)
    if use_mutual:
        target_idx, _ = match_descriptors(tgt_desc, src_desc)
        mutual = target_idx[source_idx] == ar                      # index bookkeeping: device tensor ops
        corr = torch.stack([ar[mutual], source_idx[mutual]], dim=1)
    else:
        corr = torch.stack([ar, source_idx], dim=1)
    corr = corr.contiguous()
    sp = _need(src_keypts.contiguous().float(), "src_keypts", torch.float32, 2)
    tp = _need(tgt_keypts.contiguous().float(), "tgt_keypts", torch.float32, 2)
    pose = _need(torch.as_tensor(gt_pose, device=sp.device).contiguous().float(), "gt_pose", torch.float32, 2)
    labels = torch.empty(corr.shape[0], dtype=torch.int32, device=sp.device)
    _lib.check(_lib.lib().equigspr_correspondence_labels(_p(sp), _p(tp), _p(corr), corr.shape[0], _p(pose), float(inlier_threshold),
                                                         _p(labels), None, _stream()), "correspondence_labels")
    return corr, labels, source_dis
