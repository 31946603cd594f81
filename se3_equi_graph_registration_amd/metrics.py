"""Evaluation metrics on the device: ``tools/evaluation_metrics.py:14-43`` as one launch for a
whole batch of pairs instead of a D2H copy + numpy per pair (eval loop, src/eval_egnn_metrics.py:1245-1296)."""
from __future__ import annotations

import torch

from . import _lib
from .ops import _need, _p, _stream


def evaluate_batch(rot, trans, gt_pose, src_pts, tgt_pts, tau: float = 0.09):
    """``rot [B,3,3]``, ``trans [B,3]``, ``gt_pose [B,4,4]``, ``src_pts/tgt_pts [B,n,3]`` (CUDA fp32) ->
    dict of fp64 ``[B]`` tensors: ``rotation_error`` [deg], ``translation_error`` [cm], ``recall``, ``precision``, ``f1``."""
    rot = _need(rot.contiguous().float(), "rot", torch.float32, 3)
    trans = _need(trans.contiguous().float(), "trans", torch.float32, 2)
    gt = _need(gt_pose.contiguous().float(), "gt_pose", torch.float32, 3)
    src = _need(src_pts.contiguous().float(), "src_pts", torch.float32, 3)
    tgt = _need(tgt_pts.contiguous().float(), "tgt_pts", torch.float32, 3)
    b, n, _ = src.shape
    if rot.shape != (b, 3, 3) or trans.shape != (b, 3) or gt.shape != (b, 4, 4) or tgt.shape != src.shape:
        raise RuntimeError("evaluate_batch: inconsistent shapes")
    out = torch.empty(b, 5, dtype=torch.float64, device=src.device)
    _lib.check(_lib.lib().equigspr_eval_metrics(_p(rot), _p(trans), _p(gt), _p(src), _p(tgt), b, n, float(tau), _p(out), _stream()),
               "eval_metrics")
    return {"rotation_error": out[:, 0], "translation_error": out[:, 1], "recall": out[:, 2], "precision": out[:, 3], "f1": out[:, 4]}


def calculate_pose_error(gt_pose, pred_pose):
    """Reference signature (4x4 poses) -> (rotation error [deg], translation error [cm]) as 0-d tensors."""
    pred = torch.as_tensor(pred_pose, device=gt_pose.device if torch.is_tensor(gt_pose) else "cuda").float()
    gt = torch.as_tensor(gt_pose, device=pred.device).float()
    dummy = torch.zeros(1, 1, 3, device=pred.device)
    m = evaluate_batch(pred[None, :3, :3], pred[None, :3, 3], gt[None], dummy, dummy)
    return m["rotation_error"][0], m["translation_error"][0]


def registration_recall(gt_pose, pred_pose, src_pts, tgt_pts, tau=0.09):
    """Reference signature -> (recall, precision) as 0-d tensors."""
    pred = torch.as_tensor(pred_pose, device=src_pts.device).float()
    gt = torch.as_tensor(gt_pose, device=src_pts.device).float()
    m = evaluate_batch(pred[None, :3, :3], pred[None, :3, 3], gt[None], src_pts[None], tgt_pts[None], tau)
    return m["recall"][0], m["precision"][0]
