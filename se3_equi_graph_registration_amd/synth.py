"""Synthetic correspondence clouds with the reference's data contract (SURVEY §8(d), App. C:
datasets/ThreeDMatch.py:403-409 item layout; single-axis rotation augmentation :18-43).
There are no datasets offline, so bench.py / smoke() / tests use these."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def synthetic_pair(seed: int, n: int = 2048, feat: int = 32, scale=(3.0, 3.0, 3.0), inlier: float = 0.5):
    """One pair as fp32 CPU tensors: src/tgt [n,3], feat_src/feat_tgt [n,feat], labels [n],
    gt_pose [4,4], corr [n,2]."""
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


def synthetic_batch(first_seed: int, batch: int, n: int = 2048, feat: int = 32, scale=(3.0, 3.0, 3.0), device="cpu"):
    """Stack ``batch`` pairs (seeds first_seed .. first_seed+batch-1) into [B,...] tensors."""
    ps = [synthetic_pair(first_seed + b, n, feat, scale) for b in range(batch)]
    out = {k: torch.stack([p[k] for p in ps]).to(device) for k in ps[0]}
    # source and target halves of ONE buffer each (a loader's natural layout): the package then concatenates by view
    for a, b in (("src", "tgt"), ("feat_src", "feat_tgt")):
        both = torch.stack([out[a], out[b]])
        out[a], out[b] = both[0], both[1]
    return out
