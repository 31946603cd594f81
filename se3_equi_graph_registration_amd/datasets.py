"""the pair-file formats and the dataset sampling either side of the hot path.

Host-side mirror of the reference's dataset interface (same class names, constructor arguments, returned 7-tuple
``(corr, labels, src_pts, tgt_pts, src_features, tgt_features, gt_pose)`` and the same consumption of the global
``numpy.random`` / ``random`` streams, so a seeded run draws the same samples):

* ``ThreeDMatchTrainVal`` / ``ThreeDMatchTest``   -- reference ``datasets/ThreeDMatch.py:187-412`` / ``:582-767``
* ``KITTItrainVal`` / ``KITTItest``               -- reference ``datasets/KITTI.py:188-396`` / ``:399-561``
* ``load_pair_file`` / ``save_pair_file``         -- the three on-disk layouts (below)
* ``preprocess_pair``                             -- the record the preprocessing writes (``data_preprocess/3DMatch_Feature.py:
  158-214`` train, ``:332-361`` test); descriptor matching and inlier labels run on the GPU (``correspondences.py``)
* ``split_dataset`` / ``save_file_list``          -- ``datasets/split_dataset_train_val.py:13-38``
* ``collate_fn`` / ``get_dataloader``             -- ``datasets/dataloader.py:6-37``
* ``DeviceBatches``                               -- MI355X side: pinned staging + copies on a side stream, one batch ahead

On-disk layouts (all three carry the same seven arrays):

``pickle-tensors``  3DMatch train/val ``train_3dmatch/*.pkl``: a pickled dict of torch tensors with a leading batch axis of 1
                    under ``src_keypts, tgt_keypts, src_features, tgt_features, corr, gt_labels, gt_pose``
                    (``ThreeDMatch.py:229-240``).
``pickle-arrays``   3DMatch test ``test_3dmatch/*.pkl`` and the preprocessing output: a pickled dict of numpy arrays under
                    ``xyz_0, xyz_1, feat_0, feat_1, corr, labels, gt_pose`` (+ ``file_0, file_1``) (``ThreeDMatch.py:622-634``).
``npz``             KITTI ``train_kitti/*.npz`` / ``test_kitti/*.npz``: the same keys as ``pickle-arrays``
                    (``KITTI.py:230-247``, ``:437-452``).  The reference opens these with ``np.load`` and reads the
                    members after closing the file, which works only when the ``.npz``-named file is in fact a pickled
                    dict (``np.load(allow_pickle=True)`` falls back to ``pickle``) and, for the train class (no
                    ``allow_pickle``), not at all; the KITTI classes here take either a real ``.npz`` archive or a
                    pickled dict, told apart by the file's magic bytes.

The sampling itself is numpy on the host, as in the reference: it is a few thousand integers per pair and its random
stream is part of the interface.
"""
import os
import pickle
import random
import threading
import queue

import numpy as np
import torch
import torch.utils.data as data

PAIR_KEYS = ("xyz_0", "xyz_1", "feat_0", "feat_1", "corr", "labels", "gt_pose")
_TENSOR_KEYS = ("src_keypts", "tgt_keypts", "src_features", "tgt_features", "corr", "gt_labels", "gt_pose")


# ------------------------------------------------------------------------------------------------ SE(3) helpers (numpy)
def rotation_matrix(num_axis, augment_rotation):
    """Random rotation about one of / all three axes, angles in [0, 2 pi augment_rotation) (``ThreeDMatch.py:18-43``).
    Draws ``np.random.rand(3)`` and, for one axis, one ``random.choice`` -- the reference's stream consumption."""
    if num_axis not in (0, 1, 3):
        raise AssertionError("num_axis must be 0, 1 or 3")
    if num_axis == 0:
        return np.eye(3)
    ang = np.random.rand(3) * 2 * np.pi * augment_rotation
    c, s = np.cos(ang), np.sin(ang)
    rx = np.array([[1, 0, 0], [0, c[0], -s[0]], [0, s[0], c[0]]])
    ry = np.array([[c[1], 0, s[1]], [0, 1, 0], [-s[1], 0, c[1]]])
    rz = np.array([[c[2], -s[2], 0], [s[2], c[2], 0], [0, 0, 1]])
    if num_axis == 1:
        return random.choice([rx, ry, rz])
    return rx @ ry @ rz


def translation_matrix(augment_translation):
    """[3,1] translation with components in [0, augment_translation) (``ThreeDMatch.py:45-54``)."""
    return (np.random.rand(3) * augment_translation).reshape(3, 1)


def transform(pts, trans):
    """``R @ pts + t`` for [N,3] points and a [4,4] pose, or batched [B,N,3] / [B,4,4] tensors (``ThreeDMatch.py:56-70``)."""
    if pts.ndim == 3:
        return (trans[:, :3, :3] @ pts.permute(0, 2, 1) + trans[:, :3, 3:4]).permute(0, 2, 1)
    return (trans[:3, :3] @ pts.T + trans[:3, 3:4]).T


def decompose_trans(trans):
    if trans.ndim == 3:
        return trans[:, :3, :3], trans[:, :3, 3:4]
    return trans[:3, :3], trans[:3, 3:4]


def integrate_trans(R, t):
    """[4,4] pose from R [3,3] and t [3,1] (numpy or torch; batched torch as in ``ThreeDMatch.py:86-109``)."""
    if R.ndim == 3:
        if not isinstance(R, torch.Tensor):
            raise TypeError("batched integrate_trans takes torch tensors")
        out = torch.eye(4, dtype=R.dtype, device=R.device).repeat(R.shape[0], 1, 1)
        out[:, :3, :3] = R
        out[:, :3, 3:4] = t.reshape(-1, 3, 1)
        return out
    out = torch.eye(4, dtype=R.dtype, device=R.device) if isinstance(R, torch.Tensor) else np.eye(4)
    out[:3, :3] = R
    out[:3, 3:4] = t
    return out


def concatenate(trans1, trans2):
    """``trans1 @ trans2`` through (R, t) (``ThreeDMatch.py:111-125``)."""
    r1, t1 = decompose_trans(trans1)
    r2, t2 = decompose_trans(trans2)
    return integrate_trans(r1 @ r2, r1 @ t2 + t1)


def transform_target_to_source_frame(xyz_source, xyz_target):
    """Shift the target by the source centroid (``ThreeDMatch.py:132-155``: a translation-only world->source transform
    applied in homogeneous coordinates, i.e. float64 ``target - mean(source)``)."""
    centroid = np.mean(xyz_source, axis=0)
    w2s = np.eye(4)
    w2s[:3, 3] = -centroid
    hom = np.hstack((xyz_target, np.ones((xyz_target.shape[0], 1))))
    return (w2s @ hom.T).T[:, :3]


def remap_correspondences(sampled_corr, sample_size=None):
    """Renumber both columns with ONE table in order of first appearance (``ThreeDMatch.py:157-184``)."""
    flat = np.asarray(sampled_corr).reshape(-1)
    uniq, first, inv = np.unique(flat, return_index=True, return_inverse=True)
    rank = np.empty(len(uniq), dtype=np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(len(uniq))
    return rank[inv].reshape(np.asarray(sampled_corr).shape)


# ------------------------------------------------------------------------------------------------------- file formats
def _np(v):
    return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


def load_pair_file(path, layout=None):
    """Read one pair record into a dict with the ``PAIR_KEYS`` arrays (numpy, dtypes as stored).

    ``layout``: ``'pickle-tensors'``, ``'pickle-arrays'``, ``'npz'`` or None (by content: an ``.npz`` archive, else a pickle
    whose keys decide).  The tensor layout's leading batch axis is dropped, as ``ThreeDMatch.py:234-240`` does."""
    if layout is None:
        with open(path, "rb") as f:
            layout = "npz" if f.read(2) == b"PK" else "pickle"
    if layout == "npz":
        with np.load(path, allow_pickle=True) as z:
            rec = {k: z[k] for k in z.files}
    else:
        with open(path, "rb") as f:
            rec = pickle.load(f)
    if not hasattr(rec, "keys"):
        raise ValueError(f"{path}: not a pair record (got {type(rec).__name__})")
    if layout == "pickle-tensors" or (layout == "pickle" and "src_keypts" in rec):
        missing = [k for k in _TENSOR_KEYS if k not in rec]
        if missing:
            raise KeyError(f"{path}: missing {missing}")
        return {dst: _np(rec[src]).squeeze(0) for dst, src in zip(PAIR_KEYS, _TENSOR_KEYS)}
    missing = [k for k in PAIR_KEYS if k not in rec]
    if missing:
        raise KeyError(f"{path}: missing {missing}")
    out = {k: _np(rec[k]) for k in PAIR_KEYS}
    for k in ("file_0", "file_1"):
        if k in rec:
            out[k] = str(_np(rec[k])) if not isinstance(rec[k], str) else rec[k]
    return out


def save_pair_file(path, record, layout="pickle-arrays"):
    """Write a pair record in one of the three layouts (the inverse of ``load_pair_file``)."""
    missing = [k for k in PAIR_KEYS if k not in record]
    if missing:
        raise KeyError(f"record is missing {missing}")
    extra = {k: record[k] for k in ("file_0", "file_1") if k in record}
    if layout == "npz":
        with open(path, "wb") as f:
            np.savez(f, **{k: _np(record[k]) for k in PAIR_KEYS}, **extra)
    elif layout == "pickle-arrays":
        with open(path, "wb") as f:
            pickle.dump({**extra, **{k: _np(record[k]) for k in PAIR_KEYS}}, f)
    elif layout == "pickle-tensors":
        with open(path, "wb") as f:
            pickle.dump({dst: torch.from_numpy(np.ascontiguousarray(_np(record[src]))).unsqueeze(0)
                         for dst, src in zip(_TENSOR_KEYS, PAIR_KEYS)}, f)
    else:
        raise ValueError(f"unknown layout {layout!r}")


def preprocess_pair(src_keypts, tgt_keypts, src_desc, tgt_desc, gt_pose, inlier_threshold=0.10, use_mutual=False,
                    file_0=None, file_1=None, device="cuda"):
    """The record the reference's preprocessing stores for one fragment pair (``3DMatch_Feature.py:158-214`` / ``:332-361``):
    nearest-descriptor correspondences ``argmin sqrt(2 - 2 f_s f_t^T)`` and inlier labels ``|T x_s - x_t| < threshold``.
    Matching and labelling run in the HIP kernels of ``csrc/match.hip``; the arrays come back to the host for the file."""
    from .correspondences import build_correspondences
    dev = torch.device(device)
    f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    corr, labels, _dis = build_correspondences(f32(src_desc), f32(tgt_desc), f32(src_keypts), f32(tgt_keypts), f32(gt_pose),
                                               use_mutual=use_mutual, inlier_threshold=inlier_threshold)
    rec = {"xyz_0": np.asarray(src_keypts, dtype=np.float32), "xyz_1": np.asarray(tgt_keypts, dtype=np.float32),
           "feat_0": np.asarray(src_desc), "feat_1": np.asarray(tgt_desc), "corr": corr.cpu().numpy(),
           "labels": labels.cpu().numpy().astype(np.int32), "gt_pose": np.asarray(gt_pose)}
    if file_0 is not None:
        rec["file_0"] = file_0
    if file_1 is not None:
        rec["file_1"] = file_1
    return rec


# ------------------------------------------------------------------------------------------------------------ sampling
def sample_correspondences(labels, sample_size, pos_fraction):
    """Sorted indices of ``sample_size`` correspondences with at most ``int(sample_size * pos_fraction)`` drawn inliers
    (``ThreeDMatch.py:296-328``; ``pos_fraction`` 0.7 train / 0.9 test).

    Fewer inliers than that: keep them all and fill up with outliers drawn WITH replacement.  Otherwise: that many
    inliers WITHOUT replacement, the rest outliers with replacement.  A pair without outliers that needs a fill raises
    ``ValueError`` like ``np.random.choice`` on an empty set.  One ``np.random.choice`` call per drawn set, in that order."""
    pos = np.where(labels == 1)[0]
    neg = np.where(labels == 0)[0]
    cap = int(sample_size * pos_fraction)
    if len(pos) < cap:
        pos_s = pos
        need = sample_size - len(pos) if len(pos) < sample_size else 0
    else:
        pos_s = np.random.choice(pos, cap, replace=False)
        need = sample_size - cap
    neg_s = np.random.choice(neg, need, replace=True)
    return np.sort(np.concatenate([pos_s, neg_s]))


def _rank_columns(sampled_corr):
    """Per-column renumbering to 0..(#distinct-1) in ascending order of the original index (``ThreeDMatch.py:345-358``)."""
    a = np.unique(sampled_corr[:, 0], return_inverse=True)[1]
    b = np.unique(sampled_corr[:, 1], return_inverse=True)[1]
    return np.stack((a.reshape(-1), b.reshape(-1)), axis=1)


def _unit_rows(f):
    return f / (np.linalg.norm(f, axis=1, keepdims=True) + 1e-6)


def _read_list(root, name):
    with open(os.path.join(root, name), "r") as f:
        return [line.strip() for line in f.readlines()]


class _PairDataset(data.Dataset):
    """Shared body of the four reference dataset classes; subclasses set the class attributes."""
    subdir = ""              # directory under root holding the records
    layout = None            # on-disk layout passed to load_pair_file
    pos_fraction = 0.7       # cap on drawn inliers as a fraction of num_node
    fcgf_first32 = False     # keep the first 32 descriptor channels for 'fcgf' (test / KITTI records)
    return_remapped = True   # first output: per-column renumbered pairs (train) or the stored pairs (test)
    default_descriptor = "fcgf"

    def __init__(self, root, split, descriptor=None, in_dim=6, inlier_threshold=0.10, num_node=2048, use_mutual=True,
                 downsample=0.03, augment_axis=1, augment_rotation=1.4, augment_translation=0.1):
        descriptor = self.default_descriptor if descriptor is None else descriptor
        assert descriptor in ["fpfh", "fcgf"]
        self.root, self.split, self.descriptor, self.in_dim = root, split, descriptor, in_dim
        self.inlier_threshold, self.num_node, self.use_mutual, self.downsample = inlier_threshold, num_node, use_mutual, downsample
        self.augment_axis, self.augment_rotation, self.augment_translation = augment_axis, augment_rotation, augment_translation
        self.synthetic_pose_flag = False
        self.normalize_use = False
        self.file_list = self._files()

    def _files(self):
        if self.split == "train":
            return _read_list(self.root, "train_files.txt")
        if self.split == "val":
            return _read_list(self.root, "val_files.txt")
        raise ValueError(f"Invalid split: {self.split}. Must be 'train' or 'val'.")

    def __len__(self):
        return len(self.file_list)

    def _load(self, index):
        rec = load_pair_file(os.path.join(self.root, self.subdir, self.file_list[index]), self.layout)
        fs, ft = rec["feat_0"], rec["feat_1"]
        if self.fcgf_first32 and self.descriptor == "fcgf":
            fs, ft = fs[:, :32], ft[:, :32]
        if self.descriptor == "fpfh":
            fs, ft = _unit_rows(fs), _unit_rows(ft)
        return rec["xyz_0"], rec["xyz_1"], fs, ft, rec["corr"], rec["labels"], rec["gt_pose"]

    def _augment(self, src):
        """Synthetic-pose mode (``ThreeDMatch.py:379-395``): jitter the sampled source, make the target a random rigid
        copy of it, every correspondence an inlier.  Stream order: rand(n,3), rand(3) [+ random.choice], rand(3), rand(n,3)."""
        n = self.num_node
        src += np.random.rand(n, 3) * 0.005
        aug = integrate_trans(rotation_matrix(self.augment_axis, self.augment_rotation), translation_matrix(self.augment_translation))
        tgt = transform(src, aug)
        tgt += np.random.rand(n, 3) * 0.005
        gt = concatenate(aug, np.eye(4).astype(np.float32))
        ident = np.arange(n, dtype=np.float32).reshape(-1, 1)
        return src, tgt, gt, np.ones(n, dtype=np.float32), np.hstack((ident, ident))

    def __getitem__(self, index):
        src_pts, tgt_pts, fs, ft, corr, labels, gt = self._load(index)
        idx = sample_correspondences(labels, self.num_node, self.pos_fraction)
        s_src, s_fs, s_corr, s_lab = src_pts[idx], fs[idx], corr[idx], labels[idx]
        t_idx = s_corr[:, 1].astype(int)
        s_tgt, s_ft = tgt_pts[t_idx], ft[t_idx]
        out_corr = _rank_columns(s_corr) if self.return_remapped else s_corr
        if self.synthetic_pose_flag and self.return_remapped:
            s_src, s_tgt, gt, s_lab, out_corr = self._augment(s_src)
            s_ft = s_fs
        if self.normalize_use:
            s_tgt = transform_target_to_source_frame(s_src, s_tgt)
            s_src = s_src - np.mean(s_src, axis=0)
        f32 = np.float32
        return (out_corr.astype(f32), s_lab.astype(f32), s_src.astype(f32), s_tgt.astype(f32), s_fs.astype(f32),
                s_ft.astype(f32), gt.astype(f32))


class ThreeDMatchTrainVal(_PairDataset):
    """``datasets/ThreeDMatch.py:187-412``: ``root/{train,val}_files.txt`` name records under ``root/train_3dmatch``."""
    subdir, layout, pos_fraction, fcgf_first32, return_remapped = "train_3dmatch", "pickle-tensors", 0.7, False, True


class _TestSplit(_PairDataset):
    """The test classes ignore ``split`` and read ``root/test_files.txt`` (``ThreeDMatch.py:612-613``)."""

    def __init__(self, root, split, descriptor=None, in_dim=6, inlier_threshold=0.10, num_node=2048, use_mutual=True,
                 downsample=0.03, augment_axis=1, augment_rotation=1.0, augment_translation=0.01):
        super().__init__(root, split, descriptor, in_dim, inlier_threshold, num_node, use_mutual, downsample, augment_axis,
                         augment_rotation, augment_translation)
        self.test_file_list = self.file_list

    def _files(self):
        return _read_list(self.root, "test_files.txt")

    @staticmethod
    def __loadlog__(gtpath):
        """3DMatch ``gt.log`` trajectories: 5-line blocks (header + 4x4) keyed ``'i_j'`` (``ThreeDMatch.py:769-778``)."""
        with open(gtpath) as f:
            content = f.readlines()
        traj = {}
        for i in range(len(content) // 5):
            head = content[i * 5].strip().split()
            rows = " ".join(x.strip() for x in content[i * 5 + 1:i * 5 + 5])
            traj[f"{head[0]}_{head[1]}"] = np.array(rows.split(), dtype=float).reshape(4, 4)
        return traj


class ThreeDMatchTest(_TestSplit):
    """``datasets/ThreeDMatch.py:582-767``: array pickles under ``root/test_3dmatch``; returns the STORED pair indices."""
    subdir, layout, pos_fraction, fcgf_first32, return_remapped = "test_3dmatch", "pickle-arrays", 0.9, True, False


class KITTItrainVal(_PairDataset):
    """``datasets/KITTI.py:188-396``: ``.npz`` records under ``root/train_kitti``."""
    subdir, layout, pos_fraction, fcgf_first32, return_remapped = "train_kitti", None, 0.7, True, True
    default_descriptor = "fpfh"


class KITTItest(_TestSplit):
    """``datasets/KITTI.py:399-561``: ``.npz`` records under ``root/test_kitti``.  Unlike the other three it keeps only the
    INLIER correspondences and draws ``2048`` of them uniformly (without replacement when there are enough) -- ``:487-
    531``; ``num_node`` is not consulted there, and neither is it here."""
    subdir, layout, fcgf_first32, return_remapped = "test_kitti", None, True, False
    default_descriptor = "fpfh"
    sample_num = 2048

    def __getitem__(self, index):
        src_pts, tgt_pts, fs, ft, corr, labels, gt = self._load(index)
        valid = labels.astype(bool)
        t_idx = corr[:, 1].astype(int)
        v_src, v_tgt, v_fs, v_ft = src_pts[valid], tgt_pts[t_idx][valid], fs[valid], ft[t_idx][valid]
        v_lab, v_corr = labels[valid], corr[valid]
        n_valid = v_src.shape[0]
        idx = np.random.choice(n_valid, self.sample_num, replace=n_valid < self.sample_num)
        s_src, s_tgt = v_src[idx], v_tgt[idx]
        if self.normalize_use:   # the reference's branch names undefined variables here (KITTI.py:543-546); same intent as the other classes
            s_tgt = transform_target_to_source_frame(s_src, s_tgt)
            s_src = s_src - np.mean(s_src, axis=0)
        f32 = np.float32
        return (v_corr[idx].astype(f32), v_lab[idx].astype(f32), s_src.astype(f32), s_tgt.astype(f32), v_fs[idx].astype(f32),
                v_ft[idx].astype(f32), gt.astype(f32))


# ------------------------------------------------------------------------------------------------- file lists, loaders
def split_dataset(directory, train_ratio=0.90, seed=42, suffix=".npz"):
    """Seeded shuffle of the records in ``directory`` and a train/val split (``split_dataset_train_val.py:13-31``)."""
    files = [os.path.join(directory, f) for f in os.listdir(directory) if f.endswith(suffix)]
    random.seed(seed)
    random.shuffle(files)
    cut = int(len(files) * train_ratio)
    return files[:cut], files[cut:]


def save_file_list(file_list, output_file):
    """One base name per line (``split_dataset_train_val.py:33-38``)."""
    with open(output_file, "w") as f:
        for name in file_list:
            f.write(os.path.basename(name) + "\n")


def collate_fn(list_data):
    """Clip every pair of the batch to the smallest correspondence count (``dataloader.py:6-28``; items are the 5-tuples
    ``(corr_pos, src_keypts, tgt_keypts, gt_trans, gt_labels)`` of that loader)."""
    min_num = min(len(item[4]) for item in list_data)
    corr, src, tgt, pose = [], [], [], []
    for corr_pos, src_keypts, tgt_keypts, gt_trans, gt_labels in list_data:
        sel = np.random.choice(len(gt_labels), min_num, replace=False)
        corr.append(corr_pos[sel, :][None])
        src.append(src_keypts[sel, :][None])
        tgt.append(tgt_keypts[sel, :][None])
        pose.append(gt_trans[None])
    return tuple(torch.from_numpy(np.concatenate(v, axis=0)) for v in (corr, src, tgt, pose))


def get_dataloader(dataset, batch_size, shuffle=True, num_workers=4, fix_seed=True):
    """``dataloader.py:31-37``."""
    return data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, collate_fn=collate_fn, num_workers=num_workers)


class DeviceBatches:
    """Iterate a loader of 7-tuples with the batches already resident in HBM.

    A worker thread pulls the next host batch and copies it into pinned staging buffers; the copy to the GPU is issued
    on a side stream, one batch ahead of the consumer, and the consumer's stream waits on its event -- the H2D traffic
    (``B x 2048 x (2 x 3 + 2 x 32 + 3)`` floats, 0.6 MB per pair) hides under the previous step's kernels.  Labels, points
    and descriptors arrive in the layout ``CrossAttentionPoseRegression.forward`` takes."""

    def __init__(self, loader, device="cuda", depth=2):
        self.loader, self.device, self.depth = loader, torch.device(device), depth
        if self.device.type != "cuda":
            raise RuntimeError("DeviceBatches stages batches into GPU memory; device must be a cuda (HIP) device")

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        q = queue.Queue(maxsize=self.depth)
        stop = object()
        abandoned = threading.Event()          # the consumer left the loop early: let the producer thread finish

        def put(item):
            while not abandoned.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                for batch in self.loader:
                    # a collate function may hand back None (or None members) for unreadable pairs: passed through, the loops skip them
                    if not put(None if batch is None else tuple(t.pin_memory() if isinstance(t, torch.Tensor) else t for t in batch)):
                        return
            except BaseException as e:  # surfaced in the consumer
                put(e)
            put(stop)

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        copy_stream = torch.cuda.Stream(self.device)

        def upload(batch):
            with torch.cuda.stream(copy_stream):
                dev = None if batch is None else tuple(t.to(self.device, non_blocking=True) if isinstance(t, torch.Tensor) else t for t in batch)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return dev, ev, batch   # keep the pinned source alive until the copy has been waited on

        def take():
            item = q.get()
            if isinstance(item, BaseException):
                raise item
            return item

        try:
            nxt = take()
            pending = upload(nxt) if nxt is not stop else None
            while pending is not None:
                dev, ev, _host = pending
                nxt = take()
                pending = upload(nxt) if nxt is not stop else None
                torch.cuda.current_stream(self.device).wait_event(ev)
                for t in (dev or ()):
                    if isinstance(t, torch.Tensor):
                        t.record_stream(torch.cuda.current_stream(self.device))
                yield dev
        finally:
            abandoned.set()
            th.join()
