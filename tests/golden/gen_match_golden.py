#!/usr/bin/env python3
"""Golden vectors for the correspondence construction from the REAL reference script.

`data_preprocess/3DMatch_Feature.py` is a `__main__` script with hard-coded dataset paths and knobs at its top ("change root
dir ...", `make_training_data`, `use_mutual`).  This generator -- run in the BUILD container only, where /root/reference exists --
  1. writes a small synthetic dataset tree in a temp dir in the layouts the script reads (test split: `gt_result/<scene>-evaluation/
     gt.log` + `fragments/<scene>/cloud_bin_<id>_fpfh.npz`; train split: `misc/split/test_3dmatch.txt`, `threedmatch/<name>*0.30.txt`,
     `threedmatch_feat/<id>_fcgf.npz`),
  2. executes the reference script's own text as `__main__` with ONLY its path literals and those two knobs substituted in memory
     (nothing of it is written to this repository; open3d is not needed by this script),
  3. reads the pickles it wrote -- `feat_0/feat_1/xyz_0/xyz_1/gt_pose` (what the matching lines :157-196 / :331-351 consumed) and
     `corr/labels` (what they produced) -- and stores them, plus per-row decision margins, in tests/golden/match_cases.npz.
`tests/test_gpu_parity.py::test_match_kernels_golden` grades csrc/match.hip against it on the GPU box."""
import os
import pickle
import random
import sys
import tempfile

import numpy as np

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REF = "/root/reference"
SCRIPT = os.path.join(REF, "data_preprocess", "3DMatch_Feature.py")
OLD_ROOT = "/media/eavise3d/新加卷/Datasets/eccv-data-0126/3DMatch"
OLD_OUT = OLD_ROOT + "/3DMatch_fcgf_feature_test"
SCENES = ['7-scenes-redkitchen', 'sun3d-home_at-home_at_scan1_2013_jan_1', 'sun3d-home_md-home_md_scan9_2012_sep_30',
          'sun3d-hotel_uc-scan3', 'sun3d-hotel_umd-maryland_hotel1', 'sun3d-hotel_umd-maryland_hotel3',
          'sun3d-mit_76_studyroom-76-1studyroom2', 'sun3d-mit_lab_hj-lab_hj_tea_nov_2_2012_scan1_erika']


def synthetic_fragment_pair(rng, n_src, n_tgt, dim, pose, overlap=0.6, desc_noise=0.08):
    """Two fragments: `overlap` of the target points are noisy images of source points under `pose` (src -> tgt) with noisy copies
    of their descriptors; the rest are unrelated.  Descriptors are left un-normalised (the script normalises FPFH itself)."""
    src_xyz = rng.uniform(-1.5, 1.5, (n_src, 3)).astype(np.float32)
    src_feat = rng.standard_normal((n_src, dim)).astype(np.float32)
    tgt_xyz = rng.uniform(-1.5, 1.5, (n_tgt, 3)).astype(np.float32)
    tgt_feat = rng.standard_normal((n_tgt, dim)).astype(np.float32)
    k = int(overlap * min(n_src, n_tgt))
    pick = rng.permutation(n_src)[:k]
    slot = rng.permutation(n_tgt)[:k]
    tgt_xyz[slot] = (src_xyz[pick] @ pose[:3, :3].T + pose[:3, 3] + rng.normal(0, 0.03, (k, 3))).astype(np.float32)
    tgt_feat[slot] = src_feat[pick] + desc_noise * np.linalg.norm(src_feat[pick], axis=1, keepdims=True) * rng.standard_normal((k, dim)).astype(np.float32)
    return src_xyz, src_feat, tgt_xyz, tgt_feat


def random_pose(rng):
    a = rng.uniform(0, 2 * np.pi, 3)
    Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
    Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
    Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rx @ Ry @ Rz
    T[:3, 3] = rng.uniform(-0.5, 0.5, 3)
    return T


def write_test_tree(root, rng, pairs):
    """Test split: the script reads gt.log (target -> source transforms) and cloud_bin_<id>_fpfh.npz per scene."""
    for s in SCENES:
        os.makedirs(os.path.join(root, "gt_result", s + "-evaluation"), exist_ok=True)
        os.makedirs(os.path.join(root, "fragments", s), exist_ok=True)
        open(os.path.join(root, "gt_result", s + "-evaluation", "gt.log"), "w").close()
    scene = SCENES[0]
    lines = []
    for q, (n_src, n_tgt) in enumerate(pairs):
        a, b = 2 * q, 2 * q + 1
        pose = random_pose(rng)                               # src -> tgt
        sx, sf, tx, tf = synthetic_fragment_pair(rng, n_src, n_tgt, 33, pose)
        np.savez(os.path.join(root, "fragments", scene, f"cloud_bin_{a}_fpfh.npz"), xyz=sx, feature=sf)
        np.savez(os.path.join(root, "fragments", scene, f"cloud_bin_{b}_fpfh.npz"), xyz=tx, feature=tf)
        inv = np.linalg.inv(pose)                             # the log stores target -> source (:303)
        lines.append(f"{a}\t{b}\t{len(pairs) * 2}\n")
        for r in range(4):
            lines.append(" \t".join(f"{v:.10e}" for v in inv[r]) + "\n")
    with open(os.path.join(root, "gt_result", scene + "-evaluation", "gt.log"), "w") as f:
        f.writelines(lines)


def write_train_tree(root, rng, pairs):
    os.makedirs(os.path.join(root, "misc", "split"), exist_ok=True)
    os.makedirs(os.path.join(root, "threedmatch"), exist_ok=True)
    os.makedirs(os.path.join(root, "threedmatch_feat"), exist_ok=True)
    name = "synthetic-scene"
    with open(os.path.join(root, "misc", "split", "test_3dmatch.txt"), "w") as f:
        f.write(name + "\n")
    rows = []
    for q, (n_src, n_tgt) in enumerate(pairs):
        a, b = f"{name}@seq-01_{2 * q:03d}.npz", f"{name}@seq-01_{2 * q + 1:03d}.npz"
        # training fragments are already in one world frame (orig_trans = identity, :107); FCGF descriptors are unit vectors
        sx, sf, tx, tf = synthetic_fragment_pair(rng, n_src, n_tgt, 32, np.eye(4))
        sf /= np.linalg.norm(sf, axis=1, keepdims=True)
        tf /= np.linalg.norm(tf, axis=1, keepdims=True)
        np.savez(os.path.join(root, "threedmatch_feat", a.replace(".npz", "_fcgf.npz")), xyz=sx, feature=sf)
        np.savez(os.path.join(root, "threedmatch_feat", b.replace(".npz", "_fcgf.npz")), xyz=tx, feature=tf)
        rows.append(f"{a} {b} 0.45\n")
    with open(os.path.join(root, "threedmatch", name + "@seq-01-0.30.txt"), "w") as f:
        f.writelines(rows)


def run_reference(root, out, train, mutual, seed):
    text = open(SCRIPT, encoding="utf-8").read()
    assert OLD_OUT in text and OLD_ROOT in text and "make_training_data = False" in text and "use_mutual = False" in text
    text = text.replace(OLD_OUT, out).replace(OLD_ROOT, root)
    if train:
        text = text.replace("make_training_data = False", "make_training_data = True")
    if mutual:
        text = text.replace("use_mutual = False", "use_mutual = True")
    np.random.seed(seed)
    random.seed(seed)
    sys.path.insert(0, REF)
    try:
        exec(compile(text, SCRIPT, "exec"), {"__name__": "__main__", "__file__": SCRIPT})
    finally:
        sys.path.remove(REF)
    sub = os.path.join(out, "train_3dmatch" if train else "test_3dmatch")
    recs = []
    for fn in sorted(os.listdir(sub), key=lambda s: int(s.split(".")[0])):
        with open(os.path.join(sub, fn), "rb") as f:
            recs.append(pickle.load(f))
    return recs


def margins(rec, thr=0.10):
    """Decision margins of one record, recomputed in fp64 from what the script saved: best-vs-second-best descriptor distance per
    source row (and per target column), and |point distance - threshold| per correspondence."""
    s, t = rec["feat_0"].astype(np.float64), rec["feat_1"].astype(np.float64)
    d = np.sqrt(np.maximum(2 - 2 * (s @ t.T) + 1e-6, 0))
    pr = np.partition(d, 1, axis=1)[:, :2]
    pc = np.partition(d, 1, axis=0)[:2, :]
    P = rec["gt_pose"]
    c = rec["corr"]
    w = rec["xyz_0"].astype(np.float64)[c[:, 0]] @ P[:3, :3].T + P[:3, 3]
    pd = np.sqrt(((w - rec["xyz_1"].astype(np.float64)[c[:, 1]]) ** 2).sum(1))
    return pr[:, 1] - pr[:, 0], pc[1] - pc[0], np.abs(pd - thr)


def main(out_path):
    fx = {}
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for tag, train, mutual, seed in (("test", False, False, 11), ("test_mutual", False, True, 12),
                                         ("train", True, False, 13), ("train_mutual", True, True, 14)):
            root, out = os.path.join(tmp, tag, "root"), os.path.join(tmp, tag, "out")
            os.makedirs(root)
            os.makedirs(out)
            rng = np.random.default_rng(seed)
            (write_train_tree if train else write_test_tree)(root, rng, [(700, 650), (301, 515)])
            for q, rec in enumerate(run_reference(root, out, train, mutual, seed)):
                name = f"{tag}.{q}"
                cases.append(name)
                row_gap, col_gap, thr_gap = margins(rec)
                fx[name + ".feat_0"], fx[name + ".feat_1"] = rec["feat_0"].astype(np.float32), rec["feat_1"].astype(np.float32)
                fx[name + ".xyz_0"], fx[name + ".xyz_1"] = rec["xyz_0"], rec["xyz_1"]
                fx[name + ".gt_pose"] = np.asarray(rec["gt_pose"], dtype=np.float64)
                fx[name + ".corr"], fx[name + ".labels"] = rec["corr"].astype(np.int64), rec["labels"].astype(np.int32)
                fx[name + ".mutual"] = np.array(int(mutual))
                fx[name + ".row_gap"], fx[name + ".col_gap"], fx[name + ".thr_gap"] = row_gap, col_gap, thr_gap
                print(name, rec["feat_0"].shape, rec["feat_1"].shape, "corr", rec["corr"].shape, "inliers", int(rec["labels"].sum()),
                      "min row gap %.2e" % row_gap.min(), "feat dtype", rec["feat_0"].dtype)
    fx["cases"] = np.array(cases)
    np.savez_compressed(out_path, **fx)
    print("wrote", out_path, os.path.getsize(out_path) // 1024, "KiB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "match_cases.npz"))
