"""SURVEY §8(f) row N3 on the GPU: preprocessing record -> pair files -> dataset classes -> batches resident in HBM -> the
registration step, against the oracle.  (The sampling itself is pinned bit-exact against the real reference classes in
tests/test_datasets.py; this file covers the device ends of that path.)"""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import egnn_oracle as orc  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def pk():
    import se3_equi_graph_registration_amd as pkg
    assert torch.cuda.is_available(), "these tests need the MI355X"
    pkg.lib()
    return pkg


def _fragment_pair(seed, n=700, dim=32, overlap=0.6):
    """Two keypoint sets with a known pose, `overlap` of the target being noisy copies of source points, and unit
    descriptors that agree on the overlapping part."""
    rng = np.random.default_rng(seed)
    from scipy.spatial.transform import Rotation
    pose = np.eye(4)
    pose[:3, :3] = Rotation.random(random_state=seed).as_matrix()
    pose[:3, 3] = rng.uniform(-0.5, 0.5, 3)
    src = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    fs = rng.standard_normal((n, dim)).astype(np.float32)
    fs /= np.linalg.norm(fs, axis=1, keepdims=True)
    m = n - 50
    tgt = rng.uniform(-1.5, 1.5, (m, 3)).astype(np.float32)
    ft = rng.standard_normal((m, dim)).astype(np.float32)
    k = int(m * overlap)
    pick = rng.permutation(n)[:k]
    tgt[:k] = (src[pick] @ pose[:3, :3].T + pose[:3, 3] + 0.01 * rng.standard_normal((k, 3))).astype(np.float32)
    ft[:k] = fs[pick] + 0.03 * rng.standard_normal((k, dim)).astype(np.float32)
    ft /= np.linalg.norm(ft, axis=1, keepdims=True)
    return src, tgt, fs, ft, pose


def test_preprocess_pair_record_vs_oracle(pk, tmp_path):
    src, tgt, fs, ft, pose = _fragment_pair(3)
    rec = pk.datasets.preprocess_pair(src, tgt, fs, ft, pose, inlier_threshold=0.10, file_0="cloud_bin_0", file_1="cloud_bin_1")
    corr_o, labels_o, _, dmat = orc.build_correspondences(fs, ft, src, tgt, pose.astype(np.float32), False, 0.10)
    gap = np.partition(dmat, 1, axis=1)
    assert ((gap[:, 1] - gap[:, 0]) > 1e-5).all()                      # no near-ties in this input: indices must be exact
    assert np.array_equal(rec["corr"], corr_o) and np.array_equal(rec["labels"], labels_o)
    assert rec["labels"].dtype == np.int32 and 0.3 < rec["labels"].mean() < 0.9
    assert rec["xyz_0"].dtype == np.float32 and rec["gt_pose"].shape == (4, 4)
    # the record is what the test-split class reads (3DMatch_Feature.py:353-361 -> ThreeDMatch.py:622-634)
    os.makedirs(tmp_path / "test_3dmatch")
    pk.datasets.save_pair_file(str(tmp_path / "test_3dmatch" / "0.pkl"), rec, "pickle-arrays")
    (tmp_path / "test_files.txt").write_text("0.pkl\n")
    np.random.seed(0)
    c, lab, s_src, s_tgt, s_fs, s_ft, gt = pk.datasets.ThreeDMatchTest(root=str(tmp_path), split="test", num_node=256)[0]
    assert c.shape == (256, 2) and s_src.shape == (256, 3) and s_fs.shape == (256, 32)
    ci = c.astype(np.int64)
    assert np.array_equal(s_src, src[ci[:, 0]]) and np.array_equal(s_tgt, tgt[ci[:, 1]]) and np.array_equal(s_ft, ft[ci[:, 1]])
    warped = s_src @ gt[:3, :3].T + gt[:3, 3]
    assert np.array_equal(np.linalg.norm(warped - s_tgt, axis=1) < 0.10, lab.astype(bool))


@pytest.fixture()
def train_root(pk, tmp_path):
    os.makedirs(tmp_path / "train_3dmatch")
    names = []
    for i in range(5):
        src, tgt, fs, ft, pose = _fragment_pair(10 + i)
        rec = pk.datasets.preprocess_pair(src, tgt, fs, ft, pose.astype(np.float32))
        pk.datasets.save_pair_file(str(tmp_path / "train_3dmatch" / f"{i}.pkl"), rec, "pickle-tensors")
        names.append(f"{i}.pkl")
    (tmp_path / "train_files.txt").write_text("\n".join(names) + "\n")
    (tmp_path / "val_files.txt").write_text("\n".join(names[:2]) + "\n")
    return str(tmp_path)


def test_device_batches_equal_host_batches(pk, train_root):
    d = pk.datasets.ThreeDMatchTrainVal(root=train_root, split="train", num_node=256)
    loader = torch.utils.data.DataLoader(d, batch_size=2, shuffle=False)          # 5 pairs: batches of 2, 2, 1
    np.random.seed(5); random.seed(5)
    host = [tuple(t.clone() for t in b) for b in loader]
    np.random.seed(5); random.seed(5)
    dev = list(pk.datasets.DeviceBatches(loader, device="cuda"))
    assert len(dev) == len(host) == 3 == len(pk.datasets.DeviceBatches(loader))
    for hb, db in zip(host, dev):
        assert all(t.is_cuda for t in db)
        for h, g in zip(hb, db):
            assert torch.equal(h, g.cpu())
    assert list(pk.datasets.DeviceBatches(torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.zeros(0, 3))))) == []
    import threading
    before = threading.active_count()
    for first in pk.datasets.DeviceBatches(loader, device="cuda"):               # leaving early ends the staging thread
        break
    assert first[2].is_cuda and first[2].shape == host[0][2].shape and threading.active_count() == before

    class Boom(torch.utils.data.Dataset):
        def __len__(self):
            return 2

        def __getitem__(self, i):
            raise KeyError("missing record")

    with pytest.raises(KeyError, match="missing record"):                          # loader errors surface in the consumer
        list(pk.datasets.DeviceBatches(torch.utils.data.DataLoader(Boom(), batch_size=1)))


def test_dataset_batch_through_registration_step_vs_oracle(pk, train_root):
    """train :968-1120 with this package's classes: dataset 7-tuple -> kNN graphs -> EGNN + head -> losses -> backward."""
    n, k, B, feat = 256, 8, 2, 32
    d = pk.datasets.ThreeDMatchTrainVal(root=train_root, split="train", num_node=n)
    np.random.seed(7); random.seed(7)
    batch = next(iter(pk.datasets.DeviceBatches(torch.utils.data.DataLoader(d, batch_size=B, shuffle=False))))
    corr, labels, src, tgt, fs, ft, gt = batch
    torch.manual_seed(2)
    egnn = pk.EGNN(feat, feat, feat, in_edge_nf=1, device="cuda", n_layers=2)
    with torch.no_grad():
        egnn.embedding_out.weight.mul_(0.05); egnn.embedding_out.bias.mul_(0.05)
    model = pk.CrossAttentionPoseRegression(egnn, num_nodes=n, hidden_nf=feat, device="cuda").cuda()
    graph = pk.build_pair_graph(src, tgt, k)
    out = model(fs, src, graph, None, ft, tgt, graph, None, corr, labels, gt)
    loss, rl, tl = pk.registration_loss(out[0], out[1], out[2], gt)
    loss.backward()
    assert torch.isfinite(loss) and torch.isfinite(egnn.flat_grads).all() and float(egnn.flat_grads.abs().max()) > 0
    we = {kk: v.detach().cpu().double() for kk, v in egnn.state_dict().items()}
    wc = {kk: v.detach().cpu().double() for kk, v in model.state_dict().items() if not kk.startswith("egnn.")}
    e_s = torch.stack([orc.knn_graph(src[i].cpu(), k) for i in range(B)])
    e_t = torch.stack([orc.knn_graph(tgt[i].cpu(), k) for i in range(B)])
    ea = torch.ones(B, n * k, 1, dtype=torch.float64)
    dd = lambda t: t.cpu().double()  # noqa: E731
    ref, extras = orc.pair_forward_train(we, wc, dd(fs), dd(src), e_s, ea, dd(ft), dd(tgt), e_t, ea, dd(corr), dd(labels), dd(gt))
    assert float((out[0].detach().cpu().double() - ref[0]).abs().max()) <= 1e-4
    assert float((out[1].detach().cpu().double() - ref[1]).abs().max()) <= 1e-4
    assert abs(float(out[2].detach()) - float(ref[2])) <= 2e-4 * max(1.0, abs(float(ref[2])))
    o_rl, o_tl = orc.pose_loss(ref[0].float(), ref[1].float(), gt.cpu())
    # the pairs are nearly solved (angle ~1e-3 rad), where arccos turns 1e-7 of trace round-off into ~5e-4 rad
    assert torch.allclose(rl.detach().cpu(), o_rl, atol=3e-3) and torch.allclose(tl.detach().cpu(), o_tl, rtol=1e-3, atol=1e-4)
