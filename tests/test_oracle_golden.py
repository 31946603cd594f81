"""CPU: the oracle restatement against the golden vectors produced by the real reference
(tests/golden/gen_golden.py).  fp32 results must match the reference's fp32 run to
round-off (same ATen ops -> expected bitwise; a few ulp allowed), fp64 likewise."""
import os

import numpy as np
import pytest
import torch

from oracle import egnn_oracle as orc

LAYER_FIX = ["layer_n128_k8_f32_h1.npz", "layer_n128_k8_f33_h1.npz", "layer_n128_k8_f32_h4.npz",
             "layer_n96_k6_f32_h1_dup.npz", "layer_n128_k8_f32_h2.npz", "layer_n128_k8_f32_h8.npz",
             "layer_n96_k6_f64_h2.npz", "layer_n96_k6_f16_h4.npz"]
PAIR_FIX = ["pair_n2048_k20_l4.npz", "pair_n512_k12_l2_wellcond.npz", "pair_ckpt_n2048_k16_l3_h4.npz"]


def weights(fx, dt, prefix="w."):
    return {k[len(prefix):]: torch.from_numpy(fx[k]).to(dt) for k in fx.files if k.startswith(prefix)}


@pytest.mark.parametrize("name", LAYER_FIX)
@pytest.mark.parametrize("tag,dt,tol", [("f32", torch.float32, 2e-6), ("f64", torch.float64, 1e-12)])
def test_layer_forward_and_grads(golden_dir, name, tag, dt, tol):
    fx = np.load(os.path.join(golden_dir, name))
    w = {k: v.requires_grad_(True) for k, v in weights(fx, dt).items()}
    h = torch.from_numpy(fx["h"]).to(dt).requires_grad_(True)
    x = torch.from_numpy(fx["x"]).to(dt).requires_grad_(True)
    edges = torch.from_numpy(fx["edges"])
    row, col = edges[0], edges[1]
    ea = torch.ones(row.numel(), 1, dtype=dt)
    # geometry (A4-A6)
    diff, radial, dist, dot, so3 = orc.edge_geometry(x, row, col)
    for key, val in (("so3", so3), ("radial", radial), ("dist", dist), ("dot", dot)):
        np.testing.assert_allclose(val.detach().numpy(), fx[f"{tag}.{key}"], rtol=0, atol=tol)
    # layer 0 (A7-A9)
    h0 = torch.nn.functional.linear(h, w["embedding_in.weight"], w["embedding_in.bias"])
    h1, x1, inter = orc.egcl_layer(w, "gcl_0.", h0, x, row, col, ea, return_intermediates=True)
    np.testing.assert_allclose(inter["edge_feat"].detach().numpy(), fx[f"{tag}.edge_feat0"], rtol=0, atol=10 * tol)
    np.testing.assert_allclose(h1.detach().numpy(), fx[f"{tag}.h1"], rtol=0, atol=10 * tol)
    np.testing.assert_allclose(x1.detach().numpy(), fx[f"{tag}.x1"], rtol=0, atol=10 * tol)
    # full EGNN (A3) + backward (A14)
    ho, xo = orc.egnn_forward(w, h, x, [row, col], ea)
    np.testing.assert_allclose(ho.detach().numpy(), fx[f"{tag}.h_out"], rtol=0, atol=10 * tol)
    np.testing.assert_allclose(xo.detach().numpy(), fx[f"{tag}.x_out"], rtol=0, atol=10 * tol)
    loss = (ho * torch.from_numpy(fx["gh"]).to(dt)).sum() + (xo * torch.from_numpy(fx["gx"]).to(dt)).sum()
    loss.backward()
    gscale = 50 * tol
    np.testing.assert_allclose(h.grad.numpy(), fx[f"{tag}.grad_h"], rtol=1e-4 if dt == torch.float32 else 1e-9, atol=gscale)
    np.testing.assert_allclose(x.grad.numpy(), fx[f"{tag}.grad_x"], rtol=1e-4 if dt == torch.float32 else 1e-9, atol=gscale)
    for k, v in w.items():
        ref = fx[f"{tag}.grad.{k}"]
        np.testing.assert_allclose(v.grad.numpy(), ref, rtol=1e-4 if dt == torch.float32 else 1e-9,
                                   atol=gscale * max(1.0, float(np.abs(ref).max())))


def _pair_args(fx, dt):
    k = int(fx["k"])
    src, tgt = torch.from_numpy(fx["in.src"]), torch.from_numpy(fx["in.tgt"])
    e_s, e_t = orc.knn_graph(src, k), orc.knn_graph(tgt, k)
    ea = torch.ones(1, e_s.shape[1], 1, dtype=dt)
    f = lambda n: torch.from_numpy(fx["in." + n])[None].to(dt)
    return (f("feat_src"), f("src"), e_s[None], ea, f("feat_tgt"), f("tgt"), e_t[None], ea,
            f("corr"), f("labels"), f("gt_pose"))


@pytest.mark.parametrize("name", PAIR_FIX)
@pytest.mark.parametrize("tag,dt", [("f32", torch.float32), ("f64", torch.float64)])
def test_pair_heads(golden_dir, name, tag, dt):
    fx = np.load(os.path.join(golden_dir, name))
    wall = weights(fx, dt)
    we = {k[len("egnn."):]: v for k, v in wall.items() if k.startswith("egnn.")}
    wc = {k: v for k, v in wall.items() if not k.startswith("egnn.")}
    args = _pair_args(fx, dt)
    tol = 5e-5 if dt == torch.float32 else 1e-10
    hmax = float(np.abs(fx["f64.h_src"]).max())
    with torch.no_grad():
        res, ex = orc.pair_forward_train(we, wc, *args)
    np.testing.assert_allclose(res[4][0].numpy(), fx[f"{tag}.h_src"], rtol=0, atol=tol * max(1, hmax))
    np.testing.assert_allclose(res[7][0].numpy(), fx[f"{tag}.x_tgt"], rtol=0, atol=tol)
    np.testing.assert_array_equal(ex["top_idx"].numpy(), fx[f"{tag}.train.top_idx"])
    np.testing.assert_allclose(ex["logits"].numpy(), fx[f"{tag}.train.logits"], rtol=0, atol=tol * max(1, hmax))
    np.testing.assert_allclose(res[0].numpy(), fx[f"{tag}.train.R"], rtol=0, atol=tol)
    np.testing.assert_allclose(res[1].numpy(), fx[f"{tag}.train.t"], rtol=0, atol=tol)
    np.testing.assert_allclose(res[2].numpy(), fx[f"{tag}.train.corr_sim_loss"], rtol=1e-4, atol=tol)
    np.testing.assert_allclose(res[3].numpy(), fx[f"{tag}.train.equi_loss"], rtol=1e-4, atol=tol)
    rl, tl = orc.pose_loss(res[0], res[1], args[-1])
    np.testing.assert_allclose(rl.numpy(), fx[f"{tag}.train.rot_loss"], rtol=1e-3, atol=1e-3 if dt == torch.float32 else 1e-7)
    np.testing.assert_allclose(tl.numpy(), fx[f"{tag}.train.trans_loss"], rtol=1e-3, atol=1e-3 if dt == torch.float32 else 1e-7)
    with torch.no_grad():
        rese, _ = orc.pair_forward_eval(we, wc, *args)
    assert rese[2] is None
    np.testing.assert_allclose(rese[0].numpy(), fx[f"{tag}.eval.R"], rtol=0, atol=tol)
    np.testing.assert_allclose(rese[1].numpy(), fx[f"{tag}.eval.t"], rtol=0, atol=tol)


def test_train_loss_grads_wellcond(golden_dir):
    """One full training loss (train :1120) and its parameter gradients, fp64, well-conditioned head."""
    fx = np.load(os.path.join(golden_dir, "pair_n512_k12_l2_wellcond.npz"))
    dt = torch.float64
    wall = {k: v.requires_grad_(True) for k, v in weights(fx, dt).items()}
    we = {k[len("egnn."):]: v for k, v in wall.items() if k.startswith("egnn.")}
    wc = {k: v for k, v in wall.items() if not k.startswith("egnn.")}
    args = _pair_args(fx, dt)
    res, _ = orc.pair_forward_train(we, wc, *args)
    rl, tl = orc.pose_loss(res[0], res[1], args[-1])
    loss = res[2].mean() + rl.mean() + tl.mean()
    np.testing.assert_allclose(loss.detach().numpy(), fx["f64.train.loss"], rtol=1e-9)
    loss.backward()
    checked = 0
    for k, v in wall.items():
        key = "f64.train.grad." + k
        if key in fx.files:
            ref = fx[key]
            np.testing.assert_allclose(v.grad.numpy(), ref, rtol=1e-6, atol=1e-9 * max(1.0, float(np.abs(ref).max())))
            checked += 1
    assert checked > 20


def test_kabsch_cases(golden_dir):
    fx = np.load(os.path.join(golden_dir, "kabsch_cases.npz"))
    for ci in range(int(fx["n_cases"])):
        s = torch.from_numpy(fx[f"c{ci}.src"]); t = torch.from_numpy(fx[f"c{ci}.tgt"])
        w = torch.softmax(torch.from_numpy(fx[f"c{ci}.score"]), -1)
        w = w / (w.sum() + 1e-6)
        R, tt = orc.weighted_kabsch(s, t, w)
        np.testing.assert_allclose(R.numpy(), fx[f"c{ci}.f64.R"], atol=1e-12)
        np.testing.assert_allclose(tt.numpy(), fx[f"c{ci}.f64.t"], atol=1e-12)
        assert abs(float(torch.det(R)) - 1.0) < 1e-9


# ------------------------------------------------------------------ kNN known-answer tests
def test_knn_grid_known_answer():
    # 4x4x1 unit grid: the 5 nearest of an interior point are itself + 4 axis neighbours (d2=1)
    g = torch.stack(torch.meshgrid(torch.arange(4.), torch.arange(4.), torch.arange(1.), indexing="ij"), -1).reshape(-1, 3)
    e = orc.knn_graph(g, 5, loop=True)
    assert e.shape == (2, 16 * 5)
    assert torch.equal(e[1], torch.arange(16).repeat_interleave(5))
    q = 5  # (1,1)
    nb = e[0, q * 5:(q + 1) * 5].tolist()
    assert nb[0] == q and sorted(nb[1:]) == sorted([1, 4, 6, 9])
    assert nb[1:] == sorted(nb[1:])          # equal distance -> ascending index
    e2 = orc.knn_graph(g, 4, loop=False)
    assert q not in e2[0, q * 4:(q + 1) * 4].tolist()


def test_knn_cdist_form_finds_the_same_neighbour_sets():
    """bench.py times the CPU baseline with the cdist + topk neighbour search; it must be the same graph as the bit-exact
    restatement (identical sets per centre on generic points; rows may differ only in the order of tied distances)."""
    g = torch.Generator().manual_seed(3)
    x = torch.rand(500, 3, generator=g) * 3
    for loop in (True, False):
        a, b = orc.knn_graph(x, 12, loop=loop), orc.knn_graph_cdist(x, 12, loop=loop)
        assert torch.equal(a[1], b[1])
        assert torch.equal(torch.sort(a[0].view(500, 12), dim=1).values, torch.sort(b[0].view(500, 12), dim=1).values)
        assert (a[0] == b[0]).float().mean() > 0.999


def test_knn_duplicates_and_ties():
    x = torch.tensor([[0., 0, 0], [1, 0, 0], [0, 0, 0], [2, 0, 0], [0, 0, 0]])
    e = orc.knn_graph(x, 3, loop=True)
    assert e[0, 0:3].tolist() == [0, 2, 4]       # three coincident points, index order
    assert e[0, 6:9].tolist() == [0, 2, 4]
    assert e[0, 3:6].tolist() == [1, 0, 2]       # self first (d=0), then ties by index


def test_so3_tensor_product_golden(golden_dir):
    """A15: oracle vs the real SO3TensorProductLayer (src/clean_so3_model.py:34-52)."""
    fx = np.load(os.path.join(golden_dir, "so3tp_d32.npz"))
    for dt, key, tol in ((torch.float32, "y", 2e-5), (torch.float64, "y64", 1e-11)):
        y = orc.so3_tensor_product(weights(fx, dt), torch.from_numpy(fx["x"]).to(dt))
        ref = fx[key]
        assert np.abs(y.numpy() - ref).max() <= tol * max(1.0, np.abs(ref).max())


def test_gcn_layer_golden(golden_dir):
    """A17: oracle vs the real src/gcnLayer.py (build_graph on scipy/cKDTree, GraphConvolution, GlobalPooling)."""
    fx = np.load(os.path.join(golden_dir, "gcn_n300_k9.npz"))
    coords, k = fx["coords"], int(fx["k"])
    adj = np.stack([orc.gcn_build_graph(coords[b], k) for b in range(coords.shape[0])])
    assert np.abs(adj - fx["adj64"]).max() <= 1e-12
    assert ((adj != 0) == (fx["adj64"] != 0)).all()
    w = weights(fx, torch.float64)
    x = torch.from_numpy(coords).double().transpose(1, 2)
    y = orc.gcn_graph_convolution(w, torch.from_numpy(adj), x)
    assert np.abs(y.numpy() - fx["y64"]).max() <= 1e-12
    assert np.abs(orc.gcn_global_pooling(y).numpy() - fx["pooled64"]).max() <= 1e-12
    y32 = orc.gcn_graph_convolution(weights(fx, torch.float32), torch.from_numpy(adj).float(), x.float())
    assert np.abs(y32.numpy() - fx["y"]).max() <= 2e-6


def test_eval_metrics_golden(golden_dir):
    """N2: oracle vs the real tools/evaluation_metrics.py."""
    fx = np.load(os.path.join(golden_dir, "metrics_cases.npz"))
    for p in range(fx["gt"].shape[0]):
        got = orc.eval_metrics(fx["gt"][p], fx["pred"][p], fx["src"][p], fx["tgt"][p])
        assert np.allclose(np.array(got), fx["metrics"][p], rtol=1e-12, atol=1e-12)


def test_pointnet_golden(golden_dir):
    """N1: oracle vs the real PointNet/PointNetLayer classes (aggregation by the documented PyG semantics, see gen_golden.py)."""
    fx = np.load(os.path.join(golden_dir, "pointnet_n200_k6.npz"))
    ei, batch = torch.from_numpy(fx["edge_index"]), torch.from_numpy(fx["batch"])
    for tag, dt, tol in (("f32", torch.float32, 2e-6), ("f64", torch.float64, 1e-13)):
        w = weights(fx, dt)
        pos, feats = torch.from_numpy(fx["pos"]).to(dt), torch.from_numpy(fx["feats"]).to(dt)
        h1 = orc.pointnet_layer({k[len("conv1."):]: v for k, v in w.items() if k.startswith("conv1.")}, feats, pos, ei)
        _, pooled = orc.pointnet_forward(w, pos, feats, ei, batch)
        assert np.abs(h1.numpy() - fx["conv1_" + tag]).max() <= tol * max(1.0, np.abs(fx["conv1_" + tag]).max())
        assert np.abs(pooled.numpy() - fx["pooled_" + tag]).max() <= tol * max(1.0, np.abs(fx["pooled_" + tag]).max())
        assert (h1.numpy()[3] == 0).all()                 # the node without incoming edges


def test_correspondence_construction_golden(golden_dir):
    """N3: oracle.build_correspondences vs what the REAL data_preprocess/3DMatch_Feature.py wrote (:157-196 train split,
    :331-351 test split; fixtures from tests/golden/gen_match_golden.py, which executes that script on a synthetic dataset tree)."""
    fx = np.load(os.path.join(golden_dir, "match_cases.npz"))
    assert len(fx["cases"]) == 8
    for name in fx["cases"]:
        g = lambda k: fx[f"{name}.{k}"]  # noqa: E731
        corr, labels, _, _ = orc.build_correspondences(g("feat_0"), g("feat_1"), g("xyz_0").astype(np.float64), g("xyz_1").astype(np.float64),
                                                       g("gt_pose"), bool(g("mutual")), 0.10)
        assert np.array_equal(corr, g("corr")), name
        clear = g("thr_gap") > 1e-6              # the script labelled the un-rounded fp64 points; the record holds them as fp32
        assert clear.mean() > 0.99 and np.array_equal(labels[clear], g("labels")[clear]), name
        assert 0 < g("labels").sum() < g("labels").size
