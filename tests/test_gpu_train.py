"""The callers of the hot path on the GPU (train :964-1140, :1311-1396): resume from a checkpoint -- weights + the torch.optim.Adam
state dict -- and continue training, against the steps the REAL reference + torch.optim.Adam took from the same checkpoint
(tests/golden/ckpt_resume_step.npz); checkpoint files in the reference layout; one epoch over the dataset classes."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pk():
    import se3_equi_graph_registration_amd as pkg
    assert torch.cuda.is_available(), "these tests need the MI355X"
    pkg.lib()
    return pkg


@pytest.fixture(scope="module")
def fx(golden_dir):
    with np.load(os.path.join(golden_dir, "ckpt_resume_step.npz")) as z:
        return {k: z[k] for k in z.files}


def _model(pk, fx, prefix="w0."):
    w = {k[len(prefix):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(prefix)}
    we = {k[len("egnn."):]: v for k, v in w.items() if k.startswith("egnn.")}
    feat = we["embedding_in.weight"].shape[1]
    egnn = pk.EGNN(feat, feat, feat, in_edge_nf=1, device="cuda", n_layers=int(fx["n_layers"]), num_heads=int(fx["heads"]))
    model = pk.CrossAttentionPoseRegression(egnn, num_nodes=fx["in.src"].shape[0], hidden_nf=feat, device="cuda").cuda()
    model.load_state_dict(w, strict=False)
    return model


def _adam_state(fx):
    """The checkpoint's optimizer entry, as torch.optim.Adam.state_dict() lays it out."""
    n = len(fx["names"])
    state = {i: {"step": torch.tensor(float(fx["opt.step"])), "exp_avg": torch.from_numpy(fx[f"opt.m.{i}"]),
                 "exp_avg_sq": torch.from_numpy(fx[f"opt.v.{i}"])} for i in range(n) if fx["opt.has_state"][i]}
    group = {"lr": float(fx["lr"]), "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False, "maximize": False,
             "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(n))}
    return {"state": state, "param_groups": [group]}


def _batch(fx):
    t = lambda k: torch.from_numpy(fx["in." + k])[None].cuda()  # noqa: E731
    return t("corr"), t("labels"), t("src"), t("tgt"), t("feat_src"), t("feat_tgt"), t("gt_pose")


def test_parameter_order_matches_reference_optimizer(pk, fx):
    """torch.optim.Adam state is keyed by position in model.parameters(): names and shapes must line up with the
    reference's, for the fixture model and for the layout of the shipped checkpoint-3dmatch.pth (97 parameters, 85 stepped)."""
    model = _model(pk, fx)
    assert [n for n, _ in model.named_parameters()] == fx["names"].tolist()
    egnn = pk.EGNN(32, 32, 32, in_edge_nf=1, device="cuda", n_layers=3, num_heads=4)
    ship = pk.CrossAttentionPoseRegression(egnn, num_nodes=2048, hidden_nf=32, device="cuda")
    mine = list(ship.named_parameters())
    assert [n for n, _ in mine] == fx["ship.names"].tolist()
    assert ["x".join(map(str, p.shape)) for _, p in mine] == fx["ship.shapes"].tolist()
    assert len(mine) == 97 and int(fx["ship.has_state"].sum()) == 85 and float(fx["ship.step"]) == 8295.0


def test_resume_from_checkpoint_continues_like_the_reference(pk, fx):
    from se3_equi_graph_registration_amd import train
    model = _model(pk, fx)
    model.train()
    opt = train.make_optimizer(model, lr=123.0)                     # hyper-parameters come from the checkpoint
    opt.load_state_dict(_adam_state(fx))
    assert opt.adam.lr == float(fx["lr"]) and float(opt.adam.step_dev[0]) == float(fx["opt.step"])
    # state_dict() gives back what was loaded: same entries (parameters that never saw a gradient have none), same tensors
    sd = opt.state_dict()
    assert sorted(sd["state"]) == [i for i in range(len(fx["names"])) if fx["opt.has_state"][i]]
    for i, st in sd["state"].items():
        assert float(st["step"]) == float(fx["opt.step"])
        assert torch.equal(st["exp_avg"].cpu(), torch.from_numpy(fx[f"opt.m.{i}"]))
        assert torch.equal(st["exp_avg_sq"].cpu(), torch.from_numpy(fx[f"opt.v.{i}"]))
    torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in model.parameters()], lr=1.0).load_state_dict(sd)   # torch accepts it
    batch = _batch(fx)
    k, lr, steps = int(fx["k"]), float(fx["lr"]), int(fx["steps"])
    names = fx["names"].tolist()
    for s in range(1, steps + 1):
        opt.zero_grad()
        _, loss, _, _ = train._forward(model, batch, k)
        loss.backward()
        opt.step()
        ref64 = float(fx[f"f64.loss.{s}"])
        assert abs(float(loss.detach()) - ref64) <= 4 * abs(float(fx[f"f32.loss.{s}"]) - ref64) + 1e-5 * abs(ref64), (s, float(loss.detach()), ref64)
        # Adam turns every gradient into a step of ~lr, so a weight records the gradient's relative error: after two steps the
        # reference's own fp32 and fp64 runs differ by up to 1e-5 (0.5 % of one step).  Every element: within 4x that
        # element-wise gap + 0.5 % of a step of the fp64 run, and within 2 % of a step whatever the gap.
        for nm, prm in zip(names, model.parameters()):
            got = prm.detach().cpu().double().numpy()
            r32, r64 = fx[f"f32.w{s}.{nm}"].astype(np.float64), fx[f"f64.w{s}.{nm}"]
            err = np.abs(got - r64)
            assert (err <= 4 * np.abs(r32 - r64) + 0.005 * lr).all(), (s, nm, float((err - 4 * np.abs(r32 - r64)).max()))
            assert err.max() <= 0.02 * lr * s, (s, nm, err.max())
            moved = np.abs(r64 - fx["w0." + nm].astype(np.float64)).max()
            assert moved > 0.2 * lr or not fx["opt.has_state"][names.index(nm)], nm          # the comparison is not vacuous
    assert float(opt.adam.step_dev[0]) == float(fx["opt.step"]) + steps


def test_checkpoint_files_round_trip_in_the_reference_layout(pk, fx, tmp_path):
    from se3_equi_graph_registration_amd import train
    model = _model(pk, fx)
    opt = train.make_optimizer(model, lr=float(fx["lr"]))
    opt.load_state_dict(_adam_state(fx))
    path = train.save_checkpoint(7, None, model.egnn, model, opt, save_dir=str(tmp_path / "ck"), is_best=True)
    assert os.path.basename(path) == "model_epoch_7.pth" and os.path.exists(tmp_path / "ck" / "best_checkpoint")
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert set(raw) == {"epoch", "egnn_state_dict", "cross_attention_state_dict", "optimizer_state_dict"}       # train :1330-1336
    assert set(raw["optimizer_state_dict"]) == {"state", "param_groups"} and raw["epoch"] == 7
    assert all(k in raw["cross_attention_state_dict"] for k in ("egnn.embedding_in.weight", "mlp.0.weight", "bn1.running_mean"))
    fresh = _model(pk, {**fx, **{k: np.zeros_like(v) for k, v in fx.items() if k.startswith("w0.")}})
    opt2 = train.make_optimizer(fresh, lr=1.0)
    _, epoch = train.load_checkpoint(path, egnn=fresh.egnn, cross_attention=fresh, optimizer=opt2, device="cuda:0")
    assert epoch == 7
    for (n1, a), (n2, b) in zip(model.state_dict().items(), fresh.state_dict().items()):
        assert n1 == n2 and torch.equal(a, b), n1
    assert torch.equal(opt.adam.exp_avg[0], opt2.adam.exp_avg[0]) and torch.equal(opt.adam.exp_avg_sq[1], opt2.adam.exp_avg_sq[1])
    assert float(opt2.adam.step_dev[0]) == float(fx["opt.step"]) and opt2.adam.lr == float(fx["lr"])
    with pytest.raises(FileNotFoundError):
        train.load_checkpoint(str(tmp_path / "missing.pth"))
    bad = _adam_state(fx)
    bad["state"][0]["step"] = torch.tensor(1.0)
    with pytest.raises(ValueError, match="one step count"):
        opt2.load_state_dict(bad)
    bad = _adam_state(fx)
    bad["param_groups"][0]["params"] = list(range(5))
    with pytest.raises(ValueError, match="expected one group"):
        opt2.load_state_dict(bad)


def test_one_epoch_over_the_dataset_classes(pk, tmp_path):
    """train_one_epoch / validate (train :964, :1143) driven by ThreeDMatchTrainVal files: finite losses, the weights move,
    and the correspondence loss of the validation pass falls over three epochs."""
    from se3_equi_graph_registration_amd import train
    from tests.test_gpu_datasets import _fragment_pair
    os.makedirs(tmp_path / "train_3dmatch")
    names = []
    for i in range(6):
        src, tgt, fs, ft, pose = _fragment_pair(40 + i, n=500)
        rec = pk.datasets.preprocess_pair(src, tgt, fs, ft, pose.astype(np.float32))
        pk.datasets.save_pair_file(str(tmp_path / "train_3dmatch" / f"{i}.pkl"), rec, "pickle-tensors")
        names.append(f"{i}.pkl")
    (tmp_path / "train_files.txt").write_text("\n".join(names[:4]) + "\n")
    (tmp_path / "val_files.txt").write_text("\n".join(names[4:]) + "\n")
    np.random.seed(0); random.seed(0); torch.manual_seed(0)
    mk = lambda split: torch.utils.data.DataLoader(pk.datasets.ThreeDMatchTrainVal(root=str(tmp_path), split=split, num_node=256),  # noqa: E731
                                                   batch_size=2, shuffle=False)
    egnn = pk.EGNN(32, 32, 32, in_edge_nf=1, device="cuda", n_layers=2)
    with torch.no_grad():
        egnn.embedding_out.weight.mul_(0.05); egnn.embedding_out.bias.mul_(0.05)
    model = pk.CrossAttentionPoseRegression(egnn, num_nodes=256, hidden_nf=32, device="cuda").cuda()
    opt = train.make_optimizer(model, lr=1e-3)
    before = egnn.flat_params.clone()
    v0 = train.validate(model, mk("val"), "cuda", 0, k=8)
    losses = [train.train_one_epoch(model, mk("train"), opt, "cuda", e, log_interval=0, k=8) for e in range(3)]
    v1 = train.validate(model, mk("val"), "cuda", 3, k=8)
    assert all(np.isfinite(losses)) and np.isfinite(v0[0]) and np.isfinite(v1[0])
    assert torch.isfinite(egnn.flat_params).all() and float((egnn.flat_params - before).abs().max()) > 1e-4
    assert float(opt.adam.step_dev[0]) == 6.0                       # 2 batches x 3 epochs
    assert v1[2] < v0[2]                                            # correspondence loss on held-out pairs
    assert model.training is False                                  # validate leaves the model in eval mode, like the reference
    # use_pointnet=True (train :1031-1038): a fresh PointNet(3, 32, 32) encodes the coordinates into the 32 input features
    torch.manual_seed(1)
    lp = train.train_one_epoch(model, mk("train"), opt, "cuda", 3, use_pointnet=True, log_interval=0, k=8)
    vp = train.validate(model, mk("val"), "cuda", 4, use_pointnet=True, k=8)
    assert np.isfinite(lp) and np.isfinite(vp[0]) and torch.isfinite(egnn.flat_params).all()
    b = next(iter(pk.datasets.DeviceBatches(mk("val"))))
    g = pk.build_pair_graph(b[2], b[3], 8)
    f0, f1 = train._pointnet_features(g, b[2], b[3])
    assert f0.shape == (2, 256, 32) and f1.shape == (2, 256, 32) and float(f0.min()) >= 0.0 and float(f0.max()) > 0.0   # ReLU output


def test_evaluate_model_matches_per_pair_oracle_metrics(pk, fx, tmp_path):
    """eval :1084-1296: checkpoint -> evaluation head, one pair at a time -> averages; every per-pair number against the
    oracle's restatement of tools/evaluation_metrics.py on the poses this model predicts."""
    from oracle import egnn_oracle as orc
    from se3_equi_graph_registration_amd import train
    from tests.test_gpu_datasets import _fragment_pair
    os.makedirs(tmp_path / "test_3dmatch")
    names = []
    for i in range(3):
        src, tgt, fs, ft, pose = _fragment_pair(70 + i, n=500)
        pk.datasets.save_pair_file(str(tmp_path / "test_3dmatch" / f"{i}.pkl"),
                                   pk.datasets.preprocess_pair(src, tgt, fs, ft, pose.astype(np.float32)), "pickle-arrays")
        names.append(f"{i}.pkl")
    (tmp_path / "test_files.txt").write_text("\n".join(names) + "\n")
    trained = _model(pk, fx)
    ck = train.save_checkpoint(3, None, trained.egnn, trained, train.make_optimizer(trained), save_dir=str(tmp_path / "ck"))
    egnn = pk.EGNN(32, 32, 32, in_edge_nf=1, device="cuda", n_layers=int(fx["n_layers"]), num_heads=int(fx["heads"]))
    model = pk.CrossAttentionPoseRegression(egnn, num_nodes=256, hidden_nf=32, device="cuda", variant="eval").cuda()
    mk = lambda: torch.utils.data.DataLoader(pk.datasets.ThreeDMatchTest(root=str(tmp_path), split="test", num_node=256),  # noqa: E731
                                             batch_size=1, shuffle=False)
    np.random.seed(1)
    avg = train.evaluate_model(ck, str(tmp_path / "res"), model, mk(), "cuda", k=8, verbose=False)
    assert torch.equal(model.mlp[0].weight, trained.mlp[0].weight)                       # the checkpoint was loaded
    np.random.seed(1)
    rows = []
    for corr, labels, s, t, fs_, ft_, gt in mk():
        g = pk.build_pair_graph(s.cuda(), t.cuda(), 8)
        R, tr, *_ = model(fs_.cuda(), s.cuda(), g, None, ft_.cuda(), t.cuda(), g, None, corr.cuda(), labels.cuda(), gt.cuda())
        pred = np.eye(4); pred[:3, :3] = R[0].detach().cpu().double().numpy(); pred[:3, 3] = tr[0].detach().cpu().double().numpy()
        re, te, rec, prec, f1 = orc.eval_metrics(gt[0].double().numpy(), pred, s[0].double().numpy(), t[0].double().numpy())
        rows.append([re, te, rec, f1])
    want = np.mean(np.array(rows), axis=0)
    got = [avg["Average Rotation Error"], avg["Average Translation Error"], avg["Average Recall"], avg["Average F1 Score"]]
    assert np.allclose(got, want, rtol=1e-4, atol=1e-4), (got, want)
    text = (tmp_path / "res" / "evaluation_results.txt").read_text().splitlines()
    assert text[0] == f"Average Rotation Error: {got[0]:.4f}" and len(text) == 4


def test_step_lr_scheduler_drives_the_flat_adam(pk, fx, tmp_path):
    """The reference trains with torch.optim.Adam + StepLR(optimizer, 200, 0.5) (train :1412-1413).  StepOptimizer is a
    torch.optim.Optimizer with one persistent param group: a scheduler (or a manual edit of param_groups) changes the applied
    update, shows in state_dict(), survives a checkpoint, and a non-finite step raises by default (the reference stops under
    set_detect_anomaly) or is dropped with skip_nonfinite=True."""
    from se3_equi_graph_registration_amd import train
    model = _model(pk, fx)
    model.train()
    opt = train.make_optimizer(model, lr=1e-3)
    assert isinstance(opt, torch.optim.Optimizer)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    batch = _batch(fx)

    def one_step():
        before = model.egnn.flat_params.clone()
        _, loss, _, _ = train._forward(model, batch, int(fx["k"]))
        loss.backward()
        opt.step()
        return (model.egnn.flat_params - before).abs().max().item()

    d1 = one_step()                       # first Adam step moves every stepped weight by ~lr
    assert 0.5e-3 < d1 < 1.5e-3
    sched.step()
    assert opt.param_groups[0]["lr"] == pytest.approx(0.5e-3) and opt.state_dict()["param_groups"][0]["lr"] == pytest.approx(0.5e-3)
    for g in opt.param_groups:            # the manual edit the docs of torch.optim describe
        g["lr"] = 1e-5
    d2 = one_step()
    assert d2 < 5e-5 and opt.adam.lr == pytest.approx(1e-5)
    path = train.save_checkpoint(1, None, model.egnn, model, opt, save_dir=str(tmp_path / "ck"))
    fresh = _model(pk, fx)
    opt2 = train.make_optimizer(fresh, lr=1.0)
    train.load_checkpoint(path, egnn=fresh.egnn, cross_attention=fresh, optimizer=opt2, device="cuda:0")     # weights_only=True
    assert opt2.param_groups[0]["lr"] == pytest.approx(1e-5) and float(opt2.adam.step_dev[0]) == 2.0
    # a batch without inliers and a perfect fit: the reference's arccos loss is not finite there
    bad = list(batch)
    bad[6] = bad[6].clone(); bad[6][0, :3, :3] = float("nan")
    with pytest.raises(RuntimeError, match="non-finite"):
        train.train_one_epoch(fresh, [tuple(t.cpu() for t in bad)], opt2, "cuda", 0, log_interval=0, k=int(fx["k"]))
    w = fresh.egnn.flat_params.clone()
    train.train_one_epoch(fresh, [tuple(t.cpu() for t in bad), None], opt2, "cuda", 0, log_interval=0, k=int(fx["k"]), skip_nonfinite=True)
    assert torch.equal(w, fresh.egnn.flat_params) and float(opt2.adam.step_dev[0]) == 2.0 and float(fresh.egnn.flat_grads.abs().max()) == 0.0
