"""GPU: edge_basis="irreps"  -- irrep-basis invariants as the edge MLP's geometric inputs, computed inside
the edge kernels (csrc/egnn_math.h: edge_geometry_irreps).  No reference layer uses tools/basis.py, so the mode is pinned by
  * the oracle's definition through the A16 restatement (oracle.egnn_oracle.edge_geometry_irreps = contraction of
    get_basis kernels): forward and every gradient, fp32/fp64-anchored like the rest of the parity suite;
  * the property the mode exists for: SO(3) equivariance of the layer outputs, h(Rx) = h(x) and x'(Rx) = R x'(x), at the
    BASELINE configs[2] size (2048 points, k = 20, 4 layers) -- which the reference's frame-vector features do not have."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import egnn_oracle as orc  # noqa: E402  (checker only)
from tests.test_gpu_parity import anchored_tol, _grad_check  # noqa: E402


@pytest.fixture(scope="module")
def pk():
    import se3_equi_graph_registration_amd as pkg
    assert torch.cuda.is_available(), "these tests need the MI355X"
    pkg.lib()
    return pkg


@pytest.mark.parametrize("feat,heads", [(32, 1), (33, 1), (32, 4)])
def test_irreps_forward_backward_vs_oracle(pk, feat, heads, parity):
    g = torch.Generator().manual_seed(31 + feat)
    n, k, layers = 300, 8, 3
    x = torch.rand(n, 3, generator=g) * 2 - 1
    h = torch.randn(n, feat, generator=g)
    torch.manual_seed(7)
    m = pk.EGNN(feat, feat, feat, in_edge_nf=1, device="cuda", n_layers=layers, num_heads=heads, edge_basis="irreps")
    e = orc.knn_graph(x, k)
    ea = torch.ones(e.shape[1], 1)
    gh, gx = torch.randn(n, feat, generator=g), torch.randn(n, 3, generator=g)
    refs = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        w = {kk: v.detach().cpu().to(dt).clone().requires_grad_(True) for kk, v in m.state_dict().items()}
        hh, xx = h.to(dt).clone().requires_grad_(True), x.to(dt).clone().requires_grad_(True)
        ho, xo = orc.egnn_forward(w, hh, xx, [e[0], e[1]], ea.to(dt), edge_basis="irreps")
        ((ho * gh.to(dt)).sum() + (xo * gx.to(dt)).sum()).backward()
        refs[tag] = (ho.detach(), xo.detach(), {kk: v.grad for kk, v in w.items()}, hh.grad, xx.grad)
    hh, xx = h.cuda().requires_grad_(True), x.cuda().requires_grad_(True)
    ho, xo = m(hh, xx, [e[0].cuda(), e[1].cuda()], ea.cuda())
    name = f"irreps_n{n}_k{k}_l{layers}_f{feat}_h{heads}"
    for got, idx, key in ((ho, 0, "h_out"), (xo, 1, "x_out")):
        a32, a64 = refs["f32"][idx].numpy(), refs["f64"][idx].numpy()
        err, _ = parity(name, "irreps." + key, got, a64, a32)
        assert err <= anchored_tol(a32, a64), (key, err)
    m.zero_grad()
    ((ho * gh.cuda()).sum() + (xo * gx.cuda()).sum()).backward()
    _grad_check(hh.grad, refs["f32"][3].numpy(), refs["f64"][3].numpy(), "irreps.grad_h", parity=parity, fixture=name)
    _grad_check(xx.grad, refs["f32"][4].numpy(), refs["f64"][4].numpy(), "irreps.grad_x", parity=parity, fixture=name)
    for pn, p in m.named_parameters():
        _grad_check(p.grad, refs["f32"][2][pn].numpy(), refs["f64"][2][pn].numpy(), "irreps.grad." + pn, k=8.0, rel=5e-5, parity=parity,
                    fixture=name)


def test_irreps_layer_outputs_are_rotation_covariant_fullsize(pk, parity):
    """configs[2] shapes (3DMatch clouds, 2048 points, k = 20, 4 layers, batch of clouds): rotate every cloud about the origin;
    the graph is unchanged (distances are), h is invariant and the updated coordinates rotate with the input.  The reference's
    frame mode fails the same check by orders of magnitude."""
    from scipy.spatial.transform import Rotation
    clouds, n, k, layers, feat = 4, 2048, 20, 4, 32
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(clouds, n, 3, generator=g) * 3.0 - 1.5).cuda()
    h = torch.randn(clouds, n, feat, generator=g).cuda()
    h = h / h.norm(dim=-1, keepdim=True)
    R = torch.from_numpy(np.stack([Rotation.random(random_state=10 + c).as_matrix() for c in range(clouds)])).float().cuda()
    xr = torch.einsum("cij,cnj->cni", R, x).contiguous()
    edges, _, _ = pk.ops.knn_graph_batched(x, k, loop=True, want_i64=True, want_i32=False)
    graph = pk.ops.DeviceGraph.from_edge_lists([edges[c] for c in range(clouds)], n)     # the SAME edges for both runs
    out = {}
    for basis in ("irreps", "frames"):
        torch.manual_seed(1)
        m = pk.EGNN(feat, feat, feat, in_edge_nf=1, device="cuda", n_layers=layers, edge_basis=basis)
        with torch.no_grad():
            h0, x0 = m(h.reshape(-1, feat), x.reshape(-1, 3), graph, pk.egnn.EGNN_ONES)
            h1, x1 = m(h.reshape(-1, feat), xr.reshape(-1, 3), graph, pk.egnn.EGNN_ONES)
        x0r = torch.einsum("cij,cnj->cni", R, x0.view(clouds, n, 3)).reshape(-1, 3)
        scale_h, scale_x = float(h0.abs().max()), float(x0.abs().max())
        out[basis] = (float((h1 - h0).abs().max()) / scale_h, float((x1 - x0r).abs().max()) / scale_x, float((x0 - x.reshape(-1, 3)).abs().max()))
    parity("c3_clouds_n2048_k20_l4", "irreps.rotation_covariance_rel_err_h", torch.tensor(out["irreps"][0]), torch.tensor(0.0),
           note="max|h(Rx)-h(x)| / max|h|, fp32 kernels")
    parity("c3_clouds_n2048_k20_l4", "irreps.rotation_covariance_rel_err_x", torch.tensor(out["irreps"][1]), torch.tensor(0.0),
           note="max|x'(Rx)-R x'(x)| / max|x'|")
    assert out["irreps"][2] > 1e-6                                   # the coordinate update is not trivially zero
    assert out["irreps"][0] < 2e-5 and out["irreps"][1] < 2e-5, out["irreps"]
    assert out["frames"][0] > 1e-3, out["frames"]                    # the reference's features are NOT rotation covariant


def test_irreps_pair_model_trains_and_rejects_unknown_basis(pk):
    with pytest.raises(ValueError):
        pk.EGNN(32, 32, 32, in_edge_nf=1, device="cuda", n_layers=1, edge_basis="wigner")
    torch.manual_seed(0)
    egnn = pk.EGNN(32, 32, 32, in_edge_nf=1, device="cuda", n_layers=2, edge_basis="irreps")
    with torch.no_grad():
        egnn.embedding_out.weight.mul_(0.02); egnn.embedding_out.bias.mul_(0.02)
    model = pk.CrossAttentionPoseRegression(egnn, num_nodes=512, hidden_nf=32, device="cuda").cuda()
    b = pk.synthetic_batch(11, 2, n=512, feat=32, device="cuda")
    b["tgt"] = b["tgt"] + 0.05 * torch.randn(b["tgt"].shape, generator=torch.Generator().manual_seed(2)).cuda()
    graph = pk.build_pair_graph(b["src"], b["tgt"], 12)
    out = model(b["feat_src"], b["src"], graph, None, b["feat_tgt"], b["tgt"], graph, None, b["corr"], b["labels"], b["gt_pose"])
    loss, _, _ = pk.registration_loss(out[0], out[1], out[2], b["gt_pose"])
    loss.backward()
    assert bool(torch.isfinite(egnn.flat_grads).all()) and float(egnn.flat_grads.abs().max()) > 0
