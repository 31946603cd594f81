"""The typed irrep-basis layer on the GPU (csrc/irreps.hip behind irreps.IrrepConv): HIP forward against
the fp64 definition (oracle/irreps_oracle.py), the HIP backward against the definition's fp64 autograd for EVERY input (features,
coordinates, radial and self-interaction weights), SO(3) covariance of outputs and gradients of the layer and of a small stack at full
size on kNN graphs with self loops."""
import numpy as np
import pytest
import torch

from oracle import irreps_oracle as io, sh_oracle as so

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pk():
    import se3_equi_graph_registration_amd as pkg
    assert torch.cuda.is_available(), "these tests need the MI355X"
    pkg.lib()
    return pkg


def _graph(seed, n, e):
    g = torch.Generator().manual_seed(seed)
    row = torch.randint(0, n - 3, (e,), generator=g); col = torch.randint(0, n - 3, (e,), generator=g)
    row[:90] = 2                                          # a hub with > 64 incoming edges
    col[90:110] = row[90:110]                             # self loops
    return g, row, col


def _weights(layer):
    return {k: v.detach().cpu().double() for k, v in layer.state_dict().items()}


@pytest.mark.parametrize("dims", [(5, 3, 4, 6), (8, 8, 8, 8), (32, 8, 32, 8), (32, 16, 32, 32), (1, 1, 1, 1)])
def test_forward_matches_the_definition(pk, dims):
    n, e = 300, 4000
    g, row, col = _graph(11, n, e)
    torch.manual_seed(sum(dims))
    layer = pk.IrrepConv(*dims, hidden=32)
    x = torch.randn(n, 3, generator=g) * 1.5
    f0 = torch.randn(n, dims[0], generator=g); f1 = torch.randn(n, dims[1], 3, generator=g)
    with torch.no_grad():
        o0, o1 = layer(f0.cuda(), f1.cuda(), x.cuda(), [row.cuda(), col.cuda()])
    r0, r1 = io.irrep_conv(_weights(layer), f0.double(), f1.double(), x.double(), [row, col])
    scale = max(1.0, float(r0.abs().max()), float(r1.abs().max()))
    assert float((o0.cpu().double() - r0).abs().max()) <= 2e-5 * scale, dims
    assert float((o1.cpu().double() - r1).abs().max()) <= 2e-5 * scale, dims
    # bitwise reproducible (ordered aggregation, no atomics)
    with torch.no_grad():
        p0, p1 = layer(f0.cuda(), f1.cuda(), x.cuda(), [row.cuda(), col.cuda()])
    assert torch.equal(o0, p0) and torch.equal(o1, p1)


@pytest.mark.parametrize("dims,hidden", [((6, 4, 5, 7), 16), ((32, 8, 32, 8), 32), ((16, 16, 8, 12), 32), ((3, 20, 20, 3), 16), ((1, 1, 1, 1), 32)])
def test_backward_matches_the_definitions_autograd(pk, dims, hidden):
    """HIP backward (adjoint convolution, geometry kernel, weight-gradient GEMMs) against the fp64 autograd of the definition for EVERY
    input: f0, f1, the coordinates x, radial.0 / radial.2 weights and biases, both self-interaction matrices.  Graph with a hub
    (> 64 incoming edges), isolated nodes and self loops (r = 0: only the J = 0 kernels act, no gradient through the direction)."""
    n, e = 120, 1500
    g, row, col = _graph(4, n, e)
    torch.manual_seed(9 + sum(dims))
    layer = pk.IrrepConv(*dims, hidden=hidden)
    x = torch.randn(n, 3, generator=g)
    f0 = torch.randn(n, dims[0], generator=g); f1 = torch.randn(n, dims[1], 3, generator=g)
    g0 = torch.randn(n, dims[2], generator=g); g1 = torch.randn(n, dims[3], 3, generator=g)
    a0, a1, ax = f0.cuda().requires_grad_(True), f1.cuda().requires_grad_(True), x.cuda().requires_grad_(True)
    o0, o1 = layer(a0, a1, ax, [row.cuda(), col.cuda()])
    layer.zero_grad()
    ((o0 * g0.cuda()).sum() + (o1 * g1.cuda()).sum()).backward()
    w = {k: v.clone().requires_grad_(True) for k, v in _weights(layer).items()}
    b0, b1, bx = f0.double().requires_grad_(True), f1.double().requires_grad_(True), x.double().requires_grad_(True)
    r0, r1 = io.irrep_conv(w, b0, b1, bx, [row, col])
    ((r0 * g0.double()).sum() + (r1 * g1.double()).sum()).backward()
    assert torch.isfinite(bx.grad).all()
    for name, got, want in [("f0", a0.grad, b0.grad), ("f1", a1.grad, b1.grad), ("x", ax.grad, bx.grad)] + \
            [(k, p.grad, w[k].grad) for k, p in layer.named_parameters()]:
        assert got is not None, name
        want = want.reshape(got.shape)
        tol = 5e-5 * max(1.0, float(want.abs().max()))
        assert float((got.cpu().double() - want).abs().max()) <= tol, (name, float((got.cpu().double() - want).abs().max()), tol)
    # bitwise reproducible (no atomics; slabs summed in fixed order)
    first = [p.grad.clone() for p in layer.parameters()] + [a0.grad.clone(), a1.grad.clone(), ax.grad.clone()]
    for t in (a0, a1, ax):
        t.grad = None
    layer.zero_grad()
    o0, o1 = layer(a0, a1, ax, [row.cuda(), col.cuda()])
    ((o0 * g0.cuda()).sum() + (o1 * g1.cuda()).sum()).backward()
    second = [p.grad for p in layer.parameters()] + [a0.grad, a1.grad, ax.grad]
    assert all(torch.equal(u, v) for u, v in zip(first, second))


def test_gradients_are_so3_covariant_fullsize(pk):
    """2 x 2048 points, k = 20 (self loops included): for a rotation-invariant loss of the outputs the gradient with respect to the
    coordinates rotates with the cloud, the degree-1 feature gradient with D^1, the weight gradients are invariant."""
    b = pk.synthetic_batch(33, 1, n=2048, device="cuda")
    graph = pk.build_pair_graph(b["src"], b["tgt"], 20)
    x = torch.cat([b["src"][0], b["tgt"][0]]).contiguous()
    n = x.shape[0]
    g = torch.Generator().manual_seed(5)
    f0 = torch.randn(n, 32, generator=g).cuda()
    v = torch.randn(n, 8, 3, generator=g).cuda()
    t0 = torch.randn(n, 32, generator=g).cuda()
    rm = torch.from_numpy(so._rot(1.1, 0.6, 2.9)).float().cuda()
    torch.manual_seed(1)
    layer = pk.IrrepConv(32, 8, 32, 8)

    def grads(xx, vv):
        xx, vv, ff = xx.clone().requires_grad_(True), vv.clone().requires_grad_(True), f0.clone().requires_grad_(True)
        layer.zero_grad()
        o0, o1 = layer(ff, vv, xx, graph)
        ((o0 * t0).sum() + (o1 * o1).sum()).backward()          # invariant: a fixed scalar target on degree 0, the squared norm on degree 1
        return xx.grad, vv.grad, ff.grad, [p.grad.clone() for p in layer.parameters()]

    gx, gv, gf, gw = grads(x, v)
    # degree-1 features are Cartesian vectors up to the factor of Y_1 (A = c I): they rotate with the plain rotation matrix
    rx, rv, rf, rw = grads((x @ rm.T).contiguous(), torch.einsum("ab,nib->nia", rm, v).contiguous())
    sx, sv = float(gx.abs().max()), float(gv.abs().max())
    assert sx > 1e-3 and sv > 1e-3
    # fp32 run against fp32 run: two independently rounded evaluations (the 1 / d terms of close pairs dominate the coordinate gradient)
    assert float((rx - gx @ rm.T).abs().max()) <= 5e-4 * max(1.0, sx)
    assert float((rv - torch.einsum("ab,nib->nia", rm, gv)).abs().max()) <= 5e-4 * max(1.0, sv)
    assert float((rf - gf).abs().max()) <= 5e-4 * max(1.0, float(gf.abs().max()))
    for u, w_ in zip(gw, rw):
        assert float((u - w_).abs().max()) <= 5e-4 * max(1.0, float(u.abs().max()))


def test_layer_and_stack_are_so3_covariant_on_knn_graphs(pk):
    """Full size (2 x 2048 points, k = 20, self loops): degree-0 outputs invariant, degree-1 outputs rotate with the cloud."""
    b = pk.synthetic_batch(21, 1, n=2048, device="cuda")
    graph = pk.build_pair_graph(b["src"], b["tgt"], 20, need_col_index=False)
    x = torch.cat([b["src"][0], b["tgt"][0]]).contiguous()
    n = x.shape[0]
    g = torch.Generator().manual_seed(2)
    f0 = torch.randn(n, 32, generator=g).cuda()
    v = torch.randn(n, 8, 3, generator=g).cuda()
    rm = torch.from_numpy(so._rot(0.7, 1.9, 4.1)).float().cuda()
    a = pk.sh1_matrix()
    assert torch.allclose(a.cpu().double(), io.sh1_matrix(), atol=1e-6)
    d1 = a @ rm @ torch.linalg.inv(a)
    torch.manual_seed(0)
    layer = pk.IrrepConv(32, 8, 32, 8)
    net = pk.IrrepGNN(32, 32, 16, c1=8, n_layers=3)
    xr = (x @ rm.T).contiguous()
    graph_r = pk.build_pair_graph((b["src"][0] @ rm.T)[None].contiguous(), (b["tgt"][0] @ rm.T)[None].contiguous(), 20, need_col_index=False)
    same_graph = torch.equal(graph.row, graph_r.row)       # the rotated cloud's neighbour lists may differ in near-ties: use the original lists
    with torch.no_grad():
        for mod in (layer, net):
            o0, o1 = mod(f0, v, x, graph)
            p0, p1 = mod(f0, torch.einsum("ab,nib->nia", d1, v).contiguous(), xr, graph)
            s0, s1 = float(o0.abs().max()), float(o1.abs().max())
            assert s0 > 0.1 and s1 > 0.01
            assert float((p0 - o0).abs().max()) <= 3e-5 * max(1.0, s0), type(mod).__name__
            assert float((p1 - torch.einsum("ab,nob->noa", d1, o1)).abs().max()) <= 3e-5 * max(1.0, s1), type(mod).__name__
        h, vv = net(f0, None, x, graph)                    # degree-1 channels created from the geometry alone
        hr, vr = net(f0, None, xr, graph)
    assert float(vv.abs().max()) > 1e-3 and float((hr - h).abs().max()) <= 3e-5 * max(1.0, float(h.abs().max()))
    assert float((vr - torch.einsum("ab,nob->noa", d1, vv)).abs().max()) <= 3e-5 * max(1.0, float(vv.abs().max()))
    assert isinstance(same_graph, bool)


def test_unsupported_shapes_and_cpu_tensors_fail_loudly(pk):
    with pytest.raises(NotImplementedError):
        pk.IrrepConv(33, 8, 8, 8)
    with pytest.raises(NotImplementedError):
        pk.IrrepConv(8, 8, 32, 32 + 1)
    with pytest.raises(NotImplementedError):
        pk.IrrepConv(8, 8, 8, 8, hidden=24)
    layer = pk.IrrepConv(4, 4, 4, 4)
    with pytest.raises(RuntimeError):
        layer(torch.zeros(5, 4), torch.zeros(5, 4, 3), torch.zeros(5, 3), [torch.zeros(3, dtype=torch.long), torch.zeros(3, dtype=torch.long)])


@pytest.mark.parametrize("dims", [(32, 32, 32, 8, 3, 32), (12, 20, 8, 4, 2, 16), (30, 32, 27, 7, 2, 32)])
def test_irrep_egnn_matches_the_definition_forward_and_backward(pk, dims):
    """The typed network behind the EGNN interface (equigspr_irrep_gnn_forward / _backward: convolutions, gates, coordinate head in
    HIP launches, parameter gradients accumulated into the flat bucket) against the fp64 autograd of its definition: outputs, the
    gradients of h and x and of EVERY parameter; odd channel counts (30, 27, 7) exercise the 16-byte padding of the parameter block."""
    in_nf, hid, out_nf, c1, layers, rh = dims
    n, e = 150, 2000
    g, row, col = _graph(8, n, e)
    torch.manual_seed(3 + sum(dims))
    net = pk.IrrepEGNN(in_nf, hid, out_nf, n_layers=layers, c1=c1, radial_hidden=rh)
    with torch.no_grad():
        net.coord.weight.mul_(300.0)                       # a visible coordinate update
    x = torch.randn(n, 3, generator=g)
    h = torch.randn(n, in_nf, generator=g)
    th, tx = torch.randn(n, out_nf, generator=g), torch.randn(n, 3, generator=g)
    ah, ax = h.cuda().requires_grad_(True), x.cuda().requires_grad_(True)
    ho, xo = net(ah, ax, [row.cuda(), col.cuda()], None)
    net.zero_grad()
    ((ho * th.cuda()).sum() + (xo * tx.cuda()).sum()).backward()
    w = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    bh, bx = h.double().requires_grad_(True), x.double().requires_grad_(True)
    rh_, rx_ = io.irrep_egnn(w, bh, bx, [row, col], layers)
    ((rh_ * th.double()).sum() + (rx_ * tx.double()).sum()).backward()
    s = max(1.0, float(rh_.abs().max()))
    assert float((ho.detach().cpu().double() - rh_).abs().max()) <= 3e-5 * s
    assert float((xo.detach().cpu().double() - rx_).abs().max()) <= 3e-5 * max(1.0, float(rx_.abs().max()))
    for name, got, want in [("h", ah.grad, bh.grad), ("x", ax.grad, bx.grad)] + [(k, p.grad, w[k].grad) for k, p in net.named_parameters()]:
        assert got is not None and want is not None, name
        want = want.reshape(got.shape)
        tol = 1e-4 * max(1.0, float(want.abs().max()))
        assert float((got.cpu().double() - want).abs().max()) <= tol, (name, float((got.cpu().double() - want).abs().max()), tol)
    # the parameters are views into one flat block, their gradients views into one flat bucket
    assert all(p.data_ptr() >= net.flat_params.data_ptr() and p.data_ptr() < net.flat_params.data_ptr() + 4 * net.flat_params.numel()
               for p in net.parameters())
    assert float(net.flat_grads.abs().sum()) > 0


def test_irrep_egnn_is_se3_covariant_and_drops_into_the_pair_model(pk):
    """h invariant, x covariant under rotation + translation at 2 x 2048 points, k = 20; and the network replaces the frame-vector EGNN
    inside CrossAttentionPoseRegression: one training step (forward, fused head, backward, flat Adam) runs and changes the weights."""
    from se3_equi_graph_registration_amd import train
    b = pk.synthetic_batch(5, 1, n=2048, device="cuda")
    graph = pk.build_pair_graph(b["src"], b["tgt"], 20)
    x = torch.cat([b["src"][0], b["tgt"][0]]).contiguous()
    h = torch.cat([b["feat_src"][0], b["feat_tgt"][0]]).contiguous()
    torch.manual_seed(0)
    net = pk.IrrepEGNN(32, 32, 32, n_layers=4, c1=8)
    with torch.no_grad():
        net.coord.weight.mul_(300.0)
    rm = torch.from_numpy(so._rot(0.4, 2.2, 5.0)).float().cuda()
    t = torch.tensor([0.3, -1.1, 0.7], device="cuda")
    with torch.no_grad():
        h1, x1 = net(h, x, graph, None)
        h2, x2 = net(h, (x @ rm.T + t).contiguous(), graph, None)
    assert float((x1 - x).abs().max()) > 1e-3
    assert float((h2 - h1).abs().max()) <= 5e-5 * max(1.0, float(h1.abs().max()))
    assert float((x2 - (x1 @ rm.T + t)).abs().max()) <= 5e-5 * max(1.0, float(x1.abs().max()))
    # inside the pair model
    torch.manual_seed(1)
    egnn = pk.IrrepEGNN(32, 32, 32, n_layers=2, c1=8)
    model = pk.CrossAttentionPoseRegression(egnn, num_nodes=2048, hidden_nf=32, device="cuda").cuda()
    opt = train.make_optimizer(model, lr=1e-3)
    before = egnn.flat_params.clone()
    bb = pk.synthetic_batch(77, 2, n=2048)
    batches = [(bb["corr"], bb["labels"], bb["src"], bb["tgt"], bb["feat_src"], bb["feat_tgt"], bb["gt_pose"])] * 3
    loss = train.train_one_epoch(model, batches, opt, "cuda", 0, log_interval=0, k=20, skip_nonfinite=True)
    st = opt.adam.guard_stats()
    assert st["applied"] + st["skipped"] == 3 and st["applied"] >= 1 and loss == loss
    assert not torch.equal(before, egnn.flat_params)
    gs = train.graphed_step_for(model, opt, k=20)
    assert gs.steps_replayed >= 1, (gs.graph_mode, gs.steps_eager)


def test_torch_library_irrep_conv_ops_match_the_ctypes_path(pk):
    """torch.ops.equigspr.irrep_conv_forward (autograd registered in csrc/torch_ops.cpp) and the explicit irrep_conv_backward op
    call the same C-ABI entries as irreps.IrrepConv: bit-identical outputs and gradients for every input; argument checks raise."""
    from se3_equi_graph_registration_amd import sh, torch_library
    ops = torch_library.load()
    n, e = 200, 2500
    g, row, col = _graph(21, n, e)
    torch.manual_seed(3)
    layer = pk.IrrepConv(12, 5, 9, 6, hidden=32)
    x = torch.randn(n, 3, generator=g).cuda()
    f0 = torch.randn(n, 12, generator=g).cuda(); f1 = torch.randn(n, 5, 3, generator=g).cuda()
    g0 = torch.randn(n, 9, generator=g).cuda(); g1 = torch.randn(n, 6, 3, generator=g).cuda()
    row32, col32 = row.int().cuda(), col.int().cuda()
    basis = [float(c) for c in sh.degree1_closed_form()]
    lin1, lin2 = layer.radial[0], layer.radial[2]
    names = ["radial.0.weight", "radial.0.bias", "radial.2.weight", "radial.2.bias", "self0.weight", "self1.weight"]
    params = [lin1.weight, lin1.bias, lin2.weight, lin2.bias, layer.self0.weight, layer.self1.weight]
    a0, a1, ax = f0.clone().requires_grad_(True), f1.clone().requires_grad_(True), x.clone().requires_grad_(True)
    o0, o1 = layer(a0, a1, ax, [row.cuda(), col.cuda()])
    layer.zero_grad()
    ((o0 * g0).sum() + (o1 * g1).sum()).backward()
    want = {k: p.grad.clone() for k, p in zip(names, params)}
    b0, b1, bx = f0.clone().requires_grad_(True), f1.clone().requires_grad_(True), x.clone().requires_grad_(True)
    q = [p.detach().clone().requires_grad_(True) for p in params]
    p0, p1, moments = ops.irrep_conv_forward(b0, b1, bx, row32, col32, *q, basis)
    assert torch.equal(p0, o0) and torch.equal(p1, o1) and not moments.requires_grad
    ((p0 * g0).sum() + (p1 * g1).sum()).backward()
    assert torch.equal(b0.grad, a0.grad) and torch.equal(b1.grad, a1.grad) and torch.equal(bx.grad, ax.grad)
    for k, t in zip(names, q):
        assert torch.equal(t.grad, want[k]), k
    d = ops.irrep_conv_backward(f0, f1, x, row32, col32, *[p.detach() for p in params], basis, moments, g0, g1)
    assert len(d) == 9 and torch.equal(d[0], a0.grad) and torch.equal(d[2], ax.grad) and torch.equal(d[5], want["radial.2.weight"])
    with pytest.raises(RuntimeError, match="basis"):
        ops.irrep_conv_forward(f0, f1, x, row32, col32, *[p.detach() for p in params], basis[:5])
    with pytest.raises(RuntimeError):
        ops.irrep_conv_forward(f0, f1[:, :4].contiguous(), x, row32, col32, *[p.detach() for p in params], basis)
    with pytest.raises(RuntimeError, match="moments"):
        ops.irrep_conv_backward(f0, f1, x, row32, col32, *[p.detach() for p in params], basis, moments[:-1].contiguous(), g0, g1)
