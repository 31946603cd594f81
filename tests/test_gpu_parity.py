"""GPU parity tests (run with ``-m gpu`` on the MI355X box): the HIP path, called through the
C-ABI (ctypes) from the reference-shaped Python interface, against
  * the committed golden vectors produced by the real reference (tests/golden/*.npz), and
  * the CPU oracle (oracle/egnn_oracle.py) on seeded inputs.

Tolerances (north_star: R/t within 1e-4, logits within 1e-5):
  * indices (kNN, inverse index, top-k sets): bit-exact;
  * R, t: |hip - ref64| <= 1e-4 absolute;
  * logits: |hip - ref64| <= 1e-5 ABSOLUTE wherever the reference's own fp32 run is within 5e-6 of its fp64 run on that
    fixture (the raw north_star bound); where the reference's fp32 arithmetic itself is further from fp64 than that (up to
    4.7e-5 on these inputs, gen_golden.py prints the gaps) the check is anchored on fp64:
    |hip - ref64| <= 4*|ref32 - ref64|_max + 1e-5 * max(1, |ref64|_max);
  * features: the anchored form (|h| reaches 20-180, so 1e-5 absolute is below one fp32 ulp there).
Every compared tensor's max|hip - ref64|, max|hip - ref32| and the reference's own gap go to gpurun_out/parity_report.json
(conftest.ParityRecorder; committed per round under profiles/).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import egnn_oracle as orc  # noqa: E402  (checker only)

LAYER_FIX = ["layer_n128_k8_f32_h1.npz", "layer_n128_k8_f33_h1.npz", "layer_n128_k8_f32_h4.npz",
             "layer_n96_k6_f32_h1_dup.npz", "layer_n128_k8_f32_h2.npz", "layer_n128_k8_f32_h8.npz",
             "layer_n96_k6_f64_h2.npz", "layer_n96_k6_f16_h4.npz"]
PAIR_FIX = ["pair_n2048_k20_l4.npz", "pair_n512_k12_l2_wellcond.npz", "pair_ckpt_n2048_k16_l3_h4.npz"]


@pytest.fixture(scope="module")
def pk():
    import se3_equi_graph_registration_amd as pkg
    assert torch.cuda.is_available(), "these tests need the MI355X"
    pkg.lib()          # fail loudly if the HIP extension is missing
    return pkg


def anchored_tol(ref32, ref64, k=4.0, rel=1e-5):
    gap = float(np.abs(ref32.astype(np.float64) - ref64).max())
    return k * gap + rel * max(1.0, float(np.abs(ref64).max()))


NORTH_STAR_LOGITS, NORTH_STAR_RT = 1e-5, 1e-4


def logits_tol(ref32, ref64):
    """north_star's absolute 1e-5 where the reference's own fp32 run resolves it (gap <= 5e-6), else the anchored bound."""
    gap = float(np.abs(ref32.astype(np.float64) - ref64).max())
    return NORTH_STAR_LOGITS if gap <= 5e-6 else anchored_tol(ref32, ref64)


def build_egnn(pk, fx, prefix="w."):
    w = {k[len(prefix):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(prefix)}
    feat = w["embedding_in.weight"].shape[1]
    hid = w["embedding_in.weight"].shape[0]
    heads, layers = int(fx["heads"]), int(fx["n_layers"])
    m = pk.EGNN(feat, hid, w["embedding_out.weight"].shape[0], in_edge_nf=1, device="cuda", n_layers=layers, num_heads=heads)
    missing = m.load_state_dict(w, strict=True)
    return m


# ------------------------------------------------------------------------------------ kNN
@pytest.mark.parametrize("n,k,loop", [(2048, 20, True), (2048, 16, True), (100, 8, True), (100, 8, False), (64, 64, True),
                                      (4096, 20, True), (8192, 32, True), (300, 1, True), (513, 33, False)])
def test_knn_matches_oracle_bit_exact(pk, n, k, loop):
    g = torch.Generator().manual_seed(n * 131 + k)
    x = torch.rand(n, 3, generator=g) * 3.0
    ref = orc.knn_graph(x, k, loop=loop)
    got = pk.knn_graph(x.cuda(), k, loop=loop)
    assert got.dtype == torch.int64 and tuple(got.shape) == (2, n * k)
    assert torch.equal(got.cpu(), ref)
    slow = pk.knn_graph(x.cuda(), k, loop=loop, force_slow_path=True)
    assert torch.equal(slow.cpu(), ref)


def test_knn_duplicates_grid_and_batched(pk):
    # duplicated points (ties at distance 0 and at the k-th distance), integer grid (massive ties)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(500, 3, generator=g)
    x[250:] = x[:250]
    for k in (4, 20):
        assert torch.equal(pk.knn_graph(x.cuda(), k, loop=True).cpu(), orc.knn_graph(x, k, loop=True))
    grid = torch.stack(torch.meshgrid(torch.arange(8.), torch.arange(8.), torch.arange(8.), indexing="ij"), -1).reshape(-1, 3)
    for k in (7, 27, 40):
        assert torch.equal(pk.knn_graph(grid.cuda(), k, loop=True).cpu(), orc.knn_graph(grid, k, loop=True))
    same = torch.ones(130, 3)                      # every candidate ties: exercises the slow path fallback
    assert torch.equal(pk.knn_graph(same.cuda(), 9, loop=True).cpu(), orc.knn_graph(same, 9, loop=True))
    # the tiled multi-query kernel (n > 2048): duplicated points, an integer grid (massive ties -> its slow path), a cloud that
    # is not a multiple of the 16-query workgroup or of the 2048-point tile, and a batched launch
    big = torch.rand(2500, 3, generator=g)
    big[1250:] = big[:1250]
    for k, loop in ((20, True), (5, False)):
        assert torch.equal(pk.knn_graph(big.cuda(), k, loop=loop).cpu(), orc.knn_graph(big, k, loop=loop))
    grid2 = torch.stack(torch.meshgrid(torch.arange(6.), torch.arange(20.), torch.arange(20.), indexing="ij"), -1).reshape(-1, 3)   # 2400
    for k in (7, 27):
        assert torch.equal(pk.knn_graph(grid2.cuda(), k, loop=True).cpu(), orc.knn_graph(grid2, k, loop=True))
    xb2 = torch.rand(2, 2051, 3, generator=g)
    e64b, _, _ = pk.knn_graph_batched(xb2.cuda(), 9, True, want_i64=True, want_i32=False)
    for b in range(2):
        assert torch.equal(e64b[b].cpu(), orc.knn_graph(xb2[b], 9, loop=True))
    # batched launch == per-cloud calls, and the int32 device format agrees with the int64 lists
    xb = torch.rand(3, 256, 3, generator=g)
    e64, r32, c32 = pk.knn_graph_batched(xb.cuda(), 12, True, want_i64=True, want_i32=True)
    for b in range(3):
        ref = orc.knn_graph(xb[b], 12, loop=True)
        assert torch.equal(e64[b].cpu(), ref)
        assert torch.equal(r32.view(3, -1)[b].cpu().long(), ref[0] + 256 * b)
        assert torch.equal(c32.view(3, -1)[b].cpu().long(), ref[1] + 256 * b)


@pytest.mark.parametrize("n,k,loop", [(300, 100, True), (2048, 100, True), (130, 129, False), (9000, 8, True), (8200, 70, False)])
def test_knn_beyond_the_tuned_kernels_matches_oracle_bit_exact(pk, n, k, loop):
    """k > 64 (the replaced torch_cluster call takes any k: k = 100 in the verdict's list) and n > 8 192 points per cloud: the streaming
    kernel, same (d2, index) order bit for bit; duplicated points for the ties; the device graph builds on it like on the tuned ones."""
    g = torch.Generator().manual_seed(n * 17 + k)
    x = torch.rand(n, 3, generator=g) * 3.0
    x[n // 2: n // 2 + 40] = x[:40]
    ref = orc.knn_graph(x, k, loop=loop)
    got = pk.knn_graph(x.cuda(), k, loop=loop)
    assert got.dtype == torch.int64 and tuple(got.shape) == (2, n * k) and torch.equal(got.cpu(), ref)
    if loop:
        dg = pk.build_pair_graph(x.cuda()[None], x.cuda()[None], k, loop=True)
        e = n * k
        assert torch.equal(dg.row[:e].cpu().long(), ref[0]) and torch.equal(dg.col[:e].cpu().long(), ref[1])
        assert torch.equal(dg.row[e:].cpu().long(), ref[0] + n)
        np.testing.assert_array_equal(dg.rperm.cpu().numpy()[:2 * e], np.argsort(dg.row.cpu().numpy(), kind="stable"))


def test_knn_graph_batch_vector(pk):
    """`knn_graph(x, k, batch=...)` (src/gcnn.py:31; train :1473-1474): a sorted batch vector of equal-sized clouds = one launch of the
    batched kernel, global node ids, cloud after cloud (torch_cluster's layout); a ragged batch = one launch per cloud; unsorted raises."""
    g = torch.Generator().manual_seed(11)
    nb, n, k = 3, 257, 9
    x = torch.rand(nb * n, 3, generator=g)
    batch = torch.arange(nb).repeat_interleave(n)
    got = pk.knn_graph(x.cuda(), k, batch=batch.cuda(), loop=True).cpu()
    ref = torch.cat([orc.knn_graph(x[b * n:(b + 1) * n], k, loop=True) + b * n for b in range(nb)], dim=1)
    assert got.dtype == torch.int64 and torch.equal(got, ref)
    assert bool((batch[got[0]] == batch[got[1]]).all())                       # no edge crosses clouds
    one = pk.knn_graph(x[:n].cuda(), k, batch=torch.zeros(n, dtype=torch.long).cuda(), loop=False).cpu()
    assert torch.equal(one, orc.knn_graph(x[:n], k, loop=False))
    rag = pk.knn_graph(x[:2 * n - 7].cuda(), k, batch=batch[:2 * n - 7].cuda(), loop=True).cpu()     # ragged: 257 + 250 points, a launch per cloud
    ref_rag = torch.cat([orc.knn_graph(x[:n], k, loop=True), orc.knn_graph(x[n:2 * n - 7], k, loop=True) + n], dim=1)
    assert torch.equal(rag, ref_rag)
    with pytest.raises(RuntimeError):
        pk.knn_graph(x.cuda(), k, batch=batch.flip(0).cuda(), loop=True)                       # not sorted
    with pytest.raises(RuntimeError):
        pk.knn_graph(x.cuda(), k, batch=batch[:-1].cuda(), loop=True)


def test_knn_argument_errors(pk):
    x = torch.rand(10, 3).cuda()
    with pytest.raises(RuntimeError):
        pk.knn_graph(x, 11, loop=True)
    with pytest.raises(RuntimeError):
        pk.knn_graph(x, 10, loop=False)
    with pytest.raises(RuntimeError):
        pk.knn_graph(torch.rand(10, 3), 3, loop=True)          # CPU tensor: no CPU path


def test_inverse_index_is_stable_counting_sort(pk):
    g = torch.Generator().manual_seed(3)
    n, e = 700, 9000
    row = torch.randint(0, n, (e,), generator=g)
    col = torch.randint(0, n, (e,), generator=g)
    row[:50] = 5                                                # a hub
    dg = pk.DeviceGraph.from_edge_lists([torch.stack([row, col]).cuda()], n)
    for key, ptr, perm in ((row, dg.rptr, dg.rperm), (col, dg.cptr, dg.cperm)):
        order = np.argsort(key.numpy(), kind="stable")
        cnt = np.bincount(key.numpy(), minlength=n)
        np.testing.assert_array_equal(ptr.cpu().numpy(), np.concatenate([[0], np.cumsum(cnt)]))
        np.testing.assert_array_equal(perm.cpu().numpy()[:e], order)
    inv = np.empty(e, dtype=np.int64)
    inv[np.argsort(row.numpy(), kind="stable")] = np.arange(e)
    np.testing.assert_array_equal(dg.rinv.cpu().numpy()[:e], inv)    # written by the pass that orders the segments


def test_knn_two_keys_per_lane_path_and_hub_segments_fullsize(pk):
    """Distinct synthetic pairs exposed a cliff the single fixed bench pair never hit (round 4): queries whose tau leaves 65..128
    candidates (k = 32 of 64 lane minima: a few per 8 192-point cloud, in three of eight seeds) took the slow path -- 0.6 ms each.
    The two-keys-per-lane ranking must reproduce the exact slow path bit for bit on the very seeds that showed it (hub segments of
    the index build: test_build_csr_scratch_contract_and_inverse)."""
    from se3_equi_graph_registration_amd import ops
    for seed, n, k in ((1236, 8192, 32), (1238, 8192, 32), (1239, 4096, 20)):
        b = pk.synthetic_batch(seed, 1, n=n, device="cuda")
        x = torch.cat([b["src"], b["tgt"]]).contiguous()
        _, r_fast, c_fast = ops.knn_graph_batched(x, k, True, want_i64=False, want_i32=True)
        _, r_slow, c_slow = ops.knn_graph_batched(x, k, True, want_i64=False, want_i32=True, force_slow_path=True)
        assert torch.equal(r_fast, r_slow) and torch.equal(c_fast, c_slow), (seed, n, k)
        graph = pk.build_pair_graph(b["src"], b["tgt"], k)
        row = graph.row.cpu().numpy()
        order = np.argsort(row, kind="stable")
        np.testing.assert_array_equal(graph.rperm.cpu().numpy(), order)


def test_knn_tiled_chunk_pruned_pass_shapes(pk):
    """The tiled kernel's second pass only visits the 16-candidate chunks whose minimum is <= tau (round 4): clouds that are not
    multiples of the tile / chunk / workgroup sizes, both list capacities (k <= 24, k > 24), both tile counts (n <= 4 096, <= 8 192),
    no self loops, spatially SORTED points (neighbours in consecutive indices) and the 8-queries-per-wave variant of large batches."""
    from se3_equi_graph_registration_amd import ops
    g = torch.Generator().manual_seed(11)
    for n, k, loop in ((2049, 5, True), (3000, 24, False), (4097, 25, True), (5000, 20, True), (6160, 33, False), (8191, 32, True)):
        x = torch.rand(n, 3, generator=g) * 2.0
        assert torch.equal(pk.knn_graph(x.cuda(), k, loop=loop).cpu(), orc.knn_graph(x, k, loop=loop)), (n, k, loop)
    x = torch.rand(4500, 3, generator=g)
    x = x[torch.argsort(x[:, 0])].contiguous()                       # sorted along x
    assert torch.equal(pk.knn_graph(x.cuda(), 20, loop=True).cpu(), orc.knn_graph(x, 20, loop=True))
    xb = torch.rand(3, 2500, 3, generator=g)
    xb[1, 1250:] = xb[1, :1250]                                       # duplicates inside one cloud of a batch
    e64, _, _ = pk.knn_graph_batched(xb.cuda(), 12, True, want_i64=True, want_i32=False)
    for b in range(3):
        assert torch.equal(e64[b].cpu(), orc.knn_graph(xb[b], 12, loop=True))
    for clouds, n, k in ((16, 4096, 20), (20, 4000, 30), (8, 8192, 32), (10, 5000, 7)):      # >= 2 048 32-query workgroups: 8 queries per wave
        xw = torch.rand(clouds, n, 3, generator=g).cuda()
        _, r_fast, c_fast = ops.knn_graph_batched(xw, k, True, want_i64=False, want_i32=True)
        _, r_slow, c_slow = ops.knn_graph_batched(xw, k, True, want_i64=False, want_i32=True, force_slow_path=True)
        assert torch.equal(r_fast, r_slow) and torch.equal(c_fast, c_slow), (clouds, n, k)
        ref = orc.knn_graph(xw[clouds - 1].cpu(), k, loop=True)
        assert torch.equal(r_fast.view(clouds, -1)[clouds - 1].cpu().long(), ref[0] + n * (clouds - 1))


def test_index_build_with_the_scan_inside_the_fill(pk):
    """Graphs of <= 8 192 nodes from the k-NN path take the fused scan + ranked fill launch (round 4): sizes around its per-thread
    chunking (1 .. 32 counts per thread), exactly 8 192 nodes, and just above (the two-launch path) must all give the stable
    counting sort of the edges by row, its inverse, and a scratch handed back zeroed (a second build on the same buffers)."""
    for n, k in ((64, 8), (100, 7), (129, 16), (300, 20), (2048, 20), (4096, 20), (4100, 12)):
        b = pk.synthetic_batch(77 + n, 1, n=n, device="cuda")
        for rep in range(2):
            graph = pk.build_pair_graph(b["src"], b["tgt"], k)
            row = graph.row[:graph.n_edges].cpu().numpy()
            order = np.argsort(row, kind="stable")
            np.testing.assert_array_equal(graph.rperm[:graph.n_edges].cpu().numpy(), order)
            cnt = np.bincount(row, minlength=graph.n_nodes)
            np.testing.assert_array_equal(graph.rptr.cpu().numpy(), np.concatenate([[0], np.cumsum(cnt)]))
            inv = np.empty(graph.n_edges, dtype=np.int64)
            inv[order] = np.arange(graph.n_edges)
            np.testing.assert_array_equal(graph.rinv[:graph.n_edges].cpu().numpy(), inv)


def test_build_csr_scratch_contract_and_inverse(pk):
    """ABI v3: build_csr also writes inv (same values as equigspr_invert_perm), accepts a scratch declared zero, and
    hands it back all-zero -- including hub segments longer than a wave (serial path) and isolated nodes."""
    import ctypes
    from se3_equi_graph_registration_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    n, e = 1500, 20000
    key = torch.randint(0, n - 100, (e,), generator=g)             # the last 100 nodes stay empty
    key[:300] = 7                                                   # a 300-entry segment (hub: ranked from LDS by the whole wave)
    key[300:365] = 9                                                # 65: just past one wave
    key[400:1424] = 11                                              # 1024: the largest segment of the LDS path ...
    key[1500:2600] = 13                                             # ... and 1100: beyond it (serial last resort)
    key32 = key.to(torch.int32).cuda()
    order = np.argsort(key.numpy(), kind="stable")
    inv_ref = np.empty(e, dtype=np.int64)
    inv_ref[order] = np.arange(e)
    scratch = torch.full((n + 1,), 12345, dtype=torch.int32, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    for is_zero in (0, 1, 1):                                       # first call clears the dirty buffer, the next two trust it
        ptr = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        perm = torch.empty(e, dtype=torch.int32, device="cuda")
        inv = torch.empty(e, dtype=torch.int32, device="cuda")
        _lib.check(L.equigspr_build_csr(p(key32), e, n, p(ptr), p(perm), p(inv), p(scratch), is_zero, None, ops._stream()), "build_csr")
        np.testing.assert_array_equal(perm.cpu().numpy(), order)
        np.testing.assert_array_equal(inv.cpu().numpy(), inv_ref)
        np.testing.assert_array_equal(ptr.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(key.numpy(), minlength=n))]))
        assert int(scratch.abs().sum()) == 0
        inv2 = torch.empty_like(inv)
        _lib.check(L.equigspr_invert_perm(p(perm), e, p(inv2), ops._stream()), "invert_perm")
        assert torch.equal(inv, inv2)
    # inv = NULL is allowed; the cached scratch of DeviceGraph survives many builds of different graphs
    _lib.check(L.equigspr_build_csr(p(key32), e, n, p(ptr), p(perm), None, p(scratch), 1, None, ops._stream()), "build_csr")
    np.testing.assert_array_equal(perm.cpu().numpy(), order)
    for seed in range(4):
        gg = torch.Generator().manual_seed(seed)
        x = torch.rand(2, 600, 3, generator=gg).cuda()
        graph = pk.build_pair_graph(x[:1], x[1:], 9)
        row = graph.row.cpu().numpy()
        np.testing.assert_array_equal(graph.rperm.cpu().numpy(), np.argsort(row, kind="stable"))
        assert all(int(b.abs().sum()) == 0 for b in ops._CSR_SCRATCH.values())
        inv = np.empty(row.size, dtype=np.int64)
        inv[np.argsort(row, kind="stable")] = np.arange(row.size)
        np.testing.assert_array_equal(graph.rinv.cpu().numpy(), inv)
    # the search's row histogram (what EQUIGSPR_SCRATCH_HISTOGRAM hands to build_csr), per-query and tiled kernels, += semantics
    for npts, k in ((600, 9), (2500, 6)):
        x = torch.rand(2, npts, 3, generator=g).cuda()
        hist = torch.ones(2 * npts, dtype=torch.int32, device="cuda")
        _, r32, _ = ops.knn_graph_batched(x, k, True, want_i64=False, want_i32=True, row_hist=hist)
        np.testing.assert_array_equal(hist.cpu().numpy(), 1 + np.bincount(r32.cpu().numpy(), minlength=2 * npts))
        # ABI 10: the arrival slots of the same atomics -> the index fill without atomics, same stable sort
        hist0 = torch.zeros(2 * npts + 1, dtype=torch.int32, device="cuda")
        arr = torch.full((2 * npts * k,), -1, dtype=torch.int32, device="cuda")
        _, r32b, c32b = ops.knn_graph_batched(x, k, True, want_i64=False, want_i32=True, row_hist=hist0, arrival=arr)
        a, r = arr.cpu().numpy(), r32b.cpu().numpy()
        assert (a >= 0).all() and all(sorted(a[r == j].tolist()) == list(range(int((r == j).sum()))) for j in range(0, 2 * npts, 97))
        gph = pk.DeviceGraph(r32b, c32b, 2 * npts, knn_k=k, row_hist=hist0, arrival=arr)
        np.testing.assert_array_equal(gph.rperm.cpu().numpy(), np.argsort(r, kind="stable"))
        assert int(hist0.abs().sum()) == 0
    with pytest.raises(RuntimeError, match="scratch_state"):
        _lib.check(L.equigspr_build_csr(p(key32), e, n, p(ptr), p(perm), None, p(scratch), 3, None, ops._stream()), "build_csr")


def test_captured_graph_owns_its_csr_scratch(pk):
    """A hipGraph that captured an index build keeps the zero-on-exit counter array it adopted from the eager warm-up (its replays
    use the raw pointer); later eager builds of the SAME size -- on any stream, also while the graph replays -- must get a buffer
    of their own."""
    from se3_equi_graph_registration_amd import ops
    g = torch.Generator().manual_seed(17)
    x = torch.rand(2, 700, 3, generator=g).cuda()
    y = torch.rand(2, 700, 3, generator=g).cuda()
    ref_x = pk.build_pair_graph(x[:1], x[1:], 7)                    # eager warm-up: leaves a zeroed scratch of this size in the cache
    ref_y = pk.build_pair_graph(y[:1], y[1:], 7)
    rx, ry = ref_x.rperm.clone(), ref_y.rperm.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            cap = pk.build_pair_graph(x[:1], x[1:], 7)
    owned = [b.data_ptr() for b in ops._CSR_GRAPH_OWNED]
    assert owned and all(b.data_ptr() not in owned for b in ops._CSR_SCRATCH.values())
    other = torch.cuda.Stream()
    for _ in range(20):                                             # replays on one stream, eager builds of the same size on another
        graph.replay()
        with torch.cuda.stream(other):
            eager = pk.build_pair_graph(y[:1], y[1:], 7)
    torch.cuda.synchronize()
    assert torch.equal(cap.rperm, rx) and torch.equal(eager.rperm, ry)
    assert all(int(b.abs().sum()) == 0 for b in ops._CSR_SCRATCH.values()) and all(int(b.abs().sum()) == 0 for b in ops._CSR_GRAPH_OWNED)


# ------------------------------------------------------------------------------------ EGNN forward
@pytest.mark.parametrize("name", LAYER_FIX)
def test_egnn_forward_golden(pk, golden_dir, name, parity):
    fx = np.load(os.path.join(golden_dir, name))
    m = build_egnn(pk, fx)
    h, x = torch.from_numpy(fx["h"]).cuda(), torch.from_numpy(fx["x"]).cuda()
    edges = torch.from_numpy(fx["edges"]).cuda()
    ea = torch.ones(edges.shape[1], 1, device="cuda")
    with torch.no_grad():
        ho, xo = m(h, x, [edges[0], edges[1]], ea)
    for got, key in ((ho, "h_out"), (xo, "x_out")):
        tol = anchored_tol(fx["f32." + key], fx["f64." + key])
        err, _ = parity(name, "egnn." + key, got, fx["f64." + key], fx["f32." + key])
        assert err <= tol, f"{name} {key}: err {err:.3e} > tol {tol:.3e}"
    # kNN edges rebuilt on the device give the same graph as the fixture's
    k = int(fx["k"])
    assert torch.equal(pk.knn_graph(x, k, loop=True).cpu(), torch.from_numpy(fx["edges"]))


def test_egnn_forward_vs_oracle_random_graph(pk):
    """Arbitrary (non-kNN) edge list incl. isolated nodes, multi-edges and self loops."""
    g = torch.Generator().manual_seed(21)
    n, e = 300, 2500
    torch.manual_seed(5)
    m = pk.EGNN(32, 32, 32, in_edge_nf=1, device="cuda", n_layers=3)
    h = torch.randn(n, 32, generator=g); x = torch.rand(n, 3, generator=g) * 2
    row = torch.randint(0, n - 10, (e,), generator=g); col = torch.randint(0, n - 10, (e,), generator=g)   # last 10 isolated
    row[:5] = col[:5]
    ea = torch.rand(e, 1, generator=g)
    w64 = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    w32 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    r64h, r64x = orc.egnn_forward(w64, h.double(), x.double(), [row, col], ea.double())
    r32h, r32x = orc.egnn_forward(w32, h, x, [row, col], ea)
    with torch.no_grad():
        ho, xo = m(h.cuda(), x.cuda(), [row.cuda(), col.cuda()], ea.cuda())
    for got, a32, a64 in ((ho, r32h, r64h), (xo, r32x, r64x)):
        tol = anchored_tol(a32.numpy(), a64.numpy())
        assert float((got.cpu().double() - a64).abs().max()) <= tol


# ------------------------------------------------------------------------------------ pair model
def _pair_models(pk, fx):
    w = {k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w.")}
    we = {k[len("egnn."):]: v for k, v in w.items() if k.startswith("egnn.")}
    feat = we["embedding_in.weight"].shape[1]
    egnn = pk.EGNN(feat, feat, feat, in_edge_nf=1, device="cuda", n_layers=int(fx["n_layers"]), num_heads=int(fx["heads"]))
    egnn.load_state_dict(we)
    n = fx["in.src"].shape[0]
    models = {}
    for variant in ("train", "eval"):
        c = pk.CrossAttentionPoseRegression(egnn, num_nodes=n, hidden_nf=feat, device="cuda", variant=variant).cuda()
        c.load_state_dict(w, strict=False)
        models[variant] = c
    return models


def _pair_inputs(pk, fx):
    k = int(fx["k"])
    f = lambda name: torch.from_numpy(fx["in." + name])[None].cuda()
    e_s = pk.knn_graph(f("src")[0], k, loop=True)[None]
    e_t = pk.knn_graph(f("tgt")[0], k, loop=True)[None]
    ea = torch.ones(1, e_s.shape[2], 1, device="cuda")
    return (f("feat_src"), f("src"), e_s, ea, f("feat_tgt"), f("tgt"), e_t, ea, f("corr"), f("labels"), f("gt_pose"))


@pytest.mark.parametrize("name", PAIR_FIX)
def test_pair_forward_golden(pk, golden_dir, name, parity):
    fx = np.load(os.path.join(golden_dir, name))
    models = _pair_models(pk, fx)
    args = _pair_inputs(pk, fx)
    with torch.no_grad():
        R, t, closs, equi, hs, xs, ht, xt, _ = models["train"](*args)
    mt = models["train"]
    # features / coordinates after the EGNN
    for got, key in ((hs[0], "h_src"), (xs[0], "x_src"), (ht[0], "h_tgt"), (xt[0], "x_tgt")):
        tol = anchored_tol(fx["f32." + key], fx["f64." + key])
        err, _ = parity(name, "pair." + key, got, fx["f64." + key], fx["f32." + key])
        assert err <= tol, f"{name} {key}: err {err:.3e} > tol {tol:.3e}"
    # top-k: same index set; logits compared after aligning by index (SURVEY hard part 3)
    ref_idx, ref_log = fx["f64.train.top_idx"][0], fx["f64.train.logits"][0]
    got_idx, got_log = mt.last_top_idx[0].cpu().numpy(), mt.last_logits[0].cpu().numpy().astype(np.float64)
    assert set(ref_idx.tolist()) == set(got_idx.tolist())
    ref_map = dict(zip(ref_idx.tolist(), ref_log.tolist()))
    ref32_map = dict(zip(fx["f32.train.top_idx"][0].tolist(), fx["f32.train.logits"][0].tolist())) if "f32.train.top_idx" in fx.files else None
    aligned64 = np.array([ref_map[i] for i in got_idx.tolist()])
    aligned32 = np.array([ref32_map[i] for i in got_idx.tolist()]) if ref32_map and set(ref32_map) == set(ref_map) else None
    d, _ = parity(name, "train.logits", got_log, aligned64, aligned32)
    tol = logits_tol(fx["f32.train.logits"], fx["f64.train.logits"])
    assert d <= tol, f"{name} logits: err {d:.3e} > tol {tol:.3e}"
    # pose: within 1e-4 of the reference (training head; degenerate regime gives R = I for both)
    eR, _ = parity(name, "train.R", R, fx["f64.train.R"], fx["f32.train.R"] if "f32.train.R" in fx.files else None)
    eT, _ = parity(name, "train.t", t, fx["f64.train.t"], fx["f32.train.t"] if "f32.train.t" in fx.files else None)
    assert eR <= NORTH_STAR_RT and eT <= NORTH_STAR_RT
    np.testing.assert_allclose(closs.item(), float(fx["f64.train.corr_sim_loss"]), rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(equi.item(), float(fx["f64.train.equi_loss"]), rtol=2e-4, atol=1e-5)
    # evaluation head (well-conditioned: near-uniform weights over raw coordinates)
    with torch.no_grad():
        Re, te, scores, *_ = models["eval"](*args)
    assert scores is None
    eR, _ = parity(name, "eval.R", Re, fx["f64.eval.R"], fx["f32.eval.R"] if "f32.eval.R" in fx.files else None)
    eT, _ = parity(name, "eval.t", te, fx["f64.eval.t"], fx["f32.eval.t"] if "f32.eval.t" in fx.files else None)
    assert eR <= NORTH_STAR_RT and eT <= NORTH_STAR_RT
    assert abs(float(torch.det(Re[0])) - 1.0) < 1e-5


def test_fp32_products_switch_restores_the_parity_margin_on_the_baseline_pair(pk, golden_dir, parity):
    """`equigspr_set_edge_products(1)` (ABI 14; EQUIGSPR_EDGE_PRODUCTS=fp32): the same kernels on fp32 matrix products.  On the BASELINE
    pair (2 048 points, k = 20, 4 layers; reference = the real scripts' fp64 run) the split-bf16 default sits at the reference's own
    fp32 error (logits 5.1e-6 against its 4.4e-6; north_star 1e-5); the fp32 form must be at least as close as the default on every
    compared tensor and within the north_star with a third to spare -- the margin round 5 spent is available to who wants it."""
    from se3_equi_graph_registration_amd import _lib
    name = "pair_n2048_k20_l4.npz"
    fx = np.load(os.path.join(golden_dir, name))
    L = _lib.lib()

    def run(fp32):
        _lib.check(L.equigspr_set_edge_products(1 if fp32 else 0), "set_edge_products")
        models = _pair_models(pk, fx)
        with torch.no_grad():
            R, t, _, _, hs, xs, ht, xt, _ = models["train"](*_pair_inputs(pk, fx))
        mt = models["train"]
        ref_map = dict(zip(fx["f64.train.top_idx"][0].tolist(), fx["f64.train.logits"][0].tolist()))
        got_idx = mt.last_top_idx[0].cpu().numpy().tolist()
        assert set(got_idx) == set(ref_map)
        d_log = float(np.abs(mt.last_logits[0].cpu().numpy().astype(np.float64) - np.array([ref_map[i] for i in got_idx])).max())
        d_h = float(np.abs(hs[0].cpu().numpy().astype(np.float64) - fx["f64.h_src"]).max())
        d_R = float(np.abs(R.cpu().numpy().astype(np.float64) - fx["f64.train.R"]).max())
        d_t = float(np.abs(t.cpu().numpy().astype(np.float64) - fx["f64.train.t"]).max())
        return d_log, d_h, d_R, d_t

    try:
        bf, fp = run(False), run(True)
    finally:
        L.equigspr_set_edge_products(0)
    for tag, vals in (("bf16x3", bf), ("fp32", fp)):
        for q, v in zip(("train.logits", "pair.h_src", "train.R", "train.t"), vals):
            parity(name + ":" + tag, q, np.array([v]), np.array([0.0]), None)
    assert fp[0] <= 6.7e-6 and fp[2] <= NORTH_STAR_RT and fp[3] <= NORTH_STAR_RT, fp
    assert fp[0] <= bf[0] * 1.05 and fp[1] <= bf[1] * 1.05, (bf, fp)


@pytest.mark.parametrize("name", PAIR_FIX)
def test_fused_eval_head_golden_and_vs_aten_glue(pk, golden_dir, name, parity):
    """csrc/head.hip::head_eval_pair_kernel (eval :687-818 in two launches) against the reference's outputs in the fixtures, the
    oracle's Kabsch weights (incl. the [K] vs [1,K,1] broadcast: only p[0] is used), and the same head written in ATen ops."""
    fx = np.load(os.path.join(golden_dir, name))
    ev = _pair_models(pk, fx)["eval"]
    args = _pair_inputs(pk, fx)
    out = {}
    for fused in (True, False):
        ev.fused_head = fused
        with torch.no_grad():
            R, t, scores, equi, *_ = ev(*args)
        assert scores is None
        out[fused] = dict(R=R.clone(), t=t.clone(), equi=equi.clone(), w=ev.last_weights.clone(), idx=ev.last_top_idx.clone(),
                          p0=ev.last_p0.clone())
    a, b = out[True], out[False]
    eR, _ = parity(name, "eval_fused.R", a["R"], fx["f64.eval.R"], fx["f32.eval.R"])
    eT, _ = parity(name, "eval_fused.t", a["t"], fx["f64.eval.t"], fx["f32.eval.t"])
    assert eR <= NORTH_STAR_RT and eT <= NORTH_STAR_RT
    np.testing.assert_allclose(a["equi"].item(), float(fx["f64.eval.equi_loss"]), rtol=2e-4, atol=1e-5)
    assert set(a["idx"][0].tolist()) == set(b["idx"][0].tolist()) and int(a["idx"][0, 0]) == int(b["idx"][0, 0])
    assert abs(float(a["p0"]) - float(b["p0"])) <= 1e-5 * max(1.0, abs(float(b["p0"])))
    n = a["w"].shape[1]
    assert float((a["w"] - b["w"]).abs().max()) <= 1e-5 / n             # weights are ~1/n
    assert abs(float(a["w"].double().sum()) - 1.0) < 1e-5
    assert torch.allclose(a["R"], b["R"], atol=2e-6) and torch.allclose(a["t"], b["t"], atol=2e-5)
    # the oracle's weights on the same inputs (fp64), incl. which nodes had their similarity replaced by p[0]
    from oracle import egnn_oracle as orc
    we = {k[len("w.egnn."):]: torch.from_numpy(fx[k]).double() for k in fx.files if k.startswith("w.egnn.")}
    wc = {k[2:]: torch.from_numpy(fx[k]).double() for k in fx.files
          if k.startswith("w.") and not k.startswith("w.egnn.") and fx[k].dtype.kind == "f"}
    cpu = [(x.detach().cpu().double() if x.is_floating_point() else x.detach().cpu()) for x in args]
    with torch.no_grad():
        _, ex = orc.pair_forward_eval(we, wc, *cpu)
    assert set(ex["top_idx"][0].tolist()) == set(a["idx"][0].tolist())
    assert float((a["w"][0].cpu().double() - ex["weights"]).abs().max()) <= 2e-5 / n


def test_prebuilt_pair_graph_equals_edge_list_path(pk, golden_dir):
    fx = np.load(os.path.join(golden_dir, "pair_n512_k12_l2_wellcond.npz"))
    models = _pair_models(pk, fx)
    args = list(_pair_inputs(pk, fx))
    with torch.no_grad():
        a = models["train"](*args)
        graph = pk.build_pair_graph(args[1], args[5], int(fx["k"]))
        args[2] = graph
        b = models["train"](*args)
    for u, v in zip(a[:2] + a[4:8], b[:2] + b[4:8]):
        assert torch.equal(u, v)          # same kernels, same graph -> bitwise identical


# ------------------------------------------------------------------------------------ Kabsch
def test_kabsch_forward_backward_golden(pk, golden_dir):
    fx = np.load(os.path.join(golden_dir, "kabsch_cases.npz"))
    for ci in range(int(fx["n_cases"])):
        s = torch.from_numpy(fx[f"c{ci}.src"]).float().cuda().requires_grad_(True)
        t = torch.from_numpy(fx[f"c{ci}.tgt"]).float().cuda().requires_grad_(True)
        sc = torch.from_numpy(fx[f"c{ci}.score"]).float().cuda().requires_grad_(True)
        w = torch.softmax(sc, -1)
        w = w / (w.sum() + 1e-6)
        R, tt = pk.weighted_kabsch(s[None], t[None], w[None])
        assert float(np.abs(R[0].detach().cpu().numpy() - fx[f"c{ci}.f64.R"]).max()) <= 1e-4
        assert float(np.abs(tt[0].detach().cpu().numpy() - fx[f"c{ci}.f64.t"]).max()) <= 1e-4
        gR = torch.from_numpy(fx[f"c{ci}.gR"]).float().cuda()
        gt = torch.from_numpy(fx[f"c{ci}.gt"]).float().cuda()
        ((R[0] * gR).sum() + (tt[0] * gt).sum()).backward()
        for got, key in ((s.grad, "g_src"), (t.grad, "g_tgt"), (sc.grad, "g_score")):
            ref = fx[f"c{ci}.f64.{key}"]
            gap = float(np.abs(fx[f"c{ci}.f32.{key}"] - ref).max())
            tol = 10 * gap + 1e-4 * max(1e-3, float(np.abs(ref).max()))
            err = float(np.abs(got.cpu().numpy() - ref).max())
            assert err <= tol, f"kabsch case {ci} {key}: err {err:.3e} > tol {tol:.3e}"


def test_kabsch_degenerate_and_masked(pk):
    g = torch.Generator().manual_seed(1)
    s = torch.rand(2, 50, 3, generator=g).cuda(); t = torch.rand(2, 50, 3, generator=g).cuda()
    w = torch.zeros(2, 50).cuda()
    w[0, 7] = 1.0 / (1.0 + 1e-6)            # one-hot weights -> H = 1e-6 I -> R = I, t = tgt[j]-src[j]
    R, tt = pk.weighted_kabsch(s, t, w)     # sample 1: all-zero weights -> R = I, t = 0 (train :708-711)
    assert torch.allclose(R[0].cpu(), torch.eye(3), atol=1e-6) and torch.allclose(R[1].cpu(), torch.eye(3), atol=1e-6)
    assert torch.allclose(tt[0].cpu(), (t[0, 7] - s[0, 7]).cpu() * w[0, 7].cpu(), atol=1e-5)
    assert torch.allclose(tt[1].cpu(), torch.zeros(3), atol=1e-7)


# ------------------------------------------------------------------------------------ EGNN backward
def _grad_check(got, ref32, ref64, name, k=6.0, rel=2e-5, parity=None, fixture=None):
    ref64 = np.asarray(ref64, dtype=np.float64)
    scale = max(1e-6, float(np.abs(ref64).max()))
    gap = float(np.abs(np.asarray(ref32, dtype=np.float64) - ref64).max())
    tol = k * gap + rel * scale
    err = float(np.abs(got.detach().cpu().numpy().astype(np.float64) - ref64).max())
    if parity is not None:
        parity(fixture, name, got, ref64, ref32)
    assert err <= tol, f"{name}: err {err:.3e} > tol {tol:.3e} (scale {scale:.3e}, ref fp32 gap {gap:.3e})"


@pytest.mark.parametrize("name", LAYER_FIX)
def test_egnn_backward_golden(pk, golden_dir, name, parity):
    fx = np.load(os.path.join(golden_dir, name))
    m = build_egnn(pk, fx)
    h = torch.from_numpy(fx["h"]).cuda().requires_grad_(True)
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_(True)
    edges = torch.from_numpy(fx["edges"]).cuda()
    ea = torch.ones(edges.shape[1], 1, device="cuda")
    ho, xo = m(h, x, [edges[0], edges[1]], ea)
    loss = (ho * torch.from_numpy(fx["gh"]).cuda()).sum() + (xo * torch.from_numpy(fx["gx"]).cuda()).sum()
    m.zero_grad()
    loss.backward()
    _grad_check(h.grad, fx["f32.grad_h"], fx["f64.grad_h"], "grad_h", parity=parity, fixture=name)
    _grad_check(x.grad, fx["f32.grad_x"], fx["f64.grad_x"], "grad_x", parity=parity, fixture=name)
    for pn, p in m.named_parameters():
        assert p.grad is not None, pn
        _grad_check(p.grad, fx["f32.grad." + pn], fx["f64.grad." + pn], "grad." + pn, parity=parity, fixture=name)
    # gradients are views of ONE flat bucket, in parameter-block order
    off = 0
    for pn, p in m.named_parameters():
        assert p.grad.data_ptr() == m.flat_grads.data_ptr() + 4 * off
        off += p.numel()
    # accumulation semantics (+=) and zero_grad(set_to_none=True) recovery
    g1 = m.flat_grads.clone()
    ho, xo = m(h, x, [edges[0], edges[1]], ea)
    ((ho * torch.from_numpy(fx["gh"]).cuda()).sum() + (xo * torch.from_numpy(fx["gx"]).cuda()).sum()).backward()
    assert torch.allclose(m.flat_grads, 2 * g1, rtol=1e-6, atol=1e-7)
    for p in m.parameters():
        p.grad = None
    ho, xo = m(h, x, [edges[0], edges[1]], ea)
    ((ho * torch.from_numpy(fx["gh"]).cuda()).sum() + (xo * torch.from_numpy(fx["gx"]).cuda()).sum()).backward()
    assert torch.equal(m.flat_grads, g1)          # deterministic: bitwise identical to the first backward


def test_train_step_gradients_golden(pk, golden_dir, parity):
    """Full training loss (train :1120: corr+sim, geodesic rotation, translation direction) on the
    well-conditioned pair: loss and every parameter gradient against the reference autograd."""
    fx = np.load(os.path.join(golden_dir, "pair_n512_k12_l2_wellcond.npz"))
    models = _pair_models(pk, fx)
    mt = models["train"]
    args = _pair_inputs(pk, fx)
    R, t, closs, equi, *_ = mt(*args)
    rl, tl = pk.pose_loss(R, t, args[-1])
    loss = closs.mean() + rl.mean() + tl.mean()
    np.testing.assert_allclose(loss.item(), float(fx["f64.train.loss"]), rtol=1e-4)
    assert float(np.abs(R.detach().cpu().numpy() - fx["f64.train.R"]).max()) <= 1e-4
    mt.zero_grad()
    loss.backward()
    checked = 0
    for pn, p in mt.named_parameters():
        key = "train.grad." + pn
        if "f64." + key in fx.files:
            assert p.grad is not None, pn
            _grad_check(p.grad, fx["f32." + key], fx["f64." + key], "train.grad." + pn, k=8.0, rel=5e-5, parity=parity,
                        fixture="pair_n512_k12_l2_wellcond.npz")
            checked += 1
    assert checked > 20


@pytest.mark.parametrize("feat,heads,in_nf,out_nf,n,e", [(32, 4, 32, 32, 257, 3001), (33, 1, 33, 33, 257, 3001), (33, 1, 17, 40, 257, 3001),
                                                         (32, 2, 32, 32, 257, 3001), (32, 8, 9, 32, 257, 3001), (32, 8, 32, 32, 16999, 70001),
                                                         (16, 1, 16, 16, 257, 3001), (48, 4, 20, 48, 257, 3001), (64, 8, 64, 64, 257, 3001),
                                                         (64, 1, 64, 9, 2000, 9001),
                                                         (32, 1, 5, 64, 257, 3001), (32, 1, 5, 7, 16999, 70001), (32, 4, 32, 32, 16999, 70001),
                                                         (33, 1, 17, 40, 16999, 70001),
                                                         (48, 2, 48, 48, 16999, 70001), (64, 4, 64, 64, 16999, 70001),      # lane widths above the 16-node threshold
                                                         (32, 1, 32, 32, 40, 5), (33, 1, 33, 33, 40, 17), (32, 1, 32, 32, 300, 1025),      # fewer edges than one tile; one tile + 1; 64 tiles + 1
                                                         (64, 2, 64, 64, 40, 5), (48, 1, 48, 48, 40, 17), (16, 4, 16, 16, 300, 1025), (64, 8, 30, 64, 300, 4099),   # the same for the feature-parallel kernels
                                                         (32, 1, 32, 32, 20000, 500001), (33, 1, 33, 33, 20000, 500001), (48, 2, 48, 48, 20000, 500001)])   # above the edge count below which the edge kernels write their outputs through (plain stores here)
def test_backward_random_graph_vs_oracle_autograd(pk, feat, heads, in_nf, out_nf, n, e):
    """Non-kNN edge list (hub, isolated nodes, self loops, multi-edges), random edge_attr, 3 layers; both hidden widths
    (33 = the MFMA kernels' border path), 1 and 4 heads, embedding widths that differ from the hidden width (odd ones:
    the layers' parameter blocks are then not 16-byte aligned).  More than 16 384 nodes: the 16-node MFMA node backward
    (csrc/egnn_node_bwd.hip) with 16-node tiles; hidden_nf 33 below that: its 4-node tiles."""
    g = torch.Generator().manual_seed(33)
    torch.manual_seed(9)
    m = pk.EGNN(in_nf, feat, out_nf, in_edge_nf=1, device="cuda", n_layers=3, num_heads=heads)
    h = torch.randn(n, in_nf, generator=g); x = torch.rand(n, 3, generator=g) * 2
    row = torch.randint(0, n - 7, (e,), generator=g); col = torch.randint(0, n - 7, (e,), generator=g)
    row[:min(40, e // 2)] = 3; row[40:45] = col[40:45]
    ea = torch.rand(e, 1, generator=g)
    gh = torch.randn(n, out_nf, generator=g); gx = torch.randn(n, 3, generator=g)
    refs = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        w = {k: v.detach().cpu().to(dt).requires_grad_(True) for k, v in m.state_dict().items()}
        hh = h.to(dt).clone().requires_grad_(True); xx = x.to(dt).clone().requires_grad_(True)
        ho, xo = orc.egnn_forward(w, hh, xx, [row, col], ea.to(dt))
        ((ho * gh.to(dt)).sum() + (xo * gx.to(dt)).sum()).backward()
        refs[tag] = dict(h=hh.grad.numpy(), x=xx.grad.numpy(), **{k: v.grad.numpy() for k, v in w.items()})
        refs[tag + "_out"] = (ho.detach().numpy(), xo.detach().numpy())
    hc = h.cuda().requires_grad_(True); xc = x.cuda().requires_grad_(True)
    ho, xo = m(hc, xc, [row.cuda(), col.cuda()], ea.cuda())
    for got, a32, a64 in zip((ho, xo), refs["f32_out"], refs["f64_out"]):
        assert np.abs(got.detach().cpu().numpy() - a64).max() <= anchored_tol(a32, a64)
    m.zero_grad()
    ((ho * gh.cuda()).sum() + (xo * gx.cuda()).sum()).backward()
    _grad_check(hc.grad, refs["f32"]["h"], refs["f64"]["h"], "grad_h")
    _grad_check(xc.grad, refs["f32"]["x"], refs["f64"]["x"], "grad_x")
    for pn, p in m.named_parameters():
        _grad_check(p.grad, refs["f32"][pn], refs["f64"][pn], pn)


@pytest.mark.parametrize("feat,heads", [(32, 1), (32, 4), (33, 1), (48, 1), (16, 1), (64, 4), (48, 2), (16, 8)])
def test_no_edge_attribute_forward_backward_vs_oracle(pk, feat, heads):
    """in_edge_nf = 0 (the reference E_GCL's own default, edges_in_d = 0, train :186): the edge MLP's first layer is 2F + 12 wide and
    no attribute column exists -- forward and every gradient against the oracle, on the MFMA kernels and on the lane kernels."""
    g = torch.Generator().manual_seed(71)
    torch.manual_seed(4)
    n, e = 300, 3500
    m = pk.EGNN(feat, feat, feat, in_edge_nf=0, device="cuda", n_layers=2, num_heads=heads)
    assert m.state_dict()["gcl_0.edge_mlps.0.0.weight"].shape[1] == 2 * feat + 12
    h = torch.randn(n, feat, generator=g); x = torch.rand(n, 3, generator=g) * 2
    row = torch.randint(0, n, (e,), generator=g); col = torch.randint(0, n, (e,), generator=g)
    gh = torch.randn(n, feat, generator=g); gx = torch.randn(n, 3, generator=g)
    refs = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        w = {k: v.detach().cpu().to(dt).requires_grad_(True) for k, v in m.state_dict().items()}
        hh = h.to(dt).clone().requires_grad_(True); xx = x.to(dt).clone().requires_grad_(True)
        ho, xo = orc.egnn_forward(w, hh, xx, [row, col], None)
        ((ho * gh.to(dt)).sum() + (xo * gx.to(dt)).sum()).backward()
        refs[tag] = dict(h=hh.grad.numpy(), x=xx.grad.numpy(), **{k: v.grad.numpy() for k, v in w.items()})
        refs[tag + "_out"] = (ho.detach().numpy(), xo.detach().numpy())
    hc = h.cuda().requires_grad_(True); xc = x.cuda().requires_grad_(True)
    ho, xo = m(hc, xc, [row.cuda(), col.cuda()], None)
    for got, a32, a64 in zip((ho, xo), refs["f32_out"], refs["f64_out"]):
        assert np.abs(got.detach().cpu().numpy() - a64).max() <= anchored_tol(a32, a64)
    m.zero_grad()
    ((ho * gh.cuda()).sum() + (xo * gx.cuda()).sum()).backward()
    _grad_check(hc.grad, refs["f32"]["h"], refs["f64"]["h"], "grad_h")
    _grad_check(xc.grad, refs["f32"]["x"], refs["f64"]["x"], "grad_x")
    for pn, p in m.named_parameters():
        _grad_check(p.grad, refs["f32"][pn], refs["f64"][pn], pn)


def test_smoke_entry(pk):
    import __graft_entry__
    __graft_entry__.smoke()


# ------------------------------------------------------------------------------------ fused head
@pytest.mark.parametrize("name", ["pair_n512_k12_l2_wellcond.npz", "pair_n2048_k20_l4.npz"])
def test_fused_head_matches_aten_head(pk, golden_dir, name):
    """csrc/head.hip against the same head written with PyTorch-ROCm ops (fused_head=False): forward values,
    and all gradients of corr+sim+equi+pose losses (the equi term exercises the optional g_equi path)."""
    fx = np.load(os.path.join(golden_dir, name))
    res = {}
    for fused in (True, False):
        models = _pair_models(pk, fx)
        mt = models["train"]
        mt.fused_head = fused
        args = _pair_inputs(pk, fx)
        R, t, closs, equi, hs, xs, ht, xt, _ = mt(*args)
        rl, tl = pk.pose_loss(R, t, args[-1])
        wcond = "wellcond" in name
        loss = closs + 0.5 * equi + ((rl.mean() + tl.mean()) if wcond else (R * 0.3).sum() + (t * 0.7).sum())
        mt.zero_grad()
        loss.backward()
        res[fused] = dict(R=R.detach(), t=t.detach(), closs=closs.detach(), equi=equi.detach(), idx=mt.last_top_idx.clone(),
                          logits=mt.last_logits.clone(), g=mt.egnn.flat_grads.clone(),
                          gm=torch.cat([p.grad.reshape(-1) for p in mt.mlp.parameters()]))
    a, b = res[True], res[False]
    assert set(a["idx"][0].tolist()) == set(b["idx"][0].tolist())
    o1, o2 = torch.sort(a["idx"][0])[1], torch.sort(b["idx"][0])[1]
    assert torch.allclose(a["logits"][0][o1], b["logits"][0][o2], rtol=1e-4, atol=1e-4 * float(b["logits"].abs().max()))
    assert torch.allclose(a["R"], b["R"], atol=2e-5) and torch.allclose(a["t"], b["t"], atol=2e-5)
    assert torch.allclose(a["closs"], b["closs"], rtol=2e-4, atol=1e-5)
    assert torch.allclose(a["equi"], b["equi"], rtol=2e-4, atol=1e-5)
    for key in ("g", "gm"):
        scale = float(b[key].abs().max())
        err = float((a[key] - b[key]).abs().max())
        assert err <= 2e-3 * scale + 1e-7, f"{name} {key}: err {err:.3e} scale {scale:.3e}"


@pytest.mark.parametrize("feat,chunk,n", [(32, 0, 700), (32, 64, 700), (32, 48, 700), (32, 16, 700), (33, 0, 700), (32, 0, 9000), (33, 0, 9000)])   # 9000 = 3 pairs of 3000 points: 18 000 nodes
def test_egnn_backward_is_bitwise_reproducible(pk, feat, chunk, n):
    """The edge backward recomputes the edge forward (nothing per edge is kept) and reduces the weight gradients through
    per-wave partials, per-workgroup slabs and one fixed-order reduction: no atomics, a static chunk -> wave map.  Two
    runs on the same inputs give bit-identical outputs and gradients, for every chunk size of the 16-edge kernels."""
    g = torch.Generator().manual_seed(17)
    k = 9
    B = 3 if n > 8192 else 1
    n //= B
    x = (torch.rand(2 * B, n, 3, generator=g) * 3).cuda()
    n *= B
    h = torch.randn(2 * n, feat, generator=g).cuda()
    graph = pk.build_pair_graph(x[:B], x[B:], k)
    torch.manual_seed(4)
    egnn = pk.EGNN(feat, feat, feat, in_edge_nf=1, device="cuda", n_layers=3)
    gh, gx = torch.randn(2 * n, feat, generator=g).cuda(), torch.randn(2 * n, 3, generator=g).cuda()
    from se3_equi_graph_registration_amd import _lib
    res = []
    _lib.check(_lib.lib().equigspr_set_edge_chunk(chunk), "set_edge_chunk")
    try:
        for _ in range(3):
            egnn.zero_grad()
            hh, xx = h.clone().requires_grad_(True), x.reshape(2 * n, 3).clone().requires_grad_(True)
            ho, xo = egnn(hh, xx, graph, torch.ones(graph.n_edges, 1, device="cuda"))
            ((ho * gh).sum() + (xo * gx).sum()).backward()
            res.append((ho.detach().clone(), xo.detach().clone(), egnn.flat_grads.clone(), hh.grad.clone(), xx.grad.clone()))
    finally:
        _lib.lib().equigspr_set_edge_chunk(0)
    for other in res[1:]:
        for u, v in zip(res[0], other):
            assert torch.equal(u, v)
    assert float(res[0][2].abs().max()) > 0 and bool(torch.isfinite(res[0][2]).all())


def test_head_tail_ticket_is_deterministic_and_matches_two_launch_kabsch(pk):
    """The training head finishes with ONE launch: Kabsch moments by several blocks per pair, the solve by whichever block
    of the pair finishes last (ticket + fences), the loss scalars by an extra block.  Same R, t as equigspr_kabsch_forward on
    the weights the ATen-op head computes, and bit-identical over 200 repetitions with 12 pairs x 8 blocks in flight."""
    torch.manual_seed(5)
    B, n, feat, k = 12, 2048, 32, 8
    egnn = pk.EGNN(feat, feat, feat, in_edge_nf=1, device="cuda", n_layers=1)
    with torch.no_grad():
        egnn.embedding_out.weight.mul_(0.05); egnn.embedding_out.bias.mul_(0.05)
    model = pk.CrossAttentionPoseRegression(egnn, num_nodes=n, hidden_nf=feat, device="cuda").cuda()
    b = pk.synthetic_batch(3, B, n=n, feat=feat, device="cuda")
    graph = pk.build_pair_graph(b["src"], b["tgt"], k)
    args = (b["feat_src"], b["src"], graph, None, b["feat_tgt"], b["tgt"], graph, None, b["corr"], b["labels"], b["gt_pose"])
    with torch.no_grad():
        first = model(*args)
        R0, t0, c0 = first[0].clone(), first[1].clone(), first[2].clone()
        for _ in range(200):
            out = model(*args)
            assert torch.equal(out[0], R0) and torch.equal(out[1], t0) and torch.equal(out[2], c0)
        model.fused_head = False                      # ATen-op head: weights by torch ops, then the two-launch Kabsch
        ref = model(*args)
    assert torch.allclose(R0, ref[0], atol=2e-5) and torch.allclose(t0, ref[1], atol=2e-5)
    assert torch.allclose(c0, ref[2], rtol=2e-4, atol=1e-5)


def test_head_mlp_columns_equal_standalone_mlp_launch(pk):
    """Small batches run the head's MLP blocks as extra grid columns of the pair kernel (they wait for the pair's selection);
    large ones launch them on their own.  Same rows, same arithmetic: the logits and selections of a pair are bit-identical
    whichever way its batch was run, and R / t agree."""
    torch.manual_seed(6)
    B, n, feat, k = 40, 256, 32, 8                      # 40 x (1 + 1 + 16) workgroups > 512: the stand-alone MLP launch
    egnn = pk.EGNN(feat, feat, feat, in_edge_nf=1, device="cuda", n_layers=1)
    with torch.no_grad():
        egnn.embedding_out.weight.mul_(0.05); egnn.embedding_out.bias.mul_(0.05)
    model = pk.CrossAttentionPoseRegression(egnn, num_nodes=n, hidden_nf=feat, device="cuda").cuda()
    b = pk.synthetic_batch(11, B, n=n, feat=feat, device="cuda")

    def run(lo, hi):
        graph = pk.build_pair_graph(b["src"][lo:hi], b["tgt"][lo:hi], k)
        with torch.no_grad():
            out = model(b["feat_src"][lo:hi], b["src"][lo:hi], graph, None, b["feat_tgt"][lo:hi], b["tgt"][lo:hi], graph, None,
                        b["corr"][lo:hi], b["labels"][lo:hi], b["gt_pose"][lo:hi])
        return out[0].clone(), out[1].clone(), model.last_logits.clone(), model.last_top_idx.clone()

    R, t, lg, ti = run(0, B)
    assert bool(torch.isfinite(lg).all()) and float(lg.abs().max()) > 0
    for lo in range(0, B, 8):                           # 8 x 18 workgroups: the MLP rides in the pair kernel
        r8, t8, lg8, ti8 = run(lo, lo + 8)
        assert torch.equal(ti8, ti[lo:lo + 8])
        assert torch.equal(lg8, lg[lo:lo + 8])
        assert torch.allclose(r8, R[lo:lo + 8], atol=1e-6) and torch.allclose(t8, t[lo:lo + 8], atol=1e-6)


@pytest.mark.parametrize("B,N,K,expect_fused", [(5, 8192, 128, True), (10, 8192, 128, False), (16, 2048, 128, True), (20, 2048, 128, True),
                                                  (21, 2048, 128, False)])
def test_head_mlp_fusion_is_bounded_by_residency_and_replays_are_stable(pk, B, N, K, expect_fused):
    """The MLP blocks ride in the pair kernel's launch only when the whole fused grid is resident at the launch's LDS size (at
    N = 8192 that is one workgroup per CU: 5 pairs fit, 10 do not) -- forward progress then does not depend on dispatch order.  At
    the limits: 200 forwards give finite, bit-identical logits / selections, and no in-launch wait ever runs out."""
    from se3_equi_graph_registration_amd import ops
    F = 32
    assert ops.head_fuses_mlp(B, N, F, K) == expect_fused
    g = torch.Generator().manual_seed(B * 1000 + N)
    h_all = (torch.randn(2 * B * N, F, generator=g) * 0.3).cuda()
    x_all = (torch.rand(2 * B * N, 3, generator=g) * 3).cuda()
    raw = torch.nn.functional.normalize(torch.randn(2, B, N, F, generator=g), dim=-1).cuda()
    labels = (torch.rand(B, N, generator=g) < 0.5).float().cuda()
    gt = torch.eye(4).repeat(B, 1, 1)
    gt[:, :3, 3] = torch.rand(B, 3, generator=g) + 0.1
    gt = gt.cuda()
    torch.manual_seed(1)
    mlp = torch.nn.Sequential(torch.nn.Linear(2 * F, F), torch.nn.ReLU(), torch.nn.Linear(F, F // 2), torch.nn.ReLU(), torch.nn.Linear(F // 2, 1)).cuda()
    params = [mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, mlp[4].weight, mlp[4].bias]
    before = ops.head_wait_failures()
    ref = None
    with torch.no_grad():
        for it in range(200):
            out = ops.fused_pose_head(h_all, x_all, raw[0], raw[1], labels, gt, K, params)
            logits, top_idx, loss = out[4], out[5], out[2]          # out[2]: BCE + similarity-MSE (the arccos pose terms can be NaN on random poses)
            if ref is None:
                ref = (logits.clone(), top_idx.clone(), loss.clone())
                assert bool(torch.isfinite(logits).all()) and bool(torch.isfinite(loss)) and float(logits.abs().max()) > 0
            elif it % 10 == 0 or it == 199:
                assert torch.equal(logits, ref[0]) and torch.equal(top_idx, ref[1]) and torch.equal(loss, ref[2]), it
    torch.cuda.synchronize()
    assert ops.head_wait_failures() == before


# ------------------------------------------------------------------------------------ small fused training ops
def test_fused_pose_loss_matches_aten(pk):
    g = torch.Generator().manual_seed(4)
    B = 5
    q, _ = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))
    q = q * torch.sign(torch.det(q))[:, None, None]
    gq, _ = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))
    pose = torch.eye(4).repeat(B, 1, 1); pose[:, :3, :3] = gq; pose[:, :3, 3] = torch.rand(B, 3, generator=g)
    t = torch.rand(B, 3, generator=g)
    w1, w2 = torch.rand(B, generator=g), torch.rand(B, generator=g)
    ref = {}
    for dev in ("cpu", "cuda"):
        R_ = q.clone().to(dev).requires_grad_(True); t_ = t.clone().to(dev).requires_grad_(True)
        fn = pk.pose_loss if dev == "cuda" else pk.pose.pose_loss_aten          # HIP kernels vs the reference's ATen formula
        rl, tl = fn(R_, t_, pose.to(dev))
        ((rl * w1.to(dev)).sum() + (tl * w2.to(dev)).sum()).backward()
        ref[dev] = (rl.detach().cpu(), tl.detach().cpu(), R_.grad.cpu(), t_.grad.cpu())
    for a, b in zip(ref["cpu"], ref["cuda"]):
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-5)
    o_rl, o_tl = orc.pose_loss(q, t, pose)
    assert torch.allclose(ref["cuda"][0], o_rl, rtol=2e-4, atol=2e-5) and torch.allclose(ref["cuda"][1], o_tl, rtol=2e-4, atol=2e-5)


def test_step_loss_inside_head_equals_standalone_step_loss(pk, golden_dir):
    """registration_loss on the fused head's own outputs is served by the head's launches (csrc/head.hip: per-pair terms by the
    block that solves the pair, the sum by the MLP launch's last block; backward: the pose terms join grad_R / grad_t inside
    the Kabsch-backward blocks, the corr/sim term joins g_corr_sim).  Same value and same parameter gradients as the
    stand-alone step-loss kernels on the same tensors, for an upstream gradient != 1."""
    from se3_equi_graph_registration_amd import ops
    fx = np.load(os.path.join(golden_dir, "pair_n512_k12_l2_wellcond.npz"))
    res = []
    for inside in (True, False):
        mt = _pair_models(pk, fx)["train"]
        args = _pair_inputs(pk, fx)
        out = mt(*args)
        gt = args[-1]
        if inside:
            loss, rl, tl = pk.registration_loss(out[0], out[1], out[2], gt)
            assert loss.grad_fn is out[0].grad_fn                         # one autograd node with the head: no launch of its own
        else:
            loss, rl, tl = ops.fused_step_loss(out[0].float(), out[1].float(), out[2], gt)
        mt.zero_grad()
        (loss * 0.7).backward()
        res.append((loss.detach().clone(), rl.detach().clone(), tl.detach().clone(), mt.egnn.flat_grads.clone(),
                    torch.cat([p.grad.reshape(-1) for p in mt.mlp.parameters()])))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for u, v in zip(a[3:], b[3:]):
        scale = float(v.abs().max())
        assert float((u - v).abs().max()) <= 1e-5 * scale + 1e-9


def test_registration_loss_matches_unfused(pk):
    """train :1120 as one launch: value and gradients equal corr.mean() + pose_loss means through autograd; CPU tensors raise."""
    g = torch.Generator().manual_seed(9)
    for B in (1, 5, 300):
        q, _ = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))
        gq, _ = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))
        pose = torch.eye(4).repeat(B, 1, 1); pose[:, :3, :3] = gq; pose[:, :3, 3] = torch.rand(B, 3, generator=g)
        t = torch.rand(B, 3, generator=g)
        res = []
        for fused in (True, False):
            R_ = q.clone().cuda().requires_grad_(True); t_ = t.clone().cuda().requires_grad_(True)
            c_ = torch.tensor(0.37, device="cuda", requires_grad=True)
            if fused:
                loss, rl, tl = pk.registration_loss(R_, t_, c_ * 2.0, pose.cuda())
            else:
                rl, tl = pk.pose_loss(R_, t_, pose.cuda())
                loss = (c_ * 2.0).mean() + rl.mean() + tl.mean()
            (loss * 1.5).backward()
            res.append((loss.detach(), rl.detach(), tl.detach(), R_.grad, t_.grad, c_.grad))
        for a, b in zip(*res):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        pk.registration_loss(q, t, None, pose)
    with pytest.raises(RuntimeError):
        pk.pose_loss(q, t, pose)


def test_head_gradients_accumulate_into_flat_bucket(pk, golden_dir):
    """With the head MLP re-homed in a flat bucket (dp.FlatBucket) the HIP backward accumulates into the bucket directly;
    the result equals the per-tensor autograd path, and a second backward adds on top."""
    from se3_equi_graph_registration_amd.dp import FlatBucket
    fx = np.load(os.path.join(golden_dir, "pair_n512_k12_l2_wellcond.npz"))
    grads = {}
    for mode in ("tensors", "bucket"):
        models = _pair_models(pk, fx)
        args = _pair_inputs(pk, fx)
        m = models["train"]
        bucket = FlatBucket(list(m.mlp.parameters())) if mode == "bucket" else None
        reps = 2 if mode == "bucket" else 1
        for _ in range(reps):
            out = m(*args)
            loss, _, _ = pk.registration_loss(out[0], out[1], out[2], args[-1])
            loss.backward()
        g = torch.cat([p.grad.reshape(-1) for p in m.mlp.parameters()])
        grads[mode] = g / reps
        if bucket is not None:
            assert torch.equal(bucket.grad, g)
    assert float(grads["tensors"].abs().max()) > 0
    assert torch.allclose(grads["tensors"], grads["bucket"], rtol=1e-5, atol=1e-7)


def test_flat_adam_matches_torch_adam(pk):
    g = torch.Generator().manual_seed(8)
    p0 = [torch.randn(1000, generator=g), torch.randn(77, generator=g)]
    grads = [[torch.randn(1000, generator=g), torch.randn(77, generator=g)] for _ in range(6)]
    ref = [torch.nn.Parameter(p.clone().cuda()) for p in p0]
    opt = torch.optim.Adam(ref, lr=1e-3)
    mine = [p.clone().cuda() for p in p0]
    mgr = [torch.zeros_like(p) for p in mine]
    fa = pk.FlatAdam(list(zip(mine, mgr)), lr=1e-3)
    for step in grads:
        for r, m, gg in zip(ref, mgr, step):
            r.grad = gg.cuda(); m.copy_(gg.cuda())
        opt.step(); fa.step()
    for r, m in zip(ref, mine):
        assert torch.allclose(r.detach(), m, rtol=1e-5, atol=1e-6)
    assert float(fa.step_dev[0]) == 6.0
    # zero_grad=True: the same update, and the gradient buckets come back cleared (optimizer.zero_grad() fused into the launch)
    mine2 = [p.clone().cuda() for p in p0]
    mgr2 = [torch.zeros_like(p) for p in mine2]
    fb = pk.FlatAdam(list(zip(mine2, mgr2)), lr=1e-3)
    for step in grads:
        for m, gg in zip(mgr2, step):
            assert float(m.abs().max()) == 0.0
            m.copy_(gg.cuda())
        fb.step(zero_grad=True)
    for a, b in zip(mine, mine2):
        assert torch.equal(a, b)
    assert all(float(m.abs().max()) == 0.0 for m in mgr2)


# ------------------------------------------------------------------------------------------------
# A15: SO3TensorProductLayer (src/clean_so3_model.py:34-52), MFMA forward
# ------------------------------------------------------------------------------------------------
def _so3_layer(pk, fx):
    layer = pk.SO3TensorProductLayer(32, 32)
    layer.load_state_dict({k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w.")})
    return layer.cuda()


def test_so3_tensor_product_golden(pk, golden_dir):
    fx = np.load(os.path.join(golden_dir, "so3tp_d32.npz"))
    layer = _so3_layer(pk, fx)
    with torch.no_grad():
        y = layer(torch.from_numpy(fx["x"]).cuda()).cpu().numpy().astype(np.float64)
    assert y.shape == fx["y64"].shape
    assert np.abs(y - fx["y64"]).max() <= anchored_tol(fx["y"], fx["y64"])


@pytest.mark.parametrize("n_rows", [1, 31, 33, 40960])
def test_so3_tensor_product_sizes(pk, golden_dir, n_rows):
    fx = np.load(os.path.join(golden_dir, "so3tp_d32.npz"))
    layer = _so3_layer(pk, fx)
    x = torch.randn(n_rows, 32, generator=torch.Generator().manual_seed(n_rows))
    w64 = {k[2:]: torch.from_numpy(fx[k]).double() for k in fx.files if k.startswith("w.")}
    ref64 = orc.so3_tensor_product(w64, x.double()).numpy()
    ref32 = orc.so3_tensor_product({k: v.float() for k, v in w64.items()}, x).numpy()
    with torch.no_grad():
        y = layer(x.cuda()).cpu().numpy().astype(np.float64)
    assert np.abs(y - ref64).max() <= anchored_tol(ref32, ref64)


def test_so3_tensor_product_rejects(pk):
    with pytest.raises(NotImplementedError):
        with torch.no_grad():
            pk.SO3TensorProductLayer(16, 16).cuda()(torch.zeros(4, 16, device="cuda"))


def test_so3_tensor_product_trains(pk):
    """clean_so3_model.py:34-52 is a trainable nn.Module: input and parameter gradients against the oracle's autograd."""
    torch.manual_seed(3)
    layer = pk.SO3TensorProductLayer(32, 32).cuda()
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(700, 32, generator=g) * 0.3)
    gy = torch.randn(700, 32, generator=g)
    xg = x.cuda().requires_grad_(True)
    y = layer(xg)
    (y * gy.cuda()).sum().backward()
    w = {k: v.detach().cpu().double().requires_grad_(True) for k, v in layer.state_dict().items()}
    xr = x.double().requires_grad_(True)
    yr = orc.so3_tensor_product(w, xr)
    (yr * gy.double()).sum().backward()
    assert torch.allclose(y.detach().cpu().double(), yr.detach(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(xg.grad.cpu().double(), xr.grad, rtol=2e-4, atol=2e-4 * float(xr.grad.abs().max()))
    for name, p_ in layer.named_parameters():
        ref = w[name].grad
        assert torch.allclose(p_.grad.cpu().double(), ref, rtol=2e-4, atol=2e-4 * float(ref.abs().max())), name


# ------------------------------------------------------------------------------------------------
# A17: gcnLayer (src/gcnLayer.py): build_graph / GraphConvolution / GlobalPooling
# ------------------------------------------------------------------------------------------------
def test_gcn_build_graph_golden(pk, golden_dir):
    fx = np.load(os.path.join(golden_dir, "gcn_n300_k9.npz"))
    coords, k = fx["coords"], int(fx["k"])
    adj = pk.build_graph(torch.from_numpy(coords).cuda(), k=k)
    dense = adj.to_dense().cpu().numpy()
    assert ((dense != 0) == (fx["adj64"] != 0)).all()                       # same symmetrised edge set
    assert np.abs(dense - fx["adj64"]).max() <= 2e-7                        # fp64 arithmetic, fp32 storage
    assert np.array_equal(dense, dense.transpose(0, 2, 1))
    single = pk.build_graph(coords[0], k=k)                                 # numpy [n,3], as the reference takes it
    assert np.array_equal(single.to_dense().cpu().numpy(), dense[0])


def test_gcn_conv_and_pool_golden(pk, golden_dir):
    fx = np.load(os.path.join(golden_dir, "gcn_n300_k9.npz"))
    coords, k = fx["coords"], int(fx["k"])
    layer = pk.GraphConvolution(3, fx["y"].shape[1])
    layer.load_state_dict({kk[2:]: torch.from_numpy(fx[kk]) for kk in fx.files if kk.startswith("w.")})
    layer = layer.cuda()
    x = torch.from_numpy(coords).cuda().transpose(1, 2).contiguous()
    adj = pk.build_graph(torch.from_numpy(coords).cuda(), k=k)
    tol = anchored_tol(fx["y"], fx["y64"])
    with torch.no_grad():
        y_sparse = layer(adj, x)
        y_dense = layer(torch.from_numpy(fx["adj64"]).float().cuda(), x)    # the reference's dense-tensor call form
        pooled = pk.GlobalPooling()(y_sparse)
    assert np.abs(y_sparse.cpu().numpy() - fx["y64"]).max() <= tol
    assert np.abs(y_dense.cpu().numpy() - fx["y64"]).max() <= tol
    assert np.abs(pooled.cpu().numpy() - fx["pooled64"]).max() <= tol
    # trainable: gradients of both call forms against plain device autograd of the reference formula (gcnLayer.py:76-79)
    xg = x.clone().requires_grad_(True)
    layer.zero_grad()
    gy = torch.randn_like(y_sparse)
    (layer(adj, xg) * gy).sum().backward()
    g_sparse = (xg.grad.clone(), layer.conv.weight.grad.clone(), layer.conv.bias.grad.clone())
    xr = x.clone().requires_grad_(True)
    layer.zero_grad()
    (layer.conv(torch.matmul(xr, adj.to_dense()[None] if adj.to_dense().dim() == 2 else adj.to_dense())) * gy).sum().backward()
    for a, b_ in zip(g_sparse, (xr.grad, layer.conv.weight.grad, layer.conv.bias.grad)):
        assert torch.allclose(a, b_, rtol=1e-4, atol=1e-4 * float(b_.abs().max()))
    ref_grads = (xr.grad.clone(), layer.conv.weight.grad.clone(), layer.conv.bias.grad.clone())
    # ... the dense-adjacency call form through the same HIP backward (round 6: _GcnConvFn; the adjacency is transposed explicitly there)
    dense = adj.to_dense()
    dense = (dense[None] if dense.dim() == 2 else dense).contiguous()
    xd = x.clone().requires_grad_(True)
    layer.zero_grad()
    (layer(dense, xd) * gy).sum().backward()
    for a, b_ in zip((xd.grad, layer.conv.weight.grad, layer.conv.bias.grad), ref_grads):
        assert torch.allclose(a, b_, rtol=1e-4, atol=1e-4 * float(b_.abs().max()))
    # GlobalPooling: HIP backward against autograd of max || mean (ties of the max share the gradient, as torch.amax does)
    gp = torch.randn(y_sparse.shape[0], 2 * y_sparse.shape[1], device="cuda")
    pg = y_sparse.clone()
    pg[0, 0, 5] = pg[0, 0, 9] = pg[0, 0].max() + 1.0                       # a tie of the maximum
    pg.requires_grad_(True)
    (pk.GlobalPooling()(pg) * gp).sum().backward()
    pr = pg.detach().clone().requires_grad_(True)
    (torch.cat([pr.amax(dim=2), pr.mean(dim=2)], dim=1) * gp).sum().backward()
    assert torch.allclose(pg.grad, pr.grad, rtol=1e-6, atol=1e-7) and float(pg.grad[0, 0, 5]) == float(pg.grad[0, 0, 9])


def test_gcn_backward_wide_output_and_batched_adjacency(pk):
    """GraphConvolution(3, 128) (more output channels than the kernel aggregates per thread: the input gradient mixes the channels with one
    library GEMM first) and a per-cloud adjacency of a batch of two: gradients against device autograd of the reference formula."""
    g = torch.Generator().manual_seed(4)
    coords = torch.randn(2, 300, 3, generator=g).cuda()
    x = torch.randn(2, 3, 300, generator=g).cuda()
    adj = pk.build_graph(coords, k=9)
    layer = pk.GraphConvolution(3, 128).cuda()
    gy = torch.randn(2, 128, 300, generator=g).cuda()
    xg = x.clone().requires_grad_(True)
    (layer(adj, xg) * gy).sum().backward()
    got = (xg.grad.clone(), layer.conv.weight.grad.clone(), layer.conv.bias.grad.clone())
    layer.zero_grad()
    xr = x.clone().requires_grad_(True)
    (layer.conv(torch.matmul(xr, adj.to_dense())) * gy).sum().backward()
    for a, b_ in zip(got, (xr.grad, layer.conv.weight.grad, layer.conv.bias.grad)):
        assert torch.allclose(a, b_, rtol=1e-4, atol=1e-4 * float(b_.abs().max()))


@pytest.mark.parametrize("n,k,c_in,c_out", [(1024, 25, 3, 64), (257, 5, 33, 7)])
def test_gcn_random_vs_oracle(pk, n, k, c_in, c_out):
    g = torch.Generator().manual_seed(n + k)
    coords = torch.randn(2, n, 3, generator=g)
    x = torch.randn(2, c_in, n, generator=g)
    layer = pk.GraphConvolution(c_in, c_out)
    w64 = {kk: v.detach().double() for kk, v in layer.state_dict().items()}
    adj64 = np.stack([orc.gcn_build_graph(coords[b].numpy(), k) for b in range(2)])
    y64 = orc.gcn_graph_convolution(w64, torch.from_numpy(adj64), x.double())
    y32 = orc.gcn_graph_convolution({kk: v.float() for kk, v in w64.items()}, torch.from_numpy(adj64).float(), x)
    with torch.no_grad():
        adj = pk.build_graph(coords.cuda(), k=k)
        y = layer.cuda()(adj, x.cuda())
        pooled = pk.GlobalPooling()(y)
    assert np.abs(adj.to_dense().cpu().numpy() - adj64).max() <= 2e-7
    assert np.abs(y.cpu().numpy() - y64.numpy()).max() <= anchored_tol(y32.numpy(), y64.numpy())
    assert np.abs(pooled.cpu().numpy() - orc.gcn_global_pooling(y64).numpy()).max() <= anchored_tol(y32.numpy(), y64.numpy())


# ------------------------------------------------------------------------------------------------
# A16: spherical harmonics / irrep basis (tools/spherical_harmonics.py, tools/basis.py)
# ------------------------------------------------------------------------------------------------
def test_sh_golden(pk, golden_dir):
    from math import pi
    from se3_equi_graph_registration_amd import sh
    fx = np.load(os.path.join(golden_dir, "sh_l6.npz"))
    max_l = int(fx["max_l"])
    cart = torch.from_numpy(fx["cart"]).cuda()
    sph = sh.get_spherical_from_cartesian(cart)
    assert np.abs(sph.cpu().numpy() - fx["sph_f64"]).max() <= 4 * np.abs(fx["sph_f32"] - fx["sph_f64"]).max() + 1e-6
    tol = anchored_tol(fx["y_f32"], fx["y_f64"])
    ys = sh.precompute_sh(torch.from_numpy(fx["sph_f64"]).cuda(), max_l)                    # fp64 spherical input
    y = torch.cat([ys[j] for j in range(max_l + 1)], dim=-1).cpu().numpy()
    assert y.dtype == np.float64 and np.abs(y - fx["y_f64"]).max() <= 1e-12
    ys = sh.precompute_sh(sph, max_l)                                                       # fp32 spherical input
    y = torch.cat([ys[j] for j in range(max_l + 1)], dim=-1).cpu().numpy()
    assert np.abs(y - fx["y_f64"]).max() <= tol
    theta, phi = torch.from_numpy(fx["theta"]).cuda(), torch.from_numpy(fx["phi"]).cuda()
    y = torch.cat([sh.get_spherical_harmonics(l, theta, phi) for l in range(8)], dim=-1).cpu().numpy()
    assert np.abs(y - fx["y_angles_f64"]).max() <= 1e-12
    assert np.abs(sh.get_spherical_harmonics_element(3, -2, theta, phi).cpu().numpy() - fx["y_angles_f64"][:, 9 + 1]).max() <= 1e-12


@pytest.mark.parametrize("max_degree,shape", [(1, (2, 500)), (3, (3, 7, 11)), (2, (40960,))])
def test_get_basis_vs_oracle(pk, max_degree, shape):
    from oracle import sh_oracle as so
    from se3_equi_graph_registration_amd import sh
    g = torch.Generator().manual_seed(max_degree)
    r = torch.randn(*shape, 3, generator=g)
    if len(shape) == 2:
        r[0, 0] = torch.tensor([0.0, 2.0, 0.0])            # polar axis and the zero vector
        r[0, 1] = 0.0
    n_check = min(r.reshape(-1, 3).shape[0], 2000)
    basis = sh.get_basis(r.cuda(), max_degree)
    assert len(basis) == (max_degree + 1) ** 2
    flat = r.reshape(-1, 3)[:n_check]
    ref64 = so.get_basis(flat.double(), max_degree)
    ref32 = so.get_basis(flat, max_degree)
    for key, val in basis.items():
        d_in, d_out = map(int, key.split(","))
        assert val.shape == (*shape, 1, 2 * d_out + 1, 1, 2 * d_in + 1, 2 * min(d_in, d_out) + 1)
        got = val.reshape(-1, *val.shape[len(shape):])[:n_check].cpu().numpy().astype(np.float64)
        assert np.abs(got - ref64[key].numpy()).max() <= anchored_tol(ref32[key].numpy(), ref64[key].numpy()), key


# ------------------------------------------------------------------------------------------------
# N2: evaluation metrics on the device (tools/evaluation_metrics.py)
# ------------------------------------------------------------------------------------------------
def test_eval_metrics_golden(pk, golden_dir):
    fx = np.load(os.path.join(golden_dir, "metrics_cases.npz"))
    t = lambda k: torch.from_numpy(fx[k]).cuda()
    m = pk.evaluate_batch(t("pred")[:, :3, :3], t("pred")[:, :3, 3], t("gt"), t("src"), t("tgt"))
    got = torch.stack([m["rotation_error"], m["translation_error"], m["recall"], m["precision"], m["f1"]], dim=1).cpu().numpy()
    ref = fx["metrics"]
    # counts are integers: recall / precision / f1 must agree exactly unless a distance sits on the threshold
    assert np.allclose(got[:, 2:], ref[:, 2:], rtol=0, atol=1e-12)
    assert np.allclose(got[:, 1], ref[:, 1], rtol=1e-12, atol=1e-9)
    # acos near 0 amplifies rounding of the trace: compare the angle through its cosine
    assert np.allclose(np.cos(np.radians(got[:, 0])), np.cos(np.radians(ref[:, 0])), rtol=0, atol=1e-14)
    assert np.allclose(got[:, 0], ref[:, 0], rtol=1e-6, atol=1e-5)
    re, te = pk.calculate_pose_error(t("gt")[3], t("pred")[3])
    rec, prec = pk.registration_recall(t("gt")[3], t("pred")[3], t("src")[3], t("tgt")[3])
    assert abs(float(te) - ref[3, 1]) < 1e-9 and abs(float(rec) - ref[3, 2]) < 1e-12 and abs(float(prec) - ref[3, 3]) < 1e-12


# ------------------------------------------------------------------------------------------------
# N1: PointNet encoder (train :52-96)
# ------------------------------------------------------------------------------------------------
def test_pointnet_golden(pk, golden_dir):
    fx = np.load(os.path.join(golden_dir, "pointnet_n200_k6.npz"))
    net = pk.PointNet(in_num_feature=5, hidden_num_feature=64, output_num_feature=32, device="cuda")
    net.load_state_dict({k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w.")})
    net = net.cuda()
    t = lambda k: torch.from_numpy(fx[k]).cuda()
    with torch.no_grad():
        h1 = net.conv1(t("feats"), t("pos"), t("edge_index"))
        pooled = net(t("pos"), t("feats"), t("edge_index"), t("batch"))
    assert np.abs(h1.cpu().numpy() - fx["conv1_f64"]).max() <= anchored_tol(fx["conv1_f32"], fx["conv1_f64"])
    assert (h1.cpu().numpy()[3] == 0).all()
    assert np.abs(pooled.cpu().numpy() - fx["pooled_f64"]).max() <= anchored_tol(fx["pooled_f32"], fx["pooled_f64"])
    # trainable: the whole encoder's gradients against the oracle's autograd (fp64)
    gy = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(2)).cuda()
    feats = t("feats").clone().requires_grad_(True)
    posg = t("pos").clone().requires_grad_(True)                 # round 6: HIP backward, the coordinates' gradient included
    net.zero_grad()
    (net(posg, feats, t("edge_index"), t("batch")) * gy).sum().backward()
    w = {k: v.detach().cpu().double().requires_grad_(True) for k, v in net.state_dict().items()}
    fr = t("feats").cpu().double().requires_grad_(True)
    pr = t("pos").cpu().double().requires_grad_(True)
    _, ref = orc.pointnet_forward(w, pr, fr, t("edge_index").cpu(), t("batch").cpu())
    (ref * gy.cpu().double()).sum().backward()
    assert torch.allclose(feats.grad.cpu().double(), fr.grad, rtol=2e-4, atol=2e-4 * float(fr.grad.abs().max()))
    assert torch.allclose(posg.grad.cpu().double(), pr.grad, rtol=2e-4, atol=2e-4 * float(pr.grad.abs().max()))
    for name, p_ in net.named_parameters():
        assert torch.allclose(p_.grad.cpu().double(), w[name].grad, rtol=2e-4, atol=2e-4 * float(w[name].grad.abs().max()) + 1e-9), name


def test_pointnet_reference_call_shape(pk):
    """The encoder as the training script means to use it (train :1034-1038): 3-d coordinates as features, kNN graph
    of each cloud, per-point output reshaped to [batch, 2048, 32]; checked against the oracle."""
    g = torch.Generator().manual_seed(5)
    b, n, k = 2, 2048, 16
    pos = torch.rand(b, n, 3, generator=g) * 3
    net = pk.PointNet(in_num_feature=3, hidden_num_feature=32, output_num_feature=32, device="cuda").cuda()
    edges, _, _ = pk.knn_graph_batched(pos.cuda(), k, loop=True)
    ei = torch.cat([edges[c] + c * n for c in range(b)], dim=1)
    flat = pos.reshape(b * n, 3)
    with torch.no_grad():
        enc = net.encode(flat.cuda(), flat.cuda(), ei).view(b, n, -1)
    w64 = {kk: v.detach().cpu().double() for kk, v in net.state_dict().items()}
    batch = torch.arange(b).repeat_interleave(n)
    ref64, _ = orc.pointnet_forward(w64, flat.double(), flat.double(), ei.cpu(), batch)
    ref32, _ = orc.pointnet_forward({kk: v.float() for kk, v in w64.items()}, flat, flat, ei.cpu(), batch)
    assert enc.shape == (b, n, 32)
    assert np.abs(enc.reshape(b * n, -1).cpu().numpy() - ref64.numpy()).max() <= anchored_tol(ref32.numpy(), ref64.numpy())


# ------------------------------------------------------------------------------------------------
# N3: descriptor matching + correspondence labels (data_preprocess/3DMatch_Feature.py:157-196)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,m,dim,mutual", [(2048, 2048, 32, False), (2048, 2048, 33, True), (1000, 777, 33, True), (5, 3, 7, False)])
def test_build_correspondences_vs_oracle(pk, n, m, dim, mutual):
    rng = np.random.default_rng(n + m + dim)
    src_desc = rng.standard_normal((n, dim)).astype(np.float32)
    src_desc /= np.linalg.norm(src_desc, axis=1, keepdims=True)
    tgt_desc = rng.standard_normal((m, dim)).astype(np.float32)
    k = min(n, m) // 2
    tgt_desc[:k] = src_desc[rng.permutation(n)[:k]] + 0.05 * rng.standard_normal((k, dim)).astype(np.float32)   # true matches
    tgt_desc /= np.linalg.norm(tgt_desc, axis=1, keepdims=True)
    src_pts = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    pose = np.eye(4, dtype=np.float32)
    th = 0.7
    pose[:3, :3] = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], dtype=np.float32)
    pose[:3, 3] = (0.3, -0.2, 0.5)
    tgt_pts = rng.uniform(-1, 1, (m, 3)).astype(np.float32)
    corr_o, labels_o, dis_o, dmat = orc.build_correspondences(src_desc, tgt_desc, src_pts, tgt_pts, pose, mutual, 0.10)
    # make a known fraction of the oracle's matches geometric inliers, then recompute
    sel = corr_o[: max(1, corr_o.shape[0] // 2)]
    tgt_pts[sel[:, 1]] = src_pts[sel[:, 0]] @ pose[:3, :3].T + pose[:3, 3] + 0.01 * rng.standard_normal((sel.shape[0], 3)).astype(np.float32)
    corr_o, labels_o, dis_o, dmat = orc.build_correspondences(src_desc, tgt_desc, src_pts, tgt_pts, pose, mutual, 0.10)
    t = lambda a: torch.from_numpy(a).cuda()
    corr, labels, dis = pk.build_correspondences(t(src_desc), t(tgt_desc), t(src_pts), t(tgt_pts), t(pose), use_mutual=mutual,
                                                 inlier_threshold=0.10)
    # index parity is exact wherever the best and second-best distances differ by more than fp32 round-off of the dot product
    part = np.partition(dmat, 1, axis=1)[:, :2] if m > 1 else None
    clear_rows = (part[:, 1] - part[:, 0]) > 1e-5
    idx, _ = pk.match_descriptors(t(src_desc), t(tgt_desc))
    assert clear_rows.mean() > 0.99
    assert np.array_equal(idx.cpu().numpy()[clear_rows], np.argmin(dmat, axis=1)[clear_rows])
    assert np.abs(dis.cpu().numpy() - dis_o).max() <= 2e-4            # sqrt amplifies dot round-off near distance 0
    if clear_rows.all() and (not mutual or ((np.partition(dmat, 1, axis=0)[1] - np.partition(dmat, 1, axis=0)[0]) > 1e-5).all()):
        assert np.array_equal(corr.cpu().numpy(), corr_o)
        assert np.array_equal(labels.cpu().numpy(), labels_o)
        assert 0 < labels_o.sum() < labels_o.size or n < 10


def test_match_kernels_golden(pk, golden_dir):
    """csrc/match.hip against the correspondences and labels the REAL data_preprocess/3DMatch_Feature.py produced on the same
    descriptors / key points (tests/golden/match_cases.npz, both splits, with and without the mutual check).  Index parity is exact
    on every row / column whose best and second-best descriptor distances differ by more than the fp32 round-off of a 33-term dot
    product (the fixture stores that gap); labels on every correspondence further than 1e-5 from the 0.10 m threshold."""
    fx = np.load(os.path.join(golden_dir, "match_cases.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    for name in fx["cases"]:
        g = lambda k: fx[f"{name}.{k}"]  # noqa: E731
        mutual = bool(g("mutual"))
        idx, _ = pk.match_descriptors(t(g("feat_0")), t(g("feat_1")))
        rows = g("row_gap") > 2e-6
        assert rows.mean() > 0.99
        s64, t64 = g("feat_0").astype(np.float64), g("feat_1").astype(np.float64)
        want_idx = np.argmin(2 - 2 * (s64 @ t64.T), axis=1)
        assert np.array_equal(idx.cpu().numpy()[rows], want_idx[rows]), name
        corr, labels, _ = pk.build_correspondences(t(g("feat_0")), t(g("feat_1")), t(g("xyz_0")), t(g("xyz_1")),
                                                   t(g("gt_pose").astype(np.float32)), use_mutual=mutual, inlier_threshold=0.10)
        corr, labels = corr.cpu().numpy(), labels.cpu().numpy()
        if rows.all() and (not mutual or (g("col_gap") > 2e-6).all()):
            assert np.array_equal(corr, g("corr")), name
            clear = g("thr_gap") > 1e-5
            assert clear.mean() > 0.99 and np.array_equal(labels[clear], g("labels")[clear]), name
        else:                               # a near-tie somewhere: compare the correspondences both sides agree on
            ref = {tuple(r): l for r, l in zip(g("corr").tolist(), g("labels").tolist())}
            common = [i for i, r in enumerate(corr.tolist()) if tuple(r) in ref]
            assert len(common) > 0.98 * len(ref), name


def test_egnn_degenerate_graphs_vs_oracle(pk):
    """Edge cases of the graph input: no edges at all (every node isolated), a single node, one self loop."""
    torch.manual_seed(3)
    m = pk.EGNN(32, 32, 32, in_edge_nf=1, device="cuda", n_layers=2)
    w64 = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    cases = [(17, torch.zeros(2, 0, dtype=torch.int64)), (1, torch.zeros(2, 0, dtype=torch.int64)),
             (1, torch.zeros(2, 1, dtype=torch.int64)), (5, torch.tensor([[0, 1, 1, 4], [1, 0, 1, 4]]))]
    for n, edges in cases:
        h = torch.randn(n, 32, generator=g)
        x = torch.rand(n, 3, generator=g)
        ea = torch.ones(edges.shape[1], 1)
        rh, rx = orc.egnn_forward(w64, h.double(), x.double(), [edges[0], edges[1]], ea.double())
        with torch.no_grad():
            gh, gx = m(h.cuda(), x.cuda(), [edges[0].cuda(), edges[1].cuda()], ea.cuda())
        assert torch.allclose(gh.cpu().double(), rh, rtol=1e-4, atol=1e-5) and torch.allclose(gx.cpu().double(), rx, rtol=1e-4, atol=1e-5)
    # and the backward on the empty graph: finite gradients equal to the oracle's autograd
    h = torch.randn(9, 32, generator=g); x = torch.rand(9, 3, generator=g)
    edges = torch.zeros(2, 0, dtype=torch.int64)
    wref = {k: v.clone().requires_grad_(True) for k, v in w64.items()}
    rh, rx = orc.egnn_forward(wref, h.double(), x.double(), [edges[0], edges[1]], torch.ones(0, 1, dtype=torch.float64))
    (rh.sum() + rx.sum()).backward()
    m.zero_grad()
    gh, gx = m(h.cuda(), x.cuda(), [edges[0].cuda(), edges[1].cuda()], torch.ones(0, 1).cuda())
    (gh.sum() + gx.sum()).backward()
    for k, p in m.named_parameters():
        ref = wref[k].grad if wref[k].grad is not None else torch.zeros_like(wref[k])
        assert torch.allclose(p.grad.cpu().double(), ref, rtol=1e-4, atol=1e-5), k


def test_tiny_inputs(pk):
    """Smallest legal inputs: one-point cloud, k == n, one descriptor, one correspondence, a PointNet graph without edges."""
    one = torch.rand(1, 3)
    assert torch.equal(pk.knn_graph(one.cuda(), 1, loop=True).cpu(), torch.zeros(2, 1, dtype=torch.int64))
    x = torch.rand(7, 3, generator=torch.Generator().manual_seed(0))
    assert torch.equal(pk.knn_graph(x.cuda(), 7, loop=True).cpu(), orc.knn_graph(x, 7, loop=True))
    assert torch.equal(pk.knn_graph(x.cuda(), 6, loop=False).cpu(), orc.knn_graph(x, 6, loop=False))
    idx, dist = pk.match_descriptors(torch.tensor([[1.0, 0.0]]).cuda(), torch.tensor([[0.6, 0.8]]).cuda())
    assert idx.tolist() == [0] and abs(float(dist[0]) - (2 - 1.2 + 1e-6) ** 0.5) < 1e-6
    corr, labels, _ = pk.build_correspondences(torch.tensor([[1.0, 0.0]]).cuda(), torch.tensor([[1.0, 0.0]]).cuda(), one.cuda(), one.cuda(),
                                               torch.eye(4).cuda())
    assert corr.tolist() == [[0, 0]] and labels.tolist() == [1]
    R, t = pk.weighted_kabsch(x[None].cuda(), x[None].cuda(), torch.ones(1, 7).cuda())
    assert torch.allclose(R.cpu(), torch.eye(3)[None], atol=1e-5) and torch.allclose(t.cpu(), torch.zeros(1, 3), atol=1e-5)
    net = pk.PointNetLayer(3, 8).cuda()
    with torch.no_grad():
        out = net(x.cuda(), x.cuda(), torch.zeros(2, 0, dtype=torch.int64).cuda())
    assert out.shape == (7, 8) and float(out.abs().max()) == 0.0          # no incoming edges: PyG's max-aggregation fill


def test_pair_without_valid_correspondences_matches_reference_semantics(pk):
    """labels == 0 for a whole pair: the reference sets R = I, t = 0 (train :708-711) and its pose loss then divides by |t| = 0:
    the loss is NaN there, and here (torch.clamp propagates NaN); the other pair of the batch is unaffected."""
    torch.manual_seed(0)
    egnn = pk.EGNN(32, 32, 32, in_edge_nf=1, device="cuda", n_layers=2)
    model = pk.CrossAttentionPoseRegression(egnn, num_nodes=256, hidden_nf=32, device="cuda").cuda()
    b = pk.synthetic_batch(5, 2, n=256, feat=32, device="cuda")
    b["labels"][0] = 0
    graph = pk.build_pair_graph(b["src"], b["tgt"], 8)
    with torch.no_grad():
        out = model(b["feat_src"], b["src"], graph, None, b["feat_tgt"], b["tgt"], graph, None, b["corr"], b["labels"], b["gt_pose"])
        loss, rl, tl = pk.registration_loss(out[0], out[1], out[2], b["gt_pose"])
    assert torch.equal(out[0][0].cpu(), torch.eye(3)) and torch.equal(out[1][0].cpu(), torch.zeros(3))
    o_rl, o_tl = orc.pose_loss(out[0].cpu(), out[1].cpu(), b["gt_pose"].cpu())
    assert torch.isnan(o_tl[0]) and torch.isnan(tl[0]) and torch.isnan(loss)
    assert torch.allclose(rl.cpu(), o_rl, atol=1e-5) and torch.allclose(tl[1].cpu(), o_tl[1], atol=1e-5)


@pytest.mark.parametrize("feat,B,n,k,scale", [(32, 3, 1000, 7, 3.0), (33, 2, 777, 9, 60.0), (32, 1, 130, 5, 1.0), (64, 2, 500, 6, 3.0),
                                              (16, 1, 300, 5, 3.0)])
def test_pair_train_head_odd_shapes_vs_oracle(pk, feat, B, n, k, scale):
    """Shapes that are not multiples of any tile size (points, k, batch), KITTI-scale coordinates, both hidden widths:
    R, t, the correspondence loss, the equi loss and the top-k index sets against the oracle."""
    torch.manual_seed(1)
    egnn = pk.EGNN(feat, feat, feat, in_edge_nf=1, device="cuda", n_layers=2)
    with torch.no_grad():
        egnn.embedding_out.weight.mul_(0.05); egnn.embedding_out.bias.mul_(0.05)
    model = pk.CrossAttentionPoseRegression(egnn, num_nodes=n, hidden_nf=feat, device="cuda").cuda()
    b = pk.synthetic_batch(11, B, n=n, feat=feat, scale=(scale, scale, scale / 4), device="cuda")
    graph = pk.build_pair_graph(b["src"], b["tgt"], k)
    with torch.no_grad():
        out = model(b["feat_src"], b["src"], graph, None, b["feat_tgt"], b["tgt"], graph, None, b["corr"], b["labels"], b["gt_pose"])
    we = {kk: v.detach().cpu().double() for kk, v in egnn.state_dict().items()}
    wc = {kk: v.detach().cpu().double() for kk, v in model.state_dict().items() if not kk.startswith("egnn.")}
    e_s = torch.stack([orc.knn_graph(b["src"][i].cpu(), k) for i in range(B)])
    e_t = torch.stack([orc.knn_graph(b["tgt"][i].cpu(), k) for i in range(B)])
    ea = torch.ones(B, n * k, 1, dtype=torch.float64)
    d = lambda name: b[name].cpu().double()
    ref, extras = orc.pair_forward_train(we, wc, d("feat_src"), d("src"), e_s, ea, d("feat_tgt"), d("tgt"), e_t, ea, d("corr"),
                                         d("labels"), d("gt_pose"))
    assert float((out[0].cpu().double() - ref[0]).abs().max()) <= 1e-4
    assert float((out[1].cpu().double() - ref[1]).abs().max()) <= 1e-4 * max(1.0, scale)
    assert abs(float(out[2]) - float(ref[2])) <= 2e-4 * max(1.0, abs(float(ref[2])))
    assert abs(float(out[3]) - float(ref[3])) <= 2e-4 * max(1.0, abs(float(ref[3])))
    for i in range(B):
        assert set(model.last_top_idx[i].cpu().tolist()) == set(extras["top_idx"][i].tolist())


@pytest.mark.parametrize("feat,B,n,k,layers", [(33, 2, 384, 10, 2), (32, 3, 384, 10, 2), (48, 2, 384, 10, 2),
                                                (32, 1, 2048, 20, 4)])      # the last = BASELINE configs[1] at full size
def test_train_step_gradients_vs_oracle_autograd(pk, feat, B, n, k, layers):
    """The whole training loss (train :1120) through the fused head, Kabsch, both EGNN passes: every parameter gradient
    against the oracle's autograd in fp32 and fp64 (hidden_nf = 33 exercises the border path of the MFMA backward; the
    2 048-point / k = 20 / 4-layer case is the headline workload itself, on a well-conditioned head)."""
    torch.manual_seed(2)
    egnn = pk.EGNN(feat, feat, feat, in_edge_nf=1, device="cuda", n_layers=layers)
    with torch.no_grad():
        egnn.embedding_out.weight.mul_(0.02); egnn.embedding_out.bias.mul_(0.02)          # well-conditioned head
    model = pk.CrossAttentionPoseRegression(egnn, num_nodes=n, hidden_nf=feat, device="cuda").cuda()
    b = pk.synthetic_batch(21, B, n=n, feat=feat, device="cuda")
    b["tgt"] = b["tgt"] + 0.05 * torch.randn(b["tgt"].shape, generator=torch.Generator().manual_seed(3)).cuda()   # away from arccos'(1)
    graph = pk.build_pair_graph(b["src"], b["tgt"], k)
    out = model(b["feat_src"], b["src"], graph, None, b["feat_tgt"], b["tgt"], graph, None, b["corr"], b["labels"], b["gt_pose"])
    loss, _, _ = pk.registration_loss(out[0], out[1], out[2], b["gt_pose"])
    model.zero_grad()
    loss.backward()
    e_s = torch.stack([orc.knn_graph(b["src"][i].cpu(), k) for i in range(B)])
    e_t = torch.stack([orc.knn_graph(b["tgt"][i].cpu(), k) for i in range(B)])
    refs = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        we = {kk: v.detach().cpu().to(dt).requires_grad_(True) for kk, v in egnn.state_dict().items()}
        wc = {kk: v.detach().cpu().to(dt).requires_grad_(True) for kk, v in model.state_dict().items()
              if not kk.startswith("egnn.") and v.dtype.is_floating_point}
        d = lambda name: b[name].cpu().to(dt)
        ea = torch.ones(B, n * k, 1, dtype=dt)
        ref, _ = orc.pair_forward_train(we, wc, d("feat_src"), d("src"), e_s, ea, d("feat_tgt"), d("tgt"), e_t, ea, d("corr"),
                                        d("labels"), d("gt_pose"))
        rl, tl = orc.pose_loss(ref[0], ref[1], d("gt_pose"))
        l = ref[2].mean() + rl.mean() + tl.mean()
        l.backward()
        refs[tag] = dict(loss=float(l.detach()), **{"egnn." + kk: v.grad.numpy() for kk, v in we.items() if v.grad is not None},
                         **{kk: v.grad.numpy() for kk, v in wc.items() if v.grad is not None})
    np.testing.assert_allclose(float(loss), refs["f64"]["loss"], rtol=1e-4)
    checked = 0
    for pn, p in model.named_parameters():
        if pn in refs["f64"]:
            assert p.grad is not None, pn
            _grad_check(p.grad, refs["f32"][pn], refs["f64"][pn], pn, k=8.0, rel=5e-5)
            checked += 1
    assert checked > 20


def test_standalone_egcl_layer_is_differentiable(pk):
    """E_GCL.forward (train :280-289) as a stand-alone trainable layer: outputs and every gradient against the oracle layer."""
    g = torch.Generator().manual_seed(12)
    n, k, f = 200, 7, 32
    x = torch.rand(n, 3, generator=g) * 2
    h = torch.randn(n, f, generator=g)
    e = orc.knn_graph(x, k)
    ea = torch.ones(e.shape[1], 1)
    torch.manual_seed(2)
    layer = pk.E_GCL(f, f, f, edges_in_d=1, num_heads=1, device="cuda").cuda()
    hg, xg = h.cuda().requires_grad_(True), x.cuda().requires_grad_(True)
    h2, x2, ea2 = layer(hg, [e[0].cuda(), e[1].cuda()], xg, ea.cuda())
    gh, gx = torch.randn(n, f, generator=g), torch.randn(n, 3, generator=g)
    ((h2 * gh.cuda()).sum() + (x2 * gx.cuda()).sum()).backward()
    w = {"gcl_0." + kk: v.detach().cpu().double().requires_grad_(True) for kk, v in layer.state_dict().items()}
    hr, xr = h.double().requires_grad_(True), x.double().requires_grad_(True)
    h_ref, x_ref = orc.egcl_layer(w, "gcl_0.", hr, xr, e[0], e[1], ea.double())
    ((h_ref * gh.double()).sum() + (x_ref * gx.double()).sum()).backward()
    assert torch.allclose(h2.detach().cpu().double(), h_ref.detach(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(x2.detach().cpu().double(), x_ref.detach(), rtol=1e-4, atol=1e-5)
    assert torch.allclose(hg.grad.cpu().double(), hr.grad, rtol=1e-3, atol=1e-4 * float(hr.grad.abs().max()))
    assert torch.allclose(xg.grad.cpu().double(), xr.grad, rtol=1e-3, atol=1e-4 * float(xr.grad.abs().max()))
    for name, p_ in layer.named_parameters():
        ref = w["gcl_0." + name].grad
        assert p_.grad is not None, name
        assert torch.allclose(p_.grad.cpu().double(), ref, rtol=1e-3, atol=2e-4 * float(ref.abs().max()) + 1e-9), name
    # a second call reuses the cached one-layer wrapper and accumulates into the same .grad tensors
    g1 = layer.coord_mlp[0].weight.grad.clone()
    h3, x3, _ = layer(hg, [e[0].cuda(), e[1].cuda()], xg, ea.cuda())
    ((h3 * gh.cuda()).sum() + (x3 * gx.cuda()).sum()).backward()
    assert torch.allclose(layer.coord_mlp[0].weight.grad, 2 * g1, rtol=1e-5, atol=1e-8)


def test_torch_library_ops_match_the_ctypes_path(pk):
    """torch.ops.equigspr.* (TORCH_LIBRARY registration, csrc/torch_ops.cpp) call the same C-ABI functions as the package's ctypes
    binding: identical outputs, and the dispatcher-level argument checks raise."""
    from se3_equi_graph_registration_amd import torch_library
    ops = torch_library.load()
    g = torch.Generator().manual_seed(23)
    n, k = 500, 9
    x = (torch.rand(n, 3, generator=g) * 2).cuda()
    h = torch.randn(n, 32, generator=g).cuda()
    e = ops.knn_graph(x, k, True)
    assert torch.equal(e, pk.knn_graph(x, k, loop=True))
    row32, col32 = e[0].int().contiguous(), e[1].int().contiguous()
    ptr, perm, inv = ops.build_inverse_index(row32, n)
    order = np.argsort(e[0].cpu().numpy(), kind="stable")
    np.testing.assert_array_equal(perm.cpu().numpy(), order)
    assert int(ptr[-1]) == n * k and torch.equal(inv[perm.long()].cpu(), torch.arange(n * k, dtype=torch.int32))
    torch.manual_seed(1)
    m = pk.EGNN(32, 32, 32, in_edge_nf=1, device="cuda", n_layers=2)
    with torch.no_grad():
        h_ref, x_ref = m(h, x, [e[0], e[1]], torch.ones(n * k, 1, device="cuda"))
    m._ensure_flat()
    h_o, x_o = ops.egnn_forward(m.flat_params, h, x, row32, col32, torch.ones(n * k, device="cuda"), 32, 32, 2, 1, 0)
    assert torch.equal(h_o, h_ref) and torch.equal(x_o, x_ref)
    src, tgt = torch.rand(3, 200, 3, generator=g).cuda(), torch.rand(3, 200, 3, generator=g).cuda()
    w = torch.softmax(torch.randn(3, 200, generator=g), dim=1).cuda()
    R, t, aux = ops.weighted_kabsch_forward(src, tgt, w)
    R2, t2 = pk.weighted_kabsch(src, tgt, w)
    assert torch.equal(R, R2) and torch.equal(t, t2)
    gs, gt_, gw = ops.weighted_kabsch_backward(src, tgt, w, aux, torch.ones_like(R), torch.ones_like(t))
    sr, tr, wr = src.clone().requires_grad_(True), tgt.clone().requires_grad_(True), w.clone().requires_grad_(True)
    Rr, tt = pk.weighted_kabsch(sr, tr, wr)
    (Rr.sum() + tt.sum()).backward()
    assert torch.allclose(gs, sr.grad) and torch.allclose(gt_, tr.grad) and torch.allclose(gw, wr.grad)
    # autograd through the dispatcher: weighted_kabsch_forward has an Autograd kernel (the C-ABI backward) ...
    s2, t2_, w2 = src.clone().requires_grad_(True), tgt.clone().requires_grad_(True), w.clone().requires_grad_(True)
    Rd, td, _ = ops.weighted_kabsch_forward(s2, t2_, w2)
    (Rd.sum() + td.sum()).backward()
    assert torch.allclose(s2.grad, sr.grad) and torch.allclose(t2_.grad, tr.grad) and torch.allclose(w2.grad, wr.grad)
    # ... and so has egnn_forward: the training form + equigspr_egnn_backward.  Same gradients as the package's own autograd node.
    hp, xp = h.clone().requires_grad_(True), x.clone().requires_grad_(True)
    m.zero_grad()
    ho_p, xo_p = m(hp, xp, [e[0], e[1]], torch.ones(n * k, 1, device="cuda"))
    gh_, gx_ = torch.randn(n, 32, generator=g).cuda(), torch.randn(n, 3, generator=g).cuda()
    ((ho_p * gh_).sum() + (xo_p * gx_).sum()).backward()
    flat = m.flat_params.detach().clone().requires_grad_(True)
    hd, xd = h.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ho_d, xo_d = ops.egnn_forward(flat, hd, xd, row32, col32, torch.ones(n * k, device="cuda"), 32, 32, 2, 1, 0)
    assert torch.equal(ho_d, ho_p.detach()) and torch.equal(xo_d, xo_p.detach())
    ((ho_d * gh_).sum() + (xo_d * gx_).sum()).backward()
    assert torch.equal(flat.grad, m.flat_grads) and torch.equal(hd.grad, hp.grad) and torch.equal(xd.grad, xp.grad)
    with pytest.raises(RuntimeError):
        ops.egnn_forward(m.flat_params[:-1].contiguous(), h, x, row32, col32, None, 32, 32, 2, 1, 0)
    with pytest.raises(RuntimeError, match="same length"):
        ops.egnn_forward(m.flat_params, h, x, row32, col32[:-1].contiguous(), torch.ones(n * k, device="cuda"), 32, 32, 2, 1, 0)
    with pytest.raises(RuntimeError, match="unsupported configuration"):
        ops.egnn_forward(m.flat_params, h, x, row32, col32, torch.ones(n * k, device="cuda"), 32, 32, 2, 5, 0)
    with pytest.raises(RuntimeError):
        ops.knn_graph(x.double(), k, True)


def test_torch_library_layer_head_and_loss_ops(pk):
    """The dispatcher ops SURVEY section 8(b) lists beyond the whole-model forward: one E_GCL layer (egcl_forward with autograd,
    egcl_backward as an explicit op) against the package's stand-alone E_GCL; the evaluation pair model's head and the pose loss against
    the ctypes path."""
    from se3_equi_graph_registration_amd import torch_library
    ops = torch_library.load()
    g = torch.Generator().manual_seed(5)
    n, k, F = 300, 7, 32
    x = (torch.rand(n, 3, generator=g) * 2).cuda()
    h = torch.randn(n, F, generator=g).cuda()
    e = pk.knn_graph(x, k, loop=True)
    row32, col32 = e[0].int().contiguous(), e[1].int().contiguous()
    torch.manual_seed(2)
    layer = pk.E_GCL(F, F, F, edges_in_d=1, num_heads=4).cuda()
    ea = torch.rand(n * k, 1, generator=g).cuda()
    hr, xr = h.clone().requires_grad_(True), x.clone().requires_grad_(True)
    h_ref, x_ref, _ = layer(hr, [e[0], e[1]], xr, ea)
    gh, gx = torch.randn(n, F, generator=g).cuda(), torch.randn(n, 3, generator=g).cuda()
    layer.zero_grad()
    ((h_ref * gh).sum() + (x_ref * gx).sum()).backward()
    flat = torch.cat([p.detach().reshape(-1) for p in layer.parameters()])
    h_o, x_o = ops.egcl_forward(flat, h, x, row32, col32, ea.reshape(-1), 4, 0)
    assert torch.equal(h_o, h_ref.detach()) and torch.equal(x_o, x_ref.detach())
    gp, ghi, gxi = ops.egcl_backward(flat, h, x, row32, col32, ea.reshape(-1), gh, gx, 4, 0)
    ref_gp = torch.cat([p.grad.reshape(-1) for p in layer.parameters()])
    assert torch.allclose(gp, ref_gp, rtol=1e-5, atol=1e-6) and torch.allclose(ghi, hr.grad, rtol=1e-5, atol=1e-6) and torch.allclose(gxi, xr.grad, rtol=1e-5, atol=1e-6)
    fp = flat.clone().requires_grad_(True)                      # autograd through the composite op reaches the layer block
    h2, x2 = ops.egcl_forward(fp, h, x, row32, col32, ea.reshape(-1), 4, 0)
    ((h2 * gh).sum() + (x2 * gx).sum()).backward()
    assert torch.allclose(fp.grad, ref_gp, rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError, match="layer block"):
        ops.egcl_forward(flat[:-1], h, x, row32, col32, ea.reshape(-1), 4, 0)
    # pose loss: forward, explicit backward, autograd
    B = 4
    q, _ = torch.linalg.qr(torch.randn(B, 3, 3, generator=g)); q = (q * torch.sign(torch.det(q))[:, None, None]).contiguous().cuda()
    gq, _ = torch.linalg.qr(torch.randn(B, 3, 3, generator=g)); gq = gq * torch.sign(torch.det(gq))[:, None, None]
    pose = torch.eye(4).repeat(B, 1, 1); pose[:, :3, :3] = gq; pose[:, :3, 3] = torch.rand(B, 3, generator=g); pose = pose.cuda()
    t = torch.rand(B, 3, generator=g).cuda()
    rot, tr = ops.pose_loss_forward(q, t, pose)
    rot2, tr2 = pk.pose_loss(q, t, pose)
    assert torch.equal(rot, rot2) and torch.equal(tr, tr2)
    qa, ta = q.clone().requires_grad_(True), t.clone().requires_grad_(True)
    r3, t3 = ops.pose_loss_forward(qa, ta, pose)
    (r3.sum() + 2 * t3.sum()).backward()
    gR, gt_ = ops.pose_loss_backward(q, t, pose, torch.ones(B, device="cuda"), torch.full((B,), 2.0, device="cuda"))
    assert torch.equal(qa.grad, gR) and torch.equal(ta.grad, gt_)
    # evaluation head against the package's fused evaluation head (same C entry)
    Bp, N, K = 1, 512, 128
    torch.manual_seed(3)
    egnn = pk.EGNN(F, F, F, in_edge_nf=1, device="cuda", n_layers=1)
    model = pk.CrossAttentionPoseRegression(egnn, num_nodes=N, hidden_nf=F, device="cuda", variant="eval").cuda().eval()
    b = pk.synthetic_batch(77, Bp, n=N, feat=F, device="cuda")
    graph = pk.build_pair_graph(b["src"], b["tgt"], 8)
    with torch.no_grad():
        out = model(b["feat_src"], b["src"], graph, None, b["feat_tgt"], b["tgt"], graph, None, b["corr"], b["labels"], b["gt_pose"])
        h_all = torch.cat([out[4].reshape(N, F), out[6].reshape(N, F)]).contiguous()
        x_all = torch.cat([out[5].reshape(N, 3), out[7].reshape(N, 3)]).contiguous()
        mlp = torch.cat([p.detach().reshape(-1) for p in model.mlp.parameters()])
        R, tt, equi, wts, top_idx = ops.head_eval_forward(h_all, x_all, b["feat_src"].contiguous(), b["feat_tgt"].contiguous(), b["src"].contiguous(),
                                                         b["tgt"].contiguous(), b["labels"], b["gt_pose"], mlp, K)
    assert torch.equal(R, out[0]) and torch.equal(tt, out[1]) and torch.equal(equi, out[3]) and torch.equal(top_idx, model.last_top_idx)
    assert torch.equal(wts, model.last_weights)


def test_edge_kernels_are_reproducible_when_waves_run_many_units(pk):
    """Regression guard for the split-bf16 products (csrc/egnn_mfma16.h mv16_bf, EQ_BF_GUARD): without the closed instruction groups a
    load returning early into an operand register of a queued matrix instruction corrupted 1-6 of 3 000 tiles, different ones every run
    (profiles/r05_bf16x3_edge_kernels.txt) -- only where a wave's persistent loop runs several units.  A ring graph of 40 000 nodes x 16
    edges gives every wave ~20 tiles (5 units); forward outputs and all gradients must be bit-identical run to run, and agree (2e-5)
    with the same nodes in a graph cut to 1/8 of the ring, where every wave runs ONE unit (away from the cut)."""
    g = torch.Generator().manual_seed(5)

    def run(n):
        torch.manual_seed(2)
        m = pk.EGNN(32, 32, 32, in_edge_nf=1, device="cuda", n_layers=1, num_heads=1)
        gg = torch.Generator().manual_seed(7)
        h = torch.randn(40000, 32, generator=gg)[:n].cuda()
        x = (torch.rand(40000, 3, generator=gg) * 2)[:n].cuda()
        ea = torch.rand(40000 * 16, 1, generator=gg)[:n * 16].cuda()
        gh = torch.randn(40000, 32, generator=gg)[:n].cuda()
        gx = torch.randn(40000, 3, generator=gg)[:n].cuda()
        row = torch.arange(n).repeat_interleave(16)
        col = (row + 1) % n
        hc, xc = h.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ho, xo = m(hc, xc, [row.cuda(), col.cuda()], ea)
        m.zero_grad()
        ((ho * gh).sum() + (xo * gx).sum()).backward()
        return ho.detach(), xo.detach(), hc.grad, xc.grad, m.flat_grads.clone()

    a, b = run(40000), run(40000)
    for u, v in zip(a, b):
        assert torch.isfinite(u).all() and torch.equal(u, v)
    c = run(5000)                                            # the same leading nodes in a small graph: one unit per wave, other node kernels
    sl = slice(100, 4800)                                    # away from the ring's closing edge
    for u, v in zip(a[:4], c[:4]):
        assert float((u[sl] - v[sl]).abs().max()) <= 2e-5 * max(1.0, float(v[sl].abs().max()))


@pytest.mark.parametrize("feat", [32, 33])
@pytest.mark.parametrize("n,caps", [(3000, (1, 8, 0)), (40000, (8, 0))])
def test_split_products_match_fp32_products_tile_by_tile(pk, feat, n, caps):
    """The split-bf16 products against the SAME kernels on fp32 products (equigspr_set_edge_products, ABI 14), tile by tile: a ring graph
    whose node k owns the 16-edge tile k, so one wrong tile (round 5's defect -- packed fp32 instructions miscomputing the lane-per-edge geometry
    of ~0.5 % of the chunks, profiles/r06_mfma_hazard_probe.txt: ~1e-2 of the tile's values) shows at nodes k / k + 1.  Workgroup caps (equigspr_set_edge_grid_cap) 1 / 8 put up to ~90 units on every wave with
    two waves per SIMD -- the regime the defect needed; every run must also be bit-identical to the first."""
    from se3_equi_graph_registration_amd import _lib
    L = _lib.lib()

    def run(fp32, cap):
        _lib.check(L.equigspr_set_edge_products(1 if fp32 else 0), "set_edge_products")
        _lib.check(L.equigspr_set_edge_grid_cap(cap), "set_edge_grid_cap")
        torch.manual_seed(2)
        m = pk.EGNN(feat, feat, feat, in_edge_nf=1, device="cuda", n_layers=1, num_heads=1)
        g = torch.Generator().manual_seed(5)
        h = torch.randn(n, feat, generator=g).cuda()
        x = (torch.rand(n, 3, generator=g) * 2).cuda()
        row = torch.arange(n).repeat_interleave(16)
        col = (row + 1) % n
        ea = torch.rand(row.numel(), 1, generator=g).cuda()
        gh, gx = torch.randn(n, feat, generator=g).cuda(), torch.randn(n, 3, generator=g).cuda()
        hc, xc = h.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ho, xo = m(hc, xc, [row.cuda(), col.cuda()], ea)
        m.zero_grad()
        ((ho * gh).sum() + (xo * gx).sum()).backward()
        return ho.detach(), xo.detach(), hc.grad, xc.grad, m.flat_grads.clone()

    try:
        ref = run(True, 0)
        assert all(torch.equal(u, v) for u, v in zip(ref, run(True, 0)))
        for cap in caps:
            first = run(False, cap)
            for name, u, v in zip(("h_out", "x_out", "grad_h", "grad_x"), first[:4], ref[:4]):
                d = (u - v).abs().reshape(n, -1).max(dim=1).values / v.abs().max()
                assert float(d.max()) <= 2e-5, (name, cap, int((d > 2e-5).sum()), float(d.max()))
            assert float((first[4] - ref[4]).abs().max()) <= 2e-5 * float(ref[4].abs().max()), cap
            for _ in range(2):
                again = run(False, cap)
                assert all(torch.equal(u, v) for u, v in zip(first, again)), cap
    finally:
        L.equigspr_set_edge_products(0)
        L.equigspr_set_edge_grid_cap(0)
