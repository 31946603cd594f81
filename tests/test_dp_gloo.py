"""CPU, world_size=2 over gloo: the data-parallel step (flat buckets, shard ranges, bucketed
all-reduce) used by bench.py --gpus N.  The HIP kernels cannot run here, so the EGNN is stood in
by a tiny module exposing the same flat_params / flat_grads contract; what is under test is the
N>1 host path: sharding, broadcast, ONE collective per step over the joint bucket, averaged gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from se3_equi_graph_registration_amd.dp import DataParallelStep, FlatBucket, shard_range


class FakeEgnn(nn.Module):
    """Same contract as EGNN: parameters are views of flat_params, grads views of flat_grads."""

    def __init__(self):
        super().__init__()
        self.lin = nn.Linear(4, 3)
        b = FlatBucket(self.parameters())
        self.flat_params, self.flat_grads = b.flat, b.grad
        self._bucket = b

    def zero_grad(self, set_to_none=False):
        self._bucket.zero_grad()

    def _attach_grads(self):
        self._bucket.zero_grad(fill=False)

    def rehome(self, flat, grads):                     # EGNN.rehome contract: caller-owned storage
        b = FlatBucket(self.parameters(), storage=(flat, grads))
        self.flat_params, self.flat_grads, self._bucket = b.flat, b.grad, b

    def forward(self, x):
        return self.lin(x)


class Pair(nn.Module):
    def __init__(self):
        super().__init__()
        self.egnn = FakeEgnn()
        self.mlp = nn.Sequential(nn.Linear(3, 5), nn.ReLU(), nn.Linear(5, 1))
        self.unused = nn.Linear(2, 2)

    def forward(self, x):
        return self.mlp(self.egnn(x)).sum()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)              # different init per rank: broadcast must fix it
    model = Pair()
    dp = DataParallelStep(model, model.egnn)
    dp.broadcast_parameters(0)
    data = torch.arange(8 * 4, dtype=torch.float32).view(8, 4) / 10.0
    lo, hi = shard_range(8, rank, world)
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    dp.zero_grad()
    model(data[lo:hi]).backward()
    dp.sync()
    dist.all_reduce = orig
    q.put((rank, model.egnn.flat_params.clone(), model.egnn.flat_grads.clone(), dp.head.grad.clone(), len(calls),
           [p.grad is not None and p.grad.data_ptr() != 0 for p in model.mlp.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def _run_ranks(target, world=2, attempts=3):
    """Spawn ``world`` ranks over gloo on 127.0.0.1.  The rendezvous port is picked by bind-to-0-then-close, which another
    process can grab in between (seen about once in forty runs): a failed rendezvous is retried on a fresh port."""
    ctx = mp.get_context("spawn")
    for attempt in range(attempts):
        port = _free_port()
        q = ctx.Queue()
        procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            res = [q.get(timeout=120) for _ in range(world)]
        except Exception:
            res = None
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        if res is not None and all(p.exitcode == 0 for p in procs):
            return res
    raise AssertionError(f"{world}-rank gloo job failed {attempts} times")


def test_two_rank_gradient_sync_matches_full_batch():
    world = 2
    res = sorted(_run_ranks(_worker, world), key=lambda t: t[0])
    (_, p0, g0, h0, n0, _), (_, p1, g1, h1, n1, _) = res
    assert torch.equal(p0, p1)                                   # broadcast made the replicas identical
    assert torch.equal(g0, g1) and torch.equal(h0, h1)           # all-reduced buckets agree
    assert n0 == 1 and n1 == 1                                   # joint bucket: exactly one collective per step
    # reference: single process, rank-0 weights, mean of the two shard gradients
    torch.manual_seed(100)
    ref = Pair()
    data = torch.arange(8 * 4, dtype=torch.float32).view(8, 4) / 10.0
    acc_e, acc_h = torch.zeros_like(g0), torch.zeros_like(h0)
    for r in range(world):
        lo, hi = shard_range(8, r, world)
        ref.zero_grad()
        ref(data[lo:hi]).backward()
        acc_e += torch.cat([p.grad.reshape(-1) for p in ref.egnn.parameters()])
        acc_h += torch.cat([p.grad.reshape(-1) for p in list(ref.mlp.parameters()) + list(ref.unused.parameters())
                            if p.grad is not None] + [torch.zeros(sum(p.numel() for p in ref.unused.parameters()))])[:h0.numel()]
    torch.testing.assert_close(g0, acc_e / world, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(h0[:acc_h.numel()], acc_h / world, rtol=1e-6, atol=1e-7)


def test_shard_range_partitions_contiguously():
    for n in (1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_flat_bucket_views_survive_zero_grad():
    m = nn.Linear(3, 2)
    b = FlatBucket(m.parameters())
    assert m.weight.data_ptr() == b.flat.data_ptr()
    m(torch.ones(1, 3)).sum().backward()
    assert float(b.grad.abs().sum()) > 0
    for p in m.parameters():
        p.grad = None
    b.zero_grad()
    assert all(p.grad is not None for p in m.parameters()) and float(b.grad.abs().sum()) == 0
    m(torch.ones(1, 3)).sum().backward()
    assert m.weight.grad.data_ptr() == b.grad.data_ptr() and float(b.grad.abs().sum()) > 0


# ---- the training loop's skip / stop decision must be collective (a rank that skipped a NaN batch alone left
# the others waiting in the all-reduce)
class _FakeOpt:
    def __init__(self, dp):
        self.dp = dp


def _nan_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from se3_equi_graph_registration_amd.train import _step_is_finite
    torch.manual_seed(7)
    model = Pair()
    dp = DataParallelStep(model, model.egnn)
    opt = _FakeOpt(dp)
    out = []
    for poison in (None, "grad", "loss"):
        dp.zero_grad()
        loss = model(torch.ones(2, 4))
        loss.backward()
        if rank == 1 and poison == "grad":
            model.egnn.flat_grads[0] = float("nan")
        if rank == 1 and poison == "loss":
            loss = loss * float("inf")
        ok = _step_is_finite(loss, opt, model)
        if ok:
            dp.sync()                      # would dead-lock if the ranks disagreed
        out.append(ok)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_nonfinite_step_decision_is_collective():
    res = dict(_run_ranks(_nan_worker, 2))
    assert res[0] == res[1] == [True, False, False]


# ---- the training loop under data parallelism: the None-batch skip is collective and every rank runs its
# dp.shard_range share of each step batch, weighted so that uneven shards still give the gradient of the global mean
class _SgdOpt:
    """Stands in for train.StepOptimizer: the dp bucket contract + a plain SGD update after the mean all-reduce."""

    def __init__(self, model, lr=0.1):
        self.model, self.lr = model, lr
        self.dp = DataParallelStep(model, model.egnn)
        self.dp.broadcast_parameters(0)
        self.dp.zero_grad()
        self.steps = 0

    def zero_grad(self):
        self.dp.zero_grad()

    def step(self):
        self.dp.sync()
        with torch.no_grad():
            self.dp.joint.add_(self.dp.joint_grad, alpha=-self.lr)
        self.steps += 1


def _mean_loss(model, batch):
    (x,) = batch
    return model.mlp(model.egnn(x)).mean()


def _loop_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from se3_equi_graph_registration_amd.train import run_training_batches
    torch.manual_seed(3)
    model = Pair()
    opt = _SgdOpt(model)
    g = torch.Generator().manual_seed(11)
    data = [torch.randn(b, 4, generator=g) for b in (5, 4, 1, 6)]           # 5 and 1: uneven / empty shards at world = 2
    batches = [(d,) for d in data]
    if rank == 1:
        batches[1] = None                                                   # an unreadable pair on ONE rank (train :969)
    run_training_batches(model, batches, opt, _mean_loss, "cpu", log_interval=0)
    q.put((rank, opt.steps, opt.dp.joint.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_training_loop_skips_none_batches_collectively_and_shards_by_rank():
    res = sorted(_run_ranks(_loop_worker, 2), key=lambda t: t[0])
    (_, s0, p0), (_, s1, p1) = res
    assert s0 == s1 == 3                           # batch 1 dropped by BOTH ranks, no rank left alone in a collective
    assert torch.equal(p0, p1)
    # single-process reference over the same global batches (minus the skipped one)
    torch.manual_seed(3)
    ref = Pair()
    g = torch.Generator().manual_seed(11)
    data = [torch.randn(b, 4, generator=g) for b in (5, 4, 1, 6)]
    params = [p for p in ref.parameters()]
    for i, d in enumerate(data):
        if i == 1:
            continue
        for p in params:
            p.grad = None
        _mean_loss(ref, (d,)).backward()
        with torch.no_grad():
            for p in params:
                if p.grad is not None:
                    p.add_(p.grad, alpha=-0.1)
    flat_ref = torch.cat([p.detach().reshape(-1) for p in ref.egnn.parameters()])
    torch.testing.assert_close(p0[:flat_ref.numel()], flat_ref, rtol=1e-5, atol=1e-6)
    head_ref = torch.cat([p.detach().reshape(-1) for p in list(ref.mlp.parameters()) + list(ref.unused.parameters())])
    off = (flat_ref.numel() + 63) // 64 * 64
    torch.testing.assert_close(p0[off:off + head_ref.numel()], head_ref, rtol=1e-5, atol=1e-6)
