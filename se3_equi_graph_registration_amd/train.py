"""The two callers either side of the hot path and the checkpoint format (reference ``src/3dmatch_train_egnn_with_batch.py``):

* ``train_one_epoch`` -- :964-1140: per batch kNN graphs (k = 16, self loops), model forward, ``corr + rot + trans`` loss
  (:1120), backward, optimizer step.
* ``validate``        -- :1143-1308: the same forward under ``no_grad``, running means.
* ``save_checkpoint`` / ``load_checkpoint`` -- :1311-1396: ``{'epoch', 'egnn_state_dict', 'cross_attention_state_dict',
  'optimizer_state_dict'}``; the optimizer entry is ``torch.optim.Adam.state_dict()``-shaped, so a run can resume from a
  checkpoint the reference wrote and the reference can resume from one written here.
* ``evaluate_model``  -- ``src/eval_egnn_metrics.py:1084-1296``: checkpoint -> evaluation head (batch size 1) over a test loader ->
  rotation / translation error, recall, F1 per pair on the device (``metrics.evaluate_batch``), averages, ``evaluation_results.txt``.
* ``make_optimizer``  -- flattens the model into the joint parameter/gradient bucket (``dp.DataParallelStep``) and returns
  the one-launch Adam over it (reference: ``torch.optim.Adam(model.parameters(), lr)``, :1412).

No logging back ends (tensorboard / wandb writers are accepted and ignored); ``StepOptimizer`` is a ``torch.optim.Optimizer``,
so the reference's ``StepLR`` (or any scheduler) drives it unchanged.
Everything between the loader and ``loss.item()`` runs in the HIP kernels; batches are staged to HBM one step ahead.
"""
import os
import queue
import sys
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, ops
from .datasets import DeviceBatches
from .dp import DataParallelStep, shard_range
from .graph import build_pair_graph
from .metrics import evaluate_batch
from .pose import registration_loss


class StepOptimizer(torch.optim.Optimizer):
    """``FlatAdam`` over the model's joint bucket behind the ``torch.optim.Optimizer`` interface the reference loop and its
    scheduler use (``torch.optim.Adam(model.parameters(), lr)`` :1412, ``StepLR(optimizer, 200, 0.5)`` :1413): ONE persistent
    ``param_groups`` entry whose ``lr`` / ``betas`` / ``eps`` are read at every ``step()``, so schedulers and manual
    ``for g in optimizer.param_groups: g['lr'] = ...`` edits take effect and end up in ``state_dict()``."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, process_group=None):
        self.model = model
        model.egnn._ensure_flat()
        self.dp = DataParallelStep(model, model.egnn, process_group=process_group)
        self.params = list(model.parameters())                  # torch.optim.Adam(model.parameters()) order
        self.adam = ops.FlatAdam([(model.egnn.flat_params, model.egnn.flat_grads), (self.dp.head.flat, self.dp.head.grad)],
                                 lr=lr, betas=betas, eps=eps)
        super().__init__(self.params, dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps), weight_decay=0,
                                           amsgrad=False, maximize=False))
        self.dp.zero_grad()

    def zero_grad(self, set_to_none: bool = False):
        pass        # step() clears every gradient as it consumes it (one launch); the buckets start zeroed

    def _push_hyperparameters(self):
        g = self.param_groups[0]
        if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
            raise ValueError("StepOptimizer: weight_decay / amsgrad / maximize are not supported (the reference does not use them)")
        self.adam.lr, self.adam.betas, self.adam.eps = float(g["lr"]), (float(g["betas"][0]), float(g["betas"][1])), float(g["eps"])

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("StepOptimizer.step takes no closure")
        self._push_hyperparameters()
        self.dp.sync()                                          # mean all-reduce of the joint bucket when world_size > 1
        self.adam.step(zero_grad=True)

    def state_dict(self):
        self._push_hyperparameters()
        sd = self.adam.state_dict(self.params)
        extra = {k: v for k, v in self.param_groups[0].items() if k not in sd["param_groups"][0] and k != "params"}
        sd["param_groups"][0].update(extra)                     # e.g. the scheduler's initial_lr
        return sd

    def load_state_dict(self, sd):
        self.adam.load_state_dict(sd, self.params)
        g, src = self.param_groups[0], sd["param_groups"][0]
        for key in ("lr", "betas", "eps", "initial_lr"):
            if key in src:
                g[key] = tuple(src[key]) if key == "betas" else src[key]
        self._push_hyperparameters()


def make_optimizer(model, lr=1e-4, **kw) -> StepOptimizer:
    return StepOptimizer(model, lr=lr, **kw)


def _device_batches(dataloader, device):
    first_is_cuda = isinstance(dataloader, DeviceBatches)
    return dataloader if first_is_cuda else DeviceBatches(dataloader, device=device)


def _pointnet_features(graph, xyz_0, xyz_1):
    """``use_pointnet=True`` (train :1031-1038): a FRESH, untrained ``PointNet(3, 32, 32)`` per batch encodes the coordinates
    into the 32 per-point features.  The reference's call passes three arguments to a four-argument forward and cannot
    run; this is what its reshape to ``[batch, 2048, 32]`` implies: ``encode`` (no global pooling) with the points as both
    positions and input features, on the same kNN edges.  Inference only (the encoder is not in the optimizer there either)."""
    from .pointnet import PointNet
    B, N, _ = xyz_0.shape
    enc = PointNet(in_num_feature=3, hidden_num_feature=32, output_num_feature=32, device=xyz_0.device).to(xyz_0.device)
    pos = torch.cat([xyz_0.reshape(B * N, 3), xyz_1.reshape(B * N, 3)]).contiguous()
    edge_index = torch.stack([graph.row[:graph.n_edges].long(), graph.col[:graph.n_edges].long()])
    with torch.no_grad():
        feats = enc.encode(pos, pos, edge_index)
    return feats[:B * N].view(B, N, -1), feats[B * N:].view(B, N, -1)


def _forward(model, batch, k, use_pointnet=False, loss_mode="3dmatch", graph=None):
    corr, labels, xyz_0, xyz_1, feat_0, feat_1, gt_pose = batch
    if graph is None:                                           # (GraphedStep hands over a graph it built one step ahead)
        graph = build_pair_graph(xyz_0, xyz_1, k, loop=True)    # knn_graph(xyz[i], k=16, loop=True) per cloud (:1005-1006)
    if use_pointnet:
        feat_0, feat_1 = _pointnet_features(graph, xyz_0, xyz_1)
    out = model(feat_0, xyz_0, graph, None, feat_1, xyz_1, graph, None, corr, labels, gt_pose)
    loss, rot, trans = registration_loss(out[0], out[1], out[2], gt_pose)     # corr.mean() + rot.mean() + trans.mean() (:1120)
    if loss_mode == "kitti":        # src/kitti_train_egnn_with_batch.py:1089: loss = corr_loss.mean() + equi_loss (pose losses only logged)
        loss = out[2].mean() + out[3].mean()
    elif loss_mode != "3dmatch":
        raise ValueError("loss_mode: '3dmatch' or 'kitti'")
    return out, loss, rot, trans


def _step_is_finite(loss, optimizer, model):
    """1-element device flag: loss and every local gradient finite -- MIN-reduced over the optimizer's process group, so all
    ranks of a data-parallel job take the same branch (a rank that skipped alone would leave the others in the all-reduce)."""
    grads = optimizer.dp.buckets() if hasattr(optimizer, "dp") else [p.grad for p in model.parameters() if p.grad is not None]
    ok = torch.isfinite(loss.detach()).all()
    for g in grads:
        ok = ok & torch.isfinite(g).all()
    flag = ok.to(torch.float32).reshape(1)
    group = optimizer.dp.group if hasattr(optimizer, "dp") else None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item() > 0.5)


def _process_group(optimizer):
    """(rank, world, group) of the data-parallel job this optimizer synchronises over; (0, 1, None) for a single process."""
    group = optimizer.dp.group if hasattr(optimizer, "dp") else None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        return dist.get_rank(group), dist.get_world_size(group), group
    return 0, 1, group


def _all_ranks_have(have: bool, group, device) -> bool:
    """MIN over the ranks of 'this rank holds a usable batch': a batch with a ``None`` member (an unreadable pair, train :969) is
    skipped by EVERY rank or by none -- a rank that skipped alone would leave its peers in the finite-check and gradient
    all-reduces of this step."""
    flag_dev = device if dist.get_backend(group) == "nccl" else "cpu"
    flag = torch.tensor([1.0 if have else 0.0], dtype=torch.float32, device=flag_dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item() > 0.5)


def shard_batch(batch, rank: int, world: int):
    """This rank's contiguous share (``dp.shard_range``) of a step batch whose tensors are stacked along dim 0 (SURVEY section 8(e):
    pairs are independent -- the reference asserts there are no cross-pair edges, train :1020-1028).  Returns
    ``(shard, n_local, n_global)``; ``n_local`` may be 0 when the batch holds fewer pairs than there are ranks."""
    n_global = int(batch[0].shape[0])
    lo, hi = shard_range(n_global, rank, world)
    return tuple(item[lo:hi] if isinstance(item, torch.Tensor) else item for item in batch), hi - lo, n_global


def _check_head_waits(owner):
    """A bounded in-launch wait of the fused head that ran out (csrc/head.hip) shows as a NaN loss: report it as what it is."""
    seen = ops.head_wait_failures()
    if seen > getattr(owner, "_head_wait_failures_seen", 0):
        owner._head_wait_failures_seen = seen
        raise RuntimeError(f"the fused pose head reported {seen} in-launch wait(s) that ran out (csrc/head.hip: MLP columns waiting for "
                           "their pair's selection); the affected steps carried NaN losses and were not applied -- this indicates a "
                           "device problem, not a non-finite loss of the model")


_EMPTY_SHARD = object()        # GraphedStep.run: "this rank's share of the step batch holds no pair"


class _Slot:
    """One set of static step inputs in HBM: a single fp32 packet ``[xyz (2,B,N,3) | feat (2,B,N,F) | labels (B,N) | gt_pose (B,16) |
    signal (4)]`` (one H2D copy per step) and the views the model reads.  ``signal[0]`` is 0, or NaN when this rank has nothing
    valid to contribute to the step (data parallel: it then reaches every rank through the gradient bucket)."""

    def __init__(self, B, N, F, device, pinned=False):
        sizes = [2 * B * N * 3, 2 * B * N * F, B * N, B * 16, 4]
        self.offsets = [0]
        for n in sizes:
            self.offsets.append(self.offsets[-1] + (n + 3) // 4 * 4)          # every segment 16-byte aligned
        self.packet = (torch.zeros(self.offsets[-1], dtype=torch.float32).pin_memory() if pinned
                       else torch.zeros(self.offsets[-1], dtype=torch.float32, device=device))
        seg = lambda i, shape: self.packet[self.offsets[i]:self.offsets[i] + sizes[i]].view(shape)      # noqa: E731
        self.xyz, self.feat = seg(0, (2, B, N, 3)), seg(1, (2, B, N, F))
        self.labels, self.gt_pose, self.signal = seg(2, (B, N)), seg(3, (B, 4, 4)), seg(4, (4,))
        self.last_call = 0                 # the run's guarded call that last read this slot (GraphedStep.run)

    def fill(self, batch):
        """Pack a 7-tuple ``(corr, labels, xyz_0, xyz_1, feat_0, feat_1, gt_pose)`` (``corr`` is not an input of the forward).
        Host -> pinned packets go through numpy: one plain memcpy per member on the calling (producer) thread -- ``Tensor.copy_``
        would fan a 0.5 MB copy out over an OpenMP team of all host cores, created afresh for a non-main thread (3.4 ms per
        batch on the 256-core GPU box, measured)."""
        _, labels, xyz_0, xyz_1, feat_0, feat_1, gt_pose = batch
        pairs = ((self.xyz[0], xyz_0), (self.xyz[1], xyz_1), (self.feat[0], feat_0), (self.feat[1], feat_1),
                 (self.labels, labels.reshape(self.labels.shape)), (self.gt_pose, gt_pose))
        if not self.packet.is_cuda and all(not src.is_cuda for _, src in pairs):
            for dst, src in pairs:
                np.copyto(dst.numpy(), src.detach().numpy(), casting="unsafe")
            self.signal.numpy()[:] = 0.0
            return
        for dst, src in pairs:
            dst.copy_(src)
        self.signal.zero_()


class GraphedStep:
    """The training step of ``train_one_epoch`` (train :964-1140) -- kNN graphs, inverse index, EGNN forward, pose head, losses,
    backward, [gradient all-reduce], Adam -- captured ONCE into a hipGraph per input slot and replayed per batch.

    * Static inputs: ``slots`` packets in HBM (``_Slot``).  A producer thread packs each host batch into a pinned packet; ONE
      H2D copy per step runs on a side stream while the previous step's graph is still executing (the slot it overwrites was
      released by an event two steps earlier).
    * No host round trip per step: the reference loop's "stop / skip on a non-finite loss or gradient" is taken on the device
      inside the optimizer launch (``FlatAdam.step_guarded``); the running loss, the number of applied / skipped steps and the
      first skipped step live in ``optimizer.adam.stats_dev`` and are read when the epoch ends (and, without blocking, at the log
      interval).  A step that would have been rejected leaves weights, moments and step count untouched, so raising late loses nothing.
    * The first batch of a new shape runs eagerly (a REAL step through the same kernels -- it also creates the cached index
      arrays the capture adopts); the graphs are captured before the second one.  Batches of another shape (a short last batch)
      run eagerly, batches with a ``None`` member are skipped (train :969).
    * Data parallel (one process per GPU): every rank runs its ``dp.shard_range`` share; the loss is weighted so that the mean
      all-reduce yields the gradient of the global mean; the all-reduce is captured INSIDE the step graph when RCCL allows it,
      otherwise the step is two graphs around an eager all-reduce; a rank's non-finite loss / missing batch reaches every rank
      through a spare word of the joint gradient bucket, so all ranks take the same decision without a second collective.
    * Everything is enqueued on this object's own stream (eager launches on the NULL stream between hipGraph replays fault on
      this ROCm stack); ``run`` makes the caller's stream wait for it before returning.
    Weights after n steps are bit-identical to the eager loop's (``tests/test_gpu_train.py``)."""

    def __init__(self, model, optimizer, k=16, loss_mode="3dmatch", slots=3, graph_collectives=True, use_graphs=True, depth=3,
                 force_dist=False, build_ahead=None):
        if not isinstance(optimizer, StepOptimizer):
            raise TypeError("GraphedStep needs the package's StepOptimizer (train.make_optimizer): the guarded flat Adam is part of the step")
        if loss_mode not in ("3dmatch", "kitti"):
            raise ValueError("loss_mode: '3dmatch' or 'kitti'")
        self.model, self.opt, self.k, self.loss_mode = model, optimizer, int(k), loss_mode
        self.n_slots, self.depth = max(1, int(slots)), max(int(depth), 2)
        self.graph_collectives, self.use_graphs = graph_collectives, use_graphs
        self.device = optimizer.dp.joint_grad.device if optimizer.dp.joint_grad is not None else next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("GraphedStep: the model must live on the MI355X (HIP) device; this package has no CPU path")
        self.stream = torch.cuda.Stream(self.device)
        self.copy_stream = torch.cuda.Stream(self.device)
        # build_ahead (round 6): the k-NN graph + inverse index of a batch depend on nothing but its coordinates, and are 30 us of the
        # 330 us one-pair step.  Each slot gets a SECOND captured graph holding only that build; it is replayed on `build_stream` as
        # soon as the slot's H2D copy has landed -- one step ahead, beside the previous step's kernels -- and the step graph proper
        # waits for an event recorded behind it (a wait on an already-signalled event of a compute stream costs the waiting stream
        # 1-2 us on this stack, tools/dev/ahead_probe.py; an event on the COPY stream is what was expensive, so the hand-over from
        # the copy to the build is a host-side synchronize, as before; the device offers no stream priority below the default --
        # priority_range() = (0, -1) -- and a high-priority step stream changed nothing: 3 040-3 098 pairs/s in every combination).
        # EQUIGSPR_BUILD_AHEAD=0 turns it off.
        if build_ahead is None:
            build_ahead = os.environ.get("EQUIGSPR_BUILD_AHEAD", "1") != "0"
        self.build_ahead = bool(build_ahead) and use_graphs
        self.build_stream = torch.cuda.Stream(self.device) if self.build_ahead else None
        self.build_graphs, self._built = None, []      # per slot: the captured build, the event recorded behind its last replay
        self.rank, self.world, self.group = _process_group(optimizer)
        # force_dist: run the multi-rank code path (spare-word signalling, all-reduce inside / between the graphs) on a one-rank
        # group -- how the tests exercise it on a single GPU
        self.dist_on = self.world > 1 or (bool(force_dist) and dist.is_available() and dist.is_initialized())
        self.shape = None                  # (B_local, N, F) of the captured step
        self._n_global = None              # ... and the global batch size its loss weight was computed for
        self._voted = False                # data parallel: a first batch has been accepted by all ranks
        self._capture_tried = False
        self._shape_step_done = False      # an eager step of the static shape has run (its cached index arrays exist: capture can adopt them)
        self._capture_votes = 0            # data parallel: steps at which the ranks voted on capturing (bounded: _capture_when_possible)
        self._last_idx = 0
        self.slots, self.graphs, self.graph_mode = [], None, "eager"
        self.steps_replayed = self.steps_eager = 0
        self._seed = self._pad = None
        self._call_batch = []              # guarded call -> batch index of the current epoch (for the error message)
        self.last_stats = None             # FlatAdam.guard_stats() of the last run()
        self._pinned_free, self._pinned_shapes = queue.Queue(), set()

    def reset(self):
        """Drop the captured graphs and the static slots; the next ``run`` starts over (eager first step, new capture).  Call it after
        changing anything a captured graph has baked in: which parameters require gradients, the model's structure or ``variant``, the
        process group.  (Weights, optimizer state and hyper-parameters are read at replay time: loading a checkpoint or stepping a
        scheduler needs no reset.)"""
        self.shape, self.slots, self.graphs, self.graph_mode = None, [], None, "eager"
        self.build_graphs, self._built = None, []
        self._capture_tried, self._last_idx, self._n_global, self._voted = False, 0, None, False
        self._shape_step_done, self._capture_votes = False, 0

    # ------------------------------------------------------------------ the step itself (eager or under capture)
    def _spare_words(self):
        """The unused words between the EGNN block and the head parameters of the joint gradient bucket (dp.DataParallelStep):
        part of the all-reduced message, not an optimizer bucket."""
        dp = self.opt.dp
        if dp.joint_grad is None:
            return None
        pe = int(self.model.egnn.flat_grads.numel())
        pad = dp.joint_grad[pe:dp.head_offset]
        if pad.numel() < 2:
            raise RuntimeError("GraphedStep: the joint gradient bucket has no spare words (dp.DataParallelStep leaves at least two)")
        return pad

    def _forward_backward(self, slot, seed, graph=None):
        batch = (None, slot.labels, slot.xyz[0], slot.xyz[1], slot.feat[0], slot.feat[1], slot.gt_pose)
        loss = _forward(self.model, batch, self.k, False, self.loss_mode, graph=graph)[1]
        loss.backward(gradient=seed)
        if self.dist_on:
            # two spare words of the all-reduced message carry this rank's verdict to every rank's guard: [0] = loss * 0 (NaN for
            # a non-finite local loss), [1] = the slot's signal (NaN: this rank had no batch for the step)
            torch.mul(loss.detach(), 0.0, out=self._pad[0])
            self._pad[1].copy_(slot.signal[0])
        return loss.detach()

    def _apply(self, loss):
        self.opt.adam.step_guarded(loss=loss, extra=self._pad[:2] if self.dist_on else None)

    def _body(self, slot, graph=None):
        loss = self._forward_backward(slot, self._seed, graph=graph)
        if self.dist_on:
            self.opt.dp.sync(force=True)
        self._apply(loss)
        return loss

    # ------------------------------------------------------------------ capture
    def _prepare(self, B, N, F, n_global):
        self.shape, self._n_global = (B, N, F), n_global
        self.slots = [_Slot(B, N, F, self.device) for _ in range(self.n_slots)]
        # the slots' zero-fill runs on this object's stream, their first H2D on the copy stream: order the two once, here
        torch.cuda.current_stream(self.device).synchronize()
        w = B * self.world / n_global if self.world > 1 else 1.0
        self._seed = torch.full((), float(w), dtype=torch.float32, device=self.device)
        self._loss_weight = B / n_global if self.world > 1 else 1.0
        self._pad = self._spare_words() if self.dist_on else None
        self.graphs, self.graph_mode, self._capture_tried, self._last_idx = None, "eager", False, 0
        self._shape_step_done = False

    def _capture(self):
        """Capture one copy of the step per slot (they share one memory pool: replays are sequential on one stream)."""
        if not self.use_graphs:
            return
        n_words = 2 * self.shape[0] * self.shape[1] + 1
        # only RCCL collectives can be captured (gloo moves device buffers through the host: a synchronous copy aborts the capture, and on
        # this stack an aborted capture leaves PyTorch's capture stream unusable for the rest of the process -- so a mode that cannot
        # work is not attempted; torch.cuda.Stream() objects come from a pool of 32 per device, a "fresh" one is no way around that)
        can_capture_collective = self.graph_collectives and self.dist_on and dist.get_backend(self.group) == "nccl"
        if can_capture_collective:
            # a captured collective that hangs at replay cannot be recovered from in-process: throw-away child processes try it first
            from .dp import captured_collectives_work
            nb = int(self.opt.dp.joint_grad.numel()) if self.opt.dp.joint_grad is not None else 65536
            can_capture_collective = captured_collectives_work(self.group, self.device, numel=nb)
        modes = ["one graph"] if not self.dist_on else \
            (["one graph (all-reduce captured)"] if can_capture_collective else []) + ["two graphs + eager all-reduce"]
        for mode in modes:
            graphs, pool, ok = [], None, True
            builds, bpool = [], None
            try:
                for sl in self.slots:
                    ops.reserve_csr_scratch(self.device, n_words)
                    torch.cuda.synchronize(self.device)
                    pair_graph = None
                    if self.build_ahead:
                        # the slot's graph build as a capture of its own, in a memory pool of its own: its replays run on another
                        # stream BESIDE the step graphs' replays, so the two sets must not share recycled blocks (the build graphs
                        # replay one after the other on one stream: they share theirs)
                        gbuild = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gbuild, pool=bpool, stream=self.build_stream, capture_error_mode="thread_local"):
                            pair_graph = build_pair_graph(sl.xyz[0], sl.xyz[1], self.k, loop=True)
                        builds.append((gbuild, pair_graph))
                        bpool = gbuild.pool()
                        with torch.cuda.stream(self.build_stream):
                            gbuild.replay()                        # the captured launches have not run: the step capture below reads nothing,
                        self.build_stream.synchronize()            # but the first replay of the step must find a built graph
                    if mode.startswith("one"):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                            self._body(sl, graph=pair_graph)
                        graphs.append((g,))
                    else:
                        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                        with torch.cuda.graph(ga, pool=pool, capture_error_mode="thread_local"):
                            loss = self._forward_backward(sl, self._seed, graph=pair_graph)
                        with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
                            self._apply(loss)
                        graphs.append((ga, gb))
                    pool = graphs[-1][0].pool()
            except Exception as exc:                   # capture unsupported for this mode: try the next one
                ok = False
                print(f"[GraphedStep] hipGraph capture failed in mode '{mode}' ({type(exc).__name__}: {exc})", file=sys.stderr)
                torch.cuda.synchronize(self.device)
                _lib.lib().equigspr_clear_device_error()      # the aborted capture's error must not surface in the next launch
                ops._CSR_SCRATCH.clear()                      # a scratch array the aborted capture adopted is in an unknown state
            if self.dist_on:                            # all ranks replay the same way or none does
                ok = _all_ranks_have(ok, self.group, self.device)
            if ok:
                self.graphs, self.graph_mode = graphs, mode
                self.build_graphs = builds if self.build_ahead else None
                self._built = [None] * len(self.slots)
                return
        self.graphs, self.graph_mode, self.build_graphs = None, "eager", None

    def _capture_when_possible(self):
        """Called by EVERY rank after EVERY step of a run until the graphs exist (or capture has been tried and given up).
        One process: capture right after the first eager step of the static shape.  Data parallel: `_capture` contains collectives (the
        per-mode vote, the captured-collective probe), so the decision to enter it is itself collective -- a MIN over 'this rank has run
        an eager step of its static shape': a rank whose shard of the first batch was empty (a global batch smaller than the world;
        ADVICE round 5) has nothing to capture yet, and its peers wait with theirs until it has; they all capture behind the same
        later step.  (Before: the ranks with data entered `_capture` alone and their vote paired with the next step's all-reduce.)"""
        if not self.use_graphs or self.graphs is not None or self._capture_tried:
            return
        ready = self.shape is not None and self._shape_step_done
        if self.dist_on:
            ready = _all_ranks_have(ready, self.group, self.device)
            self._capture_votes += 1
            if not ready and self._capture_votes >= 8:      # some rank never gets a batch of a static shape: stay eager, stop asking
                self._capture_tried = True                  # (every rank counts the same votes: the same decision everywhere)
        if not ready:
            return
        self._capture_tried = True
        self.stream.synchronize()
        self._capture()

    # ------------------------------------------------------------------ one step
    def _launch(self, idx):
        sl = self.slots[idx]
        if self.graphs is None:
            self._body(sl)
            self.steps_eager += 1
            self._shape_step_done = True
        else:
            g = self.graphs[idx]
            if self.build_graphs is not None:
                if self._built[idx] is None:           # no build is in flight for this slot (the first replay after the capture): build now
                    self._build(idx)
                # the build went out a step ago and has normally finished: the HOST sees that (hipEventQuery) and launches the step with
                # no device-side wait -- a wait on the event, even a signalled one, widens the idle gap between two step graphs from
                # 8.9 to 14.0 us (rocprofv3 kernel trace, tools/dev/trace_gaps2.py); only a build still in flight makes the stream wait
                if not self._built[idx].query():
                    self.stream.wait_event(self._built[idx])
            g[0].replay()
            if len(g) == 2:
                self.opt.dp.sync(force=True)
                g[1].replay()
            self.steps_replayed += 1

    def _build(self, idx):
        """Replay slot ``idx``'s graph build on the build stream (its input coordinates are in HBM: the caller has synchronized the
        copy stream) and record the event the slot's step waits for."""
        with torch.cuda.stream(self.build_stream):
            self.build_graphs[idx][0].replay()
            ev = self._built[idx]
            if ev is None:
                ev = self._built[idx] = torch.cuda.Event()
            ev.record(self.build_stream)

    def _usable_shape(self, batch):
        _, labels, xyz_0, xyz_1, feat_0, feat_1, gt_pose = batch
        return (xyz_0.dim() == 3 and xyz_0.shape == xyz_1.shape and feat_0.shape == feat_1.shape and feat_0.shape[:2] == xyz_0.shape[:2]
                and labels.numel() == xyz_0.shape[0] * xyz_0.shape[1] and tuple(gt_pose.shape) == (xyz_0.shape[0], 4, 4))

    def run(self, batches, epoch=0, log_interval=10, skip_nonfinite=False, shard_batches=True):
        """The body of ``train_one_epoch`` over an iterable of step batches (host or device 7-tuples, ``None`` allowed).
        Returns the average loss over the applied steps (the global mean under data parallel).

        Pipeline (measured on the MI355X, one 2048-point pair, tools/dev/step_timing.py): a replay is 0.363 ms; an event
        recorded on the compute stream after every replay costs +5 us, a copy-stream H2D that WAITS (on the GPU) for such an
        event +30..80 us, an independent H2D +3 us.  So nothing here makes one stream wait for another on the device: the host
        learns that a step has finished from a pinned word the optimizer launch writes (``FlatAdam.progress``) and only then
        overwrites that step's input slot; the H2D of batch t+1 is issued before the replay of batch t and has long completed
        (checked on the host) when its own replay is launched."""
        dev = self.device
        _t_entry = time.perf_counter()
        caller = torch.cuda.current_stream(dev)
        self.stream.wait_stream(caller)
        adam = self.opt.adam
        adam._guard_buffers()
        with torch.cuda.stream(self.stream):
            adam.stats_dev.zero_()
        self.stream.synchronize()
        adam.progress[0] = 0.0
        for sl in self.slots:
            sl.last_call = 0
        self._calls = 0                    # guarded optimizer calls issued in this run (= FlatAdam.progress when all are done)
        self._call_batch = []
        pinned_free, pinned_shapes = self._pinned_free, self._pinned_shapes      # pinned packets outlive the run (hipHostMalloc is slow)
        q = queue.Queue(maxsize=self.depth)
        stop, abandoned = object(), threading.Event()

        def put(item):
            while not abandoned.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def take_pinned(shape):
            """A free pinned packet of this shape (the H2D copy that last read it has completed)."""
            while not abandoned.is_set():
                if shape not in pinned_shapes:
                    pinned_shapes.add(shape)
                    for _ in range(self.depth + 3):
                        pinned_free.put((_Slot(*shape, dev, pinned=True), None))
                try:
                    pk, ev = pinned_free.get(timeout=0.1)
                except queue.Empty:
                    continue
                other = tuple(pk.xyz.shape[1:3]) + (pk.feat.shape[3],)
                if other != shape:
                    # a packet of another shape: dropped -- and forgotten, so that the shape gets fresh packets should it come back
                    # (before: a one-pair first batch, then full batches, then a one-pair tail waited for ever on packets that the
                    # full batches' requests had thrown away; found by the two-rank test with a first batch smaller than the world)
                    pinned_shapes.discard(other)
                    continue
                if ev is not None:
                    ev.synchronize()
                return pk
            return None

        def produce():
            try:
                for batch_idx, batch in enumerate(batches):
                    if batch is None or any(item is None for item in batch[1:]):
                        if not put((batch_idx, None, None)):
                            return
                        continue
                    batch = tuple(batch)
                    if self.world > 1 and shard_batches:
                        batch, _, n_global = shard_batch(batch, self.rank, self.world)
                    else:                                      # the loader yields per-rank batches: equal shares assumed
                        n_global = int(batch[2].shape[0]) * self.world
                    shape = (int(batch[2].shape[0]), int(batch[2].shape[1]), int(batch[4].shape[-1]))
                    if shape[0] == 0:
                        # fewer pairs than ranks in a short last batch: this rank's shard is empty.  Alone (no peers) the batch is
                        # skipped like a None batch; under data parallel the rank contributes zero gradients and still joins the
                        # step's collective (_eager_empty) -- its peers are already on their way into the all-reduce
                        if not put((batch_idx, _EMPTY_SHARD if self.dist_on else None, n_global)):
                            return
                        continue
                    if (not batch[2].is_cuda) and self._usable_shape(batch) and (self.shape is None or shape == self.shape):
                        pk = take_pinned(shape)
                        if pk is None:
                            return
                        pk.fill(batch)                         # host memcpy into pinned memory, off the launching thread
                        item = (batch_idx, pk, n_global)
                    else:
                        item = (batch_idx, batch, n_global)
                    if not put(item):
                        return
            except BaseException as e:      # surfaced in the consumer
                put(e)
            put(stop)

        # two Python threads share the interpreter lock: with the default 5 ms switch interval a hand-over at the wrong moment
        # stalls the launching thread for several steps' worth of GPU time (measured: 2 of 400 steps took +4.5 ms)
        switch_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(switch_interval, 1e-4))
        th = threading.Thread(target=produce, daemon=True)
        th.start()
        pending_log = []
        nan_host = torch.full((1,), float("nan"))
        staged = None                      # (slot index, pinned packet, batch index): H2D issued, step not launched yet

        tm = self.timing = {"get": 0.0, "wait_calls": 0.0, "ev_sync": 0.0, "launch": 0.0, "h2d": 0.0, "slow": []} if os.environ.get("EQUIGSPR_STEP_TIMING") == "1" else None
        if tm is not None:
            tm["phase_setup"] = time.perf_counter() - _t_entry       # entry -> producer thread running

        def note(what, dt):
            tm[what] += dt
            if dt > 5e-4:
                tm["slow"].append((what, self._calls, round(dt * 1e3, 2)))

        def wait_calls(n):
            """Block (host) until the first ``n`` guarded calls of this run have completed on the GPU."""
            prog = adam.progress
            spins = 0
            t0 = time.perf_counter()
            check_at = t0 + 2.0
            while prog[0] < n:
                spins += 1
                if spins > 2000:
                    time.sleep(2e-5)
                    if time.perf_counter() > check_at:
                        # a step that never reaches its optimizer launch (an aborted graph node, an asynchronous device error) would
                        # leave this loop spinning for ever: query() raises a pending device error; a drained stream with the count
                        # still short means the launch is lost
                        if self.stream.query() and prog[0] < n:
                            time.sleep(1e-3)               # the pinned word is written by the kernel's last workgroup: let it land
                            if prog[0] < n:
                                raise RuntimeError(f"GraphedStep: guarded optimizer call {n} of this run never reported completion "
                                                   f"(progress {int(prog[0])}, stream idle): a step's launches were lost")
                        check_at = time.perf_counter() + 2.0
            if tm is not None:
                note("wait_calls", time.perf_counter() - t0)

        def launch_staged():
            nonlocal staged
            if staged is None:
                return
            idx, packet, batch_idx = staged
            staged = None
            t0 = time.perf_counter()
            # the H2D went out one iteration ago: normally complete.  A host-side stream synchronize, not an event: an event record
            # is a marker with a system-scope release on the GPU, which measurably slows the step running beside it (+35 us at
            # one pair), and a device-side wait on such an event costs more still
            self.copy_stream.synchronize()
            pinned_free.put((packet, None))
            t1 = time.perf_counter()
            self._launch(idx)
            if tm is not None:
                note("ev_sync", t1 - t0); note("launch", time.perf_counter() - t1)
            self._calls += 1
            self.slots[idx].last_call = self._calls
            self._last_idx = idx
            self._call_batch.append(batch_idx)
            self._capture_when_possible()  # the eager step above created the cached index arrays the capture adopts
            log_point(batch_idx)

        def log_point(batch_idx):
            if log_interval and batch_idx % log_interval == 0 and self.rank == 0:
                host = torch.empty(8, dtype=torch.float32).pin_memory()
                host.copy_(adam.stats_dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
                pending_log.append((batch_idx, host, ev))
            while pending_log and pending_log[0][2].query():
                self._report(pending_log.pop(0), epoch, skip_nonfinite)

        try:
            with torch.cuda.stream(self.stream):
                while True:
                    t0 = time.perf_counter()
                    item = q.get()
                    if tm is not None:
                        note("get", time.perf_counter() - t0)
                    if isinstance(item, BaseException):
                        raise item
                    if item is stop:
                        break
                    batch_idx, payload, n_global = item
                    have = payload is not None
                    if self.dist_on and not self._voted:
                        # until a first batch has been accepted nothing is captured: the ranks agree by a collective whether the
                        # batch is usable (a rank whose shard of it is empty HOLDS it)
                        have = _all_ranks_have(have, self.group, dev)
                        self._voted = have
                        if not have:
                            if isinstance(payload, _Slot):
                                pinned_free.put((payload, None))
                            continue
                    if payload is _EMPTY_SHARD or (not have and self.dist_on and self.shape is None):
                        # no pair for this rank (void = no readable batch at all, and no captured step to replay with the void
                        # signal): zero gradients, the step's ONE collective, the same decision as its peers
                        launch_staged()
                        self.opt._push_hyperparameters()
                        self._eager_empty(void=payload is not _EMPTY_SHARD)
                        self._calls += 1
                        self._call_batch.append(batch_idx)
                        self._capture_when_possible()
                        log_point(batch_idx)
                        continue
                    if not have:
                        if self.dist_on:
                            # this rank has no batch, its peers may: run the step on the data of the previous one with the "void"
                            # signal raised -- every rank's guard then rejects the step (ONE collective per step, no host sync)
                            launch_staged()
                            sl = self.slots[self._last_idx]
                            sl.signal[:1].copy_(nan_host, non_blocking=False)
                            self._launch(self._last_idx)
                            sl.signal.zero_()
                            self._calls += 1
                            sl.last_call = self._calls
                            self._call_batch.append(batch_idx)
                            self._capture_when_possible()
                        continue
                    self.opt._push_hyperparameters()
                    packed = isinstance(payload, _Slot)
                    shape = (tuple(payload.xyz.shape[1:3]) + (payload.feat.shape[3],)) if packed else None
                    # the captured graphs have the loss weight of THEIR global batch size baked in (``_seed``): a batch whose local
                    # shape matches but whose global size differs (an uneven last batch) runs eagerly with its own weight
                    if packed and (self.shape is None or (self.shape == shape and n_global == self._n_global)):
                        if self.shape is None:
                            self._prepare(*shape, n_global)
                        adam.push_hyper(self._loss_weight)
                        idx = ((staged[0] if staged is not None else self._last_idx) + 1) % self.n_slots
                        sl = self.slots[idx]
                        if staged is not None and staged[0] == idx:
                            launch_staged()            # a single slot: no look-ahead
                        wait_calls(sl.last_call)       # the step that last read this slot has finished (host-side knowledge)
                        t0 = time.perf_counter()
                        launch_staged()                # the previous batch: its H2D went out one iteration ago
                        with torch.cuda.stream(self.copy_stream):
                            sl.packet.copy_(payload.packet, non_blocking=True)
                        if self.build_graphs is not None:
                            # the slot's k-NN graph + index one step ahead, beside the step launched just above (host-side hand-over
                            # from the copy stream: ~10 us of host time that the replay in flight hides)
                            self.copy_stream.synchronize()
                            self._build(idx)
                        if tm is not None:
                            note("h2d", time.perf_counter() - t0)
                        staged = (idx, payload, batch_idx)
                        if self.graphs is None:        # no look-ahead before the graphs exist (first step, eager mode)
                            launch_staged()
                    else:
                        # device tensors, or another shape than the captured one (one object, one static shape -- the first it
                        # sees; a packet filled before that shape was fixed is taken as a batch of another shape)
                        launch_staged()
                        if packed:
                            payload_batch = (None, payload.labels, payload.xyz[0], payload.xyz[1], payload.feat[0], payload.feat[1], payload.gt_pose)
                            self._eager_other_shape(payload_batch, n_global)
                            self.stream.synchronize()
                            pinned_free.put((payload, None))
                        else:
                            self._eager_other_shape(payload, n_global)
                        self._calls += 1
                        self._call_batch.append(batch_idx)
                        self._capture_when_possible()
                        log_point(batch_idx)
                launch_staged()
                if tm is not None:
                    tm["phase_loop"] = time.perf_counter() - _t_entry    # ... -> the last step has been launched
        finally:
            abandoned.set()
            th.join()
            sys.setswitchinterval(switch_interval)
            caller.wait_stream(self.stream)
        self.stream.synchronize()
        if tm is not None:
            tm["phase_drain"] = time.perf_counter() - _t_entry           # ... -> the stream has drained
        for entry in pending_log:
            self._report(entry, epoch, skip_nonfinite)
        st = self.last_stats = adam.guard_stats()
        if tm is not None:
            tm["phase_stats"] = time.perf_counter() - _t_entry           # ... -> the statistics have been read back
        if st["applied"] > 0:
            self.opt._opt_called = True        # what torch's lr_scheduler looks at to tell "scheduler.step() before optimizer.step()"
        _check_head_waits(self)
        self._check(st, epoch, skip_nonfinite)
        loss_sum, applied = st["loss_sum"], st["applied"]
        if self.world > 1:
            acc = torch.tensor([loss_sum], dtype=torch.float64, device=dev if dist.get_backend(self.group) == "nccl" else "cpu")
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.group)
            loss_sum = float(acc[0])
        return loss_sum / max(applied, 1)

    def _eager_other_shape(self, batch, n_global):
        """A batch the captured step cannot take (device tensors, another shape): the same step, launched eagerly on a
        throw-away slot."""
        B, N, F = int(batch[2].shape[0]), int(batch[2].shape[1]), int(batch[4].shape[-1])
        sl = _Slot(B, N, F, self.device)
        sl.fill(batch)
        w = B * self.world / n_global if self.world > 1 else 1.0
        seed = torch.full((), float(w), dtype=torch.float32, device=self.device)
        self.opt.adam.push_hyper(B / n_global if self.world > 1 else 1.0)
        if self.dist_on and self._pad is None:
            self._pad = self._spare_words()
        loss = self._forward_backward(sl, seed)
        if self.dist_on:
            self.opt.dp.sync(force=True)
        self._apply(loss)
        self.steps_eager += 1

    def _eager_empty(self, void=False):
        """This rank's shard of the step batch is empty (a short last batch with fewer pairs than ranks; ``DataLoader``'s default
        ``drop_last=False``): it contributes zero gradients -- the buckets are clear, every step clears them as it consumes them --
        and loss 0 with weight 0, joins the step's ONE collective and applies the same averaged update as its peers (the contract
        of ``run_training_batches``).  Its peers run this batch eagerly or by replay; either way one all-reduce per rank."""
        self.opt.adam.push_hyper(0.0)
        if self._pad is None:
            self._pad = self._spare_words()
        self._pad[:2].zero_()
        if void:                           # "this rank could not read the batch": every rank's guard rejects the step
            self._pad[1].fill_(float("nan"))
        loss = torch.zeros((), dtype=torch.float32, device=self.device)
        self.opt.dp.sync(force=True)
        self._apply(loss)
        self.steps_eager += 1

    def _check(self, st, epoch, skip_nonfinite):
        if st["skipped"] > 0 and not skip_nonfinite:
            k = st["first_skipped"]
            where = self._call_batch[k] + 1 if k is not None and k < len(self._call_batch) else "?"
            raise RuntimeError(f"non-finite loss or gradient at epoch {epoch + 1}, batch {where} (the reference stops here under "
                               "set_detect_anomaly; the step was not applied; pass skip_nonfinite=True to drop such steps)")

    def _report(self, entry, epoch, skip_nonfinite):
        batch_idx, host, _ = entry
        v = host.tolist()
        self._check({"skipped": int(v[2]), "first_skipped": int(v[3]) - 1 if v[3] > 0 else None}, epoch, skip_nonfinite)
        print(f"Epoch [{epoch + 1}], Batch [{batch_idx + 1}], Loss: {v[4]:.6f}")


class GraphedEval:
    """The per-pair body of ``evaluate_model`` (eval :1150-1296) -- kNN graphs, inverse index, EGNN forward, the evaluation pair
    model's head, the four metrics of ``tools/evaluation_metrics.py`` -- captured once per input slot and replayed per pair; the same
    input pipeline as ``GraphedStep`` (a producer thread packs each host batch into a pinned packet, one H2D copy per pair a step
    ahead on a side stream, no device-side wait between streams).  A slot is reused once the event recorded behind its last replay
    has completed (host-side wait).  ``run`` returns the ``[n_pairs, 5]`` float64 table (rotation error [deg], translation error
    [cm], recall, precision, F1) on the host; batches of another shape than the first run eagerly, ``None`` batches are skipped."""

    def __init__(self, model, k=16, tau=0.09, slots=3, use_graphs=True, depth=3, build_ahead=None):
        self.model, self.k, self.tau = model, int(k), float(tau)
        self.n_slots, self.depth, self.use_graphs = max(1, int(slots)), max(int(depth), 2), use_graphs
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("GraphedEval: the model must live on the MI355X (HIP) device; this package has no CPU path")
        self.stream, self.copy_stream = torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)
        self.shape, self.slots, self.graphs, self.outs = None, [], None, []
        self.steps_replayed = self.steps_eager = 0
        self._capture_tried = False
        self._pinned_free, self._pinned_shapes = queue.Queue(), set()
        # the slot's k-NN graph + index one pair ahead on a second stream, as in GraphedStep (17 % of the one-pair inference step)
        if build_ahead is None:
            build_ahead = os.environ.get("EQUIGSPR_BUILD_AHEAD", "1") != "0"
        self.build_ahead = bool(build_ahead) and use_graphs
        self.build_stream = torch.cuda.Stream(self.device) if self.build_ahead else None
        self.build_graphs, self._built = None, []

    def _build(self, idx):
        with torch.cuda.stream(self.build_stream):
            self.build_graphs[idx][0].replay()
            ev = self._built[idx]
            if ev is None:
                ev = self._built[idx] = torch.cuda.Event()
            ev.record(self.build_stream)

    def _body(self, sl, graph=None):
        if graph is None:
            graph = build_pair_graph(sl.xyz[0], sl.xyz[1], self.k, loop=True)
        out = self.model(sl.feat[0], sl.xyz[0], graph, None, sl.feat[1], sl.xyz[1], graph, None, None, sl.labels, sl.gt_pose)
        m = evaluate_batch(out[0], out[1], sl.gt_pose, sl.xyz[0], sl.xyz[1], tau=self.tau)
        return torch.stack([m["rotation_error"], m["translation_error"], m["recall"], m["precision"], m["f1"]], dim=1)

    def _capture(self):
        if not self.use_graphs:
            return
        graphs, outs, pool = [], [], None
        builds, bpool = [], None
        try:
            n_words = 2 * self.shape[0] * self.shape[1] + 1
            for sl in self.slots:
                ops.reserve_csr_scratch(self.device, n_words)
                torch.cuda.synchronize(self.device)
                pair_graph = None
                if self.build_ahead:                   # the build as a capture (and memory pool) of its own: see GraphedStep._capture
                    gbuild = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gbuild, pool=bpool, stream=self.build_stream, capture_error_mode="thread_local"):
                        pair_graph = build_pair_graph(sl.xyz[0], sl.xyz[1], self.k, loop=True)
                    builds.append((gbuild, pair_graph))
                    bpool = gbuild.pool()
                    with torch.cuda.stream(self.build_stream):
                        gbuild.replay()
                    self.build_stream.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                    o = self._body(sl, graph=pair_graph)
                graphs.append(g); outs.append(o)
                pool = g.pool()
            self.graphs, self.outs = graphs, outs
            self.build_graphs, self._built = (builds if self.build_ahead else None), [None] * len(self.slots)
        except Exception as exc:
            print(f"[GraphedEval] hipGraph capture failed ({type(exc).__name__}: {exc}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize(self.device)
            _lib.lib().equigspr_clear_device_error()
            ops._CSR_SCRATCH.clear()
            self.graphs, self.build_graphs = None, None

    @torch.no_grad()
    def run(self, batches):
        dev = self.device
        caller = torch.cuda.current_stream(dev)
        self.stream.wait_stream(caller)
        was_training = self.model.training
        self.model.eval()
        q = queue.Queue(maxsize=self.depth)
        stop, abandoned = object(), threading.Event()
        pinned_free, pinned_shapes = self._pinned_free, self._pinned_shapes

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
                for batch in batches:
                    if batch is None or any(item is None for item in batch[1:]):
                        continue
                    batch = tuple(batch)
                    shape = (int(batch[2].shape[0]), int(batch[2].shape[1]), int(batch[4].shape[-1]))
                    if (not batch[2].is_cuda) and (self.shape is None or shape == self.shape):
                        pk = None
                        while pk is None and not abandoned.is_set():
                            if shape not in pinned_shapes:
                                pinned_shapes.add(shape)
                                for _ in range(self.depth + 3):
                                    pinned_free.put(_Slot(*shape, dev, pinned=True))
                            try:
                                pk = pinned_free.get(timeout=0.1)
                            except queue.Empty:
                                continue
                            other = tuple(pk.xyz.shape[1:3]) + (pk.feat.shape[3],)
                            if other != shape:                 # dropped and forgotten (see GraphedStep.run.take_pinned)
                                pinned_shapes.discard(other)
                                pk = None
                        if pk is None:
                            return
                        pk.fill(batch)
                        item = pk
                    else:
                        item = batch
                    if not put(item):
                        return
            except BaseException as e:
                put(e)
            put(stop)

        switch_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(switch_interval, 1e-4))
        th = threading.Thread(target=produce, daemon=True)
        th.start()
        rows, staged, last_idx = [], None, -1

        def launch_staged():
            nonlocal staged, last_idx
            if staged is None:
                return
            idx, packet = staged
            staged = None
            self.copy_stream.synchronize()             # the H2D went out one iteration ago (host-side wait, no device-side one)
            pinned_free.put(packet)
            sl = self.slots[idx]
            if self.graphs is None:
                out = self._body(sl)
                self.steps_eager += 1
            else:
                if self.build_graphs is not None:
                    if self._built[idx] is None:
                        self._build(idx)
                    if not self._built[idx].query():       # normally finished a pair ago: no device-side wait (see GraphedStep._launch)
                        self.stream.wait_event(self._built[idx])
                self.graphs[idx].replay()
                out = self.outs[idx]
                self.steps_replayed += 1
            rows.append(out.clone())                   # the slot's output buffer is overwritten by its next replay
            sl.done.record(self.stream)
            last_idx = idx
            if self.graphs is None and self.use_graphs and not self._capture_tried:
                self._capture_tried = True
                self.stream.synchronize()
                self._capture()

        try:
            with torch.cuda.stream(self.stream):
                while True:
                    item = q.get()
                    if isinstance(item, BaseException):
                        raise item
                    if item is stop:
                        break
                    if isinstance(item, _Slot) and (self.shape is None or self.shape == tuple(item.xyz.shape[1:3]) + (item.feat.shape[3],)):
                        if self.shape is None:
                            self.shape = tuple(item.xyz.shape[1:3]) + (item.feat.shape[3],)
                            self.slots = [_Slot(*self.shape, dev) for _ in range(self.n_slots)]
                            for sl in self.slots:
                                sl.done = torch.cuda.Event()
                            self.stream.synchronize()          # zero-fill (this stream) before the first H2D (copy stream)
                        idx = ((staged[0] if staged is not None else last_idx) + 1) % self.n_slots
                        if staged is not None and staged[0] == idx:
                            launch_staged()
                        sl = self.slots[idx]
                        sl.done.synchronize()          # its last replay has finished reading it
                        launch_staged()
                        with torch.cuda.stream(self.copy_stream):
                            sl.packet.copy_(item.packet, non_blocking=True)
                        if self.build_graphs is not None:
                            self.copy_stream.synchronize()
                            self._build(idx)
                        staged = (idx, item)
                        if self.graphs is None:
                            launch_staged()
                    else:                                  # device tensors / another shape: the same body, eagerly
                        launch_staged()
                        if isinstance(item, _Slot):
                            batch = (None, item.labels, item.xyz[0], item.xyz[1], item.feat[0], item.feat[1], item.gt_pose)
                        else:
                            batch = item
                        B, N, F = int(batch[2].shape[0]), int(batch[2].shape[1]), int(batch[4].shape[-1])
                        sl = _Slot(B, N, F, dev)
                        sl.fill(batch)
                        rows.append(self._body(sl))
                        self.steps_eager += 1
                        if isinstance(item, _Slot):
                            self.stream.synchronize()
                            pinned_free.put(item)
                launch_staged()
        finally:
            abandoned.set()
            th.join()
            sys.setswitchinterval(switch_interval)
            caller.wait_stream(self.stream)
            self.model.train(was_training)
        self.stream.synchronize()
        return torch.cat(rows).cpu() if rows else torch.zeros(0, 5, dtype=torch.float64)


def run_training_batches(model, batches, optimizer, step_loss, device, epoch=0, log_interval=10, skip_nonfinite=False,
                         shard_batches=True):
    """The body of ``train_one_epoch`` over an iterable of step batches; ``step_loss(model, batch) -> scalar loss``.

    Data-parallel contract (one process per GPU, every rank iterates the SAME sequence of global step batches):
      * a batch is used only if every rank holds it (``_all_ranks_have``);
      * with ``shard_batches`` every rank runs the pairs ``dp.shard_range(B, rank, world)`` of it; its loss -- a mean over its
        pairs -- is weighted by ``n_local * world / B`` before ``backward`` so that the mean all-reduce of ``optimizer.step``
        yields the gradient of the mean over all B pairs also when the shards are uneven; a rank left without a pair
        contributes zero gradients and still joins both collectives.  (The training head's similarity-MSE term standardises over
        the pairs it sees, train :777-781: like batch statistics under DDP it is evaluated per shard.)
      * the non-finite decision is collective (``_step_is_finite``)."""
    rank, world, group = _process_group(optimizer)
    total, n = 0.0, 0
    for batch_idx, batch in enumerate(batches):
        have = batch is not None and not any(item is None for item in batch)
        if world > 1:
            have = _all_ranks_have(have, group, device)
        if not have:
            continue
        n_local = n_global = 1
        if world > 1 and shard_batches:
            batch, n_local, n_global = shard_batch(batch, rank, world)
        optimizer.zero_grad()
        if n_local > 0:
            loss = step_loss(model, batch)
            weight = n_local * world / n_global if (world > 1 and shard_batches) else 1.0
            (loss if weight == 1.0 else loss * weight).backward()
        else:
            loss = torch.zeros((), device=device)           # no pair on this rank: zero gradients, same collectives
        if not _step_is_finite(loss, optimizer, model):
            if not skip_nonfinite:
                raise RuntimeError(f"non-finite loss or gradient at epoch {epoch + 1}, batch {batch_idx + 1} (the reference stops "
                                   "here under set_detect_anomaly; pass skip_nonfinite=True to drop such steps)")
            if hasattr(optimizer, "dp"):
                optimizer.dp.zero_grad()
            else:
                optimizer.zero_grad()
            continue
        value = loss.item()
        optimizer.step()
        # the local loss is a mean over this rank's pairs: weighted by its share, the sum over ranks (taken once, below) is the
        # mean over the whole step batch
        total, n = total + value * (n_local / n_global if (world > 1 and shard_batches) else 1.0), n + 1
        if log_interval and batch_idx % log_interval == 0 and rank == 0:
            print(f"Epoch [{epoch + 1}], Batch [{batch_idx + 1}], Loss: {value:.6f}")
    if world > 1 and shard_batches:          # every rank returns the same epoch average (schedulers / early stopping stay in step)
        acc = torch.tensor([total], dtype=torch.float64, device=device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        total = float(acc[0])
    return total / max(n, 1)


def graphed_step_for(model, optimizer, k=16, loss_mode="3dmatch", **kw) -> "GraphedStep":
    """The ``GraphedStep`` of ``(model, optimizer, k, loss_mode)``, created on first use and kept on the optimizer (its captured
    graphs hold pointers into the optimizer's buckets)."""
    cache = optimizer.__dict__.setdefault("_graphed_steps", {})
    key = (id(model), int(k), loss_mode)
    if key not in cache:
        cache[key] = GraphedStep(model, optimizer, k=k, loss_mode=loss_mode, **kw)
    return cache[key]


def train_one_epoch(model, dataloader, optimizer, device, epoch, writer=None, use_pointnet=False, log_interval=10, beta=1, k=16,
                    skip_nonfinite=False, loss_mode="3dmatch", shard_batches=True, graph="auto"):
    """One pass over ``dataloader`` (7-tuples of the dataset classes); returns the average loss.

    The reference's arccos pose loss is NaN / has a non-finite gradient at a perfect fit and for pairs without inliers.  The
    reference runs under ``set_detect_anomaly(True)`` (:1124) and stops there: so does this loop by default (``RuntimeError``
    on every rank of a data-parallel job, the decision is collective).  ``skip_nonfinite=True`` instead drops such a step on
    all ranks together and keeps going.  Batches containing ``None`` are skipped as in the reference (:969) -- by all ranks
    together.  Under ``torch.distributed`` (one process per GPU) every rank iterates the same loader and runs its
    ``dp.shard_range`` share of each step batch (``shard_batches=False``: the loader already yields per-rank batches).
    ``loss_mode``: ``"3dmatch"`` = ``corr + rot + trans`` (train :1120); ``"kitti"`` = ``corr + equi``
    (``src/kitti_train_egnn_with_batch.py:1089``).
    ``graph``: ``"auto"`` (default) runs the loop through ``GraphedStep`` whenever the optimizer is this package's ``StepOptimizer``
    and ``use_pointnet`` is off -- same kernels, same weights bit for bit, no host round trip per step, and a non-finite step is
    reported when the epoch ends (or at the next log line) instead of at once; ``False`` (or ``EQUIGSPR_EAGER_TRAINING=1``) keeps
    the eager loop below; a ``GraphedStep`` instance is used as given."""
    model.train()
    if graph == "auto":
        graph = isinstance(optimizer, StepOptimizer) and not use_pointnet and os.environ.get("EQUIGSPR_EAGER_TRAINING") != "1"
    if graph:
        # the product loop: static input slots, the step captured once in a hipGraph and replayed per batch, the non-finite decision
        # on the device (GraphedStep); the loader's host batches are packed and copied one step ahead
        stepper = graph if isinstance(graph, GraphedStep) else graphed_step_for(model, optimizer, k=k, loss_mode=loss_mode)
        loader = dataloader.loader if isinstance(dataloader, DeviceBatches) else dataloader
        avg = stepper.run(loader, epoch=epoch, log_interval=log_interval, skip_nonfinite=skip_nonfinite, shard_batches=shard_batches)
        print(f"Epoch [{epoch + 1}], Average Loss: {avg:.6f}")
        return avg

    def step_loss(m, batch):
        return _forward(m, batch, k, use_pointnet, loss_mode)[1]

    avg = run_training_batches(model, _device_batches(dataloader, device), optimizer, step_loss, device, epoch=epoch,
                               log_interval=log_interval, skip_nonfinite=skip_nonfinite, shard_batches=shard_batches)
    print(f"Epoch [{epoch + 1}], Average Loss: {avg:.6f}")
    return avg


@torch.no_grad()
def validate(model, dataloader, device, epoch, writer=None, use_pointnet=False, beta=1, k=16, process_group=None, shard_batches=True):
    """Returns ``(avg_loss, avg_rot_loss + avg_trans_loss, avg_corr_loss)`` over the batches with a finite loss.  Under
    ``torch.distributed`` every rank evaluates its ``dp.shard_range`` share of each batch (weight ``n_local / B``) and ONE
    all-reduce at the end combines the running sums: no per-batch collective, so ranks may skip batches independently."""
    model.eval()
    was_variant = getattr(model, "variant", None)
    if was_variant is not None:
        model.variant = "train"          # the reference validates with the training forward (:1208) -- it needs the losses
    world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(process_group) if world > 1 else 0
    # running sums stay on the device (fp64): [loss, rot, trans, corr, weight] -- ONE read-back per call instead of four per batch
    acc_dev = torch.zeros(5, dtype=torch.float64, device=device)
    try:
        for batch in _device_batches(dataloader, device):
            if batch is None or any(item is None for item in batch):
                continue
            weight = 1.0
            if world > 1 and shard_batches:
                batch, n_local, n_global = shard_batch(batch, rank, world)
                if n_local == 0:
                    continue
                weight = n_local / n_global
            out, loss, rot, trans = _forward(model, batch, k, use_pointnet)
            vals = torch.stack([loss.detach().double().reshape(()), rot.mean().double(), trans.mean().double(), out[2].mean().double(),
                                torch.ones((), dtype=torch.float64, device=loss.device)])
            ok = torch.isfinite(vals[0])                       # a batch with a non-finite loss does not count (as before)
            acc_dev += torch.where(ok, vals * weight, torch.zeros_like(vals))
    finally:
        if was_variant is not None:
            model.variant = was_variant
    if world > 1:
        acc = acc_dev if dist.get_backend(process_group) == "nccl" else acc_dev.cpu()
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=process_group)
        acc_dev = acc
    sums, n = acc_dev[:4].tolist(), float(acc_dev[4])
    avg = [s / max(n, 1e-12) if n > 0 else 0.0 for s in sums]
    if rank == 0:
        print(f"Validation Loss: {avg[0]:.6f} | Pose rot Loss: {avg[1]:.6f} | Pose trans Loss: {avg[2]:.6f}")
    return avg[0], avg[1] + avg[2], avg[3]


def save_checkpoint(epoch, pointnet, egnn, cross_attention, optimizer, save_dir="./checkpoints2", is_best=False, use_pointnet=False):
    """``save_dir/model_epoch_{epoch}.pth`` (+ ``save_dir/best_checkpoint``) in the reference layout (:1311-1348)."""
    os.makedirs(save_dir, exist_ok=True)
    ckpt = {"epoch": epoch, "egnn_state_dict": egnn.state_dict(), "cross_attention_state_dict": cross_attention.state_dict(),
            "optimizer_state_dict": optimizer.state_dict()}
    if use_pointnet and pointnet is not None:
        ckpt["pointnet_state_dict"] = pointnet.state_dict()
    path = os.path.join(save_dir, f"model_epoch_{epoch}.pth")
    torch.save(ckpt, path)
    if is_best:
        torch.save(ckpt, os.path.join(save_dir, "best_checkpoint"))
    return path


def load_checkpoint(checkpoint_path, pointnet=None, egnn=None, cross_attention=None, optimizer=None, use_pointnet=False,
                    device="cuda:0", trust_pickle=False):
    """Restores what is passed in and returns ``(checkpoint, epoch)`` (:1350-1396).  ``cross_attention`` contains the EGNN:
    loading both, as the reference does, writes the EGNN weights twice with the same values.  The payload (an int epoch,
    tensor state_dicts, an Adam state of tensors and floats) loads with ``weights_only=True``; ``trust_pickle=True`` falls
    back to full unpickling for a legacy file from a source you trust."""
    if not os.path.exists(checkpoint_path):
        raise FileNotFoundError(f"Checkpoint file {checkpoint_path} not found.")
    ckpt = torch.load(checkpoint_path, map_location=device, weights_only=not trust_pickle)
    if use_pointnet and pointnet is not None and "pointnet_state_dict" in ckpt:
        pointnet.load_state_dict(ckpt["pointnet_state_dict"])
    if egnn is not None and "egnn_state_dict" in ckpt:
        egnn.load_state_dict(ckpt["egnn_state_dict"])
    if cross_attention is not None and "cross_attention_state_dict" in ckpt:
        cross_attention.load_state_dict(ckpt["cross_attention_state_dict"])
    if optimizer is not None and "optimizer_state_dict" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    return ckpt, ckpt.get("epoch", 0)


@torch.no_grad()
def evaluate_model(checkpoint_path, save_dir, model, dataloader, device, use_pointnet=False, k=16, tau=0.09, verbose=True, graph="auto"):
    """eval :1084-1296.  ``model`` is a ``CrossAttentionPoseRegression(variant="eval")`` (the evaluation head is defined for
    one pair at a time, eval :719); ``checkpoint_path=None`` evaluates the weights as they are.  Per pair the four numbers
    of ``tools/evaluation_metrics.py`` come from one kernel launch instead of a D2H copy + numpy; only the per-pair scalars
    are read back.  Returns the reference's dict of averages and writes ``save_dir/evaluation_results.txt``.
    ``graph="auto"`` (default, unless ``use_pointnet``): the per-pair body runs as captured hipGraphs over static input slots
    (``GraphedEval``, kept on the model); ``False`` launches it eagerly; a ``GraphedEval`` instance is used as given."""
    os.makedirs(save_dir, exist_ok=True)
    if checkpoint_path is not None:
        load_checkpoint(checkpoint_path, None, model.egnn, model, optimizer=None, device=device)
    model.eval()
    if graph == "auto":
        graph = not use_pointnet and os.environ.get("EQUIGSPR_EAGER_TRAINING") != "1"
    if graph:
        if not isinstance(graph, GraphedEval):
            cache = model.__dict__.setdefault("_graphed_eval", {})
            graph = cache.get((int(k), float(tau))) or cache.setdefault((int(k), float(tau)), GraphedEval(model, k=k, tau=tau))
        table = graph.run(dataloader.loader if isinstance(dataloader, DeviceBatches) else dataloader)
        vals = table[:, [0, 1, 2, 4]].numpy()
    else:
        per_pair = []
        for batch_idx, batch in enumerate(_device_batches(dataloader, device)):
            if batch is None or any(item is None for item in batch[1:]):
                continue
            corr, labels, xyz_0, xyz_1, feat_0, feat_1, gt_pose = batch
            graph_ = build_pair_graph(xyz_0, xyz_1, k, loop=True)
            if use_pointnet:
                feat_0, feat_1 = _pointnet_features(graph_, xyz_0, xyz_1)
            out = model(feat_0, xyz_0, graph_, None, feat_1, xyz_1, graph_, None, corr, labels, gt_pose)
            m = evaluate_batch(out[0], out[1], gt_pose, xyz_0, xyz_1, tau=tau)
            per_pair.append(torch.stack([m["rotation_error"], m["translation_error"], m["recall"], m["f1"]], dim=1))
        vals = torch.cat(per_pair).cpu().numpy() if per_pair else torch.zeros(0, 4).numpy()
    if verbose:
        for i, (re, te, rc, f1) in enumerate(vals):
            print(f"Batch {i}: Rot Err: {re:.4f}, Trans Err: {te:.4f}, Recall: {rc:.4f}, F1: {f1:.4f}")
    mean = vals.mean(axis=0) if len(vals) else [float("nan")] * 4
    avg = {"Average Rotation Error": float(mean[0]), "Average Translation Error": float(mean[1]),
           "Average Recall": float(mean[2]), "Average F1 Score": float(mean[3])}
    with open(os.path.join(save_dir, "evaluation_results.txt"), "w") as f:
        for name, value in avg.items():
            f.write(f"{name}: {value:.4f}\n")
    return avg
