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

import torch
import torch.distributed as dist

from . import ops
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


def _forward(model, batch, k, use_pointnet=False, loss_mode="3dmatch"):
    corr, labels, xyz_0, xyz_1, feat_0, feat_1, gt_pose = batch
    graph = build_pair_graph(xyz_0, xyz_1, k, loop=True)        # knn_graph(xyz[i], k=16, loop=True) per cloud (:1005-1006)
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
        total, n = total + value, n + 1
        if log_interval and batch_idx % log_interval == 0 and rank == 0:
            print(f"Epoch [{epoch + 1}], Batch [{batch_idx + 1}], Loss: {value:.6f}")
    return total / max(n, 1)


def train_one_epoch(model, dataloader, optimizer, device, epoch, writer=None, use_pointnet=False, log_interval=10, beta=1, k=16,
                    skip_nonfinite=False, loss_mode="3dmatch", shard_batches=True):
    """One pass over ``dataloader`` (7-tuples of the dataset classes); returns the average loss.

    The reference's arccos pose loss is NaN / has a non-finite gradient at a perfect fit and for pairs without inliers.  The
    reference runs under ``set_detect_anomaly(True)`` (:1124) and stops there: so does this loop by default (``RuntimeError``
    on every rank of a data-parallel job, the decision is collective).  ``skip_nonfinite=True`` instead drops such a step on
    all ranks together and keeps going.  Batches containing ``None`` are skipped as in the reference (:969) -- by all ranks
    together.  Under ``torch.distributed`` (one process per GPU) every rank iterates the same loader and runs its
    ``dp.shard_range`` share of each step batch (``shard_batches=False``: the loader already yields per-rank batches).
    ``loss_mode``: ``"3dmatch"`` = ``corr + rot + trans`` (train :1120); ``"kitti"`` = ``corr + equi``
    (``src/kitti_train_egnn_with_batch.py:1089``)."""
    model.train()

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
    sums, n = [0.0, 0.0, 0.0, 0.0], 0.0
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
            vals = [loss.item(), rot.mean().item(), trans.mean().item(), out[2].mean().item()]
            if vals[0] != vals[0]:
                continue
            sums, n = [a + weight * b for a, b in zip(sums, vals)], n + weight
    finally:
        if was_variant is not None:
            model.variant = was_variant
    if world > 1:
        acc = torch.tensor(sums + [n], dtype=torch.float64, device=device if dist.get_backend(process_group) == "nccl" else "cpu")
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=process_group)
        sums, n = acc[:4].tolist(), float(acc[4])
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
def evaluate_model(checkpoint_path, save_dir, model, dataloader, device, use_pointnet=False, k=16, tau=0.09, verbose=True):
    """eval :1084-1296.  ``model`` is a ``CrossAttentionPoseRegression(variant="eval")`` (the evaluation head is defined for
    one pair at a time, eval :719); ``checkpoint_path=None`` evaluates the weights as they are.  Per pair the four numbers
    of ``tools/evaluation_metrics.py`` come from one kernel launch instead of a D2H copy + numpy; only the per-pair scalars
    are read back.  Returns the reference's dict of averages and writes ``save_dir/evaluation_results.txt``."""
    os.makedirs(save_dir, exist_ok=True)
    if checkpoint_path is not None:
        load_checkpoint(checkpoint_path, None, model.egnn, model, optimizer=None, device=device)
    model.eval()
    per_pair = []
    for batch_idx, batch in enumerate(_device_batches(dataloader, device)):
        corr, labels, xyz_0, xyz_1, feat_0, feat_1, gt_pose = batch
        graph = build_pair_graph(xyz_0, xyz_1, k, loop=True)
        if use_pointnet:
            feat_0, feat_1 = _pointnet_features(graph, xyz_0, xyz_1)
        out = model(feat_0, xyz_0, graph, None, feat_1, xyz_1, graph, None, corr, labels, gt_pose)
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
