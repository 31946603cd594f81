"""One flat fp32 parameter block + one flat gradient bucket per network (MI355X-first layout): the parameters of
the module are views into ``flat_params``, their ``.grad`` views into ``flat_grads`` -- the HIP kernels read the block directly and
accumulate parameter gradients straight into the bucket, the optimizer updates it in one launch, the data-parallel exchange sends it
as one message.  Shared by ``EGNN`` (contiguous ``named_parameters()`` order = the C-ABI block of include/equigspr.h) and
``IrrepEGNN`` (every tensor on a 16-byte boundary: ``equigspr_irrep_gnn_param_offset``)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import nn


class FlatParamsMixin:
    flat_params: Optional[torch.Tensor] = None
    flat_grads: Optional[torch.Tensor] = None

    def _ordered_params(self) -> List[nn.Parameter]:
        return [p for _, p in self.named_parameters()]

    def _flat_layout(self) -> Tuple[List[int], int]:
        """(offset of every parameter of ``_ordered_params()`` in floats, size of the block).  Default: densely packed."""
        offs, off = [], 0
        for p in self._ordered_params():
            offs.append(off)
            off += p.numel()
        return offs, off

    def flat_numel(self) -> int:
        return self._flat_layout()[1]

    def rehome(self, flat: torch.Tensor, grads: torch.Tensor):
        """Move the parameter block and its gradient bucket into caller-owned storage (fp32, ``flat_numel()`` elements each, 16-byte
        aligned) -- ``dp.DataParallelStep`` uses this to put the network and the pose-head parameters in ONE buffer, so the gradient
        exchange is a single all-reduce."""
        self._ensure_flat(storage=(flat, grads))

    def _ensure_flat(self, storage=None):
        ps = self._ordered_params()
        offs, total = self._flat_layout()
        flat = self.flat_params if storage is None else None
        ok = flat is not None and flat.device == ps[0].device and flat.numel() == total
        if ok:
            base = flat.data_ptr()
            for p, off in zip(ps, offs):
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                    ok = False
                    break
        if ok:
            return
        dev = ps[0].device
        if dev.type != "cuda":
            raise RuntimeError("%s parameters are on %s: this package runs only on the MI355X (HIP) device; "
                               "move the module with .to('cuda')" % (type(self).__name__, dev))
        if storage is None:
            flat = torch.zeros(total, dtype=torch.float32, device=dev)
            grads = torch.zeros(total, dtype=torch.float32, device=dev)
        else:
            flat, grads = storage
            for t in (flat, grads):
                if (t.dtype != torch.float32 or t.numel() != total or t.device != dev or not t.is_contiguous()
                        or t.data_ptr() % 16):
                    raise RuntimeError("%s.rehome: storage must be contiguous fp32 CUDA tensors of flat_numel() elements, 16-byte aligned"
                                       % type(self).__name__)
            with torch.no_grad():
                grads.zero_()
                flat.zero_()                       # padding words between tensors stay zero (their gradients are never written)
        with torch.no_grad():
            for p, off in zip(ps, offs):
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1).float())
                old_grad = p.grad
                p.data = flat[off:off + n].view(p.shape)
                if old_grad is not None:
                    grads[off:off + n].copy_(old_grad.reshape(-1))
                p.grad = grads[off:off + n].view(p.shape)
        self.flat_params, self.flat_grads = flat, grads

    def _attach_grads(self):
        """(Re)attach ``.grad`` views into the flat bucket.  A parameter whose ``.grad`` is ``None`` was reset by
        ``optimizer.zero_grad(set_to_none=True)``: its slice of the bucket is zeroed first, because the backward kernels ACCUMULATE
        into the bucket."""
        ps = self._ordered_params()
        missing = [p.grad is None for p in ps]
        if not any(missing):
            return
        if all(missing):
            self.flat_grads.zero_()
        offs, _ = self._flat_layout()
        for p, off, miss in zip(ps, offs, missing):
            n = p.numel()
            if miss:
                if not all(missing):
                    self.flat_grads[off:off + n].zero_()
                p.grad = self.flat_grads[off:off + n].view(p.shape)

    def zero_grad(self, set_to_none: bool = False):          # keep the bucket views alive by default
        if self.flat_grads is not None and not set_to_none:
            self.flat_grads.zero_()
            self._attach_grads()
        else:
            nn.Module.zero_grad(self, set_to_none=set_to_none)
