"""``SO3TensorProductLayer`` with the reference interface (src/clean_so3_model.py:34-52), HIP forward and (round 6) HIP backward.

SURVEY §0.3 / §8(a) A15: the only piece of ``clean_so3_model.py`` that runs; it computes
``MLP(D * x (x) x)``.  The HIP kernel (csrc/so3_tp.hip) never materialises the ``[E, D*D]`` tensor product."""
from __future__ import annotations

import ctypes

import torch
from torch import nn

from . import _lib
from .ops import _need, _p, _stream


class _So3TpFn(torch.autograd.Function):
    """y = W2 relu(W1 vec(D x x^T) + b1) + b2 with both directions in HIP (csrc/so3_tp.hip); the tensor product is never materialised."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        n, d = x.shape
        w1, b1, w2, b2 = (t.contiguous() for t in (w1, b1, w2, b2))
        y = torch.empty(n, w2.shape[0], dtype=torch.float32, device=x.device)
        scratch = torch.empty(w1.numel(), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().equigspr_so3_tensor_product_forward(_p(x), n, d, w2.shape[0], _p(w1), _p(b1), _p(w2), _p(b2), _p(scratch), _p(y),
                                                                  _stream()), "so3_tensor_product_forward")
        ctx.save_for_backward(x, w1, b1, w2)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w1, b1, w2 = ctx.saved_tensors
        n, d = x.shape
        L = _lib.lib()
        need = ctx.needs_input_grad
        gx = torch.empty_like(x) if need[0] else None
        gw1 = torch.empty_like(w1) if need[1] else None
        gb1 = torch.empty_like(b1) if need[2] else None
        gw2 = torch.empty_like(w2) if need[3] else None
        gb2 = torch.empty(w2.shape[0], dtype=torch.float32, device=x.device) if need[4] else None
        work = torch.empty(int(L.equigspr_so3_tensor_product_backward_work_size(n)), dtype=torch.float32, device=x.device)
        _lib.check(L.equigspr_so3_tensor_product_backward(_p(x), _p(gy.contiguous().float()), n, d, w2.shape[0], _p(w1), _p(b1), _p(w2), _p(work),
                                                          _p(gx), _p(gw1), _p(gb1), _p(gw2), _p(gb2), _stream()), "so3_tensor_product_backward")
        return gx, gw1, gb1, gw2, gb2


class SO3TensorProductLayer(nn.Module):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.mlp = nn.Sequential(nn.Linear(input_dim * input_dim, 2 * output_dim), nn.ReLU(),
                                 nn.Linear(2 * output_dim, output_dim))

    def forward(self, x):
        """``x [num_edges, input_dim]`` -> ``[num_edges, output_dim]`` (no CPU path).  Forward and backward: csrc/so3_tp.hip
        (``equigspr_so3_tensor_product_forward`` / ``_backward``)."""
        if self.input_dim != 32 or self.output_dim != 32:
            raise NotImplementedError("SO3TensorProductLayer: the HIP kernel is compiled for input_dim=output_dim=32 "
                                      "(the reference's instantiation, clean_so3_model.py:85)")
        x = _need(x.contiguous().float(), "x", torch.float32, 2)
        params = (self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias)
        for t in params:
            _need(t, "SO3TensorProductLayer weights", torch.float32)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            return _So3TpFn.apply(x, *params)
        with torch.no_grad():
            return _So3TpFn.apply(x, *(t.detach() for t in params))
