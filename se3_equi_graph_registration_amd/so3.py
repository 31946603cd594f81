"""``SO3TensorProductLayer`` with the reference interface (src/clean_so3_model.py:34-52), HIP forward.

SURVEY §0.3 / §8(a) A15: the only piece of ``clean_so3_model.py`` that runs; it computes
``MLP(D * x (x) x)``.  The HIP kernel (csrc/so3_tp.hip) never materialises the ``[E, D*D]`` tensor product."""
from __future__ import annotations

import ctypes

import torch
from torch import nn

from . import _lib
from .ops import _need, _p, _stream


class SO3TensorProductLayer(nn.Module):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.mlp = nn.Sequential(nn.Linear(input_dim * input_dim, 2 * output_dim), nn.ReLU(),
                                 nn.Linear(2 * output_dim, output_dim))

    def forward(self, x):
        """``x [num_edges, input_dim]`` -> ``[num_edges, output_dim]`` (inference; no CPU path)."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("SO3TensorProductLayer: the HIP op is forward-only; call it under torch.no_grad()")
        if self.input_dim != 32 or self.output_dim != 32:
            raise NotImplementedError("SO3TensorProductLayer: the HIP kernel is compiled for input_dim=output_dim=32 "
                                      "(the reference's instantiation, clean_so3_model.py:85)")
        x = _need(x.contiguous().float(), "x", torch.float32, 2)
        n = x.shape[0]
        w1, b1, w2, b2 = (t.detach().contiguous() for t in (self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias))
        for t in (w1, b1, w2, b2):
            _need(t, "SO3TensorProductLayer weights", torch.float32)
        y = torch.empty(n, self.output_dim, dtype=torch.float32, device=x.device)
        scratch = torch.empty(w1.numel(), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().equigspr_so3_tensor_product_forward(_p(x), n, self.input_dim, self.output_dim, _p(w1), _p(b1),
                                                                  _p(w2), _p(b2), _p(scratch), _p(y), _stream()),
                   "so3_tensor_product_forward")
        return y
