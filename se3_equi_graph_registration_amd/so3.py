"""``SO3TensorProductLayer`` with the reference interface (src/clean_so3_model.py:34-52), HIP forward.

SURVEY §0.3 / §8(a) A15: the only piece of ``clean_so3_model.py`` that runs; it computes
``MLP(D * x (x) x)``.  The HIP kernel (csrc/so3_tp.hip) never materialises the ``[E, D*D]`` tensor product."""
from __future__ import annotations

import ctypes

import torch
from torch import nn

from . import _lib
from .ops import _need, _p, _stream, hip_forward_aten_backward


class SO3TensorProductLayer(nn.Module):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.mlp = nn.Sequential(nn.Linear(input_dim * input_dim, 2 * output_dim), nn.ReLU(),
                                 nn.Linear(2 * output_dim, output_dim))

    def _hip(self, x, w1, b1, w2, b2):
        n = x.shape[0]
        y = torch.empty(n, self.output_dim, dtype=torch.float32, device=x.device)
        scratch = torch.empty(w1.numel(), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().equigspr_so3_tensor_product_forward(_p(x), n, self.input_dim, self.output_dim, _p(w1.contiguous()),
                                                                  _p(b1.contiguous()), _p(w2.contiguous()), _p(b2.contiguous()),
                                                                  _p(scratch), _p(y), _stream()), "so3_tensor_product_forward")
        return y

    @staticmethod
    def _aten(x, w1, b1, w2, b2, chunk=8192):
        """clean_so3_model.py:41-52 in device ops, row chunks so the [E, D*D] product stays bounded (backward recompute only)."""
        d = x.shape[1]
        outs = []
        for lo in range(0, x.shape[0], chunk):
            xc = x[lo:lo + chunk]
            z = (float(d) * xc[:, :, None] * xc[:, None, :]).reshape(xc.shape[0], d * d)
            outs.append(torch.addmm(b2, torch.relu(torch.addmm(b1, z, w1.t())), w2.t()))
        return torch.cat(outs) if outs else x.new_zeros(0, w2.shape[0])

    def forward(self, x):
        """``x [num_edges, input_dim]`` -> ``[num_edges, output_dim]`` (no CPU path).  Forward: csrc/so3_tp.hip; under grad the
        backward re-runs the layer in device ops on the saved input (ops.hip_forward_aten_backward)."""
        if self.input_dim != 32 or self.output_dim != 32:
            raise NotImplementedError("SO3TensorProductLayer: the HIP kernel is compiled for input_dim=output_dim=32 "
                                      "(the reference's instantiation, clean_so3_model.py:85)")
        x = _need(x.contiguous().float(), "x", torch.float32, 2)
        params = (self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias)
        for t in params:
            _need(t, "SO3TensorProductLayer weights", torch.float32)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            return hip_forward_aten_backward(self._hip, self._aten, (x, *params))
        return self._hip(x, *(t.detach() for t in params))
