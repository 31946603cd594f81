"""CPU: the hand-derived backward (tests/manual_backward.py -- the exact decomposition the HIP
backward kernels implement) against the golden gradients of the real reference (fp64)."""
import os

import numpy as np
import pytest
import torch

from tests.manual_backward import egnn_forward_backward

FIX = ["layer_n128_k8_f32_h1.npz", "layer_n128_k8_f33_h1.npz", "layer_n128_k8_f32_h4.npz", "layer_n96_k6_f32_h1_dup.npz"]


@pytest.mark.parametrize("name", FIX)
def test_manual_backward_matches_reference_autograd(golden_dir, name):
    fx = np.load(os.path.join(golden_dir, name))
    dt = torch.float64
    w = {k[2:]: torch.from_numpy(fx[k]).to(dt) for k in fx.files if k.startswith("w.")}
    h, x = torch.from_numpy(fx["h"]).to(dt), torch.from_numpy(fx["x"]).to(dt)
    e = torch.from_numpy(fx["edges"])
    attr = torch.ones(e.shape[1], 1, dtype=dt)
    ho, xo, G, gh, gx = egnn_forward_backward(w, h, x, e[0], e[1], attr, torch.from_numpy(fx["gh"]).to(dt),
                                              torch.from_numpy(fx["gx"]).to(dt))
    np.testing.assert_allclose(ho.numpy(), fx["f64.h_out"], atol=1e-11)
    np.testing.assert_allclose(xo.numpy(), fx["f64.x_out"], atol=1e-11)
    np.testing.assert_allclose(gh.numpy(), fx["f64.grad_h"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(gx.numpy(), fx["f64.grad_x"], rtol=1e-8, atol=1e-9)
    for k in w:
        ref = fx["f64.grad." + k]
        np.testing.assert_allclose(G[k].numpy(), ref, rtol=1e-8, atol=1e-10 * max(1.0, float(np.abs(ref).max())), err_msg=k)
