"""Spherical harmonics / irrep basis with the reference interface (tools/spherical_harmonics.py, tools/basis.py,
tools/irr_repr.py), HIP forward.

Per-point work (SH of every degree, the ``K_J = Y_J Q_J^T`` contraction) runs in csrc/sh.hip.  The change-of-basis
constants ``Q_J`` are built once per ``(J, d_in, d_out)`` on the host in fp64 (as the reference does under
``torch_default_dtype(float64)`` with an lru/disk cache, basis.py:123-138) and kept on the device.

The reference's ``irr_repr`` needs a ``J_dense`` blob its repository does not ship (tools/irr_repr.py:12-20); here the
Wigner matrix is recovered from the property the reference's test pins, ``Y(R x) = D(R) Y(x)``
(tests/test_irrep_repr.py), and ``Q_J`` -- unique up to sign -- is normalised so that its first entry within 10 % of the largest magnitude is positive."""
from __future__ import annotations

from functools import lru_cache
from itertools import product
from math import pi

import numpy as np
import torch

from . import _lib
from .ops import _p, _stream

MAX_L = 8


def to_order(degree):
    return 2 * degree + 1


def _device_in(t, last):
    if not torch.is_tensor(t):
        raise TypeError("expected a tensor")
    if not t.is_cuda:
        raise RuntimeError("spherical harmonics: input must live on the GPU (there is no CPU path)")
    if t.shape[-1] != last:
        raise RuntimeError(f"expected last dimension {last}, got {tuple(t.shape)}")
    if t.dtype not in (torch.float32, torch.float64):
        t = t.float()
    return t.contiguous()


def _sh_all(inp, mode, max_l, want_sph=False, out_dtype=torch.float32):
    """All degrees 0..max_l: returns (Y [n, (max_l+1)^2], sph or None)."""
    if max_l > MAX_L:
        raise NotImplementedError(f"spherical harmonics are compiled up to degree {MAX_L}")
    flat = inp.reshape(-1, inp.shape[-1])
    n = flat.shape[0]
    y = torch.empty(n, (max_l + 1) ** 2, dtype=out_dtype, device=inp.device)
    sph = torch.empty(n, 3, dtype=torch.float32, device=inp.device) if want_sph else None
    y32, y64 = (y, None) if out_dtype == torch.float32 else (None, y)
    _lib.check(_lib.lib().equigspr_sh_forward(_p(flat), int(flat.dtype == torch.float64), n, mode, max_l, _p(sph), _p(y32), _p(y64),
                                              _stream()), "sh_forward")
    return y, sph


def get_spherical_from_cartesian(cartesian, divide_radius_by=1.0):
    """``[..., 3]`` -> ``[radius, alpha, beta]`` (tools/basis.py:57-95)."""
    c = _device_in(cartesian, 3)
    _, sph = _sh_all(c, 0, 0, want_sph=True)
    sph = sph.view(*c.shape).to(cartesian.dtype if cartesian.dtype.is_floating_point else torch.float32)
    if divide_radius_by != 1.0:
        sph[..., 0] /= divide_radius_by
    return sph


def get_spherical_harmonics(l, theta, phi):
    """Tesseral harmonics of degree ``l``: ``[*theta.shape, 2l+1]`` (tools/spherical_harmonics.py:107-123)."""
    ang = _device_in(torch.stack(torch.broadcast_tensors(theta, phi), dim=-1), 2)
    y, _ = _sh_all(ang, 1, l, out_dtype=ang.dtype)
    return y[:, l * l:(l + 1) ** 2].reshape(*ang.shape[:-1], 2 * l + 1)


def get_spherical_harmonics_element(l, m, theta, phi):
    return get_spherical_harmonics(l, theta, phi)[..., m + l]


def precompute_sh(r_ij, max_J):
    """``r_ij`` in spherical coordinates ``[..., (radius, alpha, beta)]`` -> ``{J: [..., 2J+1]}`` (tools/basis.py:140-151)."""
    s = _device_in(r_ij, 3)
    y, _ = _sh_all(s, 2, max_J, out_dtype=s.dtype)
    return {j: y[:, j * j:(j + 1) ** 2].reshape(*s.shape[:-1], 2 * j + 1) for j in range(max_J + 1)}


# --------------------------------------------------------------------------- host-side constants (fp64)
def _host_sh(l, pts):
    """Y_l at unit vectors x = rot(alpha, beta, 0) e_z, evaluated as the device kernel does (fp64 numpy)."""
    p = pts / np.linalg.norm(pts, axis=-1, keepdims=True)
    ct, st = -p[:, 2], np.sqrt(np.maximum(0.0, 1.0 - p[:, 2] ** 2))        # theta = pi - beta, beta = acos(z)
    rxy = np.hypot(p[:, 0], p[:, 1])
    cphi = np.where(rxy > 0, p[:, 0] / np.where(rxy > 0, rxy, 1.0), 1.0)
    sphi = np.where(rxy > 0, p[:, 1] / np.where(rxy > 0, rxy, 1.0), 0.0)
    out = np.zeros((p.shape[0], 2 * l + 1))
    cm, sm, pmm = np.ones_like(ct), np.zeros_like(ct), np.ones_like(ct)
    for m in range(l + 1):
        if m > 0:
            cm, sm = cm * cphi - sm * sphi, sm * cphi + cm * sphi
            pmm = pmm * (-(2 * m - 1)) * st
        p2, p1 = np.zeros_like(ct), pmm
        for ll in range(m + 1, l + 1):
            p2, p1 = p1, ((2 * ll - 1) * ct * p1 - (ll + m - 1) * p2) / (ll - m)
        norm = (2 * l + 1) / (4 * pi)
        if m > 0:
            norm *= 2.0 / np.prod(np.arange(l - m + 1, l + m + 1, dtype=np.float64))
        val = np.sqrt(norm) * p1
        if m == 0:
            out[:, l] = val
        else:
            out[:, l + m], out[:, l - m] = val * cm, val * sm
    return out


def _rot_np(a, b, c):
    def rz(g):
        return np.array([[np.cos(g), -np.sin(g), 0.0], [np.sin(g), np.cos(g), 0.0], [0.0, 0.0, 1.0]])

    def ry(g):
        return np.array([[np.cos(g), 0.0, np.sin(g)], [0.0, 1.0, 0.0], [-np.sin(g), 0.0, np.cos(g)]])
    return rz(a) @ ry(b) @ rz(c)


_DIRECTIONS = np.random.default_rng(7).standard_normal((96, 3))


def _wigner_np(l, a, b, c):
    y = _host_sh(l, _DIRECTIONS)
    yr = _host_sh(l, _DIRECTIONS @ _rot_np(a, b, c).T)
    return np.linalg.lstsq(y, yr, rcond=None)[0].T


def irr_repr(order, alpha, beta, gamma, dtype=None):
    """Irreducible representation D^order(alpha, beta, gamma) of the ZYZ rotation ``rot`` (tools/irr_repr.py:44-52)."""
    d = _wigner_np(order, float(alpha), float(beta), float(gamma))
    return torch.from_numpy(d).to(dtype or torch.get_default_dtype())


def rot(alpha, beta, gamma):
    """ZYZ Euler rotation Rz(alpha) Ry(beta) Rz(gamma) (tools/irr_repr.py:85-90)."""
    return torch.from_numpy(_rot_np(float(alpha), float(beta), float(gamma))).to(torch.get_default_dtype())


_SYLVESTER_ANGLES = [(4.4, 5.6, 4.6), (4.9, 6.1, 4.1), (0.5, 4.1, 5.4), (2.2, 3.5, 5.6), (2.5, 0.3, 3.9)]


@lru_cache(maxsize=None)
def _q_np(J, order_in, order_out):
    rows = []
    for a, b, c in _SYLVESTER_ANGLES:
        r_t = np.kron(_wigner_np(order_out, a, b, c), _wigner_np(order_in, a, b, c))
        d_j = _wigner_np(J, a, b, c)
        rows.append(np.kron(r_t, np.eye(d_j.shape[0])) - np.kron(np.eye(r_t.shape[0]), d_j.T))
    _, s, vt = np.linalg.svd(np.concatenate(rows, axis=0))
    if int((s < 1e-10).sum()) != 1:
        raise RuntimeError(f"basis_transformation_Q_J({J},{order_in},{order_out}): null space is not one-dimensional")
    q = vt[-1]
    q = q * np.sign(q[np.flatnonzero(np.abs(q) > 0.9 * np.abs(q).max())[0]])     # first large entry positive
    return q.reshape(to_order(order_out) * to_order(order_in), to_order(J))


def get_R_tensor(order_out, order_in, a, b, c):
    return torch.from_numpy(np.kron(_wigner_np(order_out, float(a), float(b), float(c)),
                                    _wigner_np(order_in, float(a), float(b), float(c)))).to(torch.get_default_dtype())


def basis_transformation_Q_J(J, order_in, order_out):
    """``[(2 order_out + 1)(2 order_in + 1), 2J + 1]`` fp32 (tools/basis.py:123-138)."""
    return torch.from_numpy(_q_np(int(J), int(order_in), int(order_out))).float()


@lru_cache(maxsize=None)
def _q_table(d_in, d_out, device):
    js = range(abs(d_in - d_out), d_in + d_out + 1)
    qm = 2 * (d_in + d_out) + 1
    tab = np.zeros((len(js), to_order(d_out) * to_order(d_in), qm), dtype=np.float32)
    for jx, j in enumerate(js):
        tab[jx, :, :2 * j + 1] = _q_np(j, d_in, d_out).astype(np.float32)
    return torch.from_numpy(tab).to(device)


def get_basis(r_ij, max_degree, differentiable=False):
    """Equivariant weight basis ``{'d_in,d_out': [*batch, 1, 2 d_out+1, 1, 2 d_in+1, 2 min(d_in,d_out)+1]}`` from Cartesian
    relative positions (tools/basis.py:153-205).  The HIP op is forward-only, so ``differentiable=True`` is rejected."""
    if differentiable:
        raise NotImplementedError("get_basis: the HIP op is forward-only")
    c = _device_in(r_ij, 3)
    max_l = 2 * max_degree
    y, _ = _sh_all(c, 0, max_l)
    n = y.shape[0]
    basis = {}
    L = _lib.lib()
    for d_in, d_out in product(range(max_degree + 1), range(max_degree + 1)):
        q = _q_table(d_in, d_out, str(c.device))
        n_j = 2 * min(d_in, d_out) + 1
        out = torch.empty(n, to_order(d_out) * to_order(d_in) * n_j, dtype=torch.float32, device=c.device)
        _lib.check(L.equigspr_sh_basis_forward(_p(y), n, max_l, d_in, d_out, _p(q), _p(out), _stream()), "sh_basis_forward")
        basis[f"{d_in},{d_out}"] = out.view(*c.shape[:-1], 1, to_order(d_out), 1, to_order(d_in), n_j).to(r_ij.dtype)
    return basis
