"""CPU restatement of the spherical-harmonic / irrep-basis helpers.  TEST INFRASTRUCTURE ONLY:
imported by tests/ (and nothing in the product package).

Follows /root/reference/tools/:
  * spherical_harmonics.py:35-73  (``lpmv``: associated Legendre with Condon-Shortley phase),
    :75-105 (``get_spherical_harmonics_element``), :107-123 (``get_spherical_harmonics``);
  * basis.py:57-95 (``get_spherical_from_cartesian``), :140-151 (``precompute_sh``), :110-138
    (``basis_transformation_Q_J``), :153-205 (``get_basis``);
  * irr_repr.py:103-104 (``spherical_harmonics(order, alpha, beta) = SH(order, theta = pi - beta, phi = alpha)``).

Pinning: the SH functions and the Cartesian->spherical map are pinned by golden vectors produced by importing the
reference's own ``tools/spherical_harmonics.py`` / ``tools/basis.py`` (tests/golden/gen_golden.py, ``sh``).
``irr_repr`` / ``basis_transformation_Q_J`` are PARITY UNPINNED: the reference's Wigner-D code needs the ``J_dense``
data blob (tools/irr_repr.py:12-20) which the repository does not ship, and the sign of ``Q_J`` is whatever LAPACK
returned on the author's machine.  Here D^l(R) is recovered from the property the reference's own test pins
(tests/test_irrep_repr.py: ``Y(R x) = D(R) Y(x)``) and ``Q_J`` is the unique unit null vector of the same Sylvester
system (basis.py:113-121), sign-normalised so that its first entry within 10 % of the largest magnitude is positive."""
from __future__ import annotations

from itertools import product
from math import pi, sqrt

import numpy as np
import torch


# ---------------------------------------------------------------------------- spherical harmonics
def _semifactorial(x: int) -> float:                       # spherical_harmonics.py:21-23
    out = 1.0
    for v in range(x, 1, -2):
        out *= v
    return out


def _pochhammer(x: int, k: int) -> float:                  # spherical_harmonics.py:25-27
    out = float(x)
    for v in range(x + 1, x + k):
        out *= v
    return out


def lpmv(l: int, m: int, x: torch.Tensor) -> torch.Tensor:
    """P_l^{|m|}(x) with Condon-Shortley phase (spherical_harmonics.py:35-73; the memo dict is a plain recursion here)."""
    m = abs(m)
    if l == 0:
        return torch.ones_like(x)
    if m == l:
        return (-1) ** m * _semifactorial(2 * m - 1) * torch.pow(1 - x * x, m / 2)
    y = ((2 * l - 1) / (l - m)) * x * lpmv(l - 1, m, x)
    if l - m > 1:
        y = y - ((l + m - 1) / (l - m)) * lpmv(l - 2, m, x)
    return y


def sh_element(l: int, m: int, theta: torch.Tensor, phi: torch.Tensor) -> torch.Tensor:
    """Tesseral harmonic Y_l^m (spherical_harmonics.py:75-105)."""
    m_abs = abs(m)
    n = sqrt((2 * l + 1) / (4 * pi))
    leg = lpmv(l, m_abs, torch.cos(theta))
    if m == 0:
        return n * leg
    y = torch.cos(m * phi) if m > 0 else torch.sin(m_abs * phi)
    return y * leg * (n * sqrt(2.0 / _pochhammer(l - m_abs + 1, 2 * m_abs)))


def get_spherical_harmonics(l: int, theta: torch.Tensor, phi: torch.Tensor) -> torch.Tensor:
    """[*theta.shape, 2l+1], m = -l..l on the last axis (spherical_harmonics.py:107-123)."""
    return torch.stack([sh_element(l, m, theta, phi) for m in range(-l, l + 1)], dim=-1)


def get_spherical_from_cartesian(c: torch.Tensor) -> torch.Tensor:
    """[radius, alpha, beta] with the axis permutation (x, y, z) = components (2, 0, 1) (basis.py:57-95)."""
    r_xy = c[..., 2] ** 2 + c[..., 0] ** 2
    beta = torch.atan2(torch.sqrt(r_xy), c[..., 1])
    alpha = torch.atan2(c[..., 0], c[..., 2])
    return torch.stack([torch.sqrt(r_xy + c[..., 1] ** 2), alpha, beta], dim=-1)


def precompute_sh(r_sph: torch.Tensor, max_j: int):
    """{J: Y_J} with theta = pi - beta, phi = alpha (basis.py:140-151, irr_repr.py:103-104)."""
    return {j: get_spherical_harmonics(j, pi - r_sph[..., 2], r_sph[..., 1]) for j in range(max_j + 1)}


# ---------------------------------------------------------------------------- Wigner D, Q_J, basis
def _rot(a: float, b: float, c: float) -> np.ndarray:
    """ZYZ Euler rotation Rz(a) Ry(b) Rz(c) (irr_repr.py:55-90)."""
    def rz(g):
        return np.array([[np.cos(g), -np.sin(g), 0.0], [np.sin(g), np.cos(g), 0.0], [0.0, 0.0, 1.0]])

    def ry(g):
        return np.array([[np.cos(g), 0.0, np.sin(g)], [0.0, 1.0, 0.0], [-np.sin(g), 0.0, np.cos(g)]])
    return rz(a) @ ry(b) @ rz(c)


def _sh_of_points(l: int, pts: np.ndarray) -> np.ndarray:
    """Y_l at unit vectors in the irr_repr convention: x = rot(alpha, beta, 0) e_z (irr_repr.py:73-80,:103-104)."""
    p = torch.from_numpy(pts / np.linalg.norm(pts, axis=-1, keepdims=True))
    beta = torch.acos(torch.clamp(p[..., 2], -1.0, 1.0))
    alpha = torch.atan2(p[..., 1], p[..., 0])
    return get_spherical_harmonics(l, pi - beta, alpha).numpy()


_SAMPLE = np.random.default_rng(20240611).standard_normal((64, 3))


def irr_repr(l: int, a: float, b: float, c: float) -> np.ndarray:
    """D^l(a, b, c) from its defining property Y_l(R x) = D^l(R) Y_l(x) (least squares over 64 fixed directions)."""
    y = _sh_of_points(l, _SAMPLE)                         # [S, 2l+1]
    yr = _sh_of_points(l, _SAMPLE @ _rot(a, b, c).T)
    return np.linalg.lstsq(y, yr, rcond=None)[0].T


_ANGLES = [(0.3 + 1.1 * i, 0.7 + 0.9 * i, 1.9 + 0.5 * i) for i in range(5)]


def basis_transformation_Q_J(j: int, d_in: int, d_out: int) -> np.ndarray:
    """Unit null vector of the stacked Sylvester systems kron(D_out (x) D_in, I) - kron(I, D_J^T) (basis.py:104-138),
    reshaped to [(2 d_out + 1)(2 d_in + 1), 2J + 1]; sign: first entry within 10 % of the largest magnitude positive."""
    rows = []
    for a, b, c in _ANGLES:
        r_t = np.kron(irr_repr(d_out, a, b, c), irr_repr(d_in, a, b, c))
        d_j = irr_repr(j, a, b, c)
        rows.append(np.kron(r_t, np.eye(d_j.shape[0])) - np.kron(np.eye(r_t.shape[0]), d_j.T))
    _, s, vt = np.linalg.svd(np.concatenate(rows, axis=0))
    null = vt[s < 1e-10] if (s < 1e-10).any() else vt[-1:]
    assert null.shape[0] == 1, null.shape
    q = null[0]
    q = q * np.sign(q[np.flatnonzero(np.abs(q) > 0.9 * np.abs(q).max())[0]])
    return q.reshape((2 * d_out + 1) * (2 * d_in + 1), 2 * j + 1)


def get_basis(r_ij: torch.Tensor, max_degree: int):
    """{'d_in,d_out': [*batch, 1, 2 d_out + 1, 1, 2 d_in + 1, 2 min(d_in, d_out) + 1]} (basis.py:153-205)."""
    y = precompute_sh(get_spherical_from_cartesian(r_ij), 2 * max_degree)
    out = {}
    for d_in, d_out in product(range(max_degree + 1), range(max_degree + 1)):
        ks = [y[j] @ torch.from_numpy(basis_transformation_Q_J(j, d_in, d_out)).to(r_ij.dtype).T
              for j in range(abs(d_in - d_out), d_in + d_out + 1)]
        out[f"{d_in},{d_out}"] = torch.stack(ks, dim=-1).view(*r_ij.shape[:-1], 1, 2 * d_out + 1, 1, 2 * d_in + 1,
                                                                2 * min(d_in, d_out) + 1)
    return out
