"""Spherical harmonics / irrep basis with the reference interface (tools/spherical_harmonics.py, tools/basis.py,
tools/irr_repr.py), HIP forward.

Per-point work (SH of every degree, the ``K_J = Y_J Q_J^T`` contraction) runs in csrc/sh.hip.  The change-of-basis
constants ``Q_J`` are built once per ``(J, d_in, d_out)`` on the host in fp64 (as the reference does under
``torch_default_dtype(float64)`` with an lru/disk cache, basis.py:123-138) and kept on the device.

The reference's ``wigner_d_matrix`` is ``Z(alpha) J Z(beta) J Z(gamma)`` with ``J`` from a ``J_dense`` blob its repository does
not ship (tools/irr_repr.py:12-31).  ``J_l`` is the degree-l representation of the half turn about (0, 1, 1)/sqrt(2) -- the
proper rotation with ``J Rz(b) J = Ry(b)``, ``J^2 = I`` -- so it is rebuilt here from the one property the reference's test
pins, ``Y(R x) = D(R) Y(x)`` (tests/test_irrep_repr.py), and ``wigner_d_matrix`` / ``irr_repr`` / ``compose`` /
``spherical_harmonics`` then follow the reference's formulas.  ``Q_J`` -- unique up to sign -- is normalised so that its first
entry within 10 % of the largest magnitude is positive, and cached on disk like the reference's (``cache_dir``, tools/utils.py:151-206:
``$CACHE_PATH`` or ``~/.cache.equivariant_attention``; ``CLEAR_CACHE`` disables it)."""
from __future__ import annotations

import fcntl
import gzip
import os
import pickle
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


def _wigner_of_matrix(l, R):
    """D^l(R) for a proper rotation matrix R from its defining property Y_l(R x) = D^l(R) Y_l(x) (least squares, fp64)."""
    y = _host_sh(l, _DIRECTIONS)
    yr = _host_sh(l, _DIRECTIONS @ np.asarray(R, dtype=np.float64).T)
    return np.linalg.lstsq(y, yr, rcond=None)[0].T


def _wigner_np(l, a, b, c):
    return _wigner_of_matrix(l, _rot_np(a, b, c))


_J3 = np.array([[-1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, 1.0, 0.0]])     # half turn about (0, 1, 1)/sqrt(2): J3 Rz(b) J3 = Ry(b)


@lru_cache(maxsize=None)
def _j_np(l):
    j = _wigner_of_matrix(l, _J3)
    return 0.5 * (j + j.T)                       # symmetric and involutory up to round-off


def J_dense(degree, dtype=None, device=None):
    """The reference's ``Jd[degree]`` (tools/irr_repr.py:14-20): symmetric, ``J @ J = I``, ``J Z(b) J = D(0, b, 0)``."""
    return torch.from_numpy(_j_np(int(degree))).to(dtype=dtype or torch.get_default_dtype(), device=device)


def z_rot_mat(angle, l):
    """tools/irr_repr.py:33-42: cos(f angle) on the diagonal, sin(f angle) on the anti-diagonal, f = l, l-1, ..., -l."""
    angle = torch.as_tensor(angle, dtype=torch.get_default_dtype()) if not torch.is_tensor(angle) else angle
    order = to_order(l)
    m = angle.new_zeros((order, order))
    inds = torch.arange(0, order, 1, dtype=torch.long, device=angle.device)
    reversed_inds = torch.arange(2 * l, -1, -1, dtype=torch.long, device=angle.device)
    frequencies = torch.arange(l, -l - 1, -1, dtype=angle.dtype, device=angle.device)
    m[inds, reversed_inds] = torch.sin(frequencies * angle)
    m[inds, inds] = torch.cos(frequencies * angle)
    return m


def wigner_d_matrix(degree, alpha, beta, gamma, dtype=None, device=None):
    """tools/irr_repr.py:22-31: ``Z(alpha) J Z(beta) J Z(gamma)`` for ZYZ Euler angles."""
    dtype = dtype or torch.get_default_dtype()
    cast = lambda t: (t if torch.is_tensor(t) else torch.tensor(float(t), dtype=dtype)).to(dtype=dtype, device=device)  # noqa: E731
    J = J_dense(degree, dtype=dtype, device=device)
    return z_rot_mat(cast(alpha), degree) @ J @ z_rot_mat(cast(beta), degree) @ J @ z_rot_mat(cast(gamma), degree)


def irr_repr(order, alpha, beta, gamma, dtype=None):
    """Irreducible representation D^order(alpha, beta, gamma) of the ZYZ rotation ``rot`` (tools/irr_repr.py:44-52): through
    ``wigner_d_matrix``, as the reference computes it."""
    return wigner_d_matrix(order, alpha, beta, gamma, dtype=dtype or torch.get_default_dtype())


def rot_z(gamma):
    g = float(gamma)
    return torch.tensor([[np.cos(g), -np.sin(g), 0.0], [np.sin(g), np.cos(g), 0.0], [0.0, 0.0, 1.0]], dtype=torch.get_default_dtype())


def rot_y(beta):
    b = float(beta)
    return torch.tensor([[np.cos(b), 0.0, np.sin(b)], [0.0, 1.0, 0.0], [-np.sin(b), 0.0, np.cos(b)]], dtype=torch.get_default_dtype())


def rot(alpha, beta, gamma):
    """ZYZ Euler rotation Rz(alpha) Ry(beta) Rz(gamma) (tools/irr_repr.py:85-90)."""
    return torch.from_numpy(_rot_np(float(alpha), float(beta), float(gamma))).to(torch.get_default_dtype())


def x_to_alpha_beta(x):
    """Point on the sphere -> (alpha, beta) (tools/irr_repr.py:76-83)."""
    x = torch.as_tensor(x, dtype=torch.get_default_dtype())
    x = x / torch.norm(x)
    return torch.atan2(x[1], x[0]), torch.acos(x[2])


def compose(a1, b1, c1, a2, b2, c2):
    """(a, b, c) = (a1, b1, c1) composed with (a2, b2, c2) (tools/irr_repr.py:92-101)."""
    comp = rot(a1, b1, c1) @ rot(a2, b2, c2)
    xyz = comp @ torch.tensor([0.0, 0.0, 1.0], dtype=comp.dtype)
    a, b = x_to_alpha_beta(xyz)
    rotz = rot(0.0, -float(b), -float(a)) @ comp
    c = torch.atan2(rotz[1, 0], rotz[0, 0])
    return a, b, c


def spherical_harmonics(order, alpha, beta, dtype=None):
    """tools/irr_repr.py:103-104: ``get_spherical_harmonics(order, theta = pi - beta, phi = alpha)`` (device op: GPU tensors)."""
    return get_spherical_harmonics(order, theta=(pi - beta), phi=alpha)


# --------------------------------------------------------------------------- disk cache of the fp64 constants
def _cache_path():
    """``$CACHE_PATH`` or ``~/.cache.equivariant_attention``; None when ``CLEAR_CACHE`` is set (tools/basis.py:15-16)."""
    if os.environ.get("CLEAR_CACHE") is not None:
        return None
    return os.environ.get("CACHE_PATH") or os.path.expanduser("~/.cache.equivariant_attention")


def _cached_on_disk(key, compute):
    """The reference's ``cache_dir`` protocol (tools/utils.py:151-206): ``index.pkl`` maps keys to ``N.pkl.gz`` files, guarded by
    a lock file (``fcntl.flock`` here; the reference's ``filelock`` package is not a dependency)."""
    dirname = _cache_path()
    if dirname is None:
        return compute()
    try:
        os.makedirs(dirname, exist_ok=True)
        with open(os.path.join(dirname, "mutex"), "a+") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                indexfile = os.path.join(dirname, "index.pkl")
                index = {}
                if os.path.exists(indexfile):
                    with open(indexfile, "rb") as f:
                        index = pickle.load(f)
                if key in index:
                    filename = index[key]
                else:
                    index[key] = filename = f"{len(index)}.pkl.gz"
                    with open(indexfile, "wb") as f:
                        pickle.dump(index, f)
                filepath = os.path.join(dirname, filename)
                if os.path.exists(filepath):
                    with gzip.open(filepath, "rb") as f:
                        return pickle.load(f)
                result = compute()
                with gzip.open(filepath, "wb") as f:
                    pickle.dump(result, f)
                return result
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    except OSError:
        return compute()                          # unwritable cache directory: compute, as without a cache


_SYLVESTER_ANGLES = [(4.4, 5.6, 4.6), (4.9, 6.1, 4.1), (0.5, 4.1, 5.4), (2.2, 3.5, 5.6), (2.5, 0.3, 3.9)]


def _q_compute(J, order_in, order_out):
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


@lru_cache(maxsize=None)
def _q_np(J, order_in, order_out):
    """RAM cache (lru) over the disk cache, as the reference stacks ``lru_cache`` on ``cache_dir`` (tools/utils.py:158-160)."""
    return _cached_on_disk(("equigspr.Q_J.v1", int(J), int(order_in), int(order_out)), lambda: _q_compute(J, order_in, order_out))


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


def _host_basis_degree1(pts):
    """The degree <= 1 basis tensors of ``get_basis`` at host points, fp64 numpy (``_host_sh`` + ``_q_np``: the same constants the HIP
    basis kernel multiplies with): K00 [n], K01 [n, 3 (out)], K10 [n, 3 (in)], K11 [n, 3 (out), 3 (in), 3 (J)]."""
    pts = np.asarray(pts, dtype=np.float64)
    # tools/basis.py:76: the polar axis is input component 1 -- (x, y, z) -> (z, x, y) is what get_spherical_from_cartesian hands the SH code
    perm = pts[:, [2, 0, 1]]
    y = {l: _host_sh(l, perm) for l in range(3)}
    k = {}
    for d_in, d_out in product(range(2), range(2)):
        ks = [y[J] @ _q_np(J, d_in, d_out).T for J in range(abs(d_in - d_out), d_in + d_out + 1)]
        k[(d_in, d_out)] = np.stack(ks, axis=-1).reshape(len(pts), 2 * d_out + 1, 2 * d_in + 1, -1)
    return k[(0, 0)].reshape(-1), k[(0, 1)].reshape(-1, 3), k[(1, 0)].reshape(-1, 3), k[(1, 1)]


@lru_cache(maxsize=None)
def degree1_closed_form():
    """The six constants of the degree <= 1 basis in closed form (what ``csrc/irreps.hip`` evaluates in registers instead of
    reading ``get_basis`` tensors).  With u = r / |r| (tools/basis.py:153-205 restricted to max_degree = 1):

        K['0,0'] = k00                       K['0,1'][a] = k01 u_a                   K['1,0'][b] = k10 u_b
        K['1,1'][a, b, J=0] = k11_0 delta_ab     [.., J=1] = k11_1 eps_abc u_c     [.., J=2] = k11_2 (u_a u_b - delta_ab / 3)

    (Schur: the unique rotation-covariant tensors of each type; degree-1 components are Cartesian up to the factor of ``Y_1``.)  The
    constants -- signs included: ``Q_J`` is only defined up to sign -- are FITTED to the basis the package itself builds (host fp64:
    ``_host_sh`` and ``_q_np``) at a set of directions, and the fit must be exact to 1e-12, so the closed form can never drift from
    ``get_basis``.  Returns ``(k00, k01, k10, k11_0, k11_1, k11_2)`` as Python floats."""
    pts = np.random.default_rng(11).standard_normal((24, 3))
    u = pts / np.linalg.norm(pts, axis=1, keepdims=True)
    k00, k01, k10, k11 = _host_basis_degree1(pts)
    eps = np.zeros((3, 3, 3))
    eps[0, 1, 2] = eps[1, 2, 0] = eps[2, 0, 1] = 1.0
    eps[0, 2, 1] = eps[2, 1, 0] = eps[1, 0, 2] = -1.0
    eye = np.eye(3)
    forms = [(k00, np.ones(len(u))), (k01, u), (k10, u), (k11[..., 0], np.broadcast_to(eye, (len(u), 3, 3))),
             (k11[..., 1], np.einsum("abc,nc->nab", eps, u)), (k11[..., 2], np.einsum("na,nb->nab", u, u) - eye / 3.0)]
    out = []
    for got, form in forms:
        c = float((got * form).sum() / (form * form).sum())
        if np.abs(got - c * form).max() > 1e-12:
            raise RuntimeError("degree1_closed_form: the degree <= 1 basis does not have the expected closed form")
        out.append(c)
    return tuple(out)


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
