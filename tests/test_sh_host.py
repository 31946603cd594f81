"""CPU: the host-side (fp64 constant) part of A16 that round 1 left out -- ``wigner_d_matrix`` through ``J`` (tools/irr_repr.py:22-31),
``irr_repr`` on top of it, ``compose`` (:92-101) and the disk cache of ``Q_J`` (tools/basis.py:123, tools/utils.py:151-206).
The reference's ``J_dense`` blob is not in its repository, so ``J_l`` is rebuilt (sh.J_dense) and pinned by the properties the
reference's own tests state: ``Y(rot x) = D Y(x)`` (tests/test_irrep_repr.py:7-33) and ``(D_out (x) D_in) Q_J = Q_J D_J``
(tests/test_basis.py:11-15)."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import sh_oracle as so
from se3_equi_graph_registration_amd import sh

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("l", [0, 1, 2, 3, 4, 6])
def test_j_dense_is_the_y_z_exchange(l):
    J = sh.J_dense(l, dtype=torch.float64).numpy()
    n = 2 * l + 1
    assert J.shape == (n, n)
    np.testing.assert_allclose(J, J.T, atol=1e-12)
    np.testing.assert_allclose(J @ J, np.eye(n), atol=1e-10)
    for b in (0.3, 1.7, 2.9):               # J Z(b) J = D(0, b, 0): the y rotation in the real-SH basis
        Z = sh.z_rot_mat(torch.tensor(b, dtype=torch.float64), l).numpy()
        np.testing.assert_allclose(J @ Z @ J, so.irr_repr(l, 0.0, b, 0.0), atol=1e-7)


@pytest.mark.parametrize("l", [1, 2, 3, 5])
def test_wigner_d_matrix_is_a_representation_compatible_with_the_harmonics(l):
    g = np.random.default_rng(l)
    for _ in range(4):
        a, b, c = g.uniform(0, 2 * np.pi), g.uniform(0, np.pi), g.uniform(0, 2 * np.pi)
        D = sh.wigner_d_matrix(l, a, b, c, dtype=torch.float64).numpy()
        np.testing.assert_allclose(D, so.irr_repr(l, a, b, c), atol=1e-7)              # the oracle's least-squares Wigner matrix
        np.testing.assert_allclose(D @ D.T, np.eye(2 * l + 1), atol=1e-7)
        # tests/test_irrep_repr.py: Y(rot(a, b, c) x) = D Y(x)
        x = g.standard_normal((7, 3))
        R = sh.rot(a, b, c).double().numpy()
        y, yr = so._sh_of_points(l, x), so._sh_of_points(l, x @ R.T)
        np.testing.assert_allclose(yr, y @ D.T, atol=1e-7)
        assert torch.allclose(sh.irr_repr(l, a, b, c, dtype=torch.float64), torch.from_numpy(D))
    # homomorphism through compose (tools/irr_repr.py:92-101)
    a1, b1, c1, a2, b2, c2 = g.uniform(0.2, 2.8, size=6)
    a, b, c = sh.compose(a1, b1, c1, a2, b2, c2)
    torch.set_default_dtype(torch.float64)
    try:
        a, b, c = sh.compose(a1, b1, c1, a2, b2, c2)
        np.testing.assert_allclose(sh.rot(float(a), float(b), float(c)).numpy(), (sh.rot(a1, b1, c1) @ sh.rot(a2, b2, c2)).numpy(), atol=1e-7)
        D12 = sh.wigner_d_matrix(l, a1, b1, c1) @ sh.wigner_d_matrix(l, a2, b2, c2)
        np.testing.assert_allclose(sh.wigner_d_matrix(l, float(a), float(b), float(c)).numpy(), D12.numpy(), atol=1e-6)
    finally:
        torch.set_default_dtype(torch.float32)


def test_q_j_intertwines_and_is_cached_on_disk(tmp_path):
    """tests/test_basis.py:11-15 of the reference: (D_out (x) D_in) Q_J = Q_J D_J; and the cache_dir protocol: a second process
    finds the constants on disk (index.pkl + N.pkl.gz), CLEAR_CACHE bypasses the directory."""
    for (j, d_in, d_out) in ((1, 1, 1), (2, 1, 1), (1, 0, 1), (2, 2, 1), (3, 2, 1)):
        q = sh.basis_transformation_Q_J(j, d_in, d_out).double().numpy()
        a, b, c = 0.4, 1.3, 2.2
        lhs = np.kron(so.irr_repr(d_out, a, b, c), so.irr_repr(d_in, a, b, c)) @ q
        np.testing.assert_allclose(lhs, q @ so.irr_repr(j, a, b, c), atol=2e-6)
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from se3_equi_graph_registration_amd import sh\n"
            "t0 = time.perf_counter(); q = sh.basis_transformation_Q_J(2, 1, 1); print('T', time.perf_counter() - t0, float(q.abs().sum()))\n" % ROOT)
    env = dict(os.environ, CACHE_PATH=str(tmp_path / "cache"))
    env.pop("CLEAR_CACHE", None)
    first = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert first.returncode == 0, first.stderr
    files = sorted(os.listdir(tmp_path / "cache"))
    assert "index.pkl" in files and any(f.endswith(".pkl.gz") for f in files) and "mutex" in files
    index = pickle.load(open(tmp_path / "cache" / "index.pkl", "rb"))
    assert ("equigspr.Q_J.v1", 2, 1, 1) in index
    second = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert second.returncode == 0 and first.stdout.split()[2] == second.stdout.split()[2]            # the same constants from disk
    n_before = len(os.listdir(tmp_path / "cache"))
    third = subprocess.run([sys.executable, "-c", code.replace("(2, 1, 1)", "(1, 1, 0)")],
                           env=dict(env, CLEAR_CACHE="1", CACHE_PATH=str(tmp_path / "cache")), capture_output=True, text=True, timeout=300)
    assert third.returncode == 0 and len(os.listdir(tmp_path / "cache")) == n_before                  # CLEAR_CACHE: nothing written


def test_degree1_closed_form_matches_the_oracle_basis():
    """The six constants the N4 kernels evaluate in registers (sh.degree1_closed_form, fitted to the package's own host basis) reproduce
    the ORACLE's get_basis tensors of degrees <= 1 (tools/basis.py:153-205 restated) at random directions: K00, K01, K10 and the three
    J-components of K11 in closed form."""
    import numpy as np
    import torch
    from oracle import sh_oracle as so
    from se3_equi_graph_registration_amd import sh
    k00, k01, k10, k110, k111, k112 = sh.degree1_closed_form()
    pts = np.random.default_rng(3).standard_normal((40, 3))
    u = pts / np.linalg.norm(pts, axis=1, keepdims=True)
    K = so.get_basis(torch.from_numpy(pts), 1)
    eps = np.zeros((3, 3, 3))
    eps[0, 1, 2] = eps[1, 2, 0] = eps[2, 0, 1] = 1.0
    eps[0, 2, 1] = eps[2, 1, 0] = eps[1, 0, 2] = -1.0
    k11 = K["1,1"].reshape(-1, 3, 3, 3).numpy()
    assert np.abs(K["0,0"].reshape(-1).numpy() - k00).max() < 1e-12
    assert np.abs(K["0,1"].reshape(-1, 3).numpy() - k01 * u).max() < 1e-12
    assert np.abs(K["1,0"].reshape(-1, 3).numpy() - k10 * u).max() < 1e-12
    assert np.abs(k11[..., 0] - k110 * np.eye(3)).max() < 1e-12
    assert np.abs(k11[..., 1] - k111 * np.einsum("abc,nc->nab", eps, u)).max() < 1e-12
    assert np.abs(k11[..., 2] - k112 * (np.einsum("na,nb->nab", u, u) - np.eye(3) / 3.0)).max() < 1e-12
    # Y_1(x) = c x / |x| with the same constant up to the sign convention of Q: |k01| = |k00| = 1 / sqrt(4 pi)
    assert abs(abs(k01) - (4 * np.pi) ** -0.5) < 1e-12 and abs(k00 - (4 * np.pi) ** -0.5) < 1e-12
