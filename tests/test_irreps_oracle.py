"""CPU: the DEFINITION of the typed irrep-basis layer (oracle/irreps_oracle.py) pinned by the property it
exists for -- degree-0 outputs invariant, degree-1 outputs covariant under rotations of the cloud -- on graphs with self loops (where
the J >= 1 kernels must vanish), hubs and isolated nodes; and the structure of the degree <= 1 basis the layer relies on."""
import numpy as np
import pytest
import torch

from oracle import irreps_oracle as io, sh_oracle as so


def _setup(seed, n, e, c0i, c1i, c0o, c1o, hid=16):
    g = torch.Generator().manual_seed(seed)
    w = {k: torch.randn(*s, generator=g, dtype=torch.float64) * 0.5 for k, s in io.param_shapes(c0i, c1i, c0o, c1o, hid).items()}
    x = torch.randn(n, 3, generator=g, dtype=torch.float64)
    row = torch.randint(0, n - 3, (e,), generator=g); col = torch.randint(0, n - 3, (e,), generator=g)      # the last 3 nodes stay isolated
    row[:20] = 2                                         # a hub
    col[20:28] = row[20:28]                              # self loops
    f0 = torch.randn(n, c0i, generator=g, dtype=torch.float64)
    f1 = torch.randn(n, c1i, 3, generator=g, dtype=torch.float64)
    return w, f0, f1, x, [row, col]


def test_sh1_matrix_is_a_multiple_of_the_identity():
    a = io.sh1_matrix()
    assert torch.allclose(a, -np.sqrt(3 / (4 * np.pi)) * torch.eye(3, dtype=torch.float64), atol=1e-12)


@pytest.mark.parametrize("dims", [(5, 3, 4, 6), (8, 8, 8, 8), (1, 1, 1, 1)])
def test_layer_is_so3_equivariant(dims):
    w, f0, f1, x, edges = _setup(3, 40, 300, *dims)
    a = io.sh1_matrix()
    o0, o1 = io.irrep_conv(w, f0, f1, x, edges)
    assert float(o1.abs().max()) > 0.1 and float(o0.abs().max()) > 0.1
    for angles in ((0.4, 1.1, 2.3), (2.9, 0.2, 5.1)):
        rm = torch.from_numpy(so._rot(*angles))
        d1 = a @ rm @ torch.linalg.inv(a)
        p0, p1 = io.irrep_conv(w, f0, torch.einsum("ab,nib->nia", d1, f1), x @ rm.T, edges)
        assert float((p0 - o0).abs().max()) < 1e-11
        assert float((p1 - torch.einsum("ab,nob->noa", d1, o1)).abs().max()) < 1e-11
    # translations leave everything unchanged (only differences of positions enter)
    q0, q1 = io.irrep_conv(w, f0, f1, x + torch.tensor([0.3, -2.0, 1.5], dtype=torch.float64), edges)
    assert float((q0 - o0).abs().max()) < 1e-11 and float((q1 - o1).abs().max()) < 1e-11


def test_every_path_contributes_and_self_loops_use_j0_only():
    w, f0, f1, x, edges = _setup(5, 30, 200, 4, 4, 4, 4)
    base0, base1 = io.irrep_conv(w, f0, f1, x, edges)
    sizes = [16, 16, 16, 48]
    for path in range(4):                                 # zeroing a path's radial weights changes the outputs it feeds
        w2 = {k: v.clone() for k, v in w.items()}
        o = sum(sizes[:path])
        w2["radial.2.weight"][o:o + sizes[path]] = 0; w2["radial.2.bias"][o:o + sizes[path]] = 0
        a0, a1 = io.irrep_conv(w2, f0, f1, x, edges)
        changed0, changed1 = float((a0 - base0).abs().max()) > 1e-6, float((a1 - base1).abs().max()) > 1e-6
        assert (changed0, changed1) == ((True, False) if path in (0, 2) else (False, True)), path
    # a graph of self loops only: degree 1 sees the J = 0 (identity) kernel and the self-interaction, nothing directional
    n = 6
    loops = [torch.arange(n), torch.arange(n)]
    l0, l1 = io.irrep_conv(w, f0[:n], f1[:n], x[:n], loops)
    rm = torch.from_numpy(so._rot(1.0, 2.0, 0.5))
    m0, m1 = io.irrep_conv(w, f0[:n], torch.einsum("ab,nib->nia", rm, f1[:n]), x[:n] @ rm.T, loops)
    assert float((m0 - l0).abs().max()) < 1e-12 and float((m1 - torch.einsum("ab,nob->noa", rm, l1)).abs().max()) < 1e-12
