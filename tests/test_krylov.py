"""EXTENSION (parity unpinned): coloured CSR assembly + CG.  The only in-tree known
answer is the P1 Laplacian of research/laplacian.cpp (exact solution sin(pi x) sin(pi y), :348, :429)."""
import numpy as np
import pytest

from nextsim_amd import krylov


def unit_square(n):
    xs = np.linspace(0, 1, n + 1)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    nid = np.arange((n + 1) ** 2).reshape(n + 1, n + 1)
    a = nid[:-1, :-1].ravel(); b = nid[1:, :-1].ravel(); c = nid[1:, 1:].ravel(); d = nid[:-1, 1:].ravel()
    tri = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]) + 1
    x, y = X.ravel(), Y.ravel()
    bnd = (x == 0) | (x == 1) | (y == 0) | (y == 1)
    return tri.astype(np.int32), x, y, bnd.astype(np.uint8)


def test_csr_pattern_and_colouring():
    tri, x, y, bnd = unit_square(8)
    rowptr, colidx = krylov.csr_pattern(tri, x.size)
    assert rowptr[0] == 0 and rowptr[-1] == colidx.size
    A = np.zeros((x.size, x.size), bool)
    for r in range(x.size):
        cols = colidx[rowptr[r]:rowptr[r + 1]]
        assert np.all(np.diff(cols) > 0) and r in cols            # ascending, diagonal present
        A[r, cols] = True
    assert np.array_equal(A, A.T)                                  # structurally symmetric
    expect = np.zeros_like(A)
    for t in tri - 1:
        expect[np.ix_(t, t)] = True
    assert np.array_equal(A, expect)
    col, ncol = krylov.colour_elements(tri, x.size)
    assert ncol <= 12
    for c in range(ncol):                                          # no node twice inside a colour
        nodes = (tri[col == c] - 1).ravel()
        assert np.unique(nodes).size == nodes.size


@pytest.mark.gpu
def test_poisson_known_answer_and_second_order_convergence():
    errs = []
    for n in (32, 64, 128):
        tri, x, y, bnd = unit_square(n)
        xb = x[tri - 1].mean(1); yb = y[tri - 1].mean(1)
        f = 2 * np.pi ** 2 * np.sin(np.pi * xb) * np.sin(np.pi * yb)      # research/laplacian.cpp:218
        u, info = krylov.poisson_solve(tri, x, y, bnd, f, rtol=1e-11)
        assert info["rel_residual"] <= 1e-10 and info["iterations"] < 2000
        errs.append(np.abs(u - np.sin(np.pi * x) * np.sin(np.pi * y)).max())
    assert errs[0] < 5e-3
    assert 3.5 < errs[0] / errs[1] < 4.5 and 3.5 < errs[1] / errs[2] < 4.5   # O(h^2)


@pytest.mark.gpu
def test_poisson_on_the_arctic_mesh_is_deterministic():
    """Coloured scatter + two-stage dot: no atomics, two solves agree bit for bit."""
    import cases
    gm = cases.global_mesh("10km")
    tri = (gm.tri + 1).astype(np.int32)
    x, y = gm.x / 2.5e6, gm.y / 2.5e6
    bnd = (gm.dirichlet | gm.neumann).astype(np.uint8)
    f = np.ones(gm.num_elements)
    u1, i1 = krylov.poisson_solve(tri, x, y, bnd, f, rtol=1e-9)
    u2, i2 = krylov.poisson_solve(tri, x, y, bnd, f, rtol=1e-9)
    assert np.array_equal(u1, u2) and i1["iterations"] == i2["iterations"]
    assert u1.min() >= -1e-12 and u1.max() > 0 and np.all(u1[bnd.astype(bool)] == 0)   # maximum principle


def test_krylov_symbols_are_declared():
    import os, re
    from nextsim_amd import dynamics
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "nxs_krylov.h")).read()
    declared = set(re.findall(r"NXS_KRYLOV_API\s+(?:const\s+char\s*\*|int)\s*(nxs_\w+)\s*\(", text))
    assert declared == set(krylov.KRYLOV_EXPORTS)
    L = dynamics.load_library()
    for n in declared:
        assert hasattr(L, n)
