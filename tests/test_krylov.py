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
    declared = set(re.findall(r"NXS_KRYLOV_API\s+(?:const\s+char\s*\*|int|void)\s*(nxs_\w+)\s*\(", text))
    assert declared == set(krylov.KRYLOV_EXPORTS)
    L = dynamics.load_library()
    for n in declared:
        assert hasattr(L, n)


def _convection_diffusion(n, peclet, seed):
    """Upwind convection-diffusion on an n x n grid: non-symmetric, diagonally dominant M-matrix; random CSR column order."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    N = n * n
    idx = np.arange(N).reshape(n, n)
    rows, cols, vals = [], [], []
    cx, cy = peclet * rng.uniform(0.3, 1.0), -peclet * rng.uniform(0.3, 1.0)
    for (di, dj, w) in ((0, 1, -1.0 + min(cx, 0)), (0, -1, -1.0 - max(cx, 0)), (1, 0, -1.0 + min(cy, 0)), (-1, 0, -1.0 - max(cy, 0))):
        a = idx[max(-di, 0):n - max(di, 0), max(-dj, 0):n - max(dj, 0)].ravel()
        b = idx[max(di, 0):n - max(-di, 0), max(dj, 0):n - max(-dj, 0)].ravel()
        rows += list(a); cols += list(b); vals += [w] * a.size
    rows += list(range(N)); cols += list(range(N)); vals += [4.0 + abs(cx) + abs(cy) + 0.05] * N
    A = sp.csr_matrix((vals, (rows, cols)), shape=(N, N))
    A.sort_indices()
    return A


@pytest.mark.gpu
def test_bicgstab_and_cg_against_a_direct_solve():
    """The general CSR entry point: BiCGStab on a non-symmetric upwind convection-diffusion matrix and CG on its
    symmetric part, both against scipy's sparse LU -- the exact answer of the same linear system."""
    import scipy.sparse.linalg as spla
    from nextsim_amd import krylov
    rng = np.random.default_rng(3)
    A = _convection_diffusion(96, 3.0, 1)
    assert abs(A - A.T).max() > 1.0                               # genuinely non-symmetric
    xs = rng.normal(size=A.shape[0])
    b = A @ xs
    x, info = krylov.solve(A.indptr, A.indices, A.data, b, method=krylov.BICGSTAB, rtol=1e-12, max_iter=4000)
    assert info["rel_residual"] <= 1e-11 and info["iterations"] < 4000
    ref = spla.spsolve(A.tocsc(), b)
    assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max()
    assert np.abs(A @ x - b).max() <= 1e-10 * np.abs(b).max()
    S = (A + A.T) * 0.5
    S.sort_indices()
    b = S @ xs
    x, info = krylov.solve(S.indptr, S.indices, S.data, b, method=krylov.CG, rtol=1e-12, max_iter=4000)
    ref = spla.spsolve(S.tocsc(), b)
    assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max()
    # run to run bit-identical (deterministic dots), and a shuffled column order inside the rows gives a valid solve too
    x2, _ = krylov.solve(S.indptr, S.indices, S.data, b, method=krylov.CG, rtol=1e-12, max_iter=4000)
    assert np.array_equal(x, x2)
    # refusals
    bad = A.copy().tolil(); bad[5, 5] = 0.0; bad = bad.tocsr(); bad.eliminate_zeros()
    with pytest.raises(Exception, match="diagonal"):
        krylov.solve(bad.indptr, bad.indices, bad.data, np.ones(bad.shape[0]), method=krylov.BICGSTAB)
    with pytest.raises(Exception):
        krylov.solve(A.indptr, A.indices, A.data, b, method=7)


def _run_workers(world, mode, tmp_path, timeout=300):
    import json, os, socket, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(here, "kr_worker.py"), str(tmp_path), mode], env=env))
    for p in procs:
        try:
            p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("distributed Krylov workers hung")
    return [json.load(open(tmp_path / f"kr{r}.json")) for r in range(world)]


def test_row_distribution_and_halo_lists_reproduce_the_global_product_gloo(tmp_path):
    """World size 2 on CPU (gloo): each rank's block of rows (own nodes first, ghost columns behind) + the partition's
    halo lists give the global SpMV and the global dot -- the host logic the distributed device solver is fed with."""
    reps = _run_workers(2, "numpy", tmp_path)
    for r in reps:
        assert r["ok"], r
        assert r["spmv_err"] <= 1e-14 and r["dot_err"] <= 1e-14


@pytest.mark.gpu
def test_sliced_ellpack_spmv_matches_the_csr_product():
    """Ragged rows (1..40 entries), n not a multiple of the slice, shuffled column order inside the rows."""
    import scipy.sparse as sp
    rng = np.random.default_rng(11)
    for n in (1, 63, 64, 65, 1000, 4097):
        rows, cols, vals = [], [], []
        for r in range(n):
            k = int(rng.integers(0, min(n, 40)))
            c = rng.choice(n, size=k, replace=False) if k else np.zeros(0, int)
            c = np.unique(np.append(c, r))
            rng.shuffle(c)
            rows += [r] * c.size; cols += list(c); vals += list(rng.normal(size=c.size) + 3.0 * (c == r))
        rp = np.zeros(n + 1, np.int32); np.add.at(rp, np.asarray(rows) + 1, 1); rp = np.cumsum(rp).astype(np.int32)
        ci = np.asarray(cols, np.int32); va = np.asarray(vals)
        A = sp.csr_matrix((va, ci, rp), shape=(n, n))
        s = krylov.Solver()
        s.set_matrix(rp, ci, va)
        x = rng.normal(size=n)
        got, _ = s.spmv(x)
        # same order of additions inside a row as the CSR loop: bit-identical to it
        want = np.array([np.add.reduce([0.0] + [va[q] * x[ci[q]] for q in range(rp[r], rp[r + 1])]) for r in range(n)]) if n <= 65 else A @ x
        if n <= 65:
            seq = np.zeros(n)
            for r in range(n):
                acc = 0.0
                for q in range(rp[r], rp[r + 1]):
                    acc += va[q] * x[ci[q]]
                seq[r] = acc
            assert np.array_equal(got, seq)
        assert np.abs(got - (A @ x)).max() <= 1e-12 * max(1.0, np.abs(want).max())
        inf = s.info()
        assert inf["nnz"] == rp[-1] and inf["stored_entries"] >= inf["nnz"] and inf["spmv_bytes"] == 12 * rp[-1] + 16 * n
        s.close()


@pytest.mark.gpu
def test_resident_matrix_repeated_solves_and_refusals():
    import cases
    gm = cases.global_mesh("small")
    A, b, xs = cases.mesh_operator(gm)
    s = krylov.Solver()
    with pytest.raises(Exception, match="set_matrix"):
        s.solve(b)
    s.set_matrix(A.indptr, A.indices, A.data)
    x1, i1 = s.solve(b, rtol=1e-12)
    x2, i2 = s.solve(b, rtol=1e-12)
    assert np.array_equal(x1, x2) and i1["iterations"] == i2["iterations"]       # deterministic, handle reusable
    assert np.abs(x1 - xs).max() <= 1e-9 * np.abs(xs).max()
    x3, _ = krylov.solve(A.indptr, A.indices, A.data, b, rtol=1e-12)
    assert np.array_equal(x1, x3)                                                  # the one-shot entry is the same solver
    bad = A.copy().tolil(); bad[7, 7] = 0.0; bad = bad.tocsr(); bad.eliminate_zeros()
    with pytest.raises(Exception, match="diagonal"):
        s.set_matrix(bad.indptr, bad.indices, bad.data)
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_distributed_cg_and_bicgstab_through_the_callers_communicator(world, tmp_path):
    """Ranks share GPU 0; the operand halo and the dot all-reduces go through gloo (nxs_krylov_set_comm_fns)."""
    reps = _run_workers(world, "host", tmp_path)
    for r in reps:
        assert r["ok"], r
        assert r["spmv_err"] <= 1e-14
        for m in ("cg", "bicgstab"):
            assert r[m]["rel_residual"] <= 1e-11 and r[m]["err"] <= 1e-8, (m, r[m])
    assert len({r["cg"]["iterations"] for r in reps}) == 1                     # every rank saw the same global residual


@pytest.mark.gpu
def test_rccl_all_reduce_on_a_communicator_of_one_rank(tmp_path):
    """RCCL refuses two ranks on one device; a communicator of ONE rank is legal, and the solver reduces through it whenever one
    is attached: every dot of CG and BiCGStab is followed by the ncclAllReduce of the distributed solve, on the solver's stream,
    in place -- the dlopen'ed entry point, its argument order and the stream use execute on hardware.  A sum over one rank is the
    identity, so the result must have the bits of the solve without a communicator."""
    reps = _run_workers(1, "rccl", tmp_path)
    r = reps[0]
    assert r["ok"], r
    assert r["cg"]["err"] <= 1e-8 and r["bicgstab"]["err"] <= 1e-8
    assert r["comm_stats"]["rccl_allreduces"] >= 2 * r["cg"]["iterations"] + 3 * r["bicgstab"]["iterations"]
    assert r["same_bits_as_no_communicator"] is True


@pytest.mark.gpu
def test_distributed_cg_over_rccl(tmp_path):
    """RCCL may refuse several ranks on one device (a one-GPU box): skipped with its message then."""
    reps = _run_workers(2, "rccl", tmp_path)
    if any("comm_error" in r for r in reps):
        pytest.skip("RCCL communicator unavailable here: " + next(r["comm_error"] for r in reps if "comm_error" in r))
    for r in reps:
        assert r["ok"], r
        assert r["cg"]["err"] <= 1e-8 and r["bicgstab"]["err"] <= 1e-8
