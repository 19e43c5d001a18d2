"""The connectivity tables the hot path reads (bamgmesh->NodalElementConnectivity / NodalConnectivity,
FE.cpp:10376-10379, 10578-10602): product code (nxs_mesh_connectivity) and the oracle's restatement
against the REAL contrib/bamg (oracle/_ref, built from /root/reference) and against fixtures that
were generated with the real bamg and committed (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

import cases
from nextsim_amd import dynamics, mesh as M
from oracle import pyoracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "bamg_connectivity.npz")


def _both(lm):
    a = O.connectivity(lm.indices, lm.num_nodes)
    b = dynamics.mesh_connectivity(lm.indices, lm.num_nodes)
    return a, b


@pytest.mark.parametrize("kind", ["tiny", "toy", "small"])
def test_product_tables_equal_oracle_tables(kind):
    lm = M.localize(cases.global_mesh(kind), 1)[0]
    (nec_o, nc_o), (nec_p, nc_p) = _both(lm)
    assert np.array_equal(nec_o, nec_p, equal_nan=True)
    assert np.array_equal(nc_o, nc_p)


@pytest.mark.skipif(O.bamg_shim() is None, reason="oracle/_ref (real contrib/bamg) not built here")
@pytest.mark.parametrize("kind,nparts", [("tiny", 1), ("toy", 1), ("small", 1), ("small", 3), ("40km", 2)])
def test_tables_equal_real_bamg(kind, nparts):
    """BamgConvertMeshx exactly as FE.cpp:77-80 calls it, on whole and on partitioned (ghosted) meshes."""
    for lm in M.localize(cases.global_mesh(kind), nparts):
        nec_b, nc_b = O.bamg_connectivity(lm.indices, lm.coord_x, lm.coord_y)
        (nec_o, nc_o), (nec_p, nc_p) = _both(lm)
        assert np.array_equal(nec_b, nec_o, equal_nan=True) and np.array_equal(nc_b, nc_o)
        assert np.array_equal(nec_b, nec_p, equal_nan=True) and np.array_equal(nc_b, nc_p)


def test_tables_equal_committed_real_bamg_fixture():
    z = np.load(GOLD)
    idx, nn = z["indices"], int(z["num_nodes"])
    for name, fn in (("oracle", O.connectivity), ("product", dynamics.mesh_connectivity)):
        nec, nc = fn(idx, nn)
        assert np.array_equal(nec, z["nec"], equal_nan=True), name
        assert np.array_equal(nc, z["nc"]), name


def test_row_order_is_what_the_summations_rely_on():
    """Fan rows are in DESCENDING element number, NaN padded (Mesh.cpp:804-811); neighbour rows hold
    `count` 1-based node ids then zeros, count in the last column (Mesh.cpp:850-865)."""
    lm = M.localize(cases.global_mesh("toy"), 1)[0]
    nec, nc = dynamics.mesh_connectivity(lm.indices, lm.num_nodes)
    tri = lm.indices.reshape(-1, 3)
    for n in (0, 7, lm.num_nodes // 2, lm.num_nodes - 1):
        row = nec[n][~np.isnan(nec[n])].astype(int)
        assert np.all(np.diff(row) < 0)
        assert set(row - 1) == set(np.flatnonzero((tri == n + 1).any(1)))
        cnt = int(nc[n, -1])
        nb = nc[n, :cnt].astype(int)
        expect = set(tri[(tri == n + 1).any(1)].ravel()) - {n + 1}
        assert set(nb) == expect and np.all(nc[n, cnt:-1] == 0)


def test_invalid_input_is_rejected():
    import ctypes as C
    from nextsim_amd import _abi
    L = dynamics.load_library()
    idx = np.array([1, 2, 9], np.int32)
    w1, w2 = C.c_int32(), C.c_int32()
    assert L.nxs_mesh_connectivity(_abi.iptr(idx), 3, 1, C.byref(w1), None, C.byref(w2), None) == -1
