"""Moorings sampling: nxs_interp_mesh_to_grid against the REAL contrib/bamg
InterpFromMeshToGridx, live and through tests/golden/bamg_mesh_to_grid.npz.  Bar: bit-exact everywhere
(same double expressions for the area coordinates, last matching element wins)."""
import os
import sys

import numpy as np
import pytest

from oracle import pyoracle as O

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "bamg_mesh_to_grid.npz")


@pytest.mark.skipif(O.bamg_shim() is None, reason="oracle/_ref (real contrib/bamg) not built here")
def test_real_bamg_reproduces_the_committed_grid_fixture():
    gm, idx, nodal, elemental, xmin, ymax, xp, yp, nrows, ncols = make_golden.grid_case()
    z = np.load(GOLD)
    assert np.array_equal(O.bamg_interp_mesh_to_grid(idx, gm.x, gm.y, nodal, xmin, ymax, xp, yp, nrows, ncols, -1e14), z["nodal"])
    assert np.array_equal(O.bamg_interp_mesh_to_grid(idx, gm.x, gm.y, elemental, xmin, ymax, xp, yp, nrows, ncols, -1e14), z["elemental"])


def test_grid_fixture_is_sane():
    z = np.load(GOLD)
    n = z["nodal"]
    inside = n[..., 0] != -1e14
    assert 0.5 < inside.mean() < 0.95                    # a disc in a slightly larger box
    assert np.abs(n[inside][:, 0]).max() <= 2.0 + 1e-12  # cos + sin
    assert not np.isnan(n).any()                         # NaN data were replaced by the default value
    # negative postings: the same points, stored in reversed order along both axes (InterpFromMeshToGridx.cpp:73-92)
    f = z["nodal_flipped"]
    assert np.allclose(f[::-1, ::-1, 0][inside], n[inside][:, 0], rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_gpu_mesh_to_grid_matches_real_bamg_fixture_bit_for_bit():
    from nextsim_amd.interp import InterpFromMeshToGridx
    gm, idx, nodal, elemental, xmin, ymax, xp, yp, nrows, ncols = make_golden.grid_case()
    z = np.load(GOLD)
    assert np.array_equal(InterpFromMeshToGridx(idx, gm.x, gm.y, nodal, xmin, ymax, xp, yp, nrows, ncols, -1e14), z["nodal"])
    assert np.array_equal(InterpFromMeshToGridx(idx, gm.x, gm.y, elemental, xmin, ymax, xp, yp, nrows, ncols, -1e14), z["elemental"])
    got = InterpFromMeshToGridx(idx, gm.x, gm.y, nodal, xmin, ymax, -xp, -yp, nrows, ncols, -1e14)
    assert np.array_equal(got, z["nodal_flipped"])


@pytest.mark.gpu
@pytest.mark.skipif(O.bamg_shim() is None, reason="oracle/_ref (real contrib/bamg) not present on this box")
def test_gpu_mesh_to_grid_matches_real_bamg_live_at_moorings_size():
    """10 km mesh -> 500 x 500 grid, 4 element variables (conc, thick, damage, ...): the Moorings shape."""
    import cases
    from nextsim_amd.interp import InterpFromMeshToGridx
    gm = cases.global_mesh("10km")
    rng = np.random.default_rng(4)
    data = rng.standard_normal((gm.num_elements, 4))
    idx = (gm.tri + 1).astype(np.int32).ravel()
    n = 500
    args = (gm.x.min(), gm.y.max(), (gm.x.max() - gm.x.min()) / (n - 1), (gm.y.max() - gm.y.min()) / (n - 1), n, n, -1e14)
    ref = O.bamg_interp_mesh_to_grid(idx, gm.x, gm.y, data, *args)
    got = InterpFromMeshToGridx(idx, gm.x, gm.y, data, *args)
    assert np.array_equal(got, ref)
