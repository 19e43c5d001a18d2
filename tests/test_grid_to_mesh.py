"""Forcing ingest: structured grid -> mesh nodes (the arrays M_wind / M_ocean / M_ssh of the hot path are filled by
InterpFromGridToMeshx, model/externaldata.cpp:1436).  The HIP gather kernel behind nxs_interp_grid_to_mesh against the REAL
contrib/bamg routine -- live when oracle/_ref is present, and through the committed fixture generated with it.  Bar: bit-exact
(the same double expressions in the same order), including the reference's own quirks (nearest-neighbour rule, NaN and
outside -> default)."""
import os
import sys

import numpy as np
import pytest

from oracle import pyoracle as O

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "bamg_grid_to_mesh.npz")
NAMES = ("bilinear", "triangle", "nearest", "contours_nan", "row_major")


@pytest.mark.skipif(O.bamg_shim() is None, reason="oracle/_ref (real contrib/bamg) not built here")
def test_real_bamg_reproduces_the_committed_fixture(capfd):
    z = np.load(GOLD)
    for name, (xs, ys, data, xm, ym, interp, rm) in make_golden.grid_to_mesh_cases().items():
        assert np.array_equal(O.bamg_interp_grid_to_mesh(xs, ys, data, xm, ym, 1e8, interp, rm), z[name]), name
    capfd.readouterr()   # the reference prints one line per node outside the grid


def test_fixture_is_sane():
    z = np.load(GOLD)
    c = make_golden.grid_to_mesh_cases()
    xs, ys, data, xm, ym, _, _ = c["bilinear"]
    lin = 1e-5 * xm - 2e-5 * ym
    inside = np.ones(xm.size, bool); inside[3] = False
    assert np.abs(z["bilinear"][inside, 0] - lin[inside]).max() < 1e-9 * np.abs(lin).max()      # bilinear / triangle reproduce a linear field
    assert np.abs(z["triangle"][inside, 0] - lin[inside]).max() < 1e-9 * np.abs(lin).max()
    assert np.all(z["bilinear"][3] == 1e8)                                                       # the node outside the grid
    assert abs(z["bilinear"][0, 0] - data[0, 0, 0]) < 1e-12 * abs(data[0, 0, 0])                 # the grid's corners: found, not "outside"
    assert abs(z["bilinear"][1, 0] - data[-1, -1, 0]) < 1e-12 * abs(data[-1, -1, 0])             # (last coordinate -> last interval)
    assert np.array_equal(z["row_major"], z["bilinear"])                                         # same field, the other memory order and ascending y
    assert (z["contours_nan"] == 1e8).any() and np.isfinite(z["contours_nan"]).all()             # NaN data -> default value
    assert np.isin(z["nearest"][inside, 2], data[:, :, 2]).all()                                 # nearest returns a grid value


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_forcing_interpolation_matches_real_bamg_fixture_bit_for_bit(name):
    from nextsim_amd.interp import InterpFromGridToMeshx
    z = np.load(GOLD)
    xs, ys, data, xm, ym, interp, rm = make_golden.grid_to_mesh_cases()[name]
    out = InterpFromGridToMeshx(xs, ys, data, xm, ym, 1e8, interp, rm)
    assert np.array_equal(out, z[name])


@pytest.mark.gpu
def test_gpu_forcing_interpolation_non_monotone_axis_and_refusals():
    """A non-monotone axis takes the reference's first-match scan; bad shapes are refused with the reference's messages."""
    from nextsim_amd.interp import InterpFromGridToMeshx
    rng = np.random.default_rng(0)
    xs = np.array([0., 1., 2., 1.5, 3.]); ys = np.array([0., 1., 2.])
    data = rng.random((3, 5, 1))
    xm = np.array([1.7, 0.5, 2.5, 3.0]); ym = np.array([0.5, 1.5, 2.0, 0.0])
    out = InterpFromGridToMeshx(xs, ys, data, xm, ym, -1., 1)
    if O.bamg_shim() is not None:
        assert np.array_equal(out, O.bamg_interp_grid_to_mesh(xs, ys, data, xm, ym, -1., 1))
    assert np.isfinite(out).all()
    with pytest.raises(Exception, match="length should be 1 or 0 more"):
        InterpFromGridToMeshx(np.arange(7.), ys, data, xm, ym)
    with pytest.raises(Exception, match="nothing to be done"):
        InterpFromGridToMeshx(np.arange(1.), np.arange(1.), np.zeros((1, 1, 1)), xm, ym)


@pytest.mark.gpu
@pytest.mark.skipif(O.bamg_shim() is None, reason="oracle/_ref (real contrib/bamg) not present on this box")
def test_gpu_forcing_interpolation_at_forcing_size(capfd):
    """A 0.25-degree-like grid (720 x 361) onto the 10 km mesh, 6 fields (two time levels of u, v, ssh): live comparison."""
    import cases
    from nextsim_amd.interp import InterpFromGridToMeshx
    gm = cases.global_mesh("10km")
    rng = np.random.default_rng(1)
    xs = np.linspace(gm.x.min() - 1e4, gm.x.max() + 1e4, 720); ys = np.linspace(gm.y.min() - 1e4, gm.y.max() + 1e4, 361)
    data = rng.normal(size=(361, 720, 6))
    out, info = InterpFromGridToMeshx(xs, ys, data, gm.x, gm.y, 1e8, 1, return_info=True)
    ref = O.bamg_interp_grid_to_mesh(xs, ys, data, gm.x, gm.y, 1e8, 1)
    capfd.readouterr()
    assert np.array_equal(out, ref)
