"""Regrid interpolation (BASELINE config 5): the HIP gather kernel behind
nxs_interp_mesh_to_mesh_2d against the REAL contrib/bamg InterpFromMeshToMesh2dx -- live when oracle/_ref
is present, and through the committed fixture tests/golden/bamg_interp.npz generated with it.

Bar: bit-exact for every target point that lies in the data mesh (the weights are ratios of the same
64-bit integer determinants, combined in the same order); default value outside when isdefault.  With
isdefault == false (the regrid call) the reference works on bamg's RECONSTRUCTED mesh -- fill triangles
between the boundary and the convex hull and inside holes, hull projection beyond (CloseBoundaryEdge): the
same completion is rebuilt here (csrc/nxs_hull.inl, pinned against the real bamg on five meshes) and
exterior points get the reference's value to the last bit or -- inside a fill triangle, whose three
vertices bamg may hold in a rotated order -- to one unit in the last place."""
import os
import sys

import numpy as np
import pytest

import cases
from oracle import pyoracle as O

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "bamg_interp.npz")


@pytest.mark.skipif(O.bamg_shim() is None, reason="oracle/_ref (real contrib/bamg) not built here")
def test_real_bamg_reproduces_the_committed_fixture():
    gm, idx, nodal, elemental, xi, yi, kind = make_golden.interp_case()
    z = np.load(GOLD)
    assert np.array_equal(z["xi"], xi) and np.array_equal(z["yi"], yi)
    assert np.array_equal(O.bamg_interp_mesh_to_mesh(idx, gm.x, gm.y, nodal, xi, yi, False), z["nodal"])
    assert np.array_equal(O.bamg_interp_mesh_to_mesh(idx, gm.x, gm.y, nodal, xi, yi, True, -999.0), z["nodal_default"])
    ins = kind == 0
    assert np.array_equal(O.bamg_interp_mesh_to_mesh(idx, gm.x, gm.y, elemental, xi[ins], yi[ins], False), z["elemental"])


def _completion_sets(x, y, tri, mode=0):
    from nextsim_amd.interp import convex_completion
    fill, hull = convex_completion((tri + 1).ravel(), x, y, mode)
    return ({tuple(sorted((r - 1).tolist())) for r in fill}, {(int(a) - 1, int(b) - 1) for a, b in hull}, fill)


def _only_cocircular_ties(x, y, mine, want):
    """True when the two triangle sets differ only in quadrilaterals whose four vertices are EXACTLY cocircular in bamg's integer plane: both
    diagonals are Delaunay there and which one bamg takes depends on the order of its point insertions."""
    a, b = mine - want, want - mine
    if len(a) != len(b):
        return False
    px0, px1, py0, py1 = x.min(), x.max(), y.min(), y.max()
    dx, dy = (px1 - px0) * 0.05, (py1 - py0) * 0.05
    coef = 1073741823. / max(px1 - px0 + 2 * dx, py1 - py0 + 2 * dy)
    ix = [int(v) for v in (coef * (x - (px0 - dx)))]; iy = [int(v) for v in (coef * (y - (py0 - dy)))]
    left = set(b)
    for t1 in a:
        mate = [t2 for t2 in a if t2 != t1 and len(set(t1) & set(t2)) == 2]
        ok = False
        for t2 in mate:
            quad = sorted(set(t1) | set(t2))
            other = [t for t in left if set(t) <= set(quad)]
            if len(quad) == 4 and len(other) == 2:
                p, q, r, d = quad
                m = [[ix[v] - ix[d], iy[v] - iy[d], (ix[v] - ix[d]) ** 2 + (iy[v] - iy[d]) ** 2] for v in (p, q, r)]
                det = (m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
                       + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]))
                ok = ok or det == 0
        if not ok:
            return False
    return True


@pytest.mark.parametrize("mode", [0, 1])
def test_convex_completion_equals_what_the_real_bamg_builds(mode):
    """Host code of the product (csrc/nxs_hull.inl) against the committed fixture made with the REAL bamg: the triangles
    ReconstructExistingMesh adds (Mesh.cpp:3135-3440) and its hull -- on a disc with an irregular coast, the same with two islands cut
    out, both at two resolutions, the toy box, AND (round 3) what is not "one outer loop with holes": two and three components, a lake
    inside an island, two holes meeting at a vertex, a hole touching the coast at a vertex.  mode 0: the pocket construction where it
    applies, else the general one (the constrained Delaunay triangulation of the boundary vertices minus the domain); mode 1: the general
    one everywhere.  Identical sets, up to the diagonal of a quadrilateral that is exactly cocircular in bamg's integer plane."""
    z = np.load(os.path.join(os.path.dirname(GOLD), "bamg_completion.npz"))
    ties = 0
    for name, (x, y, tri) in make_golden.completion_cases().items():
        mine_fill, mine_hull, fill = _completion_sets(x, y, tri, mode)
        want = {tuple(r) for r in z[name + "_fill"].tolist()}
        if mine_fill != want:
            assert _only_cocircular_ties(x, y, mine_fill, want), name
            ties += len(mine_fill - want) // 2
        assert mine_hull == {tuple(r) for r in z[name + "_hull"].tolist()}, name
        f = fill - 1
        jac = (x[f[:, 1]] - x[f[:, 0]]) * (y[f[:, 2]] - y[f[:, 0]]) - (x[f[:, 2]] - x[f[:, 0]]) * (y[f[:, 1]] - y[f[:, 0]])
        assert (jac > -1e-2).all() and (jac > 1.).mean() > 0.8, name      # (a straight piece of coast gives zero-area fill triangles, in bamg too)
    assert ties <= 3
    assert z["small_holes_fill"].shape[0] > z["small_fill"].shape[0] > 50       # the islands are filled too
    assert z["two_components_fill"].shape[0] > z["small_fill"].shape[0]          # ... and so is the gap between two components


@pytest.mark.skipif(O.bamg_shim() is None, reason="oracle/_ref (real contrib/bamg) not built here")
@pytest.mark.parametrize("mode", [0, 1])
def test_convex_completion_equals_the_real_bamg_live_at_10km(mode):
    gm = cases.global_mesh("10km")
    T, reft = O.bamg_completed_mesh((gm.tri + 1).ravel(), gm.x, gm.y)
    extra = T[gm.num_elements:]; inf = (extra < 0).any(1)
    mine_fill, mine_hull, _ = _completion_sets(gm.x, gm.y, gm.tri, mode)
    want = {tuple(sorted(r)) for r in extra[~inf].tolist()}
    assert (mine_fill == want or _only_cocircular_ties(gm.x, gm.y, mine_fill, want)) and len(mine_hull) == int(inf.sum())


def test_convex_completion_covers_components_and_pinches_and_refuses_what_is_no_mesh():
    """Round 2 refused several components and pinching boundaries (the regrid interpolation then fell back to "nearest boundary edge" outside
    the mesh); the general construction covers them.  What is still refused is what is no triangulation at all."""
    from nextsim_amd.interp import convex_completion
    from nextsim_amd.dynamics import NxsError
    x = np.array([0., 1., 0., 5., 6., 5.]); y = np.array([0., 0., 1., 0., 0., 1.])
    fill, hull = convex_completion(np.array([1, 2, 3, 4, 5, 6]), x, y)          # two components: the gap between them is filled
    assert fill.shape[0] == 2 and hull.shape[0] == 6                             # (the four vertices on y = 0 are all hull vertices: three collinear hull edges)
    with pytest.raises(NxsError, match="several outer boundary loops"):
        convex_completion(np.array([1, 2, 3, 4, 5, 6]), x, y, mode=2)            # (the pocket construction alone still says why it cannot)
    x = np.array([0., 1., 0., -1., 0.]); y = np.array([0., 0., 1., 0., -1.])
    fill, hull = convex_completion(np.array([1, 2, 3, 1, 4, 5]), x, y)          # two triangles touching at one vertex
    assert fill.shape[0] == 2 and hull.shape[0] == 4
    x = np.array([0., 1., 0., 1., 0.5]); y = np.array([0., 0., 1., 1., -1.])
    with pytest.raises(NxsError):
        convex_completion(np.array([1, 2, 3, 2, 1, 4, 1, 2, 5]), x, y)           # one edge in three triangles


def test_fixture_is_sane():
    """P1 interpolation of a linear field is exact up to bamg's integer-coordinate quantisation (~5 mm)."""
    gm, idx, nodal, elemental, xi, yi, kind = make_golden.interp_case()
    z = np.load(GOLD)
    inside = kind < 2
    lin = 1e-3 * xi - 2e-3 * yi
    assert np.abs(z["nodal"][inside, 1] - lin[inside]).max() < 1e-4
    assert np.all(z["nodal_default"][kind >= 2] == -999.0)
    assert np.array_equal(z["nodal_default"][kind == 0], z["nodal"][kind == 0])
    # a BOUNDARY vertex is a corner of hull-filling triangles too; when the reference's walk ends in one of
    # those it returns the default although the point is a node of the mesh (walk-dependent tie)
    tie = (kind == 1) & np.all(z["nodal_default"] == -999.0, axis=1)
    assert 0 < tie.sum() < 0.2 * (kind == 1).sum()
    assert np.array_equal(z["nodal_default"][(kind == 1) & ~tie], z["nodal"][(kind == 1) & ~tie])


@pytest.mark.gpu
def test_gpu_interpolation_matches_real_bamg_fixture_bit_for_bit():
    from nextsim_amd.interp import InterpFromMeshToMesh2dx
    gm, idx, nodal, elemental, xi, yi, kind = make_golden.interp_case()
    z = np.load(GOLD)
    inside = kind < 2
    out, info = InterpFromMeshToMesh2dx(idx, gm.x, gm.y, nodal, xi, yi, False, return_info=True)
    assert np.array_equal(out[inside], z["nodal"][inside])
    assert info["num_exterior"] == int((kind >= 2).sum())
    # points 50 m outside the boundary (a mesh that moved a little) and far outside, isdefault == false: the reference interpolates
    # inside bamg's fill triangles where the boundary is concave and projects on the hull (CloseBoundaryEdge) beyond it.  Same
    # completion here, so: the reference's bits wherever the triangle behind the point is a mesh triangle, and at most one unit in
    # the last place inside a fill triangle (same three products, possibly added in a rotated order).
    ext = kind >= 2
    assert info["completion_refused"] is None and info["num_stand_in"] == 0
    assert info["num_in_fill"] + info["num_on_hull"] == int(ext.sum()) and info["num_in_fill"] > 20 and info["num_on_hull"] > 20
    scale = np.abs(nodal).max(0)
    rel = (np.abs(out[ext] - z["nodal"][ext]) / scale).max(1)
    assert rel.max() <= 4.5e-16, rel.max()
    exact = np.all(out[ext] == z["nodal"][ext], axis=1)
    assert exact.mean() > 0.5, exact.mean()
    print(f"exterior points: {int(ext.sum())}, bit-identical {exact.mean():.3f}, worst {rel.max():.1e} of the field scale")
    outd = InterpFromMeshToMesh2dx(idx, gm.x, gm.y, nodal, xi, yi, True, -999.0)
    tie = (kind == 1) & np.all(z["nodal_default"] == -999.0, axis=1)   # see test_fixture_is_sane
    assert np.array_equal(outd[~tie], z["nodal_default"][~tie])
    assert np.array_equal(outd[tie], z["nodal"][tie])                  # here: the node's own value
    generic = kind == 0   # a vertex belongs to several triangles: the P0 value there depends on the walk
    oute = InterpFromMeshToMesh2dx(idx, gm.x, gm.y, elemental, xi[generic], yi[generic], False)
    assert np.array_equal(oute, z["elemental"])


@pytest.mark.gpu
@pytest.mark.skipif(O.bamg_shim() is None, reason="oracle/_ref (real contrib/bamg) not present on this box")
def test_gpu_interpolation_matches_real_bamg_live_regrid_shape():
    """The FE.cpp:3131 call shape: 6 nodal variables (M_VT, M_UM, M_UT) from a displaced old mesh onto the
    nodes of a new, differently sized mesh of the same domain; default-value mode."""
    from nextsim_amd.interp import InterpFromMeshToMesh2dx
    from nextsim_amd import mesh as M
    old = cases.global_mesh("40km")
    new = M.make_disc_mesh(61e3, seed=5, name="new")
    rng = np.random.default_rng(2)
    um = 300.0 * rng.standard_normal((2, old.num_nodes))
    xo, yo = old.x + um[0], old.y + um[1]
    data = rng.standard_normal((old.num_nodes, 6))
    idx = (old.tri + 1).astype(np.int32).ravel()
    ref = O.bamg_interp_mesh_to_mesh(idx, xo, yo, data, new.x, new.y, True, 0.0)
    got, info = InterpFromMeshToMesh2dx(idx, xo, yo, data, new.x, new.y, True, 0.0, return_info=True)
    same = np.all(got == ref, axis=1)
    # every point either equals the reference bit for bit ...
    assert same.mean() > 0.97
    # ... or sits in bamg's hull-filling triangles outside the data mesh, where the reference does not
    # return the default although the point is outside (reft >= 0 there only inside the mesh)
    assert np.all(got[~same] == 0.0)


@pytest.mark.gpu
@pytest.mark.skipif(O.bamg_shim() is None, reason="oracle/_ref (real contrib/bamg) not present on this box")
def test_gpu_interpolation_on_a_mesh_with_islands_matches_real_bamg_live():
    """isdefault == false on a domain with inner boundaries: target points inside the islands (bamg fills them with triangles
    between their shore vertices), in the pockets of the coast, beyond the hull, and inside the mesh -- against the real bamg."""
    from nextsim_amd.interp import InterpFromMeshToMesh2dx
    x, y, tri = cases.mesh_with_holes("40km")
    rng = np.random.default_rng(3)
    R = np.hypot(x, y).max()
    data = np.stack([np.sin(x / 5e5) + np.cos(y / 7e5), 1e-3 * x - 2e-3 * y, rng.standard_normal(x.size)], 1)
    th = rng.uniform(0, 2 * np.pi, 3000); rr = R * np.sqrt(rng.uniform(0, 1.3 ** 2, 3000))
    xi, yi = rr * np.cos(th), rr * np.sin(th)
    idx = (tri + 1).ravel()
    ref = O.bamg_interp_mesh_to_mesh(idx, x, y, data, xi, yi, False)
    got, info = InterpFromMeshToMesh2dx(idx, x, y, data, xi, yi, False, return_info=True)
    assert info["completion_refused"] is None and info["num_stand_in"] == 0
    assert info["num_in_fill"] > 50 and info["num_on_hull"] > 200 and info["num_exterior"] == info["num_in_fill"] + info["num_on_hull"]
    rel = (np.abs(got - ref) / np.abs(data).max(0)).max(1)
    assert rel.max() <= 4.5e-16, rel.max()
    assert np.all(got == ref, axis=1).mean() > 0.9


@pytest.mark.gpu
def test_gpu_interpolation_properties_at_full_size():
    """10 km mesh -> 1.5e5 random interior points: constants reproduced, linear fields exact to the
    integer quantisation, weights in [0,1]."""
    from nextsim_amd.interp import InterpFromMeshToMesh2dx
    gm = cases.global_mesh("10km")
    rng = np.random.default_rng(5)
    n = 150000
    t = rng.integers(0, gm.num_elements, n); w = rng.dirichlet([1, 1, 1], n)
    xi = (gm.x[gm.tri[t]] * w).sum(1); yi = (gm.y[gm.tri[t]] * w).sum(1)
    data = np.stack([np.full(gm.num_nodes, 3.25), 1e-3 * gm.x + 5e-4 * gm.y, (gm.x == gm.x).astype(float) * 0 + np.arange(gm.num_nodes) % 2], 1)
    out, info = InterpFromMeshToMesh2dx((gm.tri + 1).ravel(), gm.x, gm.y, data, xi, yi, True, np.nan, return_info=True)
    assert not np.isnan(out).any() and info["num_exterior"] == 0
    assert np.abs(out[:, 0] - 3.25).max() < 1e-14
    assert np.abs(out[:, 1] - (1e-3 * xi + 5e-4 * yi)).max() < 1e-4
    assert out[:, 2].min() >= -1e-15 and out[:, 2].max() <= 1 + 1e-15


def test_interp_rejects_bad_input_or_missing_gpu():
    from nextsim_amd.interp import InterpFromMeshToMesh2dx
    from nextsim_amd.dynamics import NxsError
    gm = cases.global_mesh("tiny")
    with pytest.raises(NxsError) as e:
        InterpFromMeshToMesh2dx((gm.tri + 1).ravel(), gm.x, gm.y, np.zeros((5, 2)), gm.x[:3], gm.y[:3])
    assert e.value.code == -1 and "lines" in str(e.value)      # InterpFromMeshToMesh2dx.cpp:39-42


@pytest.mark.gpu
@pytest.mark.skipif(O.bamg_shim() is None, reason="oracle/_ref (real contrib/bamg) not present on this box")
@pytest.mark.parametrize("name", ["two_components", "lake_in_island", "pinch_inside", "pinch_at_the_coast"])
def test_gpu_interpolation_outside_meshes_with_several_components_or_pinching_boundaries_matches_real_bamg_live(name):
    """isdefault == false (the regrid call, FE.cpp:3131-3139) on what round 2 sent to the "nearest boundary edge" stand-in (median error 1e-3): data
    meshes with several components (the reference interpolates between them inside bamg's fill triangles), a lake inside an island, holes that
    meet at a vertex or touch the coast.  With the general completion (constrained Delaunay of the boundary vertices) nothing is refused any
    more: every target point -- inside, in the gaps, in the holes, beyond the hull -- within an ulp of the real bamg's value."""
    from nextsim_amd.interp import InterpFromMeshToMesh2dx
    x, y, tri = cases.awkward_meshes()[name]
    rng = np.random.default_rng(4)
    data = np.stack([np.sin(x / 5e5) + np.cos(y / 7e5), 1e-3 * x - 2e-3 * y, rng.standard_normal(x.size)], 1)
    cx, cy = 0.5 * (x.min() + x.max()), 0.5 * (y.min() + y.max())
    hx, hy = 0.6 * np.ptp(x), 0.6 * np.ptp(y)
    xi = cx + hx * rng.uniform(-1, 1, 4000); yi = cy + hy * rng.uniform(-1, 1, 4000)
    if name.startswith("pinch"):   # ... and a cloud around the pinch: the removed triangles' barycentres
        full = cases.global_mesh("small").tri
        gone = np.array(sorted(set(map(tuple, full.tolist())) - set(map(tuple, tri.tolist()))))
        bx, by = x[gone].mean(1), y[gone].mean(1)
        xi = np.concatenate([xi, np.repeat(bx, 50) + 2e4 * rng.standard_normal(50 * bx.size)])
        yi = np.concatenate([yi, np.repeat(by, 50) + 2e4 * rng.standard_normal(50 * by.size)])
    idx = (tri + 1).ravel()
    ref = O.bamg_interp_mesh_to_mesh(idx, x, y, data, xi, yi, False)
    got, info = InterpFromMeshToMesh2dx(idx, x, y, data, xi, yi, False, return_info=True)
    assert info["completion_refused"] is None and info["num_stand_in"] == 0, info
    assert info["num_in_fill"] > 20 and info["num_exterior"] == info["num_in_fill"] + info["num_on_hull"]
    rel = (np.abs(got - ref) / np.abs(data).max(0)).max(1)
    assert rel.max() <= 4.5e-16, (rel.max(), int((rel > 4.5e-16).sum()))
