"""Generates the committed fixtures under tests/golden/.  Run in the build container (needs
/root/reference for the real bamg):  python tests/golden/make_golden.py

  bamg_connectivity.npz  REAL contrib/bamg BamgConvertMeshx output on the seeded 'tiny' mesh
                         (inputs + both tables)  -- pins nxs_mesh_connectivity and the oracle.
  bamg_interp.npz        REAL contrib/bamg InterpFromMeshToMesh2dx (FE.cpp:3131 call shape) on the seeded 'small'
                         mesh: 3 nodal fields + 2 element fields at interior points, mesh vertices, exterior
                         points, with and without default value -- pins the regrid interpolation kernel.
  bamg_mesh_to_grid.npz  REAL contrib/bamg InterpFromMeshToGridx (Moorings sampling, gridoutput.cpp:496) on the
                         'small' mesh: nodal + element data, normal and flipped grids, NaN data.
  bamg_remap.npz         REAL contrib/bamg ConservativeRemappingMeshToMesh (FE.cpp:3108) on four seeded regrid pairs
                         (adapted / coarser / finer / moved vertices), plus bamg's ElementConnectivity of the old
                         mesh and the seed triangles InterpFromMeshToMesh2dx returns -- pins the remapping kernel.
  bamg_regrid.npz        a regrid made by the REAL remesher (Bamgx as adaptMesh calls it): both meshes, PreviousNumbering, and the
                         reference's interpFields results on them (conservative remap of 5 element variables, P1 interpolation of 6
                         nodal variables) -- pins both regrid kernels on what bamg really produces.
  bamg_grid_to_mesh.npz  REAL contrib/bamg InterpFromGridToMeshx (forcing ingest, externaldata.cpp:1436): bilinear / triangle /
                         nearest, descending axes, pixel contours, NaN data, row-major data, nodes on grid lines and outside.
  bamg_completion.npz    REAL contrib/bamg Mesh(index, x, y, ...) = ReconstructExistingMesh: the triangles it adds between the boundary
                         and the convex hull and inside holes, and its hull edges, on five meshes -- pins the convex completion the
                         regrid interpolation needs for points outside the data mesh.
  mapx_lat.npz           REAL contrib/mapx inverse_mapx with mesh/NpsNextsim.mpp (GmshMesh::lat(), gmshmesh.cpp:1798-1824) at
                         600 points -- pins the latitude (Coriolis, sign of the turning angle) of the synthetic meshes.
  oracle_tiny.npz        oracle (liboracle.so) state on the 'tiny' toy case after 1 sub-step, 1 step
                         and 3 steps (beyond a few steps the algorithm amplifies 1-ulp differences to O(1), see
                         tests/test_oracle_sensitivity.py): a regression net for the oracle itself and size-0 cost
                         expected values for the GPU path.  NOT reference output: the reference's
                         model/ cannot be built here.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from nextsim_amd import mesh as M  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

KEYS = ("VT", "UM", "UT", "sigma0", "sigma1", "sigma2", "damage", "conc", "thick", "snow_thick", "ridge_ratio",
        "conc_young", "h_young", "hs_young", "conc_myi", "thick_myi")


def interp_case():
    """Seeded inputs of the interpolation fixture (shared with tests/test_interp.py)."""
    gm = cases.global_mesh("small")
    rng = np.random.default_rng(11)
    idx = (gm.tri + 1).astype(np.int32).ravel()
    nodal = np.stack([np.sin(gm.x / 7e5) * np.cos(gm.y / 9e5), 1e-3 * gm.x - 2e-3 * gm.y, rng.standard_normal(gm.num_nodes)], 1)
    elemental = np.stack([rng.standard_normal(gm.num_elements), np.arange(gm.num_elements, dtype=float)], 1)
    t = rng.integers(0, gm.num_elements, 4000)
    w = rng.dirichlet([1, 1, 1], 4000)
    xin = (gm.x[gm.tri[t]] * w).sum(1); yin = (gm.y[gm.tri[t]] * w).sum(1)      # strictly interior points
    vsel = rng.integers(0, gm.num_nodes, 300)                                     # mesh vertices themselves
    # just outside: boundary-edge points pushed 50 m along the outward normal (a mesh that moved a little)
    tri = gm.tri
    e = np.concatenate([tri[:, [1, 2]], tri[:, [2, 0]], tri[:, [0, 1]]], 0)
    key = np.sort(e, 1)[:, 0].astype(np.int64) * gm.num_nodes + np.sort(e, 1)[:, 1]
    uk, first, cnt = np.unique(key, return_index=True, return_counts=True)
    be = e[first[cnt == 1]]                                                       # oriented as in their triangle (ccw)
    be = be[rng.permutation(be.shape[0])[:150]]
    s_ = rng.uniform(0.2, 0.8, be.shape[0])
    px = gm.x[be[:, 0]] * (1 - s_) + gm.x[be[:, 1]] * s_; py = gm.y[be[:, 0]] * (1 - s_) + gm.y[be[:, 1]] * s_
    tx = gm.x[be[:, 1]] - gm.x[be[:, 0]]; ty = gm.y[be[:, 1]] - gm.y[be[:, 0]]
    nrm = np.hypot(tx, ty)
    xnear = px + 50.0 * ty / nrm; ynear = py - 50.0 * tx / nrm                    # outward of a ccw triangle: (ty, -tx)
    R = np.hypot(gm.x, gm.y).max()
    th = rng.uniform(0, 2 * np.pi, 50)
    xfar = 1.3 * R * np.cos(th); yfar = 1.3 * R * np.sin(th)                      # far outside (default-value mode only)
    xi = np.concatenate([xin, gm.x[vsel], xnear, xfar]); yi = np.concatenate([yin, gm.y[vsel], ynear, yfar])
    kind = np.concatenate([np.zeros(4000, int), np.ones(300, int), np.full(150, 2), np.full(50, 3)])
    return gm, idx, nodal, elemental, xi, yi, kind


def grid_case():
    """Moorings-like regular grid over the seeded 'small' mesh (model/gridoutput.cpp:467-505 call shape)."""
    gm = cases.global_mesh("small")
    rng = np.random.default_rng(13)
    idx = (gm.tri + 1).astype(np.int32).ravel()
    nodal = np.stack([np.cos(gm.x / 6e5) + np.sin(gm.y / 8e5), rng.standard_normal(gm.num_nodes)], 1)
    nodal[rng.integers(0, gm.num_nodes, 5), 1] = np.nan          # NaN data -> default (InterpFromMeshToGridx.cpp:172)
    elemental = np.stack([rng.standard_normal(gm.num_elements), np.arange(gm.num_elements, dtype=float)], 1)
    ncols = 97; nrows = 83
    xmin = gm.x.min() - 2e5; ymax = gm.y.max() + 1.5e5
    xpost = (gm.x.max() - gm.x.min() + 4e5) / (nrows - 1); ypost = (gm.y.max() - gm.y.min() + 3e5) / (ncols - 1)
    return gm, idx, nodal, elemental, xmin, ymax, xpost, ypost, nrows, ncols


def make_grid_fixture():
    gm, idx, nodal, elemental, xmin, ymax, xp, yp, nrows, ncols = grid_case()
    out = {}
    out["nodal"] = O.bamg_interp_mesh_to_grid(idx, gm.x, gm.y, nodal, xmin, ymax, xp, yp, nrows, ncols, -1e14)
    out["elemental"] = O.bamg_interp_mesh_to_grid(idx, gm.x, gm.y, elemental, xmin, ymax, xp, yp, nrows, ncols, -1e14)
    out["nodal_flipped"] = O.bamg_interp_mesh_to_grid(idx, gm.x, gm.y, nodal, xmin, ymax, -xp, -yp, nrows, ncols, -1e14)
    np.savez_compressed(os.path.join(HERE, "bamg_mesh_to_grid.npz"), **out)


def make_interp_fixture():
    gm, idx, nodal, elemental, xi, yi, kind = interp_case()
    out = dict(xi=xi, yi=yi, kind=kind)
    out["nodal"] = O.bamg_interp_mesh_to_mesh(idx, gm.x, gm.y, nodal, xi, yi, False)
    out["nodal_default"] = O.bamg_interp_mesh_to_mesh(idx, gm.x, gm.y, nodal, xi, yi, True, -999.0)
    # element data only at points of the mesh: the reference throws on points that fall in its hull-filling
    # triangles ("Triangle number ... not in [0 nels]", InterpFromMeshToMesh2dx.cpp:161-164)
    ins = kind == 0  # (a boundary vertex can be located in a hull triangle too)
    out["elemental"] = O.bamg_interp_mesh_to_mesh(idx, gm.x, gm.y, elemental, xi[ins], yi[ins], False)
    np.savez_compressed(os.path.join(HERE, "bamg_interp.npz"), **out)


def remap_cases():
    """Seeded (old mesh, new mesh) pairs of the conservative-remapping fixture (shared with tests/test_remap.py):
    name -> (x_old, y_old, tri_old, x_new, y_new, tri_new, previous_numbering, n_geom, data)."""
    rng = np.random.default_rng(77)
    x, y, tri, ng = cases.rect_mesh(24, 1)
    out = {}
    xn, yn, trin, prev = cases.adapted_mesh(x, y, tri, ng, 2)
    data = np.column_stack([rng.random(tri.shape[0]), rng.normal(size=tri.shape[0]) * 1e3, np.ones(tri.shape[0])])
    out["adapted"] = (x, y, tri, xn, yn, trin, prev, ng, data)                       # what a regrid looks like
    x2, y2, tri2, _ = cases.rect_mesh(17, 5)
    out["coarser"] = (x, y, tri, x2, y2, tri2, np.zeros(x2.size), ng, data)          # every triangle through checkTriangle
    x3, y3, tri3, _ = cases.rect_mesh(41, 6)
    out["finer"] = (x, y, tri, x3, y3, tri3, np.zeros(x3.size), ng, data)
    xm, ym = x.copy(), y.copy()
    xm[ng:] += rng.uniform(-800, 800, x.size - ng); ym[ng:] += rng.uniform(-800, 800, x.size - ng)
    out["moved"] = (x, y, tri, xm, ym, tri, np.arange(1, x.size + 1, dtype=np.float64), ng, data)
    return out


def make_remap_fixture():
    out = {}
    for name, (x, y, tri, xn, yn, trin, prev, ng, data) in remap_cases().items():
        out[name] = O.bamg_conservative_remap(tri + 1, x, y, trin + 1, xn, yn, prev, ng, data)
        ec, _ = O.bamg_element_connectivity(tri + 1, x, y)
        out[name + "_ec"] = ec
        bx = (((0. + xn[trin[:, 0]]) + xn[trin[:, 1]]) + xn[trin[:, 2]]) / 3.
        by = (((0. + yn[trin[:, 0]]) + yn[trin[:, 1]]) + yn[trin[:, 2]]) / 3.
        out[name + "_seed"] = np.round(O.bamg_interp_mesh_to_mesh(tri + 1, x, y, np.arange(tri.shape[0], dtype=np.float64), bx, by, True, -1)[:, 0]).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "bamg_remap.npz"), **out)


def make_regrid_fixture():
    """A regrid as the reference performs it, with the REAL remesher: bamg cleans a seeded mesh (first adaptation,
    FE.cpp:384-392), the vertices move (shear band, no flipped triangle), Bamgx adapts (FE.cpp:3760-3801).  Stored: both
    meshes, PreviousNumbering, and what the reference's interpFields computes on them -- ConservativeRemappingMeshToMesh of
    5 element variables (FE.cpp:3108) and InterpFromMeshToMesh2dx of 6 nodal variables (FE.cpp:3131-3139)."""
    x, y, tri, ng = cases.rect_mesh(24, 1)

    def sides_mm(x, y, t):
        s = np.stack([np.hypot(x[t[:, 1]] - x[t[:, 0]], y[t[:, 1]] - y[t[:, 0]]), np.hypot(x[t[:, 2]] - x[t[:, 1]], y[t[:, 2]] - y[t[:, 1]]),
                      np.hypot(x[t[:, 2]] - x[t[:, 0]], y[t[:, 2]] - y[t[:, 0]])], 1)
        return s.min(1).mean(), s.max(1).mean()                    # minMaxSide, FE.cpp:343
    hmin, hmax = sides_mm(x, y, tri)
    dirf = np.arange(1, ng + 1)
    xb, yb, trib, _, ngb = O.bamg_adapt(tri + 1, x, y, dirf, x, y, hmin, hmax)
    rng = np.random.default_rng(4)
    um = np.zeros((2, xb.size))
    um[0] = 5e3 * np.tanh((yb - yb.mean()) / 25e3); um[1] = rng.normal(0, 150, xb.size)
    d = np.minimum.reduce([xb - xb.min(), xb.max() - xb, yb - yb.min(), yb.max() - yb])
    um *= np.clip(d / 40e3, 0, 1)
    xm, ym = xb + um[0], yb + um[1]
    jac = (xm[trib[:, 1]] - xm[trib[:, 0]]) * (ym[trib[:, 2]] - ym[trib[:, 0]]) - (xm[trib[:, 2]] - xm[trib[:, 0]]) * (ym[trib[:, 1]] - ym[trib[:, 0]])
    assert (jac > 0).all()
    xc, yc, tric, prev, ngc = O.bamg_adapt(trib + 1, xb, yb, np.arange(1, ngb + 1), xm, ym, hmin, hmax)
    elt = np.column_stack([rng.random(trib.shape[0]), rng.normal(size=trib.shape[0]) * 1e3, np.ones(trib.shape[0]), rng.random(trib.shape[0]) ** 3,
                           rng.uniform(0, 4, trib.shape[0])])
    nod = np.column_stack([rng.normal(0, 0.2, xb.size) for _ in range(6)])
    out = dict(x_old=xm, y_old=ym, tri_old=trib, x_new=xc, y_new=yc, tri_new=tric, prev=prev, ngeom=np.int32(ngc), elt_in=elt, nod_in=nod)
    out["elt_out"] = O.bamg_conservative_remap(trib + 1, xm, ym, tric + 1, xc, yc, prev, ngc, elt)
    out["nod_out"] = O.bamg_interp_mesh_to_mesh(trib + 1, xm, ym, nod, xc, yc, False)
    np.savez_compressed(os.path.join(HERE, "bamg_regrid.npz"), **out)


def grid_to_mesh_cases():
    """Seeded forcing-ingest cases (shared with tests/test_grid_to_mesh.py): name -> (x_in, y_in, data, x_mesh, y_mesh, interp, row_major)."""
    rng = np.random.default_rng(21)
    gm = cases.global_mesh("small")
    xs = np.linspace(gm.x.min() - 30e3, gm.x.max() + 30e3, 41); ys = np.linspace(gm.y.max() + 30e3, gm.y.min() - 30e3, 33)   # y descending, as many datasets
    X, Y = np.meshgrid(xs, ys)
    data = np.stack([1e-5 * X - 2e-5 * Y, np.sin(X / 2e5) * np.cos(Y / 3e5), rng.normal(size=X.shape)], 2)    # [M=33, N=41, 3]
    xm, ym = gm.x.copy(), gm.y.copy()
    xm[:4] = [xs[0], xs[-1], xs[5], xs[0] - 1e3]; ym[:4] = [ys[0], ys[-1], ys[7], ys[3]]     # corners, a grid line, one node outside
    out = {"bilinear": (xs, ys, data, xm, ym, 1, False), "triangle": (xs, ys, data, xm, ym, 0, False), "nearest": (xs, ys, data, xm, ym, 2, False)}
    # pixel contours (one more coordinate than data columns / rows) and NaN data
    xc = np.linspace(xs[0], xs[-1], 42); yc = np.linspace(ys[0], ys[-1], 34)
    d2 = data.copy(); d2[10:12, 20:22, 1] = np.nan
    out["contours_nan"] = (xc, yc, d2, xm, ym, 1, False)
    out["row_major"] = (xs, ys[::-1].copy(), np.ascontiguousarray(np.transpose(data[::-1], (1, 0, 2))), xm, ym, 1, True)
    return out


def make_grid_to_mesh_fixture():
    out = {}
    for name, (xs, ys, data, xm, ym, interp, rm) in grid_to_mesh_cases().items():
        out[name] = O.bamg_interp_grid_to_mesh(xs, ys, data, xm, ym, 1e8, interp, rm)
    np.savez_compressed(os.path.join(HERE, "bamg_grid_to_mesh.npz"), **out)


def mapx_case():
    rng = np.random.default_rng(11)
    x = rng.uniform(-2.6e6, 2.6e6, 600); y = rng.uniform(-2.6e6, 2.6e6, 600)
    x[:3] = [0.0, 1.0, -2.5e6]; y[:3] = [0.0, 0.0, 2.5e6]          # the pole itself and a far corner
    return x, y


def make_mapx_fixture():
    x, y = mapx_case()
    np.savez_compressed(os.path.join(HERE, "mapx_lat.npz"), x=x, y=y, lat=O.mapx_lat(x, y))


def completion_cases():
    """Meshes of the completion fixture: name -> (x, y, tri 0-based)."""
    out = {}
    for kind in ("small", "40km"):
        gm = cases.global_mesh(kind)
        out[kind] = (gm.x, gm.y, gm.tri)
        out[kind + "_holes"] = cases.mesh_with_holes(kind)
    gm = cases.global_mesh("toy")
    out["toy"] = (gm.x, gm.y, gm.tri)
    # round 3: what the pocket construction refuses and the general one (constrained Delaunay of the boundary vertices) covers --
    # several components, a component inside a hole of another, boundaries that pinch
    out.update(cases.awkward_meshes())
    return out


def make_completion_fixture():
    """bamg_completion.npz: what the REAL bamg adds to a mesh when it reconstructs it (Mesh(index, x, y, ...), the mesh
    InterpFromMeshToMesh2dx works on): the fill triangles (sorted vertex triples, sorted) and the hull edges (from the boundary
    triangles, as counter-clockwise pairs, sorted) -- pins csrc/nxs_hull.inl."""
    out = {}
    for name, (x, y, tri) in completion_cases().items():
        T, reft = O.bamg_completed_mesh((tri + 1).ravel(), x, y)
        ne = tri.shape[0]
        assert np.array_equal(T[:ne], tri) and np.all(reft[:ne] >= 0)
        extra = T[ne:]; inf = (extra < 0).any(1)
        assert np.all(reft[ne:][~inf] < 0)
        fill = np.array(sorted(tuple(sorted(r)) for r in extra[~inf].tolist()), np.int32).reshape(-1, 3)
        hull = []
        for r in extra[inf].tolist():
            k = r.index(-1)
            hull.append((r[(k + 2) % 3], r[(k + 1) % 3]))     # the boundary triangle sees its real edge clockwise
        out[name + "_fill"] = fill
        out[name + "_hull"] = np.array(sorted(hull), np.int32).reshape(-1, 2)
    np.savez_compressed(os.path.join(HERE, "bamg_completion.npz"), **out)


def make_connectivity_fixture():
    lm = M.localize(cases.global_mesh("tiny"), 1)[0]
    nec, nc = O.bamg_connectivity(lm.indices, lm.coord_x, lm.coord_y)
    np.savez_compressed(os.path.join(HERE, "bamg_connectivity.npz"), indices=lm.indices, num_nodes=lm.num_nodes,
                        x=lm.coord_x, y=lm.coord_y, nec=nec, nc=nc)


def main():
    """No argument: every fixture.  Otherwise the named ones: connectivity interp grid remap mapx grid_to_mesh regrid oracle."""
    assert O.bamg_shim() is not None, "build oracle/_ref first (make -C oracle ref)"
    makers = {"connectivity": make_connectivity_fixture, "interp": make_interp_fixture, "grid": make_grid_fixture,
              "remap": make_remap_fixture, "mapx": make_mapx_fixture, "grid_to_mesh": make_grid_to_mesh_fixture,
              "regrid": make_regrid_fixture, "oracle": make_oracle_fixture, "completion": make_completion_fixture}
    for name in (sys.argv[1:] or list(makers)):
        makers[name]()
        print("wrote", name)


def make_oracle_fixture():
    out = {}
    for tag, nsteps, over in (("sub1", 1, dict(substeps=1, dtime_step=200. / 120.)), ("step1", 1, {}), ("step3", 3, {})):
        gm, p, g, lms, fields = cases.make_case("tiny", **over)
        r = O.OracleRank(lms[0], p, fields[0])
        for _ in range(nsteps):
            r.step()
        for k in KEYS:
            out[f"{tag}_{k}"] = r.arr[k]
    np.savez_compressed(os.path.join(HERE, "oracle_tiny.npz"), **out)


if __name__ == "__main__":
    main()
