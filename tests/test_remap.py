"""Conservative remapping of the element variables at regrid: the HIP kernel behind
nxs_interp_conservative_remap against the REAL contrib/bamg ConservativeRemappingMeshToMesh (FE.cpp:3108;
contrib/bamg/src/ConservativeRemapping.cpp:176-328) -- live when oracle/_ref is present, and through the
committed fixture tests/golden/bamg_remap.npz generated with it.

Bar: bit-exact.  The per-triangle functions of the kernel (nextsim_amd/csrc/nxs_remap_core.inl) are also built
for the host (oracle/libremap_host.so, test-only) so that the CPU suite checks the very code the GPU runs."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import cases
from nextsim_amd import _abi, dynamics
from oracle import pyoracle as O

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "bamg_remap.npz")
NAMES = ("adapted", "coarser", "finer", "moved")
IP = C.POINTER(C.c_int)


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def _tables(tri, n):
    idx = np.ascontiguousarray((tri + 1).ravel(), np.int32)
    nec, _ = dynamics.mesh_connectivity(idx, n)
    ec = dynamics.mesh_element_connectivity(idx, n)
    return nec, ec


def _host_remap(case, seed):
    """The kernel's per-triangle functions, host build."""
    x, y, tri, xn, yn, trin, prev, ng, data = case
    L = C.CDLL(os.path.join(os.path.dirname(O.__file__), "libremap_host.so"))
    D = _abi.c_double_p
    L.remap_host.argtypes = [IP, D, D, C.c_int, C.c_int, IP, C.c_int, IP, IP, D, D, C.c_int, D, C.c_int, IP, D, C.c_int, D, IP]
    nec, ec = _tables(tri, x.size)
    neci = np.ascontiguousarray(np.where(np.isnan(nec), 0, nec).astype(np.int32) - 1)
    eci = np.ascontiguousarray(np.where(np.isnan(ec), 0, ec).astype(np.int32) - 1)
    t_old = np.ascontiguousarray(tri, np.int32); t_new = np.ascontiguousarray(trin, np.int32)
    seed = np.ascontiguousarray(seed, np.int32)
    prev = np.ascontiguousarray(prev, np.float64)
    data = np.ascontiguousarray(data)
    out = np.empty((t_new.shape[0], data.shape[1])); visits = np.zeros(t_new.shape[0], np.int32)
    i = lambda a: a.ctypes.data_as(IP)  # noqa: E731
    nf = L.remap_host(i(t_old), _abi.dptr(x), _abi.dptr(y), x.size, t_old.shape[0], i(neci), neci.shape[1], i(eci), i(t_new),
                      _abi.dptr(xn), _abi.dptr(yn), t_new.shape[0], _abi.dptr(prev), ng, i(seed), _abi.dptr(data), data.shape[1],
                      _abi.dptr(out), i(visits))
    return out, visits, nf


@pytest.mark.skipif(O.bamg_shim() is None, reason="oracle/_ref (real contrib/bamg) not built here")
def test_real_bamg_reproduces_the_committed_fixture():
    z = np.load(GOLD)
    for name, (x, y, tri, xn, yn, trin, prev, ng, data) in make_golden.remap_cases().items():
        assert _same(O.bamg_conservative_remap(tri + 1, x, y, trin + 1, xn, yn, prev, ng, data), z[name]), name


def test_fixture_is_sane():
    """A constant field stays constant wherever the walk covered the whole new triangle; unchanged triangles keep
    their value to rounding; the area integral is conserved to the reference's own accuracy (its walk stops at
    the first NaN of an ElementConnectivity row, so coverage next to the boundary can be incomplete)."""
    z = np.load(GOLD)
    for name, (x, y, tri, xn, yn, trin, prev, ng, data) in make_golden.remap_cases().items():
        ref = z[name]
        assert ref.shape == (trin.shape[0], 3) and not np.isnan(ref).any()
        dev = np.abs(ref[:, 2] - 1.)
        assert np.median(dev) < 1e-12 and dev.max() < 0.9 and (dev > 1e-9).mean() < 0.05

        def area(x, y, t):
            return 0.5 * ((x[t[:, 1]] - x[t[:, 0]]) * (y[t[:, 2]] - y[t[:, 0]]) - (x[t[:, 2]] - x[t[:, 0]]) * (y[t[:, 1]] - y[t[:, 0]]))
        a_old, a_new = area(x, y, tri), area(xn, yn, trin)
        if name != "moved":
            assert abs((ref[:, 0] * a_new).sum() / (data[:, 0] * a_old).sum() - 1.) < 2e-2


def test_element_connectivity_equals_bamg():
    z = np.load(GOLD)
    x, y, tri = make_golden.remap_cases()["adapted"][:3]
    _, ec = _tables(tri, x.size)
    assert _same(ec, z["adapted_ec"])
    assert np.isnan(ec).any() and np.isnan(ec[:, 0]).any()   # boundary triangles exist, NaN also in column 0


@pytest.mark.parametrize("name", NAMES)
def test_kernel_functions_on_the_host_match_real_bamg_bit_for_bit(name):
    z = np.load(GOLD)
    case = make_golden.remap_cases()[name]
    out, visits, nf = _host_remap(case, z[name + "_seed"])
    assert nf == 0
    assert _same(out, z[name])
    if name in ("adapted", "moved"):
        assert (visits == 1).mean() > 0.8          # the PreviousNumbering shortcut carries a real regrid
    if name in ("coarser", "finer"):
        assert visits.max() >= 8                   # the recursion replay is exercised


def test_capacity_overflow_is_reported_not_hidden():
    """One huge new triangle over a fine old mesh overlaps more old triangles than the kernel's stack holds."""
    x, y, tri, ng = cases.rect_mesh(24, 1)
    xn = np.array([x.min(), x.max(), x.max(), x.min()]); yn = np.array([y.min(), y.min(), y.max(), y.max()])
    trin = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    data = np.ones((tri.shape[0], 1))
    bx = xn[trin].sum(1) / 3; by = yn[trin].sum(1) / 3
    # seeds by brute force (no bamg needed)
    seed = []
    for px, py in zip(bx, by):
        a = (x[tri[:, 1]] - x[tri[:, 0]]) * (py - y[tri[:, 0]]) - (y[tri[:, 1]] - y[tri[:, 0]]) * (px - x[tri[:, 0]])
        b = (x[tri[:, 2]] - x[tri[:, 1]]) * (py - y[tri[:, 1]]) - (y[tri[:, 2]] - y[tri[:, 1]]) * (px - x[tri[:, 1]])
        c = (x[tri[:, 0]] - x[tri[:, 2]]) * (py - y[tri[:, 2]]) - (y[tri[:, 0]] - y[tri[:, 2]]) * (px - x[tri[:, 2]])
        seed.append(int(np.flatnonzero((a >= 0) & (b >= 0) & (c >= 0))[0]))
    out, visits, nf = _host_remap((x, y, tri, xn, yn, trin, np.zeros(4), 0, data), np.array(seed))
    assert nf == 2 and np.isnan(out).all() and (visits < 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_remapping_matches_real_bamg_fixture_bit_for_bit(name):
    from nextsim_amd.interp import ConservativeRemappingMeshToMesh
    z = np.load(GOLD)
    x, y, tri, xn, yn, trin, prev, ng, data = make_golden.remap_cases()[name]
    out, info = ConservativeRemappingMeshToMesh(data, tri + 1, x, y, trin + 1, xn, yn, prev, ng, return_info=True)
    assert info["num_failed"] == 0
    assert _same(out, z[name])
    # same answer with the caller's own bamg tables
    nec, ec = _tables(tri, x.size)
    out2 = ConservativeRemappingMeshToMesh(data, tri + 1, x, y, trin + 1, xn, yn, prev, ng, nec_old=nec, ec_old=ec)
    assert _same(out2, out)


@pytest.mark.gpu
def test_gpu_remapping_second_pass_and_what_it_cannot_do():
    """New triangles that overlap more old ones than the fast path's 96-entry lists are redone by a second pass with
    4096-entry lists in global memory (a much coarser new mesh); a barycentre outside the old mesh stays a reported
    failure with a NaN row (the reference asserts there)."""
    from nextsim_amd.interp import ConservativeRemappingMeshToMesh
    x, y, tri, ng = cases.rect_mesh(24, 1)
    xn = np.array([x.min(), x.max(), x.max(), x.min(), x.max() + 5e4]); yn = np.array([y.min(), y.min(), y.max(), y.max(), y.max() + 5e4])
    trin = np.array([[0, 1, 2], [0, 2, 3], [1, 4, 2]], np.int32)   # two that cover half the domain each, one with its barycentre outside
    rng = np.random.default_rng(2)
    data = np.column_stack([np.ones(tri.shape[0]), rng.random(tri.shape[0])])
    out, info = ConservativeRemappingMeshToMesh(data, tri + 1, x, y, trin + 1, xn, yn, None, 0, return_info=True)
    assert info["num_failed"] == 1 and np.isnan(out[2]).all() and not np.isnan(out[:2]).any()
    assert (info["visits"][:2] > 96).all() and info["visits"][2] < 0
    assert np.abs(out[:2, 0] - 1.).max() < 0.2                     # a constant stays roughly constant (the reference loses coverage at the boundary)
    if O.bamg_shim() is not None:                                  # the reference has no capacity: same bits from the second pass
        ref = O.bamg_conservative_remap(tri + 1, x, y, trin[:2] + 1, xn[:4], yn[:4], np.zeros(4), 0, data)
        assert np.array_equal(out[:2], ref)


@pytest.mark.gpu
@pytest.mark.skipif(O.bamg_shim() is None, reason="oracle/_ref (real contrib/bamg) not present on this box")
def test_gpu_remapping_matches_live_bamg_on_a_larger_regrid():
    from nextsim_amd.interp import ConservativeRemappingMeshToMesh
    x, y, tri, ng = cases.rect_mesh(120, 7)
    rng = np.random.default_rng(5)
    data = rng.random((tri.shape[0], 30))                      # ~30 element variables at a regrid
    xn, yn, trin, prev = cases.adapted_mesh(x, y, tri, ng, 9, frac_touched=0.1)
    ref = O.bamg_conservative_remap(tri + 1, x, y, trin + 1, xn, yn, prev, ng, data)
    out, info = ConservativeRemappingMeshToMesh(data, tri + 1, x, y, trin + 1, xn, yn, prev, ng, return_info=True)
    assert info["num_failed"] == 0 and _same(out, ref)
    x2, y2, tri2, _ = cases.rect_mesh(100, 8)
    ref = O.bamg_conservative_remap(tri + 1, x, y, tri2 + 1, x2, y2, np.zeros(x2.size), ng, data)
    out, info = ConservativeRemappingMeshToMesh(data, tri + 1, x, y, tri2 + 1, x2, y2, np.zeros(x2.size), ng, return_info=True)
    assert info["num_failed"] == 0
    same = np.all(out == ref, axis=1)
    # a barycentre that falls exactly on an old edge in bamg's integer plane may be seeded in the other triangle
    # (walk-dependent tie): same set of contributions, another summation order
    assert same.mean() > 0.999 and np.abs(out - ref).max() < 1e-12


# ---- a regrid made by the REAL remesher (Bamgx), tests/golden/bamg_regrid.npz ----

REGRID = os.path.join(os.path.dirname(__file__), "golden", "bamg_regrid.npz")


def _seeds_brute(x, y, tri, px, py):
    out = []
    for qx, qy in zip(px, py):
        a = (x[tri[:, 1]] - x[tri[:, 0]]) * (qy - y[tri[:, 0]]) - (y[tri[:, 1]] - y[tri[:, 0]]) * (qx - x[tri[:, 0]])
        b = (x[tri[:, 2]] - x[tri[:, 1]]) * (qy - y[tri[:, 1]]) - (y[tri[:, 2]] - y[tri[:, 1]]) * (qx - x[tri[:, 1]])
        c = (x[tri[:, 0]] - x[tri[:, 2]]) * (qy - y[tri[:, 2]]) - (y[tri[:, 0]] - y[tri[:, 2]]) * (qx - x[tri[:, 2]])
        out.append(int(np.flatnonzero((a >= 0) & (b >= 0) & (c >= 0))[0]))
    return np.array(out, np.int32)


@pytest.mark.skipif(O.bamg_shim() is None, reason="oracle/_ref (real contrib/bamg) not built here")
def test_real_remesher_reproduces_the_regrid_fixture():
    import shutil, tempfile
    keep = tempfile.mkdtemp()
    shutil.copy(REGRID, os.path.join(keep, "a.npz"))
    try:
        make_golden.make_regrid_fixture()
        a, b = np.load(os.path.join(keep, "a.npz")), np.load(REGRID)
        assert all(np.array_equal(a[k], b[k]) for k in a.files)
    finally:
        shutil.copy(os.path.join(keep, "a.npz"), REGRID)


def test_regrid_fixture_is_what_a_regrid_looks_like():
    z = np.load(REGRID)
    prev, ng, tn, to = z["prev"], int(z["ngeom"]), z["tri_new"], z["tri_old"]
    assert ng == 96 and (prev[ng + 1:] >= 0).all() and prev.max() <= z["x_old"].size
    kept = prev > 0
    assert 0.5 < kept.mean() < 1.0                                     # KeepVertices = 1: most vertices survive ...
    d = np.hypot(z["x_new"][kept] - z["x_old"][prev[kept].astype(int) - 1], z["y_new"][kept] - z["y_old"][prev[kept].astype(int) - 1])
    assert d.max() < 1e-2                                               # ... where they were, up to bamg's integer grid
    old_sets = {tuple(sorted(r)) for r in to.tolist()}
    pn = np.where(np.arange(prev.size) > ng, prev.astype(int) - 1, np.arange(prev.size))
    same = sum(1 for r in tn.tolist() if tuple(sorted(pn[r].tolist())) in old_sets)
    assert 0.6 < same / tn.shape[0] < 1.0                              # most triangles survive, the sheared band is remeshed
    assert np.abs(z["elt_out"][:, 2] - 1.).max() < 0.5 and np.median(np.abs(z["elt_out"][:, 2] - 1.)) < 1e-9


def test_kernel_functions_on_the_host_match_the_reference_on_a_real_regrid():
    z = np.load(REGRID)
    bx = (((0. + z["x_new"][z["tri_new"][:, 0]]) + z["x_new"][z["tri_new"][:, 1]]) + z["x_new"][z["tri_new"][:, 2]]) / 3.
    by = (((0. + z["y_new"][z["tri_new"][:, 0]]) + z["y_new"][z["tri_new"][:, 1]]) + z["y_new"][z["tri_new"][:, 2]]) / 3.
    seed = _seeds_brute(z["x_old"], z["y_old"], z["tri_old"], bx, by)
    case = (z["x_old"], z["y_old"], z["tri_old"], z["x_new"], z["y_new"], z["tri_new"], z["prev"], int(z["ngeom"]), z["elt_in"])
    out, visits, nf = _host_remap(case, seed)
    assert nf == 0 and _same(out, z["elt_out"])
    assert 0.6 < (visits == 1).mean() < 1.0 and visits.max() >= 6


@pytest.mark.gpu
def test_gpu_regrid_kernels_match_the_reference_on_a_real_regrid():
    """Both regrid kernels on what bamg really produces: element variables (conservative) and nodal variables (P1)."""
    from nextsim_amd.interp import ConservativeRemappingMeshToMesh, InterpFromMeshToMesh2dx
    z = np.load(REGRID)
    out, info = ConservativeRemappingMeshToMesh(z["elt_in"], z["tri_old"] + 1, z["x_old"], z["y_old"], z["tri_new"] + 1, z["x_new"], z["y_new"],
                                                z["prev"], int(z["ngeom"]), return_info=True)
    assert info["num_failed"] == 0 and _same(out, z["elt_out"])
    nod, ninfo = InterpFromMeshToMesh2dx(z["tri_old"] + 1, z["x_old"], z["y_old"], z["nod_in"], z["x_new"], z["y_new"], False, return_info=True)
    assert np.array_equal(nod, z["nod_out"])      # every vertex of the new mesh, its boundary vertices included


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["small", "40km", "holes"])
def test_device_built_regrid_tables_equal_the_host_built_ones_and_a_context_serves_both_calls(kind):
    """The regrid context (nxs_regrid_*): the bucket grid of the exact locator and the two connectivity tables checkTriangle walks are built ON
    THE DEVICE (count -> scan -> fill with atomic cursors -> per-row sort): entry for entry the tables the host code builds -- NodalElementConnectivity
    (descending fans) and ElementConnectivity against nxs_mesh_connectivity / nxs_mesh_element_connectivity (which are checked against the real
    bamg), the grid against a numpy restatement of its definition.  One context then serves the conservative remapping AND the nodal
    interpolation of a regrid (FE.cpp:3071-3154), host or device-resident data, with the bits of the one-shot calls."""
    import ctypes as C
    from nextsim_amd import dynamics, interp
    if kind == "holes":
        x, y, tri = cases.mesh_with_holes("small")
    else:
        gm = cases.global_mesh(kind); x, y, tri = gm.x, gm.y, gm.tri
    nods, nels = x.size, tri.shape[0]
    idx = np.ascontiguousarray((tri + 1).ravel(), np.int32)
    rg = interp.Regrid(idx, x, y)
    # connectivity
    nec, _ = dynamics.mesh_connectivity(idx, nods)
    want_nec = np.where(np.isnan(nec), 0, nec).astype(np.int32) - 1
    assert np.array_equal(rg.debug_table(2).reshape(nods, -1), want_nec)
    ec = dynamics.mesh_element_connectivity(idx, nods)
    assert np.array_equal(rg.debug_table(3).reshape(nels, 3), np.where(np.isnan(ec), 0, ec).astype(np.int32) - 1)
    # bucket grid: bamg's integer plane (bbox + 5 %, 2^30 - 1 units, truncation), G x G cells, every triangle listed in the cells its bounding box touches, ascending
    off, lst = rg.debug_table(0), rg.debug_table(1)
    G = int(round(np.sqrt(off.size - 1)))
    assert G * G + 1 == off.size and off[0] == 0 and off[-1] == lst.size and np.all(np.diff(off) >= 0)
    px0, px1, py0, py1 = x.min(), x.max(), y.min(), y.max()
    dx, dy = (px1 - px0) * 0.05, (py1 - py0) * 0.05
    px0 -= dx; py0 -= dy; px1 += dx; py1 += dy
    coef = 1073741823. / max(px1 - px0, py1 - py0)
    ix = (coef * (x - px0)).astype(np.int64); iy = (coef * (y - py0)).astype(np.int64)
    shift = 30 - int(np.log2(G))
    cx0 = np.clip(ix[tri].min(1) >> shift, 0, G - 1); cx1 = np.clip(ix[tri].max(1) >> shift, 0, G - 1)
    cy0 = np.clip(iy[tri].min(1) >> shift, 0, G - 1); cy1 = np.clip(iy[tri].max(1) >> shift, 0, G - 1)
    assert lst.size == int(((cx1 - cx0 + 1) * (cy1 - cy0 + 1)).sum())
    rng = np.random.default_rng(0)
    for c in rng.integers(0, G * G, 400):
        cy, cx = divmod(int(c), G)
        want = np.flatnonzero((cx0 <= cx) & (cx <= cx1) & (cy0 <= cy) & (cy <= cy1))
        assert np.array_equal(lst[off[c]:off[c + 1]], want), c
    # one context, both calls of a regrid; then again with the data resident on the device
    xn, yn, trin, ng = cases.rect_mesh(9, 3, L=0.5 * np.ptp(x), H=0.4 * np.ptp(y), x0=x.mean() - 0.25 * np.ptp(x), y0=y.mean() - 0.2 * np.ptp(y))
    inside = np.ones(trin.shape[0], bool)
    elem = rng.random((nels, 5)); nodal = rng.standard_normal((nods, 3))
    a1 = rg.remap_elements(elem, trin + 1, xn, yn, np.zeros(xn.size), ng)
    a2 = rg.interp_nodes(nodal, xn, yn, False, 0.0)
    b1 = interp.ConservativeRemappingMeshToMesh(elem, idx, x, y, trin + 1, xn, yn, np.zeros(xn.size), ng)
    b2 = interp.InterpFromMeshToMesh2dx(idx, x, y, nodal, xn, yn, False, 0.0)
    assert np.array_equal(a1, b1, equal_nan=True) and np.array_equal(a2, b2)
    # results into arrays the caller keeps (out=): the same object comes back, filled with the same bits; an array of another shape, dtype or
    # layout is refused, not silently replaced by a fresh one
    k1, k2 = np.full_like(a1, -7.), np.full_like(a2, -7.)
    assert rg.remap_elements(elem, trin + 1, xn, yn, np.zeros(xn.size), ng, out=k1) is k1 and np.array_equal(k1, a1, equal_nan=True)
    assert rg.interp_nodes(nodal, xn, yn, False, 0.0, out=k2) is k2 and np.array_equal(k2, a2)
    for bad in (np.empty((a1.shape[0] + 1, a1.shape[1])), np.empty(a1.shape, np.float32), np.empty(a1.shape[::-1]).T):
        with pytest.raises(ValueError):
            rg.remap_elements(elem, trin + 1, xn, yn, np.zeros(xn.size), ng, out=bad)
    L = dynamics.load_library()
    L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; L.hipFree.argtypes = [C.c_void_p]
    L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    d_in, d_out = C.c_void_p(), C.c_void_p()
    assert L.hipMalloc(C.byref(d_in), elem.nbytes) == 0 and L.hipMalloc(C.byref(d_out), a1.nbytes) == 0
    assert L.hipMemcpy(d_in, elem.ctypes.data, elem.nbytes, 1) == 0
    rg.remap_elements(None, trin + 1, xn, yn, np.zeros(xn.size), ng, in_device=(d_in.value, elem.shape[1]), out_device=d_out.value)
    back = np.empty_like(a1)
    assert L.hipMemcpy(back.ctypes.data, d_out, back.nbytes, 2) == 0
    assert np.array_equal(back, a1, equal_nan=True)
    L.hipFree(d_in); L.hipFree(d_out)
    rg.close()
