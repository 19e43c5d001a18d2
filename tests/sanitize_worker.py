"""Worker of tests/test_sanitizers.py: runs under LD_PRELOAD=libasan with ASan/UBSan builds of the oracle
(NXS_ORACLE_LIBRARY), of the host-side sources of the product library (nxs_mesh.cpp, nxs_io.cpp) and of the
host build of the remapping functions.  Any report aborts the process (abort_on_error / -fno-sanitize-recover)."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from nextsim_amd import _abi  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

host = C.CDLL(sys.argv[1]); remap = C.CDLL(sys.argv[2]); cut = C.CDLL(sys.argv[3])
IP, D = C.POINTER(C.c_int32), _abi.c_double_p

# 1. the oracle: every rheology, young-ice category on and off, 2 ranks with ghosts, checks
for over in ({}, {"dynamics_type": _abi.NXS_DYN_EVP}, {"dynamics_type": _abi.NXS_DYN_MEVP}, {"dynamics_type": _abi.NXS_DYN_FREE_DRIFT},
             {"ice_cat_type": 1}, {"substeps": 3}):
    gm, p, g, lms, fields = cases.make_case("tiny", **over)
    r = O.OracleRank(lms[0], p, fields[0]); r.step(); r.step()   # (EVP with BBM-tuned parameters may blow up: still no UB)
gm, p, g, lms, fields = cases.make_case("toy", nparts=2)
ranks = [O.OracleRank(lm, p, f) for lm, f in zip(lms, fields)]
O.multirank_step(ranks)
nec, nc = O.connectivity(lms[0].indices, lms[0].num_nodes)

# 2. host code of the product library: connectivity tables, writers, readers
lm = lms[0]
idx = np.ascontiguousarray(lm.indices, np.int32)
w1, w2 = C.c_int32(), C.c_int32()
host.nxs_mesh_connectivity.argtypes = [IP, C.c_int32, C.c_int32, IP, D, IP, D]
assert host.nxs_mesh_connectivity(_abi.iptr(idx), lm.num_nodes, lm.num_elements, C.byref(w1), None, C.byref(w2), None) == 0
a = np.empty((lm.num_nodes, w1.value)); b = np.empty((lm.num_nodes, w2.value))
assert host.nxs_mesh_connectivity(_abi.iptr(idx), lm.num_nodes, lm.num_elements, C.byref(w1), _abi.dptr(a), C.byref(w2), _abi.dptr(b)) == 0
assert np.array_equal(a, nec, equal_nan=True) and np.array_equal(b, nc)
ec = np.empty((lm.num_elements, 3))
host.nxs_mesh_element_connectivity.argtypes = [IP, C.c_int32, C.c_int32, D]
assert host.nxs_mesh_element_connectivity(_abi.iptr(idx), lm.num_nodes, lm.num_elements, _abi.dptr(ec)) == 0
bad = idx.copy(); bad[5] = lm.num_nodes + 7
assert host.nxs_mesh_connectivity(_abi.iptr(bad), lm.num_nodes, lm.num_elements, C.byref(w1), None, C.byref(w2), None) != 0

tmp = tempfile.mkdtemp()
V = C.c_void_p
host.nxs_exporter_open.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(V)]
host.nxs_exporter_write_mesh.argtypes = [V, D, D, IP, C.c_int64, IP, C.c_int64]
host.nxs_exporter_write_field.argtypes = [V, C.c_char_p, D, C.c_int64]
host.nxs_exporter_write_field_int.argtypes = [V, C.c_char_p, IP, C.c_int64]
host.nxs_exporter_close.argtypes = [V]
host.nxs_exporter_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(V)]
host.nxs_exporter_file_get_double.argtypes = [V, C.c_char_p, D, C.c_int64]
host.nxs_exporter_file_close.argtypes = [V]
for prec in (b"double", b"float"):
    e = V()
    bp, dp = os.path.join(tmp, "f.bin").encode(), os.path.join(tmp, "f.dat").encode()
    assert host.nxs_exporter_open(bp, dp, prec, C.byref(e)) == 0
    ids = np.arange(1, lm.num_nodes + 1, dtype=np.int32)
    assert host.nxs_exporter_write_mesh(e, _abi.dptr(lm.coord_x), _abi.dptr(lm.coord_y), _abi.iptr(ids), lm.num_nodes, _abi.iptr(idx), idx.size) == 0
    assert host.nxs_exporter_write_field(e, b"M_conc", _abi.dptr(fields[0]["conc"]), lm.num_elements) == 0
    assert host.nxs_exporter_write_field(e, b"Empty", None, 0) == 0
    assert host.nxs_exporter_write_field_int(e, b"Misc_int", _abi.iptr(ids), 4) == 0
    assert host.nxs_exporter_close(e) == 0
    f = V()
    assert host.nxs_exporter_load(bp, dp, C.byref(f)) == 0
    got = np.empty(lm.num_elements)
    assert host.nxs_exporter_file_get_double(f, b"M_conc", _abi.dptr(got), got.size) == 0
    assert host.nxs_exporter_file_get_double(f, b"M_conc", _abi.dptr(got), got.size - 1) != 0      # wrong count: refused
    assert host.nxs_exporter_file_get_double(f, b"nope", _abi.dptr(got), got.size) != 0
    host.nxs_exporter_file_close(f)
    with open(bp.decode(), "r+b") as fh:
        fh.truncate(37)
    assert host.nxs_exporter_load(bp, dp, C.byref(f)) != 0                                         # truncated file: refused


class MVar(C.Structure):
    _fields_ = [(k, C.c_char_p) for k in ("name", "standard_name", "long_name", "units", "cell_methods")]


host.nxs_moorings_create.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, C.POINTER(MVar),
                                     C.c_float, C.c_double, C.c_void_p]
host.nxs_moorings_append.argtypes = [C.c_char_p, C.c_double, C.c_double, C.c_int32, C.POINTER(C.POINTER(C.c_float))]
lon = np.linspace(0, 10, 12, dtype=np.float32).reshape(3, 4); lat = lon + 70
mv = (MVar * 2)(MVar(b"sic", b"sea_ice_area_fraction", b"Concentration", b"1", b"area: mean"), MVar(b"sit", b"", b"Thickness", b"m", b""))
nc_path = os.path.join(tmp, "Moorings.nc").encode()
FP = C.POINTER(C.c_float)
assert host.nxs_moorings_create(nc_path, 4, 3, lon.ctypes.data_as(FP), lat.ctypes.data_as(FP), 2, mv, -1e14, 0.25, None) == 0
d0 = np.arange(12, dtype=np.float32); d1 = d0 * 2
for t in range(3):
    ptrs = (FP * 2)(d0.ctypes.data_as(FP), d1.ctypes.data_as(FP))
    assert host.nxs_moorings_append(nc_path, 42000.0 + t, 0.25, 2, ptrs) == 0
assert host.nxs_moorings_append(os.path.join(tmp, "missing.nc").encode(), 1.0, 0.25, 2, ptrs) != 0

# 3. the remapping functions (host build of the kernel's per-triangle code): walk, identity, overflow
x, y, tri, ng = cases.rect_mesh(10, 1)
x2, y2, tri2, _ = cases.rect_mesh(7, 5)
from nextsim_amd import dynamics as _d  # only its ctypes helpers; the sanitized host library provides the tables  # noqa: E402
I = C.POINTER(C.c_int)
remap.remap_host.argtypes = [I, D, D, C.c_int, C.c_int, I, C.c_int, I, I, D, D, C.c_int, D, C.c_int, I, D, C.c_int, D, I]


def tables(tri, n):
    ii = np.ascontiguousarray((tri + 1).ravel(), np.int32)
    assert host.nxs_mesh_connectivity(_abi.iptr(ii), n, tri.shape[0], C.byref(w1), None, C.byref(w2), None) == 0
    a = np.empty((n, w1.value)); b = np.empty((n, w2.value))
    assert host.nxs_mesh_connectivity(_abi.iptr(ii), n, tri.shape[0], C.byref(w1), _abi.dptr(a), C.byref(w2), _abi.dptr(b)) == 0
    e = np.empty((tri.shape[0], 3))
    assert host.nxs_mesh_element_connectivity(_abi.iptr(ii), n, tri.shape[0], _abi.dptr(e)) == 0
    return (np.ascontiguousarray(np.where(np.isnan(a), 0, a).astype(np.int32) - 1), np.ascontiguousarray(np.where(np.isnan(e), 0, e).astype(np.int32) - 1))


def seeds(x, y, tri, px, py):
    out = []
    for qx, qy in zip(px, py):
        a = (x[tri[:, 1]] - x[tri[:, 0]]) * (qy - y[tri[:, 0]]) - (y[tri[:, 1]] - y[tri[:, 0]]) * (qx - x[tri[:, 0]])
        b = (x[tri[:, 2]] - x[tri[:, 1]]) * (qy - y[tri[:, 1]]) - (y[tri[:, 2]] - y[tri[:, 1]]) * (qx - x[tri[:, 1]])
        c = (x[tri[:, 0]] - x[tri[:, 2]]) * (qy - y[tri[:, 2]]) - (y[tri[:, 0]] - y[tri[:, 2]]) * (qx - x[tri[:, 2]])
        out.append(int(np.flatnonzero((a >= 0) & (b >= 0) & (c >= 0))[0]))
    return np.array(out, np.int32)


neci, eci = tables(tri, x.size)
ip = lambda a: a.ctypes.data_as(I)  # noqa: E731
data = np.ones((tri.shape[0], 2))
big_x = np.array([x.min(), x.max(), x.max(), x.min()]); big_y = np.array([y.min(), y.min(), y.max(), y.max()])
big_t = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
for xn, yn, tn, prev, expect_fail in ((x2, y2, tri2, np.zeros(x2.size), 0), (x, y, tri, np.arange(1, x.size + 1, dtype=np.float64), 0),
                                      (big_x, big_y, big_t, np.zeros(4), 2)):
    tn = np.ascontiguousarray(tn, np.int32)
    sd = seeds(x, y, tri, xn[tn].sum(1) / 3, yn[tn].sum(1) / 3)
    out = np.empty((tn.shape[0], 2)); vis = np.zeros(tn.shape[0], np.int32)
    t32 = np.ascontiguousarray(tri, np.int32)
    nf = remap.remap_host(ip(t32), _abi.dptr(x), _abi.dptr(y), x.size, t32.shape[0], ip(neci), neci.shape[1], ip(eci), ip(tn), _abi.dptr(xn),
                          _abi.dptr(yn), tn.shape[0], _abi.dptr(np.ascontiguousarray(prev)), ng, ip(sd), _abi.dptr(data), 2, _abi.dptr(out), ip(vis))
    assert nf == expect_fail, (nf, expect_fail)

# 4. the host-only mesh preparation of nxs_dyn_set_mesh (nextsim_amd/csrc/nxs_patchcut.hpp) and bamg's convex completion (nxs_hull.inl):
#    tests/native/patchcut_host.cpp builds AND checks every table (each own node solved once, each element written once, complete
#    ascending fans, slots inside their rows, boundary patches first, send / ghost entries consistent with the halo lists ...)
from nextsim_amd import mesh as M  # noqa: E402
U8 = C.POINTER(C.c_uint8); I64 = C.POINTER(C.c_int64)
cut.pc_run.argtypes = [IP, U8, D, D] + [C.c_int] * 8 + [IP, IP, C.c_int, IP, IP, C.c_int, IP, IP] + [C.c_int] * 3 + [I64, C.c_char_p, C.c_int]
cut.pc_hilbert.argtypes = [D, D, C.c_int, IP, C.c_char_p, C.c_int]
cut.pc_hull.argtypes = [IP, D, D, C.c_int, C.c_int, C.c_int, I64, C.c_char_p, C.c_int]
cut.pc_guard_selftest.argtypes = [C.c_int, C.c_char_p, C.c_int]


def i32(a):
    return np.ascontiguousarray(a, np.int32)


def run_cut(lm, patch_nodes=0, want_resident=0, overlap=0, cus=256, res_ept=1, tables=False, depth_multi=0, depth_smooth=0, expect=0):
    idx = i32(lm.indices); g3 = np.ascontiguousarray(lm.ghost_nodes, np.uint8)
    x = np.ascontiguousarray(lm.coord_x, np.float64); y = np.ascontiguousarray(lm.coord_y, np.float64)
    so, si, ro, ri = i32(lm.send_offsets), i32(lm.send_index), i32(lm.recv_offsets), i32(lm.recv_index)
    n2n = cnt = None; W2 = 0
    if tables:   # NodalConnectivity as nxs_dyn_set_mesh converts it: [W2][Nn] 0-based rows + counts
        _, nc_ = O.connectivity(lm.indices, lm.num_nodes)
        W2 = nc_.shape[1] - 1
        cnt = i32(nc_[:, -1]); n2n = i32(np.maximum(nc_[:, :W2].T - 1, 0))
    stats = np.zeros(20, np.int64); msg = C.create_string_buffer(512)
    rc = cut.pc_run(_abi.iptr(idx), g3.ctypes.data_as(U8), _abi.dptr(x), _abi.dptr(y), lm.num_nodes, lm.num_elements, lm.local_ndof, patch_nodes,
                    want_resident, cus, res_ept, len(lm.send_procs), _abi.iptr(so), _abi.iptr(si) if si.size else None, len(lm.recv_procs), _abi.iptr(ro),
                    _abi.iptr(ri) if ri.size else None, overlap, _abi.iptr(n2n) if n2n is not None else None, _abi.iptr(cnt) if cnt is not None else None,
                    W2, depth_multi, depth_smooth, stats.ctypes.data_as(I64), msg, 512)
    assert rc == expect, (rc, msg.value.decode())
    return dict(zip(("nP", "Pmax", "Emax", "Mmax", "Wp", "hilbert", "lds", "P", "res_ok", "res_nbr", "n_boundary", "reordered", "m_nP", "m_EDmax",
                     "s_nP", "s_NDmax", "cut_big", "sum_elem", "over_24_passes", "max_halo"), stats.tolist()))


def shuffled(gm, seed):
    rng = np.random.default_rng(seed)
    pn = rng.permutation(gm.num_nodes); pe = rng.permutation(gm.num_elements)
    inv = np.empty_like(pn); inv[pn] = np.arange(pn.size)
    return M.GlobalMesh(x=gm.x[pn].copy(), y=gm.y[pn].copy(), tri=np.ascontiguousarray(inv[gm.tri][pe].astype(np.int32)),
                        dirichlet=gm.dirichlet[pn].copy(), neumann=gm.neumann[pn].copy(), lat=gm.lat[pn].copy(), name="shuffled")


# 4a. the sequence of the GPU suite's abort (gpurun_out/r2_suite5.log, round 2): a 550 k-triangle mesh cut for the resident loop, then the small
#     mesh with a numbering WITHOUT locality (tests/test_gpu_parity.py::test_shuffled_numbering_still_matches_the_oracle: seed 7) -- first the
#     caller's numbering (rejected: ~6 elements per own node), then the Hilbert curve through the coordinates
big = M.localize(M.make_mesh("h9000"), 1)[0]
r = run_cut(big, want_resident=1)
assert r["nP"] > 512 and r["cut_big"] == 0, r   # 550 k triangles: several rounds of patches (the occupancy check then refuses the resident loop: one kernel per sub-step)
r = run_cut(M.localize(M.make_mesh("h11000"), 1)[0], want_resident=1)
assert r["res_ok"] == 1 and r["cut_big"] == 1 and r["nP"] <= 256 and r["Emax"] <= 2048 and r["Pmax"] <= 1024 and r["Mmax"] <= 1024, r   # 367 k: one large patch per CU
sm = cases.global_mesh("small")
r = run_cut(M.localize(shuffled(sm, 7), 1)[0], tables=True, depth_multi=4, depth_smooth=10)
assert r["hilbert"] == 1 and r["Mmax"] <= 1024, r
for seed in (1, 2, 3):                                             # more permutations, explicit sizes, the resident cut, the toy box
    for kind in ("toy", "small"):
        lm_s = M.localize(shuffled(cases.global_mesh(kind), seed), 1)[0]
        assert run_cut(lm_s, tables=True, depth_multi=2, depth_smooth=5)["hilbert"] == 1
        run_cut(lm_s, patch_nodes=64); run_cut(lm_s, patch_nodes=1024); run_cut(lm_s, want_resident=1)
r = run_cut(M.localize(shuffled(cases.global_mesh("40km"), 5), 1)[0], tables=True, depth_multi=4, depth_smooth=10)
assert r["hilbert"] == 1, r
# 4b. regular numbering: every explicit size, D-ring patches, smoother patches, a device with few CUs
lm1 = M.localize(sm, 1)[0]
for pn_ in (64, 100, 200, 512, 1024):
    run_cut(lm1, patch_nodes=pn_, tables=True, depth_multi=2 + pn_ % 3, depth_smooth=5)
for cus_ in (1, 8, 104, 256, 304):
    run_cut(M.localize(cases.global_mesh("40km"), 1)[0], cus=cus_, tables=True, depth_multi=4, depth_smooth=10)
    run_cut(M.localize(cases.global_mesh("40km"), 1)[0], cus=cus_, want_resident=1)
# 4c. ragged 3- and 4-rank partitions (every rank a neighbour of every other, disconnected pieces, ghost elements without an own node -> orphan
#     patches, corner nodes sent to several ranks), the exchange tables, the overlap variant of the resident loop
for kind, nparts, seed in (("toy", 3, 11), ("toy", 4, 12), ("small", 3, 13), ("small", 4, 14), ("40km", 4, 15)):
    gmk = cases.global_mesh(kind)
    for lm_r in M.localize(gmk, nparts, elem_part=cases.ragged_partition(gmk, nparts, seed)):
        r0 = run_cut(lm_r); r1 = run_cut(lm_r, want_resident=1, overlap=1); run_cut(lm_r, patch_nodes=64, overlap=1)
        cut.pc_set_band(16); run_cut(lm_r, want_resident=1, overlap=1); run_cut(lm_r, patch_nodes=200, want_resident=1); cut.pc_set_band(0)
        assert r0["n_boundary"] >= 1 and r1["n_boundary"] >= 1, (r0, r1)
for lm_r in M.localize(sm, 2):
    run_cut(lm_r, want_resident=1, overlap=1)
# 4d. the partitions the 8-GPU runs have (BASELINE configs 3 and 4): all eight RCB parts of the 10 km and of the 2 km mesh, cut for the resident
#     loop -- the 2 km parts are the ones whose patches are closed at 480 elements (Ecap): every one must qualify
for kind in ("10km", "2km"):
    for lm_p in M.localize(M.make_mesh(kind), 8):
        r = run_cut(lm_p, want_resident=1)
        assert r["res_ok"] == 1 and r["Emax"] <= 512 and r["nP"] <= 512 and r["cut_big"] == 0, (kind, lm_p.rank, r)
        cut.pc_set_band(48)            # ... and with the sent nodes in patches of 48 of their own (what the resident loop of several ranks takes)
        rb = run_cut(lm_p, want_resident=1, overlap=1)
        cut.pc_set_band(0)
        assert rb["res_ok"] == 1 and rb["Emax"] <= 512 and rb["nP"] <= 512 and rb["cut_big"] == 0 and rb["n_boundary"] >= 1, (kind, lm_p.rank, r, rb)
for lm_p in M.localize(M.make_mesh("2km"), 4):     # ... and the four parts of a 4-GPU run: one large patch per CU (k_substep_resident_big)
    r = run_cut(lm_p, want_resident=1, overlap=1)   # (with the interior-first lists: whole slices of 512 elements)
    assert r["res_ok"] == 1 and r["cut_big"] == 1 and r["nP"] <= 256, (lm_p.rank, r)
# 4f. the two-ring patches of k_substep_pair<HALO> (two sub-steps per launch on several ranks): regular and ragged partitions (ghost elements without an own
#     node -> patches of their own, corner nodes sent to several ranks, sent nodes that touch no ghost), explicit sizes, a numbering without locality
cut.pc_pair_mr.argtypes = [IP, U8, D, D] + [C.c_int] * 5 + [IP, C.c_int, C.c_int, I64, C.c_char_p, C.c_int]


def run_pair_mr(lm, pair_nodes=0, cus=256, hilbert=0, expect=0):
    idx = i32(lm.indices); g3 = np.ascontiguousarray(lm.ghost_nodes, np.uint8)
    x = np.ascontiguousarray(lm.coord_x, np.float64); y = np.ascontiguousarray(lm.coord_y, np.float64)
    si = i32(lm.send_index)
    st = np.zeros(8, np.int64); m_ = C.create_string_buffer(512)
    rc = cut.pc_pair_mr(_abi.iptr(idx), g3.ctypes.data_as(U8), _abi.dptr(x), _abi.dptr(y), lm.num_nodes, lm.num_elements, lm.local_ndof, pair_nodes, cus,
                        _abi.iptr(si) if si.size else None, si.size, hilbert, st.ctypes.data_as(I64), m_, 512)
    assert rc == expect, (rc, m_.value.decode())
    return dict(zip(("nP", "nG", "nBand", "P", "lds", "sumE2", "sumN2", "orphan_patches"), st.tolist()))


for kind, nparts, seed in (("toy", 3, 11), ("toy", 4, 12), ("small", 3, 13), ("small", 4, 14), ("40km", 4, 15)):
    gmk = cases.global_mesh(kind)
    for lm_r in M.localize(gmk, nparts, elem_part=cases.ragged_partition(gmk, nparts, seed)):
        r = run_pair_mr(lm_r); run_pair_mr(lm_r, pair_nodes=64); run_pair_mr(lm_r, pair_nodes=200, hilbert=1); run_pair_mr(lm_r, cus=8)
        assert r["nBand"] >= 1 and r["nG"] >= r["nBand"], r
for nparts in (2, 3, 8):
    for lm_r in M.localize(cases.global_mesh("10km"), nparts):
        r = run_pair_mr(lm_r)
        assert 1 <= r["nBand"] <= r["nG"] < r["nP"], r
# 4e. the Hilbert order on coordinates nobody should pass: NaN, infinities, one point, all equal, empty
msg = C.create_string_buffer(256)
for xs, ys in ((np.array([0., np.nan, 1., np.inf, -np.inf, 2.]), np.array([np.nan, 0., 1., 5., -5., np.inf])), (np.zeros(7), np.zeros(7)),
               (np.array([3.]), np.array([4.])), (np.zeros(0), np.zeros(0)), (np.array([1e308, -1e308, 0.]), np.array([-1e308, 1e308, 0.]))):
    out = np.zeros(max(xs.size, 1), np.int32)
    assert cut.pc_hilbert(_abi.dptr(np.ascontiguousarray(xs)), _abi.dptr(np.ascontiguousarray(ys)), xs.size, _abi.iptr(out), msg, 256) == 0, msg.value
# 4f. bamg's convex completion: the disc with its irregular coast, with islands, the toy box -- by the pocket construction and by the general one
#     (constrained Delaunay of the boundary vertices); several components, a lake inside an island, pinching boundaries (general only);
#     what is no mesh says why
hs = np.zeros(3, np.int64)
toy_ = cases.global_mesh("toy")
hull_cases = [(sm.x, sm.y, sm.tri), cases.mesh_with_holes("small"), (toy_.x, toy_.y, toy_.tri), cases.mesh_with_holes("40km")]
for xk, yk, tk in hull_cases:
    ti = i32((tk + 1).ravel())
    sizes = []
    for mode in (0, 1, 2):
        assert cut.pc_hull(_abi.iptr(ti), _abi.dptr(np.ascontiguousarray(xk)), _abi.dptr(np.ascontiguousarray(yk)), xk.size, tk.shape[0], mode, hs.ctypes.data_as(I64), msg, 256) == 0, msg.value
        assert hs[0] == 1 and hs[2] >= 3, (hs, msg.value)
        sizes.append((int(hs[1]), int(hs[2])))
    assert sizes[0] == sizes[1] == sizes[2], sizes
for name, (xk, yk, tk) in cases.awkward_meshes().items():
    ti = i32((tk + 1).ravel())
    for mode in (0, 1):
        assert cut.pc_hull(_abi.iptr(ti), _abi.dptr(np.ascontiguousarray(xk)), _abi.dptr(np.ascontiguousarray(yk)), xk.size, tk.shape[0], mode, hs.ctypes.data_as(I64), msg, 256) == 0, (name, msg.value)
        assert hs[0] == 1 and hs[1] > 0, (name, hs, msg.value)
    assert cut.pc_hull(_abi.iptr(ti), _abi.dptr(np.ascontiguousarray(xk)), _abi.dptr(np.ascontiguousarray(yk)), xk.size, tk.shape[0], 2, hs.ctypes.data_as(I64), msg, 256) == 0
    assert hs[0] == 0 and msg.value, name          # the pocket construction alone refuses them, with a reason
bad_t = i32(np.array([1, 2, 3, 2, 1, 4, 1, 2, 5]))  # one edge in three triangles
assert cut.pc_hull(_abi.iptr(bad_t), _abi.dptr(np.array([0., 1., 0., 1., 0.5])), _abi.dptr(np.array([0., 0., 1., 1., -1.])), 5, 3, 0, hs.ctypes.data_as(I64), msg, 256) == 0 and hs[0] == 0
line_t = i32(np.array([1, 2, 3]))                   # a degenerate triangle: every boundary vertex on one line
cut.pc_hull(_abi.iptr(line_t), _abi.dptr(np.array([0., 1., 2.])), _abi.dptr(np.array([0., 1., 2.])), 3, 1, 1, hs.ctypes.data_as(I64), msg, 256)
assert hs[0] == 0
# 4g. the guard of the ABI (nxs_guard.hpp): length_error / bad_alloc -> NXS_ERR_NOMEM, anything else -> NXS_ERR_INTERNAL, with a text
# nxs_dyn_set_halo's padding of one-directional neighbours (nxs_cut::pad_halo_directions, the very text libnxsdyn.so includes): every partition of the test
# meshes as initUpdateGhosts leaves its lists, hand-made lists (both directions missing, nothing missing, no neighbour at all), and lists it must refuse
cut.pc_pad_halo.argtypes = [IP, IP, C.POINTER(C.c_int), IP, IP, C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_int]
def pad(sp, so, rp, ro, expect=0):
    cap = 64
    a = [np.zeros(cap + 1, np.int32) for _ in range(4)]
    for buf, v in zip(a, (sp, so, rp, ro)): buf[:len(v)] = v
    ns, nr = C.c_int(len(sp)), C.c_int(len(rp))
    rc = cut.pc_pad_halo(_abi.iptr(a[0]), _abi.iptr(a[1]), C.byref(ns), _abi.iptr(a[2]), _abi.iptr(a[3]), C.byref(nr), cap, msg, 256)
    assert rc == expect, (rc, msg.value)
    return a[0][:ns.value].tolist(), a[1][:ns.value + 1].tolist(), a[2][:nr.value].tolist(), a[3][:nr.value + 1].tolist()
one_directional = 0
for kind, nparts, seed in (("small", 4, None), ("small", 3, 1), ("small", 4, 2), ("small", 8, None), ("40km", 4, 3)):
    lms_p = M.localize(cases.global_mesh(kind), nparts, elem_part=cases.ragged_partition(cases.global_mesh(kind), nparts, seed) if seed is not None else None)
    padded = [pad(lm.send_procs.tolist(), lm.send_offsets.tolist(), lm.recv_procs.tolist(), lm.recv_offsets.tolist()) for lm in lms_p]
    for r, (sp, so, rp, ro) in enumerate(padded):
        one_directional += len(sp) != len(lms_p[r].send_procs) or len(rp) != len(lms_p[r].recv_procs)
        for k, q in enumerate(sp):     # what I send to q is what q expects from me, the added directions included: both ends padded alike without talking
            qsp, qso, qrp, qro = padded[q]
            kk = qrp.index(r)
            assert so[k + 1] - so[k] == qro[kk + 1] - qro[kk], (kind, nparts, r, q)
assert one_directional > 0               # (the regular 4-rank partition of 'small' has one: rank 0 -> rank 3)
assert pad([1, 2], [0, 3, 5], [3], [0, 4]) == ([1, 2, 3], [0, 3, 5, 5], [3, 1, 2], [0, 4, 4, 4])
assert pad([2, 1], [0, 1, 2], [1, 2], [0, 5, 9]) == ([2, 1], [0, 1, 2], [1, 2], [0, 5, 9])
assert pad([], [0], [], [0]) == ([], [0], [], [0])
pad([1, 1], [0, 1, 2], [1], [0, 2], expect=1); assert b"twice" in msg.value
assert cut.pc_guard_selftest(0, msg, 256) == -6 and b"length_error" in msg.value
assert cut.pc_guard_selftest(1, msg, 256) == -6 and b"bad_alloc" in msg.value
assert cut.pc_guard_selftest(2, msg, 256) == -7 and b"boom" in msg.value
assert cut.pc_guard_selftest(3, msg, 256) == -7
print("sanitize worker ok")
