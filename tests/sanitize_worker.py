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

host = C.CDLL(sys.argv[1]); remap = C.CDLL(sys.argv[2])
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
print("sanitize worker ok")
