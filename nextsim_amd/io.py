"""Output writers over the C ABI of include/nxs_io.h: the reference's Exporter
(.bin/.dat records, core/src/exporter.cpp) and the Moorings NetCDF file (model/gridoutput.cpp:805-1035).
Host-side I/O in C++ inside libnxsdyn.so; this module only marshals arguments."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from .dynamics import NxsError, load_library


class MooringVar(C.Structure):
    _fields_ = [(k, C.c_char_p) for k in ("name", "standard_name", "long_name", "units", "cell_methods")]


class MooringProj(C.Structure):
    _fields_ = [("semi_major_axis", C.c_double), ("semi_minor_axis", C.c_double), ("lat0", C.c_double), ("lat_ts", C.c_double),
                ("rotation", C.c_double), ("false_easting", C.c_int32)]


IO_EXPORTS = ("nxs_exporter_open", "nxs_exporter_write_mesh", "nxs_exporter_write_field", "nxs_exporter_write_field_int",
              "nxs_exporter_close", "nxs_exporter_load", "nxs_exporter_file_num_records", "nxs_exporter_file_record",
              "nxs_exporter_file_get_double", "nxs_exporter_file_get_int", "nxs_exporter_file_close", "nxs_restart_write",
              "nxs_restart_read", "nxs_moorings_create", "nxs_moorings_create_format", "nxs_moorings_file_format", "nxs_moorings_append",
              "nxs_io_last_error")
NC_AUTO, NC_CLASSIC, NC_NETCDF4 = 0, 3, 4
_decl = False


def _lib():
    global _decl
    L = load_library()
    if not _decl:
        P = C.POINTER
        L.nxs_exporter_open.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, P(C.c_void_p)]
        L.nxs_exporter_write_mesh.argtypes = [C.c_void_p, _abi.c_double_p, _abi.c_double_p, _abi.c_int32_p, C.c_int64, _abi.c_int32_p, C.c_int64]
        L.nxs_exporter_write_field.argtypes = [C.c_void_p, C.c_char_p, _abi.c_double_p, C.c_int64]
        L.nxs_exporter_write_field_int.argtypes = [C.c_void_p, C.c_char_p, _abi.c_int32_p, C.c_int64]
        L.nxs_exporter_close.argtypes = [C.c_void_p]
        L.nxs_moorings_create.argtypes = [C.c_char_p, C.c_int32, C.c_int32, P(C.c_float), P(C.c_float), C.c_int32, P(MooringVar), C.c_float,
                                          C.c_double, P(MooringProj)]
        L.nxs_moorings_create_format.argtypes = [C.c_char_p, C.c_int32, C.c_int32, P(C.c_float), P(C.c_float), C.c_int32, P(MooringVar), C.c_float,
                                                 C.c_double, P(MooringProj), C.c_int32]
        L.nxs_moorings_file_format.argtypes = [C.c_char_p]
        L.nxs_moorings_append.argtypes = [C.c_char_p, C.c_double, C.c_double, C.c_int32, P(P(C.c_float))]
        L.nxs_io_last_error.restype = C.c_char_p
        V = C.c_void_p
        L.nxs_exporter_load.argtypes = [C.c_char_p, C.c_char_p, P(V)]
        L.nxs_exporter_file_num_records.argtypes = [V]
        L.nxs_exporter_file_record.argtypes = [V, C.c_int, P(C.c_char_p), P(C.c_char_p), P(C.c_int64)]
        L.nxs_exporter_file_get_double.argtypes = [V, C.c_char_p, _abi.c_double_p, C.c_int64]
        L.nxs_exporter_file_get_int.argtypes = [V, C.c_char_p, _abi.c_int32_p, C.c_int64]
        L.nxs_exporter_file_close.argtypes = [V]
        L.nxs_restart_write.argtypes = [C.c_char_p, C.c_char_p, _abi.c_double_p, _abi.c_double_p, _abi.c_int32_p, C.c_int64, _abi.c_int32_p,
                                        C.c_int64, _abi.c_int32_p, _abi.c_int32_p, C.c_int64, C.c_double, C.c_int32, P(C.c_char_p),
                                        P(_abi.c_double_p), _abi.c_double_p, _abi.c_double_p, _abi.c_double_p, _abi.c_double_p]
        L.nxs_restart_read.argtypes = [C.c_char_p, C.c_char_p, P(V), P(V)]
        _decl = True
    return L


def _chk(L, rc):
    if rc:
        raise NxsError(rc, (L.nxs_io_last_error() or b"").decode())


class Exporter:
    """core/include/exporter.hpp: Exporter(precision); writeMesh / writeField / writeRecord."""

    def __init__(self, bin_path: str, dat_path: str, precision: str = "double"):
        self.L = _lib()
        self.h = C.c_void_p()
        _chk(self.L, self.L.nxs_exporter_open(bin_path.encode(), dat_path.encode(), precision.encode(), C.byref(self.h)))

    def writeMesh(self, xnod, ynod, idnod, elements):
        x = np.ascontiguousarray(xnod, np.float64); y = np.ascontiguousarray(ynod, np.float64)
        i = np.ascontiguousarray(idnod, np.int32); e = np.ascontiguousarray(elements, np.int32).ravel()
        _chk(self.L, self.L.nxs_exporter_write_mesh(self.h, _abi.dptr(x), _abi.dptr(y), _abi.iptr(i), x.size, _abi.iptr(e), e.size))

    def writeField(self, values, name: str):
        v = np.asarray(values)
        if np.issubdtype(v.dtype, np.integer):
            v = np.ascontiguousarray(v, np.int32)
            _chk(self.L, self.L.nxs_exporter_write_field_int(self.h, name.encode(), _abi.iptr(v) if v.size else None, v.size))
        else:
            v = np.ascontiguousarray(v, np.float64)
            _chk(self.L, self.L.nxs_exporter_write_field(self.h, name.encode(), _abi.dptr(v) if v.size else None, v.size))

    def close(self):
        if self.h.value:
            rc = self.L.nxs_exporter_close(self.h)
            self.h = C.c_void_p()
            _chk(self.L, rc)


def read_exported(bin_path: str, dat_path: str) -> dict:
    """Exporter::readRecord + loadFile (core/src/exporter.cpp:191-260): name -> array."""
    out = {}
    raw = open(bin_path, "rb").read()
    pos = 0
    for line in open(dat_path):
        name, typ, size, _, _ = line.split()
        n = int(np.frombuffer(raw, np.int32, 1, pos)[0]); pos += 4
        assert n == int(float(size)), (name, n, size)
        dt = {"int": np.int32, "float": np.float32, "double": np.float64}[typ]
        out[name] = np.frombuffer(raw, dt, n, pos).copy(); pos += n * np.dtype(dt).itemsize
    assert pos == len(raw)
    return out


def _file_to_dict(L, fh) -> dict:
    out = {}
    for i in range(L.nxs_exporter_file_num_records(fh)):
        name, typ, cnt = C.c_char_p(), C.c_char_p(), C.c_int64()
        _chk(L, L.nxs_exporter_file_record(fh, i, C.byref(name), C.byref(typ), C.byref(cnt)))
        key = name.value.decode()
        if key in out:
            continue                       # field_map.emplace keeps the first record of a name
        if typ.value == b"int":
            a = np.empty(cnt.value, np.int32)
            _chk(L, L.nxs_exporter_file_get_int(fh, name.value, _abi.iptr(a) if a.size else None, a.size))
        else:
            a = np.empty(cnt.value, np.float64)
            _chk(L, L.nxs_exporter_file_get_double(fh, name.value, _abi.dptr(a) if a.size else None, a.size))
        out[key] = a
    return out


def load_exported(bin_path: str, dat_path: str) -> dict:
    """Exporter::readRecord + loadFile through the library (nxs_exporter_load): name -> array, record order kept."""
    L = _lib()
    fh = C.c_void_p()
    _chk(L, L.nxs_exporter_load(bin_path.encode(), dat_path.encode(), C.byref(fh)))
    try:
        return _file_to_dict(L, fh)
    finally:
        L.nxs_exporter_file_close(fh)


def write_restart(directory, name_str, xnod, ynod, idnod, elements, misc_int, dirichlet_flags, current_time, elt_vars: dict,
                  VT, UM, UT, previous_numbering):
    """FiniteElement::writeRestart (FE.cpp:9518-9695).  elt_vars: ordered {restart name: [Ne] values}."""
    L = _lib()
    f64 = lambda a: np.ascontiguousarray(a, np.float64)  # noqa: E731
    x, y = f64(xnod), f64(ynod)
    ids = np.ascontiguousarray(idnod, np.int32); el = np.ascontiguousarray(elements, np.int32).ravel()
    mi = np.ascontiguousarray(misc_int, np.int32); df = np.ascontiguousarray(dirichlet_flags, np.int32)
    assert mi.size == 4
    names = (C.c_char_p * max(len(elt_vars), 1))(*[k.encode() for k in elt_vars])
    vals = [f64(v) for v in elt_vars.values()]
    ptrs = (_abi.c_double_p * max(len(vals), 1))(*[_abi.dptr(v) for v in vals])
    VT, UM, UT, pn = f64(VT), f64(UM), f64(UT), f64(previous_numbering)
    _chk(L, L.nxs_restart_write(str(directory).encode(), name_str.encode(), _abi.dptr(x), _abi.dptr(y), _abi.iptr(ids), x.size, _abi.iptr(el),
                                el.size, _abi.iptr(mi), _abi.iptr(df) if df.size else None, df.size, float(current_time), len(vals), names, ptrs,
                                _abi.dptr(VT), _abi.dptr(UM), _abi.dptr(UT), _abi.dptr(pn)))


def read_restart(directory, name_str):
    """The file part of FiniteElement::readRestart (FE.cpp:9699-9790): (mesh records, field records) as dicts."""
    L = _lib()
    mh, fh = C.c_void_p(), C.c_void_p()
    _chk(L, L.nxs_restart_read(str(directory).encode(), name_str.encode(), C.byref(mh), C.byref(fh)))
    try:
        return _file_to_dict(L, mh), _file_to_dict(L, fh)
    finally:
        L.nxs_exporter_file_close(mh); L.nxs_exporter_file_close(fh)


def moorings_file_format(path) -> int:
    """NC_CLASSIC or NC_NETCDF4, from the magic bytes."""
    L = _lib()
    r = L.nxs_moorings_file_format(str(path).encode())
    if r < 0:
        _chk(L, r)
    return r


def moorings_create(path, lon, lat, variables, miss_val=-1e14, averaging_period=0.0, proj=None, format=NC_AUTO):
    """variables: list of dicts with name, standard_name, long_name, units, cell_methods.  format: NC_AUTO (NetCDF-4 as the
    reference writes it when the HDF5 library can be loaded, else NetCDF-3 classic), NC_CLASSIC, NC_NETCDF4."""
    L = _lib()
    lon = np.ascontiguousarray(lon, np.float32); lat = np.ascontiguousarray(lat, np.float32)
    nrows, ncols = lon.shape
    arr = (MooringVar * len(variables))()
    for a, v in zip(arr, variables):
        for k in ("name", "standard_name", "long_name", "units", "cell_methods"):
            setattr(a, k, v.get(k, "").encode())
    pj = None
    if proj is not None:
        pj = MooringProj(**proj)
    _chk(L, L.nxs_moorings_create_format(path.encode(), ncols, nrows, lon.ctypes.data_as(C.POINTER(C.c_float)), lat.ctypes.data_as(C.POINTER(C.c_float)),
                                         len(variables), arr, float(miss_val), float(averaging_period), C.byref(pj) if pj is not None else None,
                                         int(format)))


def moorings_append(path, timestamp, fields, averaging_period=0.0):
    L = _lib()
    keep = [np.ascontiguousarray(f, np.float32) for f in fields]
    ptrs = (C.POINTER(C.c_float) * len(keep))(*[k.ctypes.data_as(C.POINTER(C.c_float)) for k in keep])
    _chk(L, L.nxs_moorings_append(path.encode(), float(timestamp), float(averaging_period), len(keep), ptrs))
