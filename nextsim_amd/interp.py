"""Mesh-to-mesh interpolation at regrid over the C ABI of include/nxs_interp.h.

`InterpFromMeshToMesh2dx` keeps the name and argument meaning of the reference routine
(contrib/bamg/src/InterpFromMeshToMesh2dx.cpp:17-24, called at FE.cpp:3131-3139); the work is done by the
HIP kernel in libnxsdyn.so -- no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from .dynamics import NxsError, load_library

_declared = False


def _lib():
    global _declared
    L = load_library()
    if not _declared:
        L.nxs_interp_mesh_to_mesh_2d.argtypes = [_abi.c_double_p, _abi.c_int32_p, _abi.c_double_p, _abi.c_double_p, C.c_int32,
                                                 C.c_int32, _abi.c_double_p, C.c_int32, C.c_int32, _abi.c_double_p, _abi.c_double_p,
                                                 C.c_int32, C.c_int32, C.c_double, C.c_int32, C.POINTER(C.c_int32),
                                                 C.POINTER(C.c_double)]
        L.nxs_interp_mesh_to_mesh_2d.restype = C.c_int
        L.nxs_interp_last_error.restype = C.c_char_p
        _declared = True
    return L


def InterpFromMeshToMesh2dx(index_data, x_data, y_data, data, x_interp, y_interp, isdefault=False, defaultvalue=1e-24,
                            device=0, return_info=False):
    """index_data: 1-based [3*nels]; data: [M_data, N_data] with M_data == nods (P1) or nels (P0).
    Returns data_interp [N_interp, N_data] (and {'num_exterior', 'kernel_ms'} when return_info)."""
    L = _lib()
    index_data = np.ascontiguousarray(index_data, np.int32).ravel()
    x_data = np.ascontiguousarray(x_data, np.float64); y_data = np.ascontiguousarray(y_data, np.float64)
    data = np.ascontiguousarray(data, np.float64)
    if data.ndim == 1:
        data = data[:, None]
    x_interp = np.ascontiguousarray(x_interp, np.float64); y_interp = np.ascontiguousarray(y_interp, np.float64)
    out = np.empty((x_interp.size, data.shape[1]))
    next_, ms = C.c_int32(0), C.c_double(0.0)
    rc = L.nxs_interp_mesh_to_mesh_2d(_abi.dptr(out), _abi.iptr(index_data), _abi.dptr(x_data), _abi.dptr(y_data), x_data.size,
                                      index_data.size // 3, _abi.dptr(data), data.shape[0], data.shape[1], _abi.dptr(x_interp),
                                      _abi.dptr(y_interp), x_interp.size, int(bool(isdefault)), float(defaultvalue), device,
                                      C.byref(next_), C.byref(ms))
    if rc:
        raise NxsError(rc, (L.nxs_interp_last_error() or b"").decode())
    if return_info:
        return out, {"num_exterior": next_.value, "kernel_ms": ms.value}
    return out


def InterpFromMeshToGridx(index_mesh, x_mesh, y_mesh, data, xmin, ymax, xposting, yposting, nrows, ncols, default_value,
                          device=0, return_info=False):
    """Mesh -> regular grid (Moorings) sampling, argument meaning of contrib/bamg's InterpFromMeshToGridx
    (model/gridoutput.cpp:496-505).  Returns griddata [nrows, ncols, N_data]."""
    L = _lib()
    if not hasattr(L, "_grid_declared"):
        L.nxs_interp_mesh_to_grid.argtypes = [_abi.c_double_p, _abi.c_int32_p, _abi.c_double_p, _abi.c_double_p, C.c_int32, C.c_int32,
                                              _abi.c_double_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                                              C.c_int32, C.c_int32, C.c_double, C.c_int32, C.POINTER(C.c_double)]
        L.nxs_interp_mesh_to_grid.restype = C.c_int
        L._grid_declared = True
    index_mesh = np.ascontiguousarray(index_mesh, np.int32).ravel()
    x_mesh = np.ascontiguousarray(x_mesh, np.float64); y_mesh = np.ascontiguousarray(y_mesh, np.float64)
    data = np.ascontiguousarray(data, np.float64)
    if data.ndim == 1:
        data = data[:, None]
    out = np.empty((nrows, ncols, data.shape[1]))
    ms = C.c_double(0.0)
    rc = L.nxs_interp_mesh_to_grid(_abi.dptr(out), _abi.iptr(index_mesh), _abi.dptr(x_mesh), _abi.dptr(y_mesh), x_mesh.size,
                                   index_mesh.size // 3, _abi.dptr(data), data.shape[0], data.shape[1], float(xmin), float(ymax),
                                   float(xposting), float(yposting), int(nrows), int(ncols), float(default_value), device, C.byref(ms))
    if rc:
        raise NxsError(rc, (L.nxs_interp_last_error() or b"").decode())
    return (out, {"kernel_ms": ms.value}) if return_info else out
