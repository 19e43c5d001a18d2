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
        I = C.POINTER(C.c_int32)
        L.nxs_interp_last_info.argtypes = [I, I, I, I, I, I, C.POINTER(C.c_char_p)]
        L.nxs_mesh_convex_completion_mode.argtypes = [_abi.c_int32_p, _abi.c_double_p, _abi.c_double_p, C.c_int32, C.c_int32, I, _abi.c_int32_p,
                                                      C.c_int32, I, _abi.c_int32_p, C.c_int32, C.c_int32]
        _declared = True
    return L


def last_info() -> dict:
    """Of this thread's last InterpFromMeshToMesh2dx call: the completion's size and which way the exterior points went."""
    L = _lib()
    v = [C.c_int32(0) for _ in range(6)]
    why = C.c_char_p()
    L.nxs_interp_last_info(*[C.byref(q) for q in v], C.byref(why))
    keys = ("num_fill_triangles", "num_hull_edges", "num_exterior", "num_in_fill", "num_on_hull", "num_stand_in")
    out = {k: q.value for k, q in zip(keys, v)}
    out["completion_refused"] = why.value.decode() if why.value else None
    return out


def convex_completion(index, x, y, mode=0):
    """bamg's convex completion of a mesh (index 1-based): (fill triangles [n, 3], hull edges [m, 2]), 1-based, counter-clockwise.
    Host code of the product library (csrc/nxs_hull.inl).  mode 0: automatic, 1: the general construction (constrained Delaunay of the boundary
    vertices), 2: the pocket construction only."""
    L = _lib()
    index = np.ascontiguousarray(index, np.int32).ravel()
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    nf, nh = C.c_int32(0), C.c_int32(0)
    rc = L.nxs_mesh_convex_completion_mode(_abi.iptr(index), _abi.dptr(x), _abi.dptr(y), x.size, index.size // 3, C.byref(nf), None, 0, C.byref(nh), None, 0, mode)
    if rc:
        raise NxsError(rc, (L.nxs_interp_last_error() or b"").decode())
    fill = np.zeros((max(nf.value, 1), 3), np.int32); hull = np.zeros((max(nh.value, 1), 2), np.int32)
    rc = L.nxs_mesh_convex_completion_mode(_abi.iptr(index), _abi.dptr(x), _abi.dptr(y), x.size, index.size // 3, C.byref(nf), _abi.iptr(fill), nf.value,
                                           C.byref(nh), _abi.iptr(hull), nh.value, mode)
    if rc:
        raise NxsError(rc, (L.nxs_interp_last_error() or b"").decode())
    return fill[:nf.value], hull[:nh.value]


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
        return out, dict(last_info(), num_exterior=next_.value, kernel_ms=ms.value)
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


def InterpFromMeshToGridx_device(index_mesh, x_mesh, y_mesh, data_device_ptr, data_length, N_data, xmin, ymax, xposting, yposting, nrows, ncols,
                                 default_value, device=0):
    """InterpFromMeshToGridx with the mesh data already on the device: `data_device_ptr` is a device address of [data_length][N_data]
    rows (e.g. the second value of FiniteElementDynamics.updateIceDiagnostics)."""
    L = _lib()
    if not hasattr(L, "_grid_dev_declared"):
        L.nxs_interp_mesh_to_grid_device.argtypes = [_abi.c_double_p, _abi.c_int32_p, _abi.c_double_p, _abi.c_double_p, C.c_int32, C.c_int32,
                                                     C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                                                     C.c_int32, C.c_int32, C.c_double, C.c_int32, C.POINTER(C.c_double)]
        L.nxs_interp_mesh_to_grid_device.restype = C.c_int
        L._grid_dev_declared = True
    index_mesh = np.ascontiguousarray(index_mesh, np.int32).ravel()
    x_mesh = np.ascontiguousarray(x_mesh, np.float64); y_mesh = np.ascontiguousarray(y_mesh, np.float64)
    out = np.empty((nrows, ncols, N_data))
    ms = C.c_double(0.0)
    rc = L.nxs_interp_mesh_to_grid_device(_abi.dptr(out), _abi.iptr(index_mesh), _abi.dptr(x_mesh), _abi.dptr(y_mesh), x_mesh.size,
                                          index_mesh.size // 3, C.c_void_p(data_device_ptr), int(data_length), int(N_data), float(xmin), float(ymax),
                                          float(xposting), float(yposting), int(nrows), int(ncols), float(default_value), device, C.byref(ms))
    if rc:
        raise NxsError(rc, (L.nxs_interp_last_error() or b"").decode())
    return out


def ConservativeRemappingMeshToMesh(interp_in, index_old, x_old, y_old, index_new, x_new, y_new, previous_numbering=None,
                                    n_geom_vertices=0, nec_old=None, ec_old=None, device=0, return_info=False):
    """Element variables old mesh -> new mesh, argument meaning of contrib/bamg's ConservativeRemappingMeshToMesh
    (FE.cpp:3108) with the BamgMesh members spelled out: index_* 1-based [3*nels], previous_numbering 1-based
    (0 = vertex created by the remesher).  interp_in [nels_old, nb_var] -> [nels_new, nb_var]."""
    L = _lib()
    if not hasattr(L, "_remap_declared"):
        D, I = _abi.c_double_p, _abi.c_int32_p
        L.nxs_interp_conservative_remap.argtypes = [D, D, C.c_int32, I, D, D, C.c_int32, C.c_int32, D, C.c_int32, D, I, D, D, C.c_int32,
                                                    C.c_int32, D, C.c_int32, C.c_int32, C.POINTER(C.c_int32), I, C.POINTER(C.c_double)]
        L.nxs_interp_conservative_remap.restype = C.c_int
        L._remap_declared = True
    f64 = lambda a: np.ascontiguousarray(a, np.float64)  # noqa: E731
    index_old = np.ascontiguousarray(index_old, np.int32).ravel(); index_new = np.ascontiguousarray(index_new, np.int32).ravel()
    x_old, y_old, x_new, y_new = f64(x_old), f64(y_old), f64(x_new), f64(y_new)
    interp_in = f64(interp_in)
    if interp_in.ndim == 1:
        interp_in = interp_in[:, None]
    ne_new = index_new.size // 3
    out = np.empty((ne_new, interp_in.shape[1]))
    visits = np.zeros(ne_new, np.int32)
    prev = None if previous_numbering is None else f64(previous_numbering)
    nec = None if nec_old is None else f64(nec_old)
    ec = None if ec_old is None else f64(ec_old)
    nfail, ms = C.c_int32(0), C.c_double(0.0)
    rc = L.nxs_interp_conservative_remap(_abi.dptr(out), _abi.dptr(interp_in), interp_in.shape[1], _abi.iptr(index_old), _abi.dptr(x_old),
                                         _abi.dptr(y_old), x_old.size, index_old.size // 3, None if nec is None else _abi.dptr(nec),
                                         0 if nec is None else nec.shape[1], None if ec is None else _abi.dptr(ec), _abi.iptr(index_new),
                                         _abi.dptr(x_new), _abi.dptr(y_new), x_new.size, ne_new, None if prev is None else _abi.dptr(prev),
                                         int(n_geom_vertices), device, C.byref(nfail), _abi.iptr(visits), C.byref(ms))
    if rc:
        raise NxsError(rc, (L.nxs_interp_last_error() or b"").decode())
    if return_info:
        return out, {"num_failed": nfail.value, "visits": visits, "kernel_ms": ms.value}
    return out


def last_timing() -> dict:
    """Where this thread's last regrid call spent its time (ms)."""
    L = _lib()
    v = (C.c_double * 8)()
    L.nxs_interp_last_timing.argtypes = [C.POINTER(C.c_double)]
    L.nxs_interp_last_timing(v)
    keys = ("connectivity_ms", "plane_and_grid_ms", "completion_ms", "h2d_ms", "kernel_ms", "d2h_ms", "call_ms")
    return {k: float(v[i]) for i, k in enumerate(keys)}


def _result_array(out, shape):
    """A caller-supplied result array (C-contiguous float64 of exactly `shape`) or a fresh one."""
    if out is None:
        return np.empty(shape)
    if not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.flags.c_contiguous and out.flags.writeable and out.shape == tuple(shape)):
        raise ValueError(f"out must be a writeable C-contiguous float64 array of shape {tuple(shape)}")
    return out


class Regrid:
    """A regrid's context (include/nxs_interp.h, nxs_regrid_*): the OLD mesh's search tables -- integer plane, bucket grid, convex completion,
    the two connectivity tables -- built once, on the device, and shared by interpFields' two interpolation calls (FE.cpp:3071-3154)."""

    IN_DEVICE, OUT_DEVICE = 1, 2

    def __init__(self, index_old, x_old, y_old, device=0):
        L = self.L = _lib()
        if not hasattr(L, "_regrid_declared"):
            D, I, V = _abi.c_double_p, _abi.c_int32_p, C.c_void_p
            L.nxs_regrid_create.argtypes = [I, D, D, C.c_int32, C.c_int32, C.c_int32, C.POINTER(V)]
            L.nxs_regrid_destroy.argtypes = [V]
            L.nxs_regrid_interp_nodes.argtypes = [V, V, V, C.c_int32, C.c_int32, D, D, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
            L.nxs_regrid_remap_elements.argtypes = [V, V, V, C.c_int32, D, C.c_int32, D, I, D, D, C.c_int32, C.c_int32, D, C.c_int32, C.c_int32,
                                                    C.POINTER(C.c_int32), I, C.POINTER(C.c_double)]
            L.nxs_regrid_debug_tables.argtypes = [V, C.c_int32, I, C.c_int64, C.POINTER(C.c_int64)]
            for n in ("nxs_regrid_create", "nxs_regrid_destroy", "nxs_regrid_interp_nodes", "nxs_regrid_remap_elements", "nxs_regrid_debug_tables"):
                getattr(L, n).restype = C.c_int
            L._regrid_declared = True
        self.index = np.ascontiguousarray(index_old, np.int32).ravel()
        self.x = np.ascontiguousarray(x_old, np.float64); self.y = np.ascontiguousarray(y_old, np.float64)
        self.h = C.c_void_p()
        rc = L.nxs_regrid_create(_abi.iptr(self.index), _abi.dptr(self.x), _abi.dptr(self.y), self.x.size, self.index.size // 3, device, C.byref(self.h))
        if rc:
            raise NxsError(rc, (L.nxs_interp_last_error() or b"").decode())

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.nxs_regrid_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise NxsError(rc, (self.L.nxs_interp_last_error() or b"").decode())

    def interp_nodes(self, data, x_interp, y_interp, isdefault=False, defaultvalue=1e-24, data_device=None, out_device=None, return_info=False, out=None):
        """InterpFromMeshToMesh2dx.  data_device = (device pointer, M_data, N_data) instead of `data`; out_device = a device pointer for the result."""
        x_interp = np.ascontiguousarray(x_interp, np.float64); y_interp = np.ascontiguousarray(y_interp, np.float64)
        flags = 0
        if data_device is not None:
            dptr, M, N = data_device
            src = C.c_void_p(dptr); flags |= self.IN_DEVICE
        else:
            data = np.ascontiguousarray(data, np.float64)
            if data.ndim == 1:
                data = data[:, None]
            M, N = data.shape
            src = C.c_void_p(data.ctypes.data)
        if out_device is not None:
            out = None
            dst = C.c_void_p(out_device); flags |= self.OUT_DEVICE
        else:
            out = _result_array(out, (x_interp.size, N))
            dst = C.c_void_p(out.ctypes.data)
        next_, ms = C.c_int32(0), C.c_double(0.0)
        self._chk(self.L.nxs_regrid_interp_nodes(self.h, dst, src, M, N, _abi.dptr(x_interp), _abi.dptr(y_interp), x_interp.size, int(bool(isdefault)),
                                                 float(defaultvalue), flags, C.byref(next_), C.byref(ms)))
        if return_info:
            return out, dict(last_info(), num_exterior=next_.value, kernel_ms=ms.value, timing=last_timing())
        return out

    def remap_elements(self, interp_in, index_new, x_new, y_new, previous_numbering=None, n_geom_vertices=0, nec_old=None, ec_old=None,
                       in_device=None, out_device=None, return_info=False, out=None):
        """ConservativeRemappingMeshToMesh.  out = an array of the result's shape to fill (a caller that regrids repeatedly keeps it: a fresh
        370 MB array costs its page faults inside the copy back).  in_device = (device pointer, nb_var) instead of `interp_in`; out_device = a device pointer for the result."""
        f64 = lambda a: np.ascontiguousarray(a, np.float64)  # noqa: E731
        index_new = np.ascontiguousarray(index_new, np.int32).ravel()
        x_new, y_new = f64(x_new), f64(y_new)
        flags = 0
        if in_device is not None:
            dptr, nv = in_device
            src = C.c_void_p(dptr); flags |= self.IN_DEVICE
        else:
            interp_in = f64(interp_in)
            if interp_in.ndim == 1:
                interp_in = interp_in[:, None]
            nv = interp_in.shape[1]
            src = C.c_void_p(interp_in.ctypes.data)
        ne_new = index_new.size // 3
        if out_device is not None:
            out = None
            dst = C.c_void_p(out_device); flags |= self.OUT_DEVICE
        else:
            out = _result_array(out, (ne_new, nv))
            dst = C.c_void_p(out.ctypes.data)
        visits = np.zeros(ne_new, np.int32)
        prev = None if previous_numbering is None else f64(previous_numbering)
        nec = None if nec_old is None else f64(nec_old)
        ec = None if ec_old is None else f64(ec_old)
        nfail, ms = C.c_int32(0), C.c_double(0.0)
        self._chk(self.L.nxs_regrid_remap_elements(self.h, dst, src, nv, None if nec is None else _abi.dptr(nec), 0 if nec is None else nec.shape[1],
                                                   None if ec is None else _abi.dptr(ec), _abi.iptr(index_new), _abi.dptr(x_new), _abi.dptr(y_new), x_new.size, ne_new,
                                                   None if prev is None else _abi.dptr(prev), int(n_geom_vertices), flags, C.byref(nfail), _abi.iptr(visits), C.byref(ms)))
        if return_info:
            return out, {"num_failed": nfail.value, "visits": visits, "kernel_ms": ms.value, "timing": last_timing()}
        return out

    def debug_table(self, which: int) -> np.ndarray:
        """0 bucket-grid offsets, 1 its lists, 2 NodalElementConnectivity, 3 ElementConnectivity -- as the device built them (ints, -1 = NaN)."""
        n = C.c_int64(0)
        self._chk(self.L.nxs_regrid_debug_tables(self.h, which, None, 0, C.byref(n)))
        out = np.empty(max(n.value, 1), np.int32)
        self._chk(self.L.nxs_regrid_debug_tables(self.h, which, _abi.iptr(out), out.size, C.byref(n)))
        return out[:n.value]


TRIANGLE, BILINEAR, NEAREST = 0, 1, 2


def InterpFromGridToMeshx(x_in, y_in, data, x_mesh, y_mesh, default_value=1e8, interp=BILINEAR, row_major=False, device=0, return_info=False):
    """Structured grid -> mesh nodes (forcing ingest), argument meaning of contrib/bamg's InterpFromGridToMeshx
    (model/externaldata.cpp:1436): data [M, N, N_data] ([N, M, N_data] when row_major), x_in / y_in centres or contours."""
    L = _lib()
    if not hasattr(L, "_g2m_declared"):
        D = _abi.c_double_p
        L.nxs_interp_grid_to_mesh.argtypes = [D, D, C.c_int32, D, C.c_int32, D, C.c_int32, C.c_int32, C.c_int32, D, D, C.c_int32, C.c_double, C.c_int32,
                                              C.c_int32, C.c_int32, C.POINTER(C.c_double)]
        L.nxs_interp_grid_to_mesh.restype = C.c_int
        L._g2m_declared = True
    f64 = lambda a: np.ascontiguousarray(a, np.float64)  # noqa: E731
    x_in, y_in, data, xm, ym = f64(x_in), f64(y_in), f64(data), f64(x_mesh), f64(y_mesh)
    if data.ndim == 2:
        data = data[:, :, None]
    M, N = (data.shape[1], data.shape[0]) if row_major else (data.shape[0], data.shape[1])
    out = np.empty((xm.size, data.shape[2]))
    ms = C.c_double(0.0)
    rc = L.nxs_interp_grid_to_mesh(_abi.dptr(out), _abi.dptr(x_in), x_in.size, _abi.dptr(y_in), y_in.size, _abi.dptr(data), M, N, data.shape[2],
                                   _abi.dptr(xm), _abi.dptr(ym), xm.size, float(default_value), int(interp), int(bool(row_major)), device, C.byref(ms))
    if rc:
        raise NxsError(rc, (L.nxs_interp_last_error() or b"").decode())
    return (out, {"kernel_ms": ms.value}) if return_info else out
