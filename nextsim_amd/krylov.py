"""EXTENSION: coloured CSR assembly + CG kernels of include/nxs_krylov.h.
Not part of the reference's live path; parity unpinned (see the header)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from .dynamics import NxsError, load_library

KRYLOV_EXPORTS = ("nxs_fem_csr_pattern", "nxs_fem_colour_elements", "nxs_fem_poisson_solve", "nxs_krylov_solve", "nxs_krylov_last_error")
CG, BICGSTAB = 0, 1
_decl = False


def _lib():
    global _decl
    L = load_library()
    if not _decl:
        P = C.POINTER
        L.nxs_fem_csr_pattern.argtypes = [_abi.c_int32_p, C.c_int32, C.c_int32, _abi.c_int32_p, _abi.c_int32_p, P(C.c_int64)]
        L.nxs_fem_colour_elements.argtypes = [_abi.c_int32_p, C.c_int32, C.c_int32, _abi.c_int32_p, P(C.c_int32)]
        L.nxs_fem_poisson_solve.argtypes = [_abi.c_int32_p, _abi.c_double_p, _abi.c_double_p, C.c_int32, C.c_int32, _abi.c_uint8_p,
                                            _abi.c_double_p, _abi.c_double_p, C.c_double, C.c_int32, C.c_int32, P(C.c_int32),
                                            P(C.c_double), P(C.c_double), P(C.c_double)]
        L.nxs_krylov_solve.argtypes = [C.c_int32, _abi.c_int32_p, _abi.c_int32_p, _abi.c_double_p, _abi.c_double_p, _abi.c_double_p, C.c_int32,
                                       C.c_double, C.c_int32, C.c_int32, P(C.c_int32), P(C.c_double), P(C.c_double)]
        L.nxs_krylov_last_error.restype = C.c_char_p
        _decl = True
    return L


def _chk(L, rc):
    if rc:
        raise NxsError(rc, (L.nxs_krylov_last_error() or b"").decode())


def csr_pattern(indices, num_nodes):
    L = _lib()
    idx = np.ascontiguousarray(indices, np.int32).ravel()
    rowptr = np.empty(num_nodes + 1, np.int32)
    nnz = C.c_int64()
    _chk(L, L.nxs_fem_csr_pattern(_abi.iptr(idx), num_nodes, idx.size // 3, _abi.iptr(rowptr), None, C.byref(nnz)))
    colidx = np.empty(nnz.value, np.int32)
    _chk(L, L.nxs_fem_csr_pattern(_abi.iptr(idx), num_nodes, idx.size // 3, _abi.iptr(rowptr), _abi.iptr(colidx), C.byref(nnz)))
    return rowptr, colidx


def colour_elements(indices, num_nodes):
    L = _lib()
    idx = np.ascontiguousarray(indices, np.int32).ravel()
    col = np.empty(idx.size // 3, np.int32)
    n = C.c_int32()
    _chk(L, L.nxs_fem_colour_elements(_abi.iptr(idx), num_nodes, idx.size // 3, _abi.iptr(col), C.byref(n)))
    return col, n.value


def poisson_solve(indices, x, y, dirichlet, f_elem, rtol=1e-10, max_iter=20000, device=0):
    L = _lib()
    idx = np.ascontiguousarray(indices, np.int32).ravel()
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    d = np.ascontiguousarray(dirichlet, np.uint8); f = np.ascontiguousarray(f_elem, np.float64)
    u = np.empty(x.size)
    it, res, ma, ms = C.c_int32(), C.c_double(), C.c_double(), C.c_double()
    _chk(L, L.nxs_fem_poisson_solve(_abi.iptr(idx), _abi.dptr(x), _abi.dptr(y), x.size, idx.size // 3, _abi.bptr(d), _abi.dptr(f), _abi.dptr(u),
                                    rtol, max_iter, device, C.byref(it), C.byref(res), C.byref(ma), C.byref(ms)))
    return u, {"iterations": it.value, "rel_residual": res.value, "ms_assembly": ma.value, "ms_solve": ms.value}


def solve(rowptr, colidx, val, b, method=CG, rtol=1e-10, max_iter=20000, device=0):
    """A x = b for a CSR matrix (nxs_krylov_solve): returns x and {'iterations', 'rel_residual', 'ms_solve'}."""
    L = _lib()
    rp = np.ascontiguousarray(rowptr, np.int32); ci = np.ascontiguousarray(colidx, np.int32)
    v = np.ascontiguousarray(val, np.float64); b = np.ascontiguousarray(b, np.float64)
    x = np.empty(b.size)
    it, res, ms = C.c_int32(), C.c_double(), C.c_double()
    _chk(L, L.nxs_krylov_solve(b.size, _abi.iptr(rp), _abi.iptr(ci), _abi.dptr(v), _abi.dptr(b), _abi.dptr(x), int(method), float(rtol),
                               int(max_iter), device, C.byref(it), C.byref(res), C.byref(ms)))
    return x, {"iterations": it.value, "rel_residual": res.value, "ms_solve": ms.value}
