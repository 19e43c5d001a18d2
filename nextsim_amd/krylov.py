"""EXTENSION: coloured CSR assembly + CG kernels of include/nxs_krylov.h.
Not part of the reference's live path; parity unpinned (see the header)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from .dynamics import NxsError, load_library

KRYLOV_EXPORTS = ("nxs_fem_csr_pattern", "nxs_fem_colour_elements", "nxs_fem_poisson_solve", "nxs_krylov_solve", "nxs_krylov_last_error",
                  "nxs_krylov_create", "nxs_krylov_destroy", "nxs_krylov_set_matrix", "nxs_krylov_set_halo", "nxs_krylov_comm_init",
                  "nxs_krylov_set_comm_fns", "nxs_krylov_spmv", "nxs_krylov_run", "nxs_krylov_info", "nxs_krylov_comm_stats")
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32)
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
        L.nxs_krylov_create.argtypes = [C.c_int32, P(C.c_void_p)]
        L.nxs_krylov_destroy.argtypes = [C.c_void_p]
        L.nxs_krylov_destroy.restype = None
        L.nxs_krylov_set_matrix.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _abi.c_int32_p, _abi.c_int32_p, _abi.c_double_p]
        L.nxs_krylov_set_halo.argtypes = [C.c_void_p, C.c_void_p]
        L.nxs_krylov_comm_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32]
        L.nxs_krylov_set_comm_fns.argtypes = [C.c_void_p, EXCHANGE_FN, ALLREDUCE_FN, C.c_void_p]
        L.nxs_krylov_spmv.argtypes = [C.c_void_p, _abi.c_double_p, _abi.c_double_p, C.c_int32, P(C.c_double)]
        L.nxs_krylov_run.argtypes = [C.c_void_p, _abi.c_double_p, _abi.c_double_p, C.c_int32, C.c_double, C.c_int32, P(C.c_int32), P(C.c_double), P(C.c_double)]
        L.nxs_krylov_info.argtypes = [C.c_void_p, P(C.c_int64), P(C.c_int64), P(C.c_int64)]
        L.nxs_krylov_comm_stats.argtypes = [C.c_void_p, P(C.c_int64), P(C.c_int64)]
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


class Solver:
    """The handle API of include/nxs_krylov.h: a resident matrix (sliced ELLPACK on the device), optionally one block of
    rows of a matrix distributed over ranks (own rows first, ghost columns behind them)."""

    def __init__(self, device=0):
        self.L = _lib()
        h = C.c_void_p()
        _chk(self.L, self.L.nxs_krylov_create(device, C.byref(h)))
        self.h = h
        self.n = 0
        self._keep = []

    def close(self):
        if self.h:
            self.L.nxs_krylov_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def set_matrix(self, rowptr, colidx, val, n_cols=None):
        rp = np.ascontiguousarray(rowptr, np.int32); ci = np.ascontiguousarray(colidx, np.int32)
        v = np.ascontiguousarray(val, np.float64)
        self.n = rp.size - 1
        _chk(self.L, self.L.nxs_krylov_set_matrix(self.h, self.n, self.n if n_cols is None else int(n_cols), _abi.iptr(rp), _abi.iptr(ci), _abi.dptr(v)))

    def set_halo(self, lm):
        """lm: anything with the halo lists of nextsim_amd.mesh.LocalMesh (rank, nranks, send_*/recv_*), one dof per entry."""
        hs = _abi.halo_struct(lm)
        self._keep.append(hs)
        _chk(self.L, self.L.nxs_krylov_set_halo(self.h, C.byref(hs)))

    def comm_init(self, unique_id: bytes, rank: int, nranks: int):
        _chk(self.L, self.L.nxs_krylov_comm_init(self.h, unique_id, rank, nranks))

    def set_comm_fns(self, exchange, allreduce, n_send, n_recv):
        """exchange(send: ndarray, recv: ndarray) fills recv; allreduce(vals: ndarray) sums in place over the ranks."""
        def _ex(_user, send, recv):
            try:
                exchange(np.ctypeslib.as_array(send, shape=(max(n_send, 1),))[:n_send], np.ctypeslib.as_array(recv, shape=(max(n_recv, 1),))[:n_recv])
                return 0
            except Exception:  # noqa: BLE001
                return 1

        def _ar(_user, vals, n):
            try:
                allreduce(np.ctypeslib.as_array(vals, shape=(n,)))
                return 0
            except Exception:  # noqa: BLE001
                return 1
        self._fns = (EXCHANGE_FN(_ex), ALLREDUCE_FN(_ar))
        _chk(self.L, self.L.nxs_krylov_set_comm_fns(self.h, self._fns[0], self._fns[1], None))

    def spmv(self, x, reps=1):
        x = np.ascontiguousarray(x, np.float64)
        if self.n and x.size != self.n:
            raise ValueError(f"operand has {x.size} entries, the matrix {self.n} rows")
        out = np.empty(max(self.n, 1))
        ms = C.c_double()
        _chk(self.L, self.L.nxs_krylov_spmv(self.h, _abi.dptr(x), _abi.dptr(out), reps, C.byref(ms)))
        return out, ms.value

    def solve(self, b, method=CG, rtol=1e-10, max_iter=20000):
        b = np.ascontiguousarray(b, np.float64)
        if self.n and b.size != self.n:
            raise ValueError(f"right-hand side has {b.size} entries, the matrix {self.n} rows")
        x = np.empty(max(self.n, 1))
        it, res, ms = C.c_int32(), C.c_double(), C.c_double()
        _chk(self.L, self.L.nxs_krylov_run(self.h, _abi.dptr(b), _abi.dptr(x), int(method), float(rtol), int(max_iter), C.byref(it), C.byref(res), C.byref(ms)))
        return x, {"iterations": it.value, "rel_residual": res.value, "ms_solve": ms.value}

    def comm_stats(self):
        a, b = C.c_int64(), C.c_int64()
        _chk(self.L, self.L.nxs_krylov_comm_stats(self.h, C.byref(a), C.byref(b)))
        return {"rccl_allreduces": a.value, "rccl_exchanges": b.value}

    def info(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _chk(self.L, self.L.nxs_krylov_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"nnz": a.value, "stored_entries": b.value, "spmv_bytes": c.value}


def localize_system(rowptr, colidx, val, lm):
    """One rank's block of a node-numbered global CSR matrix (one dof per node) on the partition `lm`
    (nextsim_amd.mesh.LocalMesh): the rows of its own nodes, columns in local numbering -- own nodes first, ghosts behind
    them, so that lm's halo lists are the exchange of the SpMV operand.  Every own node has its complete element fan
    locally (core/src/gmshmesh.cpp:856-1498), hence all the columns of its row.  Host-side set-up (numpy)."""
    rowptr = np.asarray(rowptr); colidx = np.asarray(colidx); val = np.asarray(val)
    gid = np.asarray(lm.node_gid)
    n_own = int(lm.local_ndof)
    g2l = np.full(int(rowptr.size - 1), -1, np.int64)
    g2l[gid] = np.arange(gid.size)
    rows = gid[:n_own]
    cnt = rowptr[rows + 1] - rowptr[rows]
    rp = np.zeros(n_own + 1, np.int32)
    rp[1:] = np.cumsum(cnt)
    take = np.concatenate([np.arange(rowptr[r], rowptr[r + 1]) for r in rows]) if n_own else np.zeros(0, np.int64)
    ci = g2l[colidx[take]]
    if (ci < 0).any():
        raise ValueError("a row of an own node has a column outside the partition's node set")
    return rp, ci.astype(np.int32), val[take].astype(np.float64), int(gid.size)
