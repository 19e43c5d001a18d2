"""Python door into the CPU oracle (oracle/liboracle*.so) -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, nowhere
else: the product path (nextsim_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from nextsim_amd import _abi

HERE = os.path.dirname(os.path.abspath(__file__))


class Work(C.Structure):
    _fields_ = [("Nn", C.c_int32), ("Ne", C.c_int32), ("Dunit", C.c_double * 9)] + [
        (k, _abi.c_double_p) for k in (
            "delta_x", "surface", "shape_coeff", "B0T", "element_mass", "rlmass_matrix", "node_mass",
            "C_bu", "grad_ssh", "grad_terms", "fcor", "VTM", "tmp", "D_tau_a", "D_tau_w",
            "D_del_ci_ridge_myi")] + [("trace", C.POINTER(C.c_uint64))]


GHOST_FN = C.CFUNCTYPE(None, C.c_void_p, _abi.c_double_p)


def build(fast: bool = False) -> str:
    if os.environ.get("NXS_ORACLE_LIBRARY"):   # another build of the same source (tests/test_sanitizers.py: ASan/UBSan)
        return os.environ["NXS_ORACLE_LIBRARY"]
    name = "liboracle_fast.so" if fast else "liboracle.so"
    path = os.path.join(HERE, name)
    src = os.path.join(HERE, "dyn_ref.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, name])
    return path


_libs: dict = {}


def lib(fast: bool = False):
    if fast in _libs:
        return _libs[fast]
    L = C.CDLL(build(fast))
    P = C.POINTER
    Mp, Pp, Sp, Fp, Wp = P(_abi.Mesh), P(_abi.Params), P(_abi.State), P(_abi.Forcing), P(Work)
    L.ref_work_create.restype = Wp
    L.ref_work_create.argtypes = [C.c_int32, C.c_int32]
    L.ref_work_destroy.argtypes = [Wp]
    L.ref_work_enable_trace.argtypes = [Wp]
    L.ref_work_enable_trace.restype = C.c_int
    L.ref_default_params.argtypes = [Pp]
    L.ref_prep.argtypes = [Mp, Pp, Sp, Fp, Wp]
    L.ref_update_sigma_damage.argtypes = [Mp, Pp, Sp, Wp, C.c_double]
    L.ref_update_sigma_vp.argtypes = [Mp, Pp, Sp, Wp, C.c_double, C.c_double]
    L.ref_substep_solve.argtypes = [Mp, Pp, Sp, Fp, Wp]
    L.ref_move_mesh.argtypes = [Mp, Sp, Wp, C.c_double]
    L.ref_smoother_sweep.argtypes = [Mp, Sp, Wp]
    L.ref_ow_tail.argtypes = [Mp, Pp, Sp, Fp, Wp]
    L.ref_explicit_solve.argtypes = [Mp, Pp, Sp, Fp, Wp, C.c_void_p, C.c_void_p]
    L.ref_update.argtypes = [Mp, Pp, Sp, Wp]
    L.ref_free_drift.argtypes = [Mp, Pp, Sp, Fp]
    L.ref_step.argtypes = [Mp, Pp, Sp, Fp, Wp, C.c_void_p, C.c_void_p]
    L.ref_check_regridding.argtypes = [Mp, Pp, Sp, P(C.c_double), P(C.c_int32)]
    L.ref_check_regridding.restype = C.c_int
    L.ref_check_fields_fast.argtypes = [Mp, Pp, Sp]
    L.ref_check_fields_fast.restype = C.c_int
    L.ref_ghosts_pack.argtypes = [P(_abi.Halo), C.c_int32, _abi.c_double_p, C.c_int, _abi.c_double_p]
    L.ref_ghosts_unpack.argtypes = [P(_abi.Halo), C.c_int32, _abi.c_double_p, C.c_int, _abi.c_double_p]
    L.ref_mesh_connectivity.argtypes = [_abi.c_int32_p, C.c_int32, C.c_int32, P(C.c_int32), _abi.c_double_p,
                                        P(C.c_int32), _abi.c_double_p]
    L.ref_mesh_connectivity.restype = C.c_int
    L.ref_ice_diagnostics.argtypes = [Mp, Pp, Sp] + [_abi.c_double_p] * 6
    L.ref_multirank_steps.argtypes = [C.c_int, P(Mp), Pp, P(Sp), P(Fp), P(Wp), P(P(_abi.Halo)), C.c_int, C.c_int]
    L.ref_multirank_steps.restype = C.c_int
    _libs[fast] = L
    return L


def connectivity(indices: np.ndarray, num_nodes: int):
    """(nec, nc) double tables in bamg layout from the oracle's restatement of Mesh::WriteMesh."""
    L = lib()
    ne = indices.size // 3
    w1, w2 = C.c_int32(), C.c_int32()
    rc = L.ref_mesh_connectivity(_abi.iptr(indices), num_nodes, ne, C.byref(w1), None, C.byref(w2), None)
    assert rc == 0
    nec = np.empty((num_nodes, w1.value)); nc = np.empty((num_nodes, w2.value))
    rc = L.ref_mesh_connectivity(_abi.iptr(indices), num_nodes, ne, C.byref(w1), _abi.dptr(nec), C.byref(w2), _abi.dptr(nc))
    assert rc == 0
    return nec, nc


def bamg_shim():
    """The REAL contrib/bamg behind oracle/_ref/libbamg_shim.so, or None when it was never built
    (it is built from /root/reference, which only exists in the build container)."""
    path = os.path.join(HERE, "_ref", "libbamg_shim.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    P = C.POINTER
    L.shim_bamg_connectivity.argtypes = [P(C.c_int), _abi.c_double_p, _abi.c_double_p, C.c_int, C.c_int,
                                         P(C.c_int), _abi.c_double_p, P(C.c_int), _abi.c_double_p]
    L.shim_bamg_connectivity.restype = C.c_int
    L.shim_bamg_interp_mesh_to_mesh.argtypes = [P(C.c_int), _abi.c_double_p, _abi.c_double_p, C.c_int, C.c_int,
                                                _abi.c_double_p, C.c_int, C.c_int, _abi.c_double_p, _abi.c_double_p,
                                                C.c_int, C.c_int, C.c_double, _abi.c_double_p]
    L.shim_bamg_interp_mesh_to_mesh.restype = C.c_int
    return L


def bamg_connectivity(indices: np.ndarray, x: np.ndarray, y: np.ndarray):
    L = bamg_shim()
    assert L is not None
    nn, ne = x.size, indices.size // 3
    idx = np.ascontiguousarray(indices.astype(np.intc))
    w1, w2 = C.c_int(), C.c_int()
    ip = idx.ctypes.data_as(C.POINTER(C.c_int))
    assert L.shim_bamg_connectivity(ip, _abi.dptr(x), _abi.dptr(y), nn, ne, C.byref(w1), None, C.byref(w2), None) == 0
    nec = np.empty((nn, w1.value)); nc = np.empty((nn, w2.value))
    assert L.shim_bamg_connectivity(ip, _abi.dptr(x), _abi.dptr(y), nn, ne, C.byref(w1), _abi.dptr(nec), C.byref(w2), _abi.dptr(nc)) == 0
    return nec, nc


class OracleRank:
    """One rank of the restated reference: owns copies of the state arrays and a ref_work."""

    def __init__(self, lm, params: _abi.Params, fields: dict, tables=None, fast: bool = False):
        self.L = lib(fast)
        self.lm = lm
        self.params = params.copy()
        self.arr = {k: np.array(v, dtype=np.float64, copy=True) for k, v in fields.items()}
        self.tables = tables if tables is not None else connectivity(lm.indices, lm.num_nodes)
        self.mesh = _abi.mesh_struct(lm, self.tables)
        self.state = _abi.state_struct(self.arr)
        self.forcing = _abi.forcing_struct(self.arr)
        self.halo = _abi.halo_struct(lm)
        self.work = self.L.ref_work_create(lm.num_nodes, lm.num_elements)

    def __del__(self):
        try:
            self.L.ref_work_destroy(self.work)
        except Exception:
            pass

    def _a(self):
        return C.byref(self.mesh), C.byref(self.params), C.byref(self.state), C.byref(self.forcing), self.work

    # whole functions (single rank: no ghosts callback)
    def step(self):
        m, p, s, f, w = self._a()
        self.L.ref_step(m, p, s, f, w, None, None)

    def explicit_solve(self):
        m, p, s, f, w = self._a()
        self.L.ref_explicit_solve(m, p, s, f, w, None, None)

    def update(self):
        m, p, s, f, w = self._a()
        self.L.ref_update(m, p, s, w)

    # phases
    def prep(self):
        m, p, s, f, w = self._a(); self.L.ref_prep(m, p, s, f, w)

    def substep_solve(self):
        m, p, s, f, w = self._a(); self.L.ref_substep_solve(m, p, s, f, w)

    def move_mesh(self, dt):
        m, p, s, f, w = self._a(); self.L.ref_move_mesh(m, s, w, dt)

    def smoother_sweep(self):
        m, p, s, f, w = self._a(); self.L.ref_smoother_sweep(m, s, w)

    def ow_tail(self):
        m, p, s, f, w = self._a(); self.L.ref_ow_tail(m, p, s, f, w)

    def update_sigma_damage(self, dt):
        m, p, s, f, w = self._a(); self.L.ref_update_sigma_damage(m, p, s, w, dt)

    def ice_diagnostics(self) -> dict:
        """updateIceDiagnostics(), FE.cpp:7860-7905 (without D_tsurf / FSD)."""
        out = {k: np.empty(self.lm.num_elements) for k in _abi.ICE_DIAG}
        self.L.ref_ice_diagnostics(C.byref(self.mesh), C.byref(self.params), C.byref(self.state), *[_abi.dptr(out[k]) for k in _abi.ICE_DIAG])
        return out

    def check_regridding(self):
        m, p, s, f, w = self._a()
        ang, flip = C.c_double(), C.c_int32()
        r = self.L.ref_check_regridding(m, p, s, C.byref(ang), C.byref(flip))
        return ang.value, flip.value, r

    def check_fields_fast(self):
        m, p, s, f, w = self._a()
        return self.L.ref_check_fields_fast(m, p, s)

    def enable_branch_trace(self):
        """Start (or restart) the per-element branch trace of updateSigmaDamage (ref_work.trace in dyn_ref.h)."""
        assert self.L.ref_work_enable_trace(self.work) == 0

    def branch_trace(self) -> dict:
        """{'hash', 'damage_substeps', 'flags', 'substeps'}: one uint64 per element each."""
        t = np.ctypeslib.as_array(self.work.contents.trace, shape=(self.lm.num_elements, 4)).copy()
        return {"hash": t[:, 0], "damage_substeps": t[:, 1], "flags": t[:, 2], "substeps": t[:, 3]}

    def work_array(self, name: str, n: int) -> np.ndarray:
        ptr = getattr(self.work.contents, name)
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()

    def pack(self, k: int) -> np.ndarray:
        n = int(self.lm.send_offsets[k + 1] - self.lm.send_offsets[k])
        buf = np.empty(2 * n)
        self.L.ref_ghosts_pack(C.byref(self.halo), self.lm.num_nodes, _abi.dptr(self.arr["VT"]), k, _abi.dptr(buf))
        return buf

    def unpack(self, k: int, buf: np.ndarray):
        self.L.ref_ghosts_unpack(C.byref(self.halo), self.lm.num_nodes, _abi.dptr(self.arr["VT"]), k, _abi.dptr(buf))


def exchange_ghosts(ranks: list):
    """updateGhosts(M_VT) across in-process oracle ranks (FE.cpp:13963-13996)."""
    bufs = {}
    for r in ranks:
        for k, q in enumerate(r.lm.send_procs):
            bufs[(r.lm.rank, int(q))] = r.pack(k)
    for r in ranks:
        for k, q in enumerate(r.lm.recv_procs):
            r.unpack(k, bufs[(int(q), r.lm.rank)])


def multirank_step(ranks: list):
    """step() on P in-process ranks in lock-step (explicitSolve with real halo exchanges + update)."""
    p = ranks[0].params
    steps = p.substeps
    dte = p.dtime_step / float(steps)
    for r in ranks:
        r.prep()
    for _ in range(steps):
        for r in ranks:
            r.substep_solve()
        exchange_ghosts(ranks)
        if p.dynamics_type != _abi.NXS_DYN_MEVP:
            for r in ranks:
                r.move_mesh(dte)
    if p.dynamics_type == _abi.NXS_DYN_MEVP:
        for r in ranks:
            r.move_mesh(p.dtime_step)
    for _ in range(50):
        for r in ranks:
            r.smoother_sweep()
        exchange_ghosts(ranks)
    for r in ranks:
        r.ow_tail()
        r.update()


def bamg_completed_mesh(index, x, y):
    """The REAL bamg's reconstructed mesh (Mesh(index, x, y, ...), what InterpFromMeshToMesh2dx locates its points in): every
    triangle in bamg's order and vertex order, 0-based, -1 = the NULL vertex of a boundary triangle; and TriangleReferenceList."""
    L = C.CDLL(os.path.join(HERE, "_ref", "libbamg_shim.so"))
    idx = np.ascontiguousarray(np.asarray(index).ravel().astype(np.intc))
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    cap = 4 * (idx.size // 3) + 64
    tri = np.zeros(3 * cap, np.intc); reft = np.zeros(cap, np.int64)
    L.shim_bamg_completed_mesh.restype = C.c_int
    L.shim_bamg_completed_mesh.argtypes = [C.POINTER(C.c_int), _abi.c_double_p, _abi.c_double_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_long)]
    n = L.shim_bamg_completed_mesh(idx.ctypes.data_as(C.POINTER(C.c_int)), _abi.dptr(x), _abi.dptr(y), x.size, idx.size // 3, cap,
                                   tri.ctypes.data_as(C.POINTER(C.c_int)), reft.ctypes.data_as(C.POINTER(C.c_long)))
    assert n >= 0, n
    return tri[:3 * n].reshape(-1, 3).copy(), reft[:n].copy()


def bamg_interp_mesh_to_mesh(index_data, x_data, y_data, data, x_interp, y_interp, isdefault=False, defaultvalue=1e-24):
    """The REAL InterpFromMeshToMesh2dx (contrib/bamg/src/InterpFromMeshToMesh2dx.cpp) through the shim."""
    L = bamg_shim()
    assert L is not None
    idx = np.ascontiguousarray(np.asarray(index_data).ravel().astype(np.intc))
    x_data = np.ascontiguousarray(x_data, np.float64); y_data = np.ascontiguousarray(y_data, np.float64)
    data = np.ascontiguousarray(data, np.float64)
    if data.ndim == 1:
        data = data[:, None]
    xi = np.ascontiguousarray(x_interp, np.float64); yi = np.ascontiguousarray(y_interp, np.float64)
    out = np.empty((xi.size, data.shape[1]))
    rc = L.shim_bamg_interp_mesh_to_mesh(idx.ctypes.data_as(C.POINTER(C.c_int)), _abi.dptr(x_data), _abi.dptr(y_data), x_data.size,
                                         idx.size // 3, _abi.dptr(data), data.shape[0], data.shape[1], _abi.dptr(xi), _abi.dptr(yi),
                                         xi.size, int(bool(isdefault)), float(defaultvalue), _abi.dptr(out))
    assert rc == 0
    return out


def bamg_interp_mesh_to_grid(index_mesh, x_mesh, y_mesh, data, xmin, ymax, xposting, yposting, nrows, ncols, default_value):
    """The REAL InterpFromMeshToGridx (contrib/bamg/src/InterpFromMeshToGridx.cpp) through the shim."""
    path = os.path.join(HERE, "_ref", "libbamg_shim.so")
    L = C.CDLL(path)
    idx = np.ascontiguousarray(np.asarray(index_mesh).ravel().astype(np.intc))
    x_mesh = np.ascontiguousarray(x_mesh, np.float64); y_mesh = np.ascontiguousarray(y_mesh, np.float64)
    data = np.ascontiguousarray(data, np.float64)
    if data.ndim == 1:
        data = data[:, None]
    out = np.empty((nrows, ncols, data.shape[1]))
    L.shim_bamg_interp_mesh_to_grid.argtypes = [C.POINTER(C.c_int), _abi.c_double_p, _abi.c_double_p, C.c_int, C.c_int, _abi.c_double_p,
                                                C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                                C.c_double, _abi.c_double_p]
    rc = L.shim_bamg_interp_mesh_to_grid(idx.ctypes.data_as(C.POINTER(C.c_int)), _abi.dptr(x_mesh), _abi.dptr(y_mesh), x_mesh.size,
                                         idx.size // 3, _abi.dptr(data), data.shape[0], data.shape[1], float(xmin), float(ymax),
                                         float(xposting), float(yposting), int(nrows), int(ncols), float(default_value), _abi.dptr(out))
    assert rc == 0
    return out


def multirank_step_threaded(ranks: list, pool) -> None:
    """multirank_step with the per-rank phases run concurrently on `pool` (a ThreadPoolExecutor; ctypes
    releases the GIL): one mesh partition per host core, halos exchanged through shared memory between the
    phases -- the CPU analogue of the reference's MPI run, used by bench.py's cpu_baseline."""
    p = ranks[0].params
    steps = p.substeps
    dte = p.dtime_step / float(steps)

    def run(fn):
        list(pool.map(fn, ranks))

    run(lambda r: r.prep())
    for _ in range(steps):
        run(lambda r: r.substep_solve())
        exchange_ghosts(ranks)
        if p.dynamics_type != _abi.NXS_DYN_MEVP:
            run(lambda r: r.move_mesh(dte))
    if p.dynamics_type == _abi.NXS_DYN_MEVP:
        run(lambda r: r.move_mesh(p.dtime_step))
    for _ in range(50):
        run(lambda r: r.smoother_sweep())
        exchange_ghosts(ranks)
    run(lambda r: (r.ow_tail(), r.update()))


def multirank_steps_native(ranks: list, nsteps: int = 1, nthreads: int = 0) -> None:
    """`nsteps` lock-step steps of the in-process ranks on `nthreads` host threads inside the oracle library itself
    (ref_multirank_steps: pthreads + barriers, shared-memory updateGhosts) -- what multirank_step does, without a Python call per
    phase and rank: the cpu_baseline of bench.py at the core counts of a GPU box."""
    n = len(ranks)
    P = C.POINTER
    Mp, Sp, Fp, Wp, Hp = P(_abi.Mesh), P(_abi.State), P(_abi.Forcing), P(Work), P(_abi.Halo)
    m = (Mp * n)(*[C.pointer(r.mesh) for r in ranks])
    s = (Sp * n)(*[C.pointer(r.state) for r in ranks])
    f = (Fp * n)(*[C.pointer(r.forcing) for r in ranks])
    w = (Wp * n)(*[r.work for r in ranks])
    h = (Hp * n)(*[C.pointer(r.halo) for r in ranks])
    rc = ranks[0].L.ref_multirank_steps(n, m, C.byref(ranks[0].params), s, f, w, h, nsteps, nthreads or n)
    if rc != 0:
        raise RuntimeError(f"ref_multirank_steps failed ({rc})")


class MultirankContext:
    """ref_mr_create / ref_mr_run / ref_mr_destroy: the lock-step run of the in-process ranks on threads that are KEPT, PINNED (physical cores first, the
    sockets in turn) and that have first touched their partitions' arrays -- the CPU baseline of bench.py as an MPI run of the reference would use the host.
    run(n): n steps (what is timed); close(): the state back into the ranks' arrays.  Bit for bit multirank_steps_native."""

    def __init__(self, ranks: list, nthreads: int = 0, pin: bool = True):
        n = len(ranks)
        P = C.POINTER
        Mp, Sp, Fp, Hp = P(_abi.Mesh), P(_abi.State), P(_abi.Forcing), P(_abi.Halo)
        self._keep = ranks   # (the context reads the ranks' structs while it is made and writes their arrays when it is closed)
        self._m = (Mp * n)(*[C.pointer(r.mesh) for r in ranks]); self._s = (Sp * n)(*[C.pointer(r.state) for r in ranks])
        self._f = (Fp * n)(*[C.pointer(r.forcing) for r in ranks]); self._h = (Hp * n)(*[C.pointer(r.halo) for r in ranks])
        self.L = ranks[0].L
        self.L.ref_mr_create.restype = C.c_void_p
        self.L.ref_mr_create.argtypes = [C.c_int, P(Mp), P(_abi.Params), P(Sp), P(Fp), P(Hp), C.c_int, C.c_int]
        self.L.ref_mr_run.argtypes = [C.c_void_p, C.c_int]; self.L.ref_mr_destroy.argtypes = [C.c_void_p, C.c_int]; self.L.ref_mr_destroy.restype = None
        self.L.ref_mr_info.argtypes = [C.c_void_p, P(C.c_int), P(C.c_int), P(C.c_int), C.c_int]
        self.ctx = self.L.ref_mr_create(n, self._m, C.byref(ranks[0].params), self._s, self._f, self._h, nthreads or n, 1 if pin else 0)
        if not self.ctx:
            raise RuntimeError("ref_mr_create failed (inconsistent halo lists, memory or threads)")

    def run(self, nsteps: int = 1):
        if self.L.ref_mr_run(self.ctx, nsteps) != 0:
            raise RuntimeError("ref_mr_run failed")

    def configure(self, barrier: str = "spin", pin: bool = True):
        """Between runs: the barrier the phases meet at ("spin": sense-reversing spin barrier; "sleep": pthread_barrier_t, the waiting threads sleep) and whether
        the threads are pinned -- the same partitions, first touched as they were."""
        self.L.ref_mr_configure.argtypes = [C.c_void_p, C.c_int, C.c_int]
        if self.L.ref_mr_configure(self.ctx, {"spin": 0, "sleep": 1}[barrier], 1 if pin else 0) != 0:
            raise RuntimeError("ref_mr_configure failed")

    def info(self) -> dict:
        nt, so = C.c_int(), C.c_int()
        cpus = (C.c_int * 1024)()
        self.L.ref_mr_info(self.ctx, C.byref(nt), C.byref(so), cpus, 1024)
        return {"threads": nt.value, "sockets_used": so.value, "cpus": [cpus[i] for i in range(min(nt.value, 1024))]}

    def close(self, copy_back: bool = True):
        if self.ctx:
            self.L.ref_mr_destroy(self.ctx, 1 if copy_back else 0)
            self.ctx = None


def bamg_element_connectivity(indices, x, y):
    """bamgmesh->ElementConnectivity and ->Triangles of the REAL BamgConvertMeshx (doubles, NaN on the boundary)."""
    L = C.CDLL(os.path.join(HERE, "_ref", "libbamg_shim.so"))
    idx = np.ascontiguousarray(np.asarray(indices).ravel().astype(np.intc))
    x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
    ne = idx.size // 3
    ec = np.empty((ne, 3)); tri = np.empty((ne, 3))
    L.shim_bamg_element_connectivity.argtypes = [C.POINTER(C.c_int), _abi.c_double_p, _abi.c_double_p, C.c_int, C.c_int, _abi.c_double_p, _abi.c_double_p]
    rc = L.shim_bamg_element_connectivity(idx.ctypes.data_as(C.POINTER(C.c_int)), _abi.dptr(x), _abi.dptr(y), x.size, ne, _abi.dptr(ec), _abi.dptr(tri))
    assert rc == 0
    return ec, tri


def bamg_conservative_remap(index_old, x_old, y_old, index_new, x_new, y_new, previous_numbering, n_geom_vertices, data):
    """The REAL ConservativeRemappingMeshToMesh (contrib/bamg/src/ConservativeRemapping.cpp:176-328), as called at
    FE.cpp:3108.  previous_numbering: 1-based old vertex number of every new vertex (0 = created by the remesher)."""
    L = C.CDLL(os.path.join(HERE, "_ref", "libbamg_shim.so"))
    io = np.ascontiguousarray(np.asarray(index_old).ravel().astype(np.intc)); inw = np.ascontiguousarray(np.asarray(index_new).ravel().astype(np.intc))
    xo = np.ascontiguousarray(x_old, np.float64); yo = np.ascontiguousarray(y_old, np.float64)
    xn = np.ascontiguousarray(x_new, np.float64); yn = np.ascontiguousarray(y_new, np.float64)
    pn = np.ascontiguousarray(previous_numbering, np.float64)
    data = np.ascontiguousarray(data, np.float64)
    if data.ndim == 1:
        data = data[:, None]
    out = np.empty((inw.size // 3, data.shape[1]))
    IP = C.POINTER(C.c_int)
    L.shim_bamg_conservative_remap.argtypes = [IP, _abi.c_double_p, _abi.c_double_p, C.c_int, C.c_int, IP, _abi.c_double_p, _abi.c_double_p,
                                               C.c_int, C.c_int, _abi.c_double_p, C.c_int, _abi.c_double_p, C.c_int, _abi.c_double_p]
    rc = L.shim_bamg_conservative_remap(io.ctypes.data_as(IP), _abi.dptr(xo), _abi.dptr(yo), xo.size, io.size // 3, inw.ctypes.data_as(IP),
                                        _abi.dptr(xn), _abi.dptr(yn), xn.size, inw.size // 3, _abi.dptr(pn), int(n_geom_vertices),
                                        _abi.dptr(data), data.shape[1], _abi.dptr(out))
    assert rc == 0
    return out


def mapx_lat(x, y, mppfile="/root/reference/mesh/NpsNextsim.mpp"):
    """GmshMesh::lat() (core/src/gmshmesh.cpp:1798-1824) with the REAL contrib/mapx: inverse_mapx of every (x, y).
    Needs oracle/_ref/libmapx_ref.so AND the reference's .mpp parameter file (build container only)."""
    L = C.CDLL(os.path.join(HERE, "_ref", "libmapx_ref.so"))
    L.init_mapx.restype = C.c_void_p; L.init_mapx.argtypes = [C.c_char_p]
    L.inverse_mapx.argtypes = [C.c_void_p, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.close_mapx.argtypes = [C.c_void_p]
    m = L.init_mapx(mppfile.encode())
    assert m, "init_mapx failed"
    lat = np.empty(len(x)); a, b = C.c_double(), C.c_double()
    for i, (xi, yi) in enumerate(zip(x, y)):
        L.inverse_mapx(m, float(xi), float(yi), C.byref(a), C.byref(b))
        lat[i] = a.value
    L.close_mapx(m)
    return lat


def bamg_adapt(index, x0, y0, dirichlet_flags, x_moved, y_moved, hmin, hmax):
    """One regrid with the REAL remesher (Bamgx as FiniteElement::adaptMesh calls it, FE.cpp:3760-3801, options of
    initBamg FE.cpp:992-1038) on the mesh (index 1-based, x0, y0) whose vertices have moved to (x_moved, y_moved).
    Returns x, y, tri (0-based), PreviousNumbering (1-based, 0 = new), VerticesOnGeomVertexSize[0]."""
    L = C.CDLL(os.path.join(HERE, "_ref", "libbamg_shim.so"))
    IP, D = C.POINTER(C.c_int), _abi.c_double_p
    L.shim_bamg_adapt.argtypes = [IP, D, D, C.c_int, C.c_int, IP, C.c_int, D, D, C.c_double, C.c_double, C.c_int, C.c_int, IP, IP, IP, D, D, D, IP]
    idx = np.ascontiguousarray(np.asarray(index).ravel().astype(np.intc)); dirf = np.ascontiguousarray(np.asarray(dirichlet_flags).astype(np.intc))
    f64 = lambda a: np.ascontiguousarray(a, np.float64)  # noqa: E731
    x0, y0, xm, ym = f64(x0), f64(y0), f64(x_moved), f64(y_moved)
    cap_n, cap_e = 4 * x0.size + 1000, 4 * (idx.size // 3) + 1000
    on, oe, og = C.c_int(), C.c_int(), C.c_int()
    oi = np.zeros(3 * cap_e, np.intc); ox = np.zeros(cap_n); oy = np.zeros(cap_n); op = np.zeros(cap_n)
    rc = L.shim_bamg_adapt(idx.ctypes.data_as(IP), _abi.dptr(x0), _abi.dptr(y0), x0.size, idx.size // 3, dirf.ctypes.data_as(IP), dirf.size,
                           _abi.dptr(xm), _abi.dptr(ym), float(hmin), float(hmax), cap_n, cap_e, C.byref(on), C.byref(oe),
                           oi.ctypes.data_as(IP), _abi.dptr(ox), _abi.dptr(oy), _abi.dptr(op), C.byref(og))
    assert rc == 0, rc
    nn, ne = on.value, oe.value
    return ox[:nn].copy(), oy[:nn].copy(), (oi[:3 * ne].reshape(-1, 3) - 1).astype(np.int32), op[:nn].copy(), og.value


def bamg_interp_grid_to_mesh(x_in, y_in, data, x_mesh, y_mesh, default_value=1e8, interp=1, row_major=False):
    """The REAL InterpFromGridToMeshx (forcing ingest, model/externaldata.cpp:1436).  data [M, N, N_data] (or [N, M, N_data]
    when row_major); prints a line per node outside the grid, as the reference does."""
    L = C.CDLL(os.path.join(HERE, "_ref", "libbamg_shim.so"))
    D = _abi.c_double_p
    L.shim_bamg_interp_grid_to_mesh.argtypes = [D, C.c_int, D, C.c_int, D, C.c_int, C.c_int, C.c_int, D, D, C.c_int, C.c_double, C.c_int, C.c_int, D]
    f64 = lambda a: np.ascontiguousarray(a, np.float64)  # noqa: E731
    x_in, y_in, data, xm, ym = f64(x_in), f64(y_in), f64(data), f64(x_mesh), f64(y_mesh)
    M, N = (data.shape[1], data.shape[0]) if row_major else (data.shape[0], data.shape[1])
    out = np.empty((xm.size, data.shape[2]))
    rc = L.shim_bamg_interp_grid_to_mesh(_abi.dptr(x_in), x_in.size, _abi.dptr(y_in), y_in.size, _abi.dptr(data), M, N, data.shape[2], _abi.dptr(xm),
                                         _abi.dptr(ym), xm.size, float(default_value), int(interp), int(bool(row_major)), _abi.dptr(out))
    assert rc == 0
    return out
