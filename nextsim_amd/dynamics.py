"""Host-side mirror of the reference's call surface for the dynamics path, over the C ABI.

`FiniteElementDynamics` keeps the names of FiniteElement's methods on this path
(model/finiteelement.hpp:162-164 and FE.cpp:8197-8214): `explicitSolve()`, `update()`, `step()`,
`checkRegridding()`, `checkFieldsFast()`, `updateGhosts` happens inside.  Every call goes through
libnxsdyn.so (include/nxs_dyn.h); there is no Python or CPU compute path here, and a missing
library or GPU is an error, never a fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import _abi

_LIB = None
# NXS_DYN_LIBRARY: another build of the same library (kernel experiments); never a different implementation
_LIB_PATH = os.environ.get("NXS_DYN_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libnxsdyn.so")


class NxsError(RuntimeError):
    def __init__(self, code: int, what: str):
        super().__init__(f"{_abi.ERRORS.get(code, code)}: {what}")
        self.code = code


def build_library(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of nextsim_amd/csrc (cross-compiles without a GPU)."""
    src_dir = os.path.dirname(_LIB_PATH)
    args = ["make", "-s", "-C", src_dir]
    if force:
        args.insert(1, "-B")
    subprocess.check_call(args)
    return _LIB_PATH


def load_library():
    """dlopen libnxsdyn.so and declare every symbol of include/nxs_dyn.h.  Raises if absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_LIB_PATH):
        raise NxsError(-2, f"{_LIB_PATH} is missing: build it with __graft_entry__.build() "
                           "(there is no CPU fallback for the dynamics path)")
    L = C.CDLL(_LIB_PATH)
    P = C.POINTER
    H = C.c_void_p
    L.nxs_dyn_abi_version.restype = C.c_int
    L.nxs_dyn_last_error.restype = C.c_char_p
    L.nxs_dyn_last_error.argtypes = [H]
    L.nxs_dyn_default_params.argtypes = [P(_abi.Params)]
    L.nxs_dyn_create.argtypes = [P(_abi.Params), C.c_int, P(H)]
    L.nxs_dyn_destroy.argtypes = [H]
    L.nxs_dyn_set_params.argtypes = [H, P(_abi.Params)]
    L.nxs_dyn_set_mesh.argtypes = [H, P(_abi.Mesh)]
    L.nxs_dyn_set_halo.argtypes = [H, P(_abi.Halo)]
    L.nxs_dyn_comm_unique_id.argtypes = [C.c_void_p]
    L.nxs_dyn_comm_init.argtypes = [H, C.c_void_p, C.c_int, C.c_int]
    L.nxs_dyn_comm_selftest.argtypes = [H, P(C.c_int32)]
    L.nxs_dyn_set_halo_exchange_fn.argtypes = [H, HALO_FN, C.c_void_p]
    L.nxs_dyn_ipc_export.argtypes = [H, C.c_void_p]
    L.nxs_dyn_ipc_connect.argtypes = [H, C.c_void_p, _abi.c_int32_p, _abi.c_int32_p, _abi.c_int32_p]
    L.nxs_dyn_ipc_selftest.argtypes = [H, C.c_int, P(C.c_int32)]
    L.nxs_dyn_ipc_record_bytes.argtypes = [H, P(C.c_int32)]
    L.nxs_dyn_ipc_export_record.argtypes = [H, C.c_void_p, C.c_int32]
    L.nxs_dyn_ipc_connect_records.argtypes = [H, C.c_void_p, C.c_int64, C.c_int32]
    L.nxs_dyn_ipc_loopback.argtypes = [H]
    L.nxs_dyn_put_state.argtypes = [H, P(_abi.State)]
    L.nxs_dyn_get_state.argtypes = [H, P(_abi.State)]
    L.nxs_dyn_set_forcing.argtypes = [H, P(_abi.Forcing)]
    L.nxs_dyn_get_diag.argtypes = [H, P(_abi.Diag)]
    L.nxs_dyn_ice_diagnostics.argtypes = [H, P(_abi.IceDiag), P(C.c_void_p)]
    L.nxs_dyn_step.argtypes = [H]
    L.nxs_dyn_explicit_solve.argtypes = [H]
    L.nxs_dyn_update.argtypes = [H]
    L.nxs_dyn_synchronize.argtypes = [H]
    L.nxs_dyn_step_host.argtypes = [H, P(_abi.State), P(_abi.Forcing)]
    L.nxs_dyn_set_forcing_pair.argtypes = [H, P(_abi.Forcing), P(_abi.Forcing)]
    L.nxs_dyn_set_forcing_time.argtypes = [H, C.c_double, C.c_double, P(C.c_double), P(C.c_double)]
    L.nxs_dyn_check_regridding.argtypes = [H, P(C.c_double), P(C.c_int32), P(C.c_int32)]
    L.nxs_dyn_check_fields_fast.argtypes = [H, P(C.c_int32)]
    L.nxs_dyn_get_timing.argtypes = [H, P(_abi.Timing)]
    L.nxs_dyn_get_step_times.argtypes = [H, _abi.c_double_p, C.c_int32, P(C.c_int32)]
    L.nxs_dyn_get_traffic_model.argtypes = [H, P(_abi.Traffic)]
    L.nxs_dyn_physical_constants.argtypes = [P(C.c_double), C.c_int32]
    L.nxs_dyn_selftest_quotients.argtypes = [C.c_int32, C.c_int64, C.c_uint64, C.c_int32, P(C.c_int64)]
    L.nxs_dyn_set_option.argtypes = [H, C.c_char_p, C.c_int64]
    L.nxs_dyn_debug_array.argtypes = [H, C.c_char_p, _abi.c_double_p, C.c_int64]
    L.nxs_dyn_get_branch_trace.argtypes = [H, C.POINTER(C.c_uint64), C.c_int64]
    L.nxs_mesh_connectivity.argtypes = [_abi.c_int32_p, C.c_int32, C.c_int32, P(C.c_int32), _abi.c_double_p,
                                        P(C.c_int32), _abi.c_double_p]
    L.nxs_mesh_element_connectivity.argtypes = [_abi.c_int32_p, C.c_int32, C.c_int32, _abi.c_double_p]
    L.nxs_calc_cohesion.argtypes = [C.c_double, C.c_double, _abi.c_int32_p, C.c_int64, C.c_int64, _abi.c_double_p]
    for name in EXPORTS:
        getattr(L, name)  # raises AttributeError if a declared symbol is not exported
        if name not in ("nxs_dyn_last_error",):
            getattr(L, name).restype = C.c_int
    L.nxs_dyn_last_error.restype = C.c_char_p
    if L.nxs_dyn_abi_version() != 2:
        raise NxsError(-1, "libnxsdyn.so ABI version mismatch")
    _LIB = L
    return L


HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _abi.c_double_p, _abi.c_double_p)

# every symbol include/nxs_dyn.h declares
IPC_BLOB_BYTES = 128

EXPORTS = (
    "nxs_dyn_set_halo_exchange_fn", "nxs_dyn_ipc_export", "nxs_dyn_ipc_connect", "nxs_dyn_ipc_selftest",
    "nxs_dyn_ipc_record_bytes", "nxs_dyn_ipc_export_record", "nxs_dyn_ipc_connect_records", "nxs_dyn_ipc_loopback",
    "nxs_dyn_abi_version", "nxs_dyn_last_error", "nxs_dyn_default_params", "nxs_dyn_physical_constants", "nxs_dyn_selftest_quotients", "nxs_dyn_create", "nxs_dyn_destroy",
    "nxs_dyn_set_params", "nxs_dyn_set_mesh", "nxs_dyn_set_halo", "nxs_dyn_comm_unique_id", "nxs_dyn_comm_init", "nxs_dyn_comm_selftest",
    "nxs_dyn_put_state", "nxs_dyn_get_state", "nxs_dyn_set_forcing", "nxs_dyn_set_forcing_pair", "nxs_dyn_set_forcing_time",
    "nxs_dyn_get_diag", "nxs_dyn_ice_diagnostics", "nxs_dyn_step",
    "nxs_dyn_explicit_solve", "nxs_dyn_update", "nxs_dyn_synchronize", "nxs_dyn_step_host",
    "nxs_dyn_check_regridding", "nxs_dyn_check_fields_fast", "nxs_dyn_get_timing", "nxs_dyn_get_step_times", "nxs_dyn_get_traffic_model", "nxs_dyn_set_option",
    "nxs_dyn_debug_array", "nxs_dyn_get_branch_trace", "nxs_mesh_connectivity", "nxs_mesh_element_connectivity", "nxs_calc_cohesion",
)
INTERP_EXPORTS = ("nxs_interp_mesh_to_mesh_2d", "nxs_interp_mesh_to_grid", "nxs_interp_mesh_to_grid_device", "nxs_interp_conservative_remap", "nxs_interp_grid_to_mesh",
                  "nxs_interp_last_error", "nxs_interp_last_info", "nxs_mesh_convex_completion", "nxs_mesh_convex_completion_mode", "nxs_regrid_create", "nxs_regrid_destroy",
                  "nxs_regrid_interp_nodes", "nxs_regrid_remap_elements", "nxs_interp_last_timing", "nxs_regrid_debug_tables")


def selftest_quotients(n: int, seed: int = 1, mode: int = 0, device: int = 0) -> int:
    """nxs_dyn_selftest_quotients: n sextuples of numerators over one divisor computed with the shared reciprocal and with six divisions on `device`;
    the number of quotients whose bits differ."""
    L = load_library()
    bad = C.c_int64(-1)
    rc = L.nxs_dyn_selftest_quotients(device, n, seed, mode, C.byref(bad))
    if rc != 0:
        raise NxsError(rc, (L.nxs_dyn_last_error(None) or b"").decode(errors="replace"))
    return bad.value


def mesh_connectivity(indices: np.ndarray, num_nodes: int):
    """(NodalElementConnectivity, NodalConnectivity) in bamg's layout and row order (host code of the
    product library; the stand-in for BamgConvertMeshx at FE.cpp:77-80)."""
    L = load_library()
    ne = indices.size // 3
    w1, w2 = C.c_int32(), C.c_int32()
    rc = L.nxs_mesh_connectivity(_abi.iptr(indices), num_nodes, ne, C.byref(w1), None, C.byref(w2), None)
    if rc:
        raise NxsError(rc, "nxs_mesh_connectivity")
    nec = np.empty((num_nodes, w1.value))
    nc = np.empty((num_nodes, w2.value))
    rc = L.nxs_mesh_connectivity(_abi.iptr(indices), num_nodes, ne, C.byref(w1), _abi.dptr(nec), C.byref(w2), _abi.dptr(nc))
    if rc:
        raise NxsError(rc, "nxs_mesh_connectivity")
    return nec, nc


def calc_cohesion(C_fix: float, C_alea: float, global_element_id: np.ndarray, num_global_elements: int) -> np.ndarray:
    """calcCohesion() (FE.cpp:3909-3914): C_fix + C_alea * (the reference's minstd/uniform_01 draw of each global element)."""
    L = load_library()
    ids = np.ascontiguousarray(global_element_id, np.int32)
    out = np.empty(ids.size)
    rc = L.nxs_calc_cohesion(float(C_fix), float(C_alea), _abi.iptr(ids), ids.size, int(num_global_elements), _abi.dptr(out))
    if rc:
        raise NxsError(rc, "nxs_calc_cohesion")
    return out


def mesh_element_connectivity(indices: np.ndarray, num_nodes: int) -> np.ndarray:
    """bamgmesh->ElementConnectivity ([Ne,3] doubles, 1-based, NaN on the boundary), bamg's column order."""
    L = load_library()
    indices = np.ascontiguousarray(indices, np.int32)
    ec = np.empty((indices.size // 3, 3))
    rc = L.nxs_mesh_element_connectivity(_abi.iptr(indices), num_nodes, indices.size // 3, _abi.dptr(ec))
    if rc:
        raise NxsError(rc, "nxs_mesh_element_connectivity")
    return ec


class FiniteElementDynamics:
    """The dynamics part of FiniteElement on one GPU (one MPI-rank equivalent)."""

    def __init__(self, params: _abi.Params, device: int = 0):
        self.L = load_library()
        self.h = C.c_void_p()
        self.params = params.copy()
        rc = self.L.nxs_dyn_create(C.byref(self.params), device, C.byref(self.h))
        if rc:
            raise NxsError(rc, (self.L.nxs_dyn_last_error(None) or b"").decode())
        self.lm = None
        self._keep = []

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.nxs_dyn_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc:
            raise NxsError(rc, (self.L.nxs_dyn_last_error(self.h) or b"").decode())

    # ---- setup (distributedMeshProcessing / initUpdateGhosts) ----
    def set_params(self, params: _abi.Params):
        self.params = params.copy()
        self._chk(self.L.nxs_dyn_set_params(self.h, C.byref(self.params)))

    def set_mesh(self, lm, tables=None):
        m = _abi.mesh_struct(lm, tables)
        self._chk(self.L.nxs_dyn_set_mesh(self.h, C.byref(m)))
        self.lm = lm
        if lm.nranks > 1:
            hs = _abi.halo_struct(lm)
            self._chk(self.L.nxs_dyn_set_halo(self.h, C.byref(hs)))

    def comm_init(self, unique_id: bytes, rank: int, nranks: int):
        buf = C.create_string_buffer(unique_id, 128)
        self._chk(self.L.nxs_dyn_comm_init(self.h, buf, rank, nranks))

    def comm_selftest(self) -> int:
        """Coded payloads through the RCCL communicator (self send/recv + the halo segments); returns the number of wrong values."""
        err = C.c_int32(-1)
        self._chk(self.L.nxs_dyn_comm_selftest(self.h, C.byref(err)))
        return err.value

    def set_halo_exchange(self, fn):
        """fn(send: np.ndarray, recv: np.ndarray) -> None : host-staged updateGhosts through the caller's
        communicator (send/recv are views laid out per neighbour as [u-block | v-block])."""
        lm = self.lm
        ns, nr = 2 * int(lm.send_offsets[-1]), 2 * int(lm.recv_offsets[-1])

        def tramp(ctx, send, recv):
            try:
                sv = np.ctypeslib.as_array(send, shape=(max(ns, 1),))[:ns]
                rv = np.ctypeslib.as_array(recv, shape=(max(nr, 1),))[:nr]
                fn(sv, rv)
                return 0
            except Exception:  # noqa: BLE001 -- never unwind through the C ABI
                import traceback
                traceback.print_exc()
                return 1
        self._halo_cb = HALO_FN(tramp)
        self._chk(self.L.nxs_dyn_set_halo_exchange_fn(self.h, self._halo_cb, None))

    def ipc_setup(self, all_gather, selftest_rounds: int = 64, low_level: bool = False) -> bool:
        """Collective setup of the device-direct halo transport (peer-mapped mailboxes).
        all_gather(obj) -> list of every rank's obj (the launcher's communicator, e.g.
        torch.distributed.all_gather_object).  Returns True when the transport is connected and its
        self-test passed on EVERY rank; otherwise the handle is left on its previous transport.
        Default: the record form (nxs_dyn_ipc_export_record / nxs_dyn_ipc_connect_records) -- the library finds every neighbour's segment, totals and
        flag slot itself, also for the directions nxs_dyn_set_halo added on a ragged partition.  low_level: the caller's own bookkeeping through
        nxs_dyn_ipc_connect (tables for the caller's send neighbours; refused by the library where a neighbour exists in one direction only)."""
        lm = self.lm
        ok = 1
        self._ipc_error = ""
        if low_level:
            blob = C.create_string_buffer(IPC_BLOB_BYTES)
            try:
                self._chk(self.L.nxs_dyn_ipc_export(self.h, blob))
            except NxsError as e:
                self._ipc_error = f"rank {lm.rank}: export: {e}"
                ok = 0
            infos = all_gather({"ok": ok, "blob": blob.raw, "recv_procs": lm.recv_procs.tolist(),
                                "recv_offsets": lm.recv_offsets.tolist(), "err": self._ipc_error})
        else:
            n = C.c_int32(0)
            rec = b""
            try:
                self._chk(self.L.nxs_dyn_ipc_record_bytes(self.h, C.byref(n)))
            except NxsError as e:
                self._ipc_error = f"rank {lm.rank}: record: {e}"
                ok = 0
            stride = max(all_gather(int(n.value)))       # (the launcher's MPI_Allreduce(MAX))
            if ok:
                buf = C.create_string_buffer(stride)
                try:
                    self._chk(self.L.nxs_dyn_ipc_export_record(self.h, buf, stride))
                    rec = buf.raw
                except NxsError as e:
                    self._ipc_error = f"rank {lm.rank}: export: {e}"
                    ok = 0
            infos = all_gather({"ok": ok, "rec": rec if ok else bytes(stride), "err": self._ipc_error})
        if not all(i["ok"] for i in infos):
            self._ipc_error = "; ".join(i["err"] for i in infos if i["err"])   # (every rank reports what any rank saw)
            return False
        ok = 1
        try:
            if low_level:
                blobs, off, tot, slot = b"", [], [], []
                for q in lm.send_procs.tolist():
                    inf = infos[q]
                    k = inf["recv_procs"].index(lm.rank)
                    blobs += inf["blob"]
                    off.append(inf["recv_offsets"][k]); tot.append(inf["recv_offsets"][-1]); slot.append(k)
                a_off, a_tot, a_slot = (np.ascontiguousarray(np.asarray(v, np.int32)) for v in (off, tot, slot))
                bbuf = C.create_string_buffer(blobs, max(len(blobs), 1))
                self._chk(self.L.nxs_dyn_ipc_connect(self.h, bbuf, _abi.iptr(a_off), _abi.iptr(a_tot), _abi.iptr(a_slot)))
            else:
                recs = b"".join(i["rec"] for i in infos)
                self._chk(self.L.nxs_dyn_ipc_connect_records(self.h, C.create_string_buffer(recs, len(recs)), stride, len(infos)))
        except NxsError as e:
            self._ipc_error = f"rank {lm.rank}: connect: {e}"
            ok = 0
        oks = all_gather((ok, self._ipc_error))
        if not all(o for o, _ in oks):
            self._ipc_error = "; ".join(m for _, m in oks if m)
            self.L.nxs_dyn_set_halo(self.h, C.byref(_abi.halo_struct(lm)))  # drops the half-made transport
            return False
        err = C.c_int32(0)
        try:
            self._chk(self.L.nxs_dyn_ipc_selftest(self.h, selftest_rounds, C.byref(err)))
        except NxsError as e:   # still take part in the gather below: the other ranks are waiting in it
            self._ipc_error = f"rank {lm.rank}: self-test: {e}"
            err.value = err.value or -1
        if err.value and not self._ipc_error:
            self._ipc_error = f"rank {lm.rank}: self-test: {err.value} payload(s) wrong"
        errs = all_gather((int(err.value), self._ipc_error))
        good = all(e == 0 for e, _ in errs)
        if not good:
            self._ipc_error = "; ".join(m for _, m in errs if m)
            self.L.nxs_dyn_set_halo(self.h, C.byref(_abi.halo_struct(lm)))
        return good

    def ipc_loopback(self) -> bool:
        """Profiling aid (nxs_dyn_ipc_loopback): the device-direct mailboxes of this handle connected to THEMSELVES, so that a rank's partition can be stepped
        alone on a device with the exchange inside its kernels.  The ghosts receive meaningless velocities (results are NOT the model's); the launches walk
        the same tables and move the same bytes.  False when the partition's lists do not allow it (no neighbour)."""
        rc = self.L.nxs_dyn_ipc_loopback(self.h)
        if rc == -1:
            return False
        self._chk(rc)
        return True

    @staticmethod
    def comm_unique_id() -> bytes:
        L = load_library()
        buf = C.create_string_buffer(128)
        rc = L.nxs_dyn_comm_unique_id(buf)
        if rc:
            raise NxsError(rc, (L.nxs_dyn_last_error(None) or b"").decode())
        return buf.raw

    def set_option(self, key: str, value: int):
        self._chk(self.L.nxs_dyn_set_option(self.h, key.encode(), int(value)))

    # ---- data movement ----
    def put_state(self, arrays: dict):
        s = _abi.state_struct(arrays)
        self._chk(self.L.nxs_dyn_put_state(self.h, C.byref(s)))

    def set_forcing(self, arrays: dict):
        f = _abi.forcing_struct(arrays)
        self._chk(self.L.nxs_dyn_set_forcing(self.h, C.byref(f)))

    def set_forcing_pair(self, arrays0: dict, arrays1: dict):
        """The two snapshots of a forcing interval become resident (ExternalData's interpolated_data[0], [1])."""
        f0, f1 = _abi.forcing_struct(arrays0), _abi.forcing_struct(arrays1)
        self._keep_forcing = (arrays0, arrays1)
        self._chk(self.L.nxs_dyn_set_forcing_pair(self.h, C.byref(f0), C.byref(f1)))

    def set_forcing_time(self, fcoeff0: float, fcoeff1: float, factor=None, bias=None):
        """Per step: M_factor*(fcoeff0*d0 + fcoeff1*d1) + M_bias_correction on the device (externaldata.cpp:360-401)."""
        fa = (C.c_double * 3)(*factor) if factor is not None else None
        bi = (C.c_double * 3)(*bias) if bias is not None else None
        self._chk(self.L.nxs_dyn_set_forcing_time(self.h, float(fcoeff0), float(fcoeff1), fa, bi))

    def get_state(self) -> dict:
        Nn, Ne = self.lm.num_nodes, self.lm.num_elements
        out = {k: np.empty(2 * Nn) for k in _abi.STATE_NODAL}
        out.update({k: np.empty(Ne) for k in _abi.STATE_ELEMENT})
        s = _abi.State()
        for k in _abi.STATE_NODAL:
            setattr(s, k, _abi.dptr(out[k]))
        for k in _abi.STATE_ELEMENT:
            if k.startswith("sigma"):
                s.sigma[int(k[-1])] = _abi.dptr(out[k])
            else:
                setattr(s, k, _abi.dptr(out[k]))
        self._chk(self.L.nxs_dyn_get_state(self.h, C.byref(s)))
        return out

    def get_diag(self) -> dict:
        Nn, Ne = self.lm.num_nodes, self.lm.num_elements
        out = {"surface": np.empty(Ne), "delta_x": np.empty(Ne), "D_tau_a": np.empty(2 * Nn),
               "D_tau_w": np.empty(2 * Nn), "D_del_ci_ridge_myi": np.empty(Ne)}
        d = _abi.Diag()
        for k, v in out.items():
            setattr(d, k, _abi.dptr(v))
        self._chk(self.L.nxs_dyn_get_diag(self.h, C.byref(d)))
        return out

    def updateIceDiagnostics(self, want_host: bool = True):
        """updateIceDiagnostics() (FE.cpp:7860-7905) on the device-resident state.  Returns (dict of host arrays or None, device pointer of the
        [Ne][6] interleaved rows D_conc, D_thick, D_snow_thick, D_sigma0, D_sigma1, D_divergence -- what interp.InterpFromMeshToGridx_device samples)."""
        Ne = self.lm.num_elements
        out, d = None, None
        if want_host:
            out = {k: np.empty(Ne) for k in _abi.ICE_DIAG}
            d = _abi.IceDiag()
            for k in _abi.ICE_DIAG:
                setattr(d, k, _abi.dptr(out[k]))
        dev = C.c_void_p()
        self._chk(self.L.nxs_dyn_ice_diagnostics(self.h, C.byref(d) if d is not None else None, C.byref(dev)))
        return out, dev.value

    def branch_trace(self) -> dict:
        """The record option "trace_branches" keeps (include/nxs_dyn.h): {'hash', 'damage_substeps', 'flags', 'substeps'}."""
        t = np.zeros((self.lm.num_elements, 4), np.uint64)
        self._chk(self.L.nxs_dyn_get_branch_trace(self.h, t.ctypes.data_as(C.POINTER(C.c_uint64)), t.size))
        return {"hash": t[:, 0], "damage_substeps": t[:, 1], "flags": t[:, 2], "substeps": t[:, 3]}

    def debug_array(self, name: str) -> np.ndarray:
        Nn, Ne = self.lm.num_nodes, self.lm.num_elements
        n = {"rlmass": Nn, "node_mass": Nn, "C_bu": Nn, "grad_ssh": 2 * Nn, "fcor": Nn, "VTM": 2 * Nn,
             "shape": 6 * Ne, "emass": Ne, "ecbu": Ne, "force": 6 * Ne, "volume": Ne, "expC": Ne,
             "erec": 6 * Ne, "nrec": 10 * Nn, "xy": 2 * Nn, "delta_x": Ne, "surface": Ne, "tau_a": 2 * Nn,
             "phase_times": 8 * 8192, "phase_times_prep": 8 * 8192, "shape_range": 1}[name]
        out = np.empty(n)
        self._chk(self.L.nxs_dyn_debug_array(self.h, name.encode(), _abi.dptr(out), n))
        return out

    # ---- the reference's call surface ----
    def step(self):
        """FE.cpp:8197-8214: UM_P = M_UM; explicitSolve(); update(UM_P) (or free drift / no motion)."""
        self._chk(self.L.nxs_dyn_step(self.h))

    def explicitSolve(self):
        self._chk(self.L.nxs_dyn_explicit_solve(self.h))

    def update(self):
        self._chk(self.L.nxs_dyn_update(self.h))

    def synchronize(self):
        self._chk(self.L.nxs_dyn_synchronize(self.h))

    def checkRegridding(self):
        ang, flip, rg = C.c_double(), C.c_int32(), C.c_int32()
        self._chk(self.L.nxs_dyn_check_regridding(self.h, C.byref(ang), C.byref(flip), C.byref(rg)))
        return ang.value, flip.value, rg.value

    def checkFieldsFast(self) -> int:
        c = C.c_int32()
        self._chk(self.L.nxs_dyn_check_fields_fast(self.h, C.byref(c)))
        return c.value

    def timing(self) -> dict:
        t = _abi.Timing()
        self._chk(self.L.nxs_dyn_get_timing(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _abi.Timing._fields_}

    def step_times(self) -> np.ndarray:
        """Device milliseconds of every step since the last "timing_reset" (nxs_dyn_get_step_times)."""
        n = C.c_int32()
        self._chk(self.L.nxs_dyn_get_step_times(self.h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.float64)
        self._chk(self.L.nxs_dyn_get_step_times(self.h, _abi.dptr(out), n.value, C.byref(n)))
        return out[:n.value]

    KERNEL_NAMES = {0: "none", 1: "k_sigma + k_solve_move", 2: "k_substep_fused", 3: "k_substep_multi", 4: "k_substep_pair",
                    5: "k_substep_resident", 6: "k_substep_resident_big", 7: "k_substep_flow"}
    PREP_NAMES = {0: "none", 1: "k_prep_elements + k_prep_nodes (work arrays)", 2: "k_prep_elements + k_prep_nodes", 3: "k_prep_fused"}

    def traffic_model(self) -> dict:
        """Bytes per launch the kernels of the last step had to move (nxs_dyn_get_traffic_model; include/nxs_dyn.h says what each figure counts)."""
        t = _abi.Traffic()
        self._chk(self.L.nxs_dyn_get_traffic_model(self.h, C.byref(t)))
        d = {k: getattr(t, k) for k, _ in _abi.Traffic._fields_ if k != "reserved0"}
        d["substep_kernel_name"] = self.KERNEL_NAMES.get(t.substep_kernel, "?")
        d["prep_kernel_name"] = self.PREP_NAMES.get(t.prep_kernel, "?")
        return d
