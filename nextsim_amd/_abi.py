"""ctypes mirror of include/nxs_dyn.h (the C ABI of libnxsdyn.so).

Pure declarations: struct layouts, argument types, numpy <-> pointer helpers.  No compute here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)

NXS_DYN_BBM, NXS_DYN_NO_MOTION, NXS_DYN_FREE_DRIFT, NXS_DYN_EVP, NXS_DYN_MEVP = range(5)
NXS_BASAL_NONE, NXS_BASAL_LEMIEUX = 0, 1
NXS_ICECAT_CLASSIC, NXS_ICECAT_YOUNG_ICE = 0, 1

DYNAMICS_TYPES = {"bbm": NXS_DYN_BBM, "no_motion": NXS_DYN_NO_MOTION, "free_drift": NXS_DYN_FREE_DRIFT,
                  "evp": NXS_DYN_EVP, "mevp": NXS_DYN_MEVP}

ERRORS = {0: "NXS_OK", -1: "NXS_ERR_INVALID", -2: "NXS_ERR_NO_DEVICE", -3: "NXS_ERR_HIP",
          -4: "NXS_ERR_STATE", -5: "NXS_ERR_COMM", -6: "NXS_ERR_NOMEM", -7: "NXS_ERR_INTERNAL"}


class Params(C.Structure):
    _fields_ = [
        ("dtime_step", C.c_double),
        ("substeps", C.c_int32),
        ("dynamics_type", C.c_int32),
        ("basal_stress_type", C.c_int32),
        ("ice_cat_type", C.c_int32),
        ("newice_type", C.c_int32),
        ("equal_ridging", C.c_int32),
        ("use_young_ice_in_myi_reset", C.c_int32),
        ("reserved0", C.c_int32),
        ("young", C.c_double),
        ("nu0", C.c_double),
        ("tan_phi", C.c_double),
        ("compr_strength", C.c_double),
        ("compaction_param", C.c_double),
        ("undamaged_time_relaxation_sigma", C.c_double),
        ("exponent_relaxation_sigma", C.c_double),
        ("compression_factor", C.c_double),
        ("exponent_compression_factor", C.c_double),
        ("min_h", C.c_double),
        ("min_c", C.c_double),
        ("quad_drag_coef_water", C.c_double),
        ("lin_drag_coef_water", C.c_double),
        ("quad_drag_coef_air", C.c_double),
        ("lin_drag_coef_air", C.c_double),
        ("ocean_turning_angle_rad", C.c_double),
        ("basal_k1", C.c_double),
        ("basal_k2", C.c_double),
        ("basal_Cb", C.c_double),
        ("basal_u_0", C.c_double),
        ("evp_e", C.c_double),
        ("evp_Pstar", C.c_double),
        ("evp_C", C.c_double),
        ("evp_dmin", C.c_double),
        ("mevp_alpha", C.c_double),
        ("mevp_beta", C.c_double),
        ("regrid_angle", C.c_double),
    ]

    def copy(self) -> "Params":
        q = Params()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(Params))
        return q


class Mesh(C.Structure):
    _fields_ = [
        ("num_nodes", C.c_int32),
        ("num_elements", C.c_int32),
        ("local_ndof", C.c_int32),
        ("local_nelements", C.c_int32),
        ("indices", c_int32_p),
        ("ghost_nodes", c_uint8_p),
        ("coord_x", c_double_p),
        ("coord_y", c_double_p),
        ("lat", c_double_p),
        ("mask_dirichlet", c_uint8_p),
        ("num_neumann_flags", C.c_int32),
        ("reserved0", C.c_int32),
        ("neumann_flags", c_int32_p),
        ("nodal_element_connectivity", c_double_p),
        ("nodal_connectivity", c_double_p),
        ("nec_width", C.c_int32),
        ("nc_width", C.c_int32),
    ]


class Halo(C.Structure):
    _fields_ = [
        ("rank", C.c_int32),
        ("nranks", C.c_int32),
        ("num_send_procs", C.c_int32),
        ("num_recv_procs", C.c_int32),
        ("send_procs", c_int32_p),
        ("send_offsets", c_int32_p),
        ("send_index", c_int32_p),
        ("recv_procs", c_int32_p),
        ("recv_offsets", c_int32_p),
        ("recv_index", c_int32_p),
    ]


class State(C.Structure):
    _fields_ = [
        ("VT", c_double_p), ("UM", c_double_p), ("UT", c_double_p),
        ("conc", c_double_p), ("thick", c_double_p), ("snow_thick", c_double_p),
        ("damage", c_double_p), ("ridge_ratio", c_double_p),
        ("sigma", c_double_p * 3),
        ("conc_young", c_double_p), ("h_young", c_double_p), ("hs_young", c_double_p),
        ("conc_myi", c_double_p), ("thick_myi", c_double_p),
        ("cohesion", c_double_p), ("time_relaxation_damage", c_double_p),
        ("drag_ui", c_double_p), ("drag_ui_young", c_double_p),
    ]


class Forcing(C.Structure):
    _fields_ = [("wind", c_double_p), ("ocean", c_double_p), ("ssh", c_double_p),
                ("element_depth", c_double_p)]


class Diag(C.Structure):
    _fields_ = [("surface", c_double_p), ("delta_x", c_double_p), ("D_tau_a", c_double_p),
                ("D_tau_w", c_double_p), ("D_del_ci_ridge_myi", c_double_p)]


ICE_DIAG = ("D_conc", "D_thick", "D_snow_thick", "D_sigma0", "D_sigma1", "D_divergence")


class IceDiag(C.Structure):
    _fields_ = [(k, c_double_p) for k in ICE_DIAG]


class Timing(C.Structure):
    _fields_ = [("prep_ms", C.c_double), ("substeps_ms", C.c_double), ("smoother_ms", C.c_double),
                ("update_ms", C.c_double), ("total_ms", C.c_double),
                ("substep_launches", C.c_int32), ("steps_averaged", C.c_int32), ("ring_flush_ms", C.c_double)]


class Traffic(C.Structure):   # nxs_dyn_traffic
    _fields_ = [("substep_kernel", C.c_int32), ("substeps_per_launch", C.c_int32), ("halo_in_kernel", C.c_int32), ("prep_kernel", C.c_int32),
                ("substep_scheme_bytes", C.c_double), ("substep_reread_bytes", C.c_double), ("substep_unique_bytes", C.c_double),
                ("survey_model_bytes", C.c_double), ("move_ring_slots", C.c_int32), ("reserved0", C.c_int32), ("move_ring_bytes", C.c_double),
                ("prep_scheme_bytes", C.c_double), ("prep_unique_bytes", C.c_double), ("update_bytes", C.c_double)]


# ---- numpy helpers --------------------------------------------------------------------------

def dptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous, (a.dtype, a.flags)
    return a.ctypes.data_as(c_double_p)


def iptr(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(c_int32_p)


def bptr(a: np.ndarray):
    assert a.dtype == np.uint8 and a.flags.c_contiguous
    return a.ctypes.data_as(c_uint8_p)


# names of the nodal / elemental members of State, in ABI order
STATE_NODAL = ("VT", "UM", "UT")
STATE_ELEMENT = ("conc", "thick", "snow_thick", "damage", "ridge_ratio", "sigma0", "sigma1", "sigma2",
                 "conc_young", "h_young", "hs_young", "conc_myi", "thick_myi")
STATE_INPUT = ("cohesion", "time_relaxation_damage", "drag_ui", "drag_ui_young")


def state_struct(arrays: dict) -> State:
    """Build a State struct pointing at the numpy arrays in `arrays` (which must stay alive)."""
    s = State()
    for k in STATE_NODAL + STATE_INPUT:
        setattr(s, k, dptr(arrays[k]))
    for k in STATE_ELEMENT:
        if k.startswith("sigma"):
            s.sigma[int(k[-1])] = dptr(arrays[k])
        else:
            setattr(s, k, dptr(arrays[k]))
    return s


def forcing_struct(arrays: dict) -> Forcing:
    f = Forcing()
    for k in ("wind", "ocean", "ssh", "element_depth"):
        setattr(f, k, dptr(arrays[k]))
    return f


def mesh_struct(lm, tables=None) -> Mesh:
    """lm: a nextsim_amd.mesh.LocalMesh; tables: optional (nec, nc) bamg-layout double tables."""
    m = Mesh()
    m.num_nodes = lm.num_nodes
    m.num_elements = lm.num_elements
    m.local_ndof = lm.local_ndof
    m.local_nelements = lm.local_nelements
    m.indices = iptr(lm.indices)
    m.ghost_nodes = bptr(lm.ghost_nodes)
    m.coord_x = dptr(lm.coord_x)
    m.coord_y = dptr(lm.coord_y)
    m.lat = dptr(lm.lat)
    m.mask_dirichlet = bptr(lm.mask_dirichlet)
    m.num_neumann_flags = int(lm.neumann_flags.size)
    m.neumann_flags = iptr(lm.neumann_flags)
    if tables is not None:
        nec, nc = tables
        m.nodal_element_connectivity = dptr(nec)
        m.nodal_connectivity = dptr(nc)
        m.nec_width = nec.shape[1]
        m.nc_width = nc.shape[1]
    return m


def halo_struct(lm) -> Halo:
    h = Halo()
    h.rank, h.nranks = lm.rank, lm.nranks
    h.num_send_procs = int(lm.send_procs.size)
    h.num_recv_procs = int(lm.recv_procs.size)
    h.send_procs = iptr(lm.send_procs)
    h.send_offsets = iptr(lm.send_offsets)
    h.send_index = iptr(lm.send_index)
    h.recv_procs = iptr(lm.recv_procs)
    h.recv_offsets = iptr(lm.recv_offsets)
    h.recv_index = iptr(lm.recv_index)
    return h


def repo_root() -> str:
    return os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
