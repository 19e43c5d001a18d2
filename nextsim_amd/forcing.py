"""Synthetic initial state, forcing and parameters for the dynamics path.

The reference gets these from NetCDF datasets through ExternalData (out of scope); the hot path
only ever sees flat per-step snapshots (model/externaldata.cpp:441-459), which is what this
module fabricates: seeded, analytic, identical on every rank for a given global mesh.
"""
from __future__ import annotations

import numpy as np

from . import _abi
from .mesh import GlobalMesh, LocalMesh

DAYS_IN_SEC = 86400.0


def default_params(**over) -> _abi.Params:
    """model/options.cpp defaults (lines 43, 80, 109-111, 314-376, 397, 545-547), bbm."""
    p = _abi.Params()
    p.dtime_step = 200.0
    p.substeps = 120
    p.dynamics_type = _abi.NXS_DYN_BBM
    p.basal_stress_type = _abi.NXS_BASAL_LEMIEUX
    p.ice_cat_type = _abi.NXS_ICECAT_YOUNG_ICE
    p.newice_type = 4
    p.equal_ridging = 0
    p.use_young_ice_in_myi_reset = 1
    p.young = 5.9605e08
    p.nu0 = 1.0 / 3.0
    p.tan_phi = 0.7
    p.compr_strength = 1e10
    p.compaction_param = -20.0
    p.undamaged_time_relaxation_sigma = 1e7
    p.exponent_relaxation_sigma = 5.0
    p.compression_factor = 10e3
    p.exponent_compression_factor = 1.5
    p.min_h = 0.05
    p.min_c = 0.01
    p.quad_drag_coef_water = 0.0055
    p.lin_drag_coef_water = 0.0
    p.quad_drag_coef_air = 0.0049
    p.lin_drag_coef_air = 0.0
    p.ocean_turning_angle_rad = (np.pi / 180.0) * 25.0
    p.basal_k1, p.basal_k2, p.basal_Cb, p.basal_u_0 = 10.0, 15.0, 20.0, 5e-5
    p.evp_e, p.evp_Pstar, p.evp_C, p.evp_dmin = 2.0, 27.5e3, 20.0, 1e-9
    p.mevp_alpha, p.mevp_beta = 500.0, 500.0
    p.regrid_angle = 10.0
    for k, v in over.items():
        if k == "dynamics_type" and isinstance(v, str):
            v = _abi.DYNAMICS_TYPES[v]
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def scale_params_to_mesh(p: _abi.Params, mesh: GlobalMesh, C_lab: float = 2.0e6, alea_factor: float = 0.0):
    """FE.cpp:6993-6999: scale_coef = sqrt(0.1/res); C_fix = C_lab*scale_coef; compr_strength *= scale_coef.
    Returns (params_scaled, C_fix, C_alea)."""
    res = mesh.resolution()
    scale_coef = np.sqrt(0.1 / res)
    q = p.copy()
    q.compr_strength = p.compr_strength * scale_coef
    C_fix = C_lab * scale_coef
    return q, C_fix, alea_factor * C_fix


def minstd_uniform01(n: int) -> np.ndarray:
    """boost::minstd_rand (a = 48271, m = 2^31-1, default seed 1) through uniform_01, one draw per
    global element id (FE.cpp:11459-11475).  Boost 1.67's backward_compatible_uniform_01 (the class an engine as first
    template argument selects) multiplies by a stored reciprocal: r_k = double(x_k - 1) * (1 / 2147483646.0), and draws
    again while r_k >= 1 -- which cannot happen here: x_k <= m - 1 gives at most 2147483645 * fl(1/2147483646) < 1."""
    m = np.uint64(2147483647)
    a = 48271
    seq = np.array([a % 2147483647], np.uint64)  # x_1
    mult = a
    while seq.size < n:
        # seq[k + len] = seq[k] * a^len mod m
        mult_len = pow(a, int(seq.size), 2147483647)
        seq = np.concatenate([seq, (seq * np.uint64(mult_len)) % m])
    x = seq[:n].astype(np.float64)
    return (x - 1.0) * (1.0 / 2147483646.0)


def global_fields(mesh: GlobalMesh, p: _abi.Params, kind: str, C_fix: float, C_alea: float,
                  time: float = 0.0) -> dict:
    """State + forcing on the GLOBAL mesh.  kind: 'toy' (nextsim.toy.cfg semantics: wind (20,0),
    ocean 0, ssh 0, depth 200 m, A=1/H=1 with the x < xmin+0.3L strip ice free, FE.cpp:11706-11738)
    or 'arctic' (analytic cyclone wind, gyre ocean, ssh dome, shelf bathymetry)."""
    Nn, Ne = mesh.num_nodes, mesh.num_elements
    x, y, tri = mesh.x, mesh.y, mesh.tri
    cx, cy = x[tri].mean(1), y[tri].mean(1)
    g: dict = {}
    z_n = lambda: np.zeros(2 * Nn)
    z_e = lambda: np.zeros(Ne)
    g["VT"], g["UM"], g["UT"] = z_n(), z_n(), z_n()
    conc = np.ones(Ne); thick = np.ones(Ne); snow = z_e()
    if kind == "toy":
        xedge = x.min() + 0.3 * (x.max() - x.min())
        ow = cx < xedge
        conc[ow] = 0.0; thick[ow] = 0.0; snow[ow] = 0.0
        wind = np.concatenate([np.full(Nn, 20.0), np.zeros(Nn)])
        ocean = z_n(); ssh = np.zeros(Nn); depth = np.full(Ne, 200.0)
        cyoung = z_e(); hyoung = z_e()
    elif kind in ("arctic", "arctic_ow"):
        R = np.hypot(x, y).max()
        # ice: full cover in the basin, marginal ice zone + open water towards the open boundary
        th = np.arctan2(cy, cx); r = np.hypot(cx, cy)
        if kind == "arctic":     # an 80-degree sector: ~9 % of the triangles ice free
            edge = np.clip((r / R - 0.55) / 0.25, 0.0, 1.0) * (np.abs(th - np.deg2rad(20.0)) < np.deg2rad(40.0))
            conc = np.clip(1.0 - 1.25 * edge, 0.0, 1.0)
        else:                    # half the rim: ~30 % of the triangles ice free, ~3 % in the 0 < A <= 0.1 band that
            #                      updateSigmaDamage skips (FE.cpp:4146-4159) while update() and the drag still see its ice
            dth = np.abs(np.angle(np.exp(1j * (th - np.deg2rad(20.0)))))
            edge = np.clip((r / R - 0.36) / 0.25, 0.0, 1.0) * (dth < np.deg2rad(90.0))
            conc = np.clip(1.0 - 1.25 * edge, 0.0, 1.0)
            low = (conc > 0.0) & (conc <= 0.3)
            conc[low] = conc[low] / 3.0
        thick = conc * (1.0 + 1.5 * np.exp(-((cx + 0.3 * R) ** 2 + cy ** 2) / (0.5 * R) ** 2))
        snow = 0.1 * conc
        cyoung = np.clip(0.5 * edge * (1 - conc), 0.0, 1.0 - conc)
        hyoung = 0.15 * cyoung
        # cyclone translating at 10 m/s
        Rc = 400e3
        xc, yc = -0.2 * R + 10.0 * time, 0.1 * R
        dx, dy = x - xc, y - yc
        rr = np.hypot(dx, dy) + 1e-9
        sp = 15.0 * (rr / Rc) * np.exp(1.0 - rr / Rc)
        wind = np.concatenate([-sp * dy / rr, sp * dx / rr])
        # solid-body gyre 0.1 m/s at the rim
        ocean = np.concatenate([-0.1 * y / R, 0.1 * x / R])
        ssh = 0.1 * np.cos(np.pi * np.hypot(x, y) / R)
        depth = np.where(r > 0.85 * R, 20.0, 3000.0)
    else:
        raise ValueError(kind)
    g.update(conc=conc, thick=thick, snow_thick=snow, damage=z_e(), ridge_ratio=z_e(),
             sigma0=z_e(), sigma1=z_e(), sigma2=z_e(),
             conc_young=cyoung, h_young=hyoung, hs_young=z_e(),
             conc_myi=0.3 * conc, thick_myi=0.3 * thick)
    g["cohesion"] = C_fix + C_alea * minstd_uniform01(Ne)   # FE.cpp:3909-3914
    g["time_relaxation_damage"] = np.full(Ne, 25.0 * DAYS_IN_SEC)  # options.cpp:330, FE.cpp:1187
    g["drag_ui"] = np.full(Ne, p.quad_drag_coef_air)        # FE.cpp:565
    g["drag_ui_young"] = np.full(Ne, p.quad_drag_coef_air)  # FE.cpp:570
    g.update(wind=wind, ocean=ocean, ssh=ssh, element_depth=depth)
    return g


NODAL2 = ("VT", "UM", "UT", "wind", "ocean")
NODAL1 = ("ssh",)


def localize_fields(g: dict, lm: LocalMesh, Nn_global: int) -> dict:
    """Restrict global fields to one rank's local numbering ([u|v] blocks keep their layout)."""
    out = {}
    nid = lm.node_gid
    for k, v in g.items():
        if k in NODAL2:
            out[k] = np.ascontiguousarray(np.concatenate([v[nid], v[Nn_global + nid]]))
        elif k in NODAL1:
            out[k] = np.ascontiguousarray(v[nid])
        else:
            out[k] = np.ascontiguousarray(v[lm.elem_gid])
    return out
