"""Randomised parity: the HIP path (through the C ABI) against the oracle on seeded random meshes, states and
parameters -- the corners a hand-picked case does not reach: high-valence vertices (fans longer than the eight
prefetched entries), open water, land-locked triangles, zero thickness, damage 1, southern latitudes, shallow
water with basal stress, young-ice category, every rheology, odd sub-step counts, non-zero initial velocity and
displaced meshes, several patch sizes.  Horizons are short (<= 5 sub-steps + the tail) because the algorithm amplifies
1-ulp differences (tests/test_oracle_sensitivity.py); bar: 1e-11 of each field's scale."""
import numpy as np
import pytest

import cases
from nextsim_amd import _abi, forcing as F, mesh as M
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu
KEYS = ("VT", "UM", "UT", "sigma0", "sigma1", "sigma2", "damage", "conc", "thick", "snow_thick", "ridge_ratio", "conc_young",
        "h_young", "hs_young", "conc_myi", "thick_myi")


def random_mesh(rng, n_pts, hubs):
    """Delaunay mesh of random points in a 600 x 450 km box centred on the pole-ish; `hubs` vertices get a ring of
    10-16 close neighbours (valence well above 8)."""
    L, H = 600e3, 450e3
    x0, y0 = rng.uniform(-1500e3, 900e3), rng.uniform(-1200e3, 800e3)
    nb = 14
    s = np.arange(nb) / nb
    bx = np.concatenate([x0 + L * s, np.full(nb, x0 + L), x0 + L * (1 - s), np.full(nb, x0)])
    by = np.concatenate([np.full(nb, y0), y0 + H * s, np.full(nb, y0 + H), y0 + H * (1 - s)])
    px = x0 + L * rng.uniform(0.04, 0.96, n_pts); py = y0 + H * rng.uniform(0.04, 0.96, n_pts)
    hx, hy = [], []
    for _ in range(hubs):
        cx, cy = x0 + L * rng.uniform(0.25, 0.75), y0 + H * rng.uniform(0.25, 0.75)
        k = int(rng.integers(10, 17)); rad = rng.uniform(9e3, 14e3)
        keep = np.hypot(px - cx, py - cy) > 2.2 * rad                      # clear the neighbourhood: the ring owns the hub
        px, py = px[keep], py[keep]
        ang = 2 * np.pi * (np.arange(k) + rng.uniform(0, 1)) / k
        hx += [cx] + list(cx + rad * np.cos(ang)); hy += [cy] + list(cy + rad * np.sin(ang))
    x = np.concatenate([bx, px, hx]); y = np.concatenate([by, py, hy])
    tri = cases._delaunay(x, y)
    on_b = np.zeros(x.size, bool); on_b[:4 * nb] = True
    neumann = on_b & (x >= x0 + L - 1.0)                                   # the east side is the open boundary
    dirichlet = on_b & ~neumann
    lat = rng.choice([1.0, -1.0]) * (60.0 + 25.0 * rng.random(x.size))      # one hemisphere per mesh (sign enters beta)
    return M.GlobalMesh(x=x, y=y, tri=tri, dirichlet=dirichlet, neumann=neumann, lat=lat, name="fuzz")


def random_case(seed, substeps=None, dynamics_type=None):
    rng = np.random.default_rng(1000 + seed)
    gm = random_mesh(rng, int(rng.integers(150, 900)), int(rng.integers(1, 4)))
    over = {"dynamics_type": [_abi.NXS_DYN_BBM, _abi.NXS_DYN_BBM, _abi.NXS_DYN_MEVP, _abi.NXS_DYN_EVP][seed % 4],
            "substeps": int(rng.integers(1, 6)), "ice_cat_type": int(rng.integers(0, 2)),
            "basal_stress_type": int(rng.integers(0, 2))}
    if substeps is not None: over["substeps"] = substeps
    if dynamics_type is not None: over["dynamics_type"] = dynamics_type
    over["dtime_step"] = 200.0 * over["substeps"] / 120.0                  # the reference's dte
    p = F.default_params(**over)
    p, C_fix, C_alea = F.scale_params_to_mesh(p, gm, alea_factor=0.33)
    g = F.global_fields(gm, p, "arctic", C_fix, C_alea)
    Ne, Nn = gm.num_elements, gm.num_nodes
    # corners of the state space
    pick = lambda frac: rng.random(Ne) < frac  # noqa: E731
    g["conc"] = np.clip(rng.random(Ne) * 1.3 - 0.1, 0., 1.); g["thick"] = g["conc"] * rng.uniform(0.05, 4.0, Ne)
    ow = pick(0.15); g["conc"][ow] = 0.; g["thick"][ow] = 0.
    thin = pick(0.05); g["thick"][thin] = 0.                               # ice area without volume
    low = pick(0.05); g["conc"][low] = 0.1                                 # exactly the hard-coded threshold (Q5)
    g["snow_thick"] = 0.2 * g["thick"] * rng.random(Ne)
    g["damage"] = np.where(pick(0.1), 1.0, rng.random(Ne) ** 3)
    g["ridge_ratio"] = rng.random(Ne) * 0.5
    for k in ("sigma0", "sigma1", "sigma2"):
        g[k] = rng.normal(0, 2e3, Ne) * (g["conc"] > 0.1)
    g["conc_young"] = np.clip(rng.random(Ne) * 0.4, 0., 1. - g["conc"]) * (over["ice_cat_type"] == 1)
    g["h_young"] = 0.2 * g["conc_young"]; g["hs_young"] = 0.02 * g["conc_young"]
    g["conc_myi"] = g["conc"] * rng.random(Ne); g["thick_myi"] = g["thick"] * rng.random(Ne)
    g["element_depth"] = np.where(pick(0.3), rng.uniform(1.0, 15.0, Ne), 2000.)   # shoals: basal stress switches on
    g["VT"] = rng.normal(0, 0.15, 2 * Nn); g["UM"] = rng.normal(0, 40.0, 2 * Nn); g["UT"] = g["UM"] + rng.normal(0, 5.0, 2 * Nn)
    g["wind"] = g["wind"] * rng.uniform(0.2, 2.5) + rng.normal(0, 2.0, 2 * Nn)
    g["ssh"] = g["ssh"] + rng.normal(0, 0.02, Nn)
    lms = M.localize(gm, 1)
    f = F.localize_fields(g, lms[0], Nn)
    bnd = lms[0].mask_dirichlet[:Nn].astype(bool)
    f["VT"][:Nn][bnd] = 0.; f["VT"][Nn:][bnd] = 0.                         # a coast does not move
    return gm, p, lms[0], f, over


@pytest.mark.parametrize("seed", range(24))
def test_random_case_matches_the_oracle(seed):
    from nextsim_amd import dynamics
    gm, p, lm, f, over = random_case(seed)
    nec, _ = dynamics.mesh_connectivity(np.ascontiguousarray(lm.indices, np.int32), lm.num_nodes)
    valence = (~np.isnan(nec)).sum(1).max()
    assert valence >= 10, valence                                          # the fan tail loop (k >= 8) is exercised
    r = O.OracleRank(lm, p, f)
    r.step()
    fe = dynamics.FiniteElementDynamics(p)
    if seed % 3 == 1:
        fe.set_option("patch_nodes", [64, 96, 128, 200][seed % 4])
    if seed % 5 == 2:
        fe.set_option("um_ring", 3)
    fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
    fe.step(); fe.synchronize()
    got = fe.get_state()
    crashed = r.check_fields_fast() != 0
    assert (fe.checkFieldsFast() != 0) == crashed
    if crashed:          # a random state that blows up (|u| > 5 m/s): both sides must say so; its digits mean nothing
        fe.close()
        return
    worst = {}
    for k in KEYS:
        worst[k] = cases.rel_err(got[k], r.arr[k])
    fe.close()
    bad = {k: v for k, v in worst.items() if not (v <= 1e-11)}
    assert not bad, (seed, over, bad)


ROUND3 = {"pair": {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1},        # k_substep_pair: two sub-steps per launch, stresses in registers
          "prep_fused": {"fused": 1, "prep_fused": 1},                            # k_prep_fused: prep elements + prep nodes through LDS
          "resident": {"fused": 4},                                               # k_substep_resident
          "resident_big": {"fused": 4, "patch_nodes": 600},                       # k_substep_resident_big, interior elements one exchange ahead
          "resident_big_plain": {"fused": 4, "patch_nodes": 600, "resident_overlap": 0}}


@pytest.mark.parametrize("variant", sorted(ROUND3))
@pytest.mark.parametrize("seed", range(8))
def test_random_case_on_the_kernels_of_round_3_matches_the_oracle(seed, variant):
    """The same random meshes (hubs of valence 10-16: the fan loops beyond eight entries, the bamg rows beyond ten), states and parameters through
    the kernels round 3 added, each forced by its option: one step against the oracle to 1e-11, and the kernel that ran is the one asked for."""
    from nextsim_amd import dynamics
    substeps = [2, 4, 6][seed % 3]                                         # (even: what k_substep_pair needs; the others do not mind)
    dyn = [_abi.NXS_DYN_BBM, _abi.NXS_DYN_EVP][seed % 2]                   # (no mEVP: its sub-steps need the step's first velocity -- one kernel per sub-step)
    gm, p, lm, f, over = random_case(seed, substeps, dyn)
    r = O.OracleRank(lm, p, f)
    r.step()
    fe = dynamics.FiniteElementDynamics(p)
    for k, v in ROUND3[variant].items(): fe.set_option(k, v)
    fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
    fe.step(); fe.synchronize()
    got = fe.get_state()
    launches = fe.timing()["substep_launches"]
    assert launches == {"pair": substeps // 2, "prep_fused": substeps}.get(variant, 1), (variant, launches)
    crashed = r.check_fields_fast() != 0
    assert (fe.checkFieldsFast() != 0) == crashed
    fe.close()
    if crashed:
        return
    bad = {k: cases.rel_err(got[k], r.arr[k]) for k in KEYS}
    bad = {k: v for k, v in bad.items() if not (v <= 1e-11)}
    assert not bad, (seed, variant, over, bad)
