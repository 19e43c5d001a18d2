"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (fp64).  The device and the oracle perform the same IEEE operations
in the same order (no FMA contraction on either side, node-gather == ascending element order);
the only source of difference is the last-ulp behaviour of libm functions (device OCML vs
glibc: sin, exp, pow, hypot), amplified by the 120 sub-steps:
    after 1 sub-step : rel <= 1e-13 (v, sigma), abs <= 1e-15 (damage)
    after 1 step     : rel <= 1e-10   (measured: 1e-15 .. 5e-14)
    long horizons    : rel <= 1e-12 per SUB-STEP along the oracle's trajectory (the device state is
                       re-seeded from the oracle before every sub-step; 600 sub-steps into heavily
                       damaged ice), plus statistics of a free 10-step run.
"rel" = max|a-b| / max|b| per field.

Why no tight free-running 10-step bound: the reference algorithm itself turns a 1-ulp change of the
wind into O(1e-3..1e-1) differences within 5-10 steps (damage-criterion branches, and the integer
truncation of M_delta_x, Q1) -- tests/test_oracle_sensitivity.py measures that on the oracle alone.
From an evolved (damaged) state even ONE step of 120 sub-steps amplifies 1 ulp to 1e-2.  SURVEY.md
section 8d's "1e-8 after 10 steps" cannot hold for ANY second implementation, including the reference
rebuilt with another libm.
"""
import os

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STATE_KEYS = ("VT", "UM", "UT", "sigma0", "sigma1", "sigma2", "damage", "conc", "thick", "snow_thick",
              "ridge_ratio", "conc_young", "h_young", "hs_young", "conc_myi", "thick_myi")


def _pair(kind="small", nsteps=1, forcing_kind=None, tables=None, options=None, **over):
    from nextsim_amd import dynamics
    from oracle import pyoracle as O
    gm, p, g, lms, fields = cases.make_case(kind, forcing_kind, **over)
    lm, f = lms[0], fields[0]
    fe = dynamics.FiniteElementDynamics(p)
    for k, v in (options or {}).items():
        fe.set_option(k, v)
    fe.set_mesh(lm, tables)
    fe.put_state(f)
    fe.set_forcing(f)
    ref = O.OracleRank(lm, p, f)
    for _ in range(nsteps):
        fe.step()
        ref.step()
    fe.synchronize()
    return fe, ref, lm


def _assert_close(got, ref, keys, tol, what):
    for k in keys:
        err = cases.rel_err(got[k], ref[k])
        assert err <= tol, f"{what}: {k} rel err {err:.3e} > {tol:.1e}"


def test_prep_arrays_bit_exact():
    """K1/K2: everything without a libm call is bit-identical to the serial loops (the node gather
    adds in ascending element order, FE.cpp:10309-10340); fcor (sin) within 2 ulp."""
    fe, ref, lm = _pair("small", 1, substeps=1, dtime_step=200. / 120., options={"work_arrays": 1})
    Nn, Ne = lm.num_nodes, lm.num_elements
    for name, rname, n in (("rlmass", "rlmass_matrix", Nn), ("node_mass", "node_mass", Nn),
                           ("grad_ssh", "grad_ssh", 2 * Nn)):
        assert np.array_equal(fe.debug_array(name), ref.work_array(rname, n)), name
    shape = fe.debug_array("shape").reshape(6, Ne).T.ravel()
    assert np.array_equal(shape, ref.work_array("shape_coeff", 6 * Ne))
    dg = fe.get_diag()
    assert np.array_equal(dg["delta_x"], ref.work_array("delta_x", Ne))      # Q1: integer metres
    assert np.all(dg["delta_x"] == np.floor(dg["delta_x"]))
    np.testing.assert_allclose(fe.debug_array("C_bu"), ref.work_array("C_bu", Nn), rtol=4e-16, atol=0)
    np.testing.assert_allclose(fe.debug_array("fcor"), ref.work_array("fcor", Nn), rtol=4e-16, atol=0)
    np.testing.assert_allclose(dg["D_tau_a"], ref.work_array("D_tau_a", 2 * Nn), rtol=1e-15, atol=1e-18)
    fe.close()


def test_one_substep():
    fe, ref, lm = _pair("small", 1, substeps=1, dtime_step=200. / 120.)
    got = fe.get_state()
    _assert_close(got, ref.arr, ("VT", "sigma0", "sigma1", "sigma2", "UM", "UT"), 1e-13, "1 sub-step")
    assert np.abs(got["damage"] - ref.arr["damage"]).max() <= 1e-15
    fe.close()


@pytest.mark.parametrize("dyn", ["bbm", "evp", "mevp"])
def test_one_step(dyn):
    fe, ref, lm = _pair("small", 1, dynamics_type=dyn)
    got = fe.get_state()
    _assert_close(got, ref.arr, STATE_KEYS, 1e-10, f"1 step {dyn}")
    dg = fe.get_diag()
    Nn, Ne = lm.num_nodes, lm.num_elements
    for k, n in (("surface", Ne), ("D_tau_w", 2 * Nn), ("D_del_ci_ridge_myi", Ne)):
        assert cases.rel_err(dg[k], ref.work_array(k, n)) <= 1e-10, k
    fe.close()


def _micro_step_run(kind, nmicro, **over):
    """Parity along the ORACLE's trajectory, one sub-step at a time: dynamics.substeps = 1 with
    dtime_step = 200/120 s makes every step() a single sub-step (+ prep, smoother, update); before each
    one the device is re-seeded with the oracle's state, so round-off cannot be amplified and a wrong
    branch anywhere in the kernels shows up as an O(1) error at that very sub-step."""
    from nextsim_amd import dynamics
    from oracle import pyoracle as O
    gm, p, g, lms, fields = cases.make_case(kind, substeps=1, dtime_step=200. / 120., **over)
    lm, f = lms[0], fields[0]
    fe = dynamics.FiniteElementDynamics(p); fe.set_mesh(lm); fe.set_forcing(f)
    ref = O.OracleRank(lm, p, f)
    worst = {}
    for it in range(nmicro):
        fe.put_state(ref.arr)
        fe.step(); ref.step()
        got = fe.get_state()
        for k in STATE_KEYS:
            worst[k] = max(worst.get(k, 0.0), cases.rel_err(got[k], ref.arr[k]))
    dmax = float(ref.arr["damage"].max())
    fe.close()
    return worst, dmax


def test_toy_config1_trajectory_sub_step_by_sub_step():
    """BASELINE config 1 (nextsim.toy.cfg semantics: partial ice cover, wind (20,0), alea .33): 600
    sub-steps (= 5 reference steps of 120) along the oracle trajectory, into heavily damaged ice."""
    worst, dmax = _micro_step_run("toy", 600)
    assert dmax > 0.5            # the damage branches are really exercised
    for k, e in worst.items():
        assert e <= 1e-12, f"{k}: {e:.3e}"


@pytest.mark.parametrize("over", [dict(ice_cat_type=0, newice_type=1, basal_stress_type=0), dict(dynamics_type="mevp")])
def test_arctic_trajectory_sub_step_by_sub_step_variants(over):
    # (EVP is not run this way: its relaxation factor 0.5*dte/T, T = dtime_step/3, is unstable when a whole
    #  step is one sub-step; EVP parity is covered by test_one_step)
    worst, dmax = _micro_step_run("small", 240, **over)
    for k, e in worst.items():
        assert e <= 1e-12, f"{k}: {e:.3e}"


def test_toy_free_running_ten_steps_statistics():
    """Free run, 10 steps: the fields decorrelate point-wise (chaos, see module docstring) but the
    run must stay physical and statistically the same as the oracle's."""
    fe, ref, lm = _pair("toy", 10)
    got = fe.get_state()
    assert fe.checkFieldsFast() == 0 == ref.check_fields_fast()
    Nn = lm.num_nodes
    sp_g = np.hypot(got["VT"][:Nn], got["VT"][Nn:]).mean(); sp_r = np.hypot(ref.arr["VT"][:Nn], ref.arr["VT"][Nn:]).mean()
    assert abs(sp_g - sp_r) <= 0.02 * sp_r
    assert abs(got["damage"].mean() - ref.arr["damage"].mean()) <= 0.02
    surf_g = fe.get_diag()["surface"]; surf_r = ref.work_array("surface", lm.num_elements)
    vg, vr = (got["thick"] * surf_g).sum(), (ref.arr["thick"] * surf_r).sum()
    assert abs(vg - vr) <= 1e-4 * vr          # ice volume
    fe.close()


def test_equal_ridging_branch():
    fe, ref, lm = _pair("small", 2, equal_ridging=1)
    _assert_close(fe.get_state(), ref.arr, STATE_KEYS, 1e-9, "equal_ridging")
    fe.close()


def test_free_drift_and_no_motion():
    fe, ref, lm = _pair("small", 2, dynamics_type="free_drift")
    _assert_close(fe.get_state(), ref.arr, ("VT", "UT", "UM"), 1e-14, "free drift")
    fe.close()
    fe, ref, lm = _pair("small", 1, dynamics_type="no_motion")
    got = fe.get_state()
    for k in STATE_KEYS:
        assert np.array_equal(got[k], ref.arr[k])
    fe.close()


def test_run_to_run_bitwise_and_graph_equals_eager():
    """No atomics anywhere: two runs agree bit for bit, and the hipGraph replay of the sub-step loop
    gives the same bits as plain launches."""
    outs = []
    for opts in ({"graph": 1}, {"graph": 1}, {"graph": 0}):
        fe, ref, lm = _pair("small", 2, options=opts)
        outs.append(fe.get_state())
        fe.close()
    for k in STATE_KEYS:
        assert np.array_equal(outs[0][k], outs[1][k]), k
        assert np.array_equal(outs[0][k], outs[2][k]), k


def test_caller_supplied_bamg_tables_equal_internal_ones():
    """Passing bamgmesh's own tables (here: the oracle's restatement, identical to the real bamg, see
    test_connectivity) or letting the library build them must not change a bit."""
    from oracle import pyoracle as O
    gm, p, g, lms, fields = cases.make_case("small")
    tables = O.connectivity(lms[0].indices, lms[0].num_nodes)
    fe1, _, _ = _pair("small", 1, tables=tables)
    fe2, _, _ = _pair("small", 1)
    a, b = fe1.get_state(), fe2.get_state()
    for k in STATE_KEYS:
        assert np.array_equal(a[k], b[k]), k
    fe1.close(); fe2.close()


def test_check_regridding_and_fields_fast():
    fe, ref, lm = _pair("small", 1)
    ang, flip, rg = fe.checkRegridding()
    rang, rflip, rrg = ref.check_regridding()
    assert abs(ang - rang) <= 1e-10 * rang and flip == rflip and rg == rrg
    assert fe.checkFieldsFast() == 0
    # poison: a NaN velocity and an out-of-range concentration must raise the crash flag
    st = fe.get_state()
    full = dict(ref.arr); full.update(st)
    bad = dict(full); bad["VT"] = full["VT"].copy(); bad["VT"][3] = np.nan
    fe.put_state(bad); assert fe.checkFieldsFast() == 1
    bad = dict(full); bad["conc"] = full["conc"].copy(); bad["conc"][5] = 1.5
    fe.put_state(bad); assert fe.checkFieldsFast() == 1
    # a flipped triangle: push one node across the opposite edge
    bad = dict(full); um = full["UM"].copy()
    n0 = lm.indices[0] - 1
    um[n0] += 10 * 125e3
    bad["UM"] = um
    fe.put_state(bad)
    ang, flip, rg = fe.checkRegridding()
    ref.arr["UM"][:] = um
    rang, rflip, rrg = ref.check_regridding()
    assert flip == rflip == 1 and rg == rrg == 1
    fe.close()


def test_ice_diagnostics_kernel_and_a_moorings_record_from_device_arrays():
    """K14 updateIceDiagnostics (FE.cpp:7860-7905) as a kernel on the device-resident state -- sigma read from the records the sub-step loop leaves
    behind, the divergence from shapeCoeff on the displaced mesh: bit for bit the oracle's restatement on the SAME state, D_sigma[1] to an ulp
    (OCML's hypot against glibc's).  And the Moorings sampling (gridoutput.cpp:496) fed from the device rows gives, bit for bit, the grid it gives
    from the host arrays: a record needs no round trip of the element state."""
    from nextsim_amd import interp
    fe, ref, lm = _pair("small", 2)
    got = fe.get_state()
    host, dev = fe.updateIceDiagnostics()
    only_dev, dev2 = fe.updateIceDiagnostics(want_host=False)
    assert only_dev is None and dev2 == dev and dev
    # the oracle's diagnostics of the DEVICE's state (the two states differ in the last bits after two steps)
    from oracle import pyoracle as O
    f = dict(ref.arr); f.update({k: got[k] for k in got})
    want = O.OracleRank(lm, fe.params, f).ice_diagnostics()
    for k in ("D_conc", "D_thick", "D_snow_thick", "D_sigma0", "D_divergence"):
        assert np.array_equal(host[k], want[k]), k
    np.testing.assert_allclose(host["D_sigma1"], want["D_sigma1"], rtol=4e-16, atol=0)
    assert np.abs(host["D_divergence"]).max() > 0 and np.abs(host["D_sigma1"]).max() > 0
    tri = lm.indices.reshape(-1, 3)
    x0, x1, y0, y1 = lm.coord_x.min(), lm.coord_x.max(), lm.coord_y.min(), lm.coord_y.max()
    nrows, ncols = 120, 90
    args = (x0, y1, (x1 - x0) / (nrows - 1), (y1 - y0) / (ncols - 1), nrows, ncols, -1e14)
    rows = np.column_stack([host[k] for k in ("D_conc", "D_thick", "D_snow_thick", "D_sigma0", "D_sigma1", "D_divergence")])
    a = interp.InterpFromMeshToGridx(tri, lm.coord_x, lm.coord_y, rows, *args)
    b = interp.InterpFromMeshToGridx_device(tri, lm.coord_x, lm.coord_y, dev, lm.num_elements, 6, *args)
    assert np.array_equal(a, b) and (a[..., 0] != -1e14).mean() > 0.3
    fe.close()


def test_step_host_drop_in():
    """nxs_dyn_step_host = the three lines of FiniteElement::step() on host vectors."""
    import ctypes as C
    from nextsim_amd import _abi, dynamics
    from oracle import pyoracle as O
    gm, p, g, lms, fields = cases.make_case("small")
    lm = lms[0]
    f = {k: v.copy() for k, v in fields[0].items()}
    fe = dynamics.FiniteElementDynamics(p)
    fe.set_mesh(lm)
    s = _abi.state_struct(f); fo = _abi.forcing_struct(f)
    assert fe.L.nxs_dyn_step_host(fe.h, C.byref(s), C.byref(fo)) == 0
    ref = O.OracleRank(lm, p, fields[0]); ref.step()
    _assert_close(f, ref.arr, STATE_KEYS, 1e-10, "step_host")
    fe.close()


def test_error_codes_on_gpu():
    from nextsim_amd import dynamics
    gm, p, g, lms, fields = cases.make_case("tiny")
    fe = dynamics.FiniteElementDynamics(p)
    with pytest.raises(dynamics.NxsError) as e:
        fe.step()
    assert e.value.code == -4  # NXS_ERR_STATE: step before set_mesh
    lm = lms[0]
    bad = type(lm)(**{**lm.__dict__, "indices": lm.indices.copy()})
    bad.indices[0] = lm.num_nodes + 7
    with pytest.raises(dynamics.NxsError) as e:
        fe.set_mesh(bad)
    assert e.value.code == -1
    q = p.copy(); q.substeps = 0
    with pytest.raises(dynamics.NxsError):
        fe.set_params(q)
    fe.close()


def test_allocation_failure_in_set_mesh_is_a_status_code_and_the_handle_lives_on():
    """'Never throws across the ABI': with the address space of a child process capped 512 MB above what it uses, nxs_dyn_set_mesh is handed a
    NodalElementConnectivity of 2^21 columns (its int table alone would take 12 GB): the std::bad_alloc is caught at the boundary
    (csrc/nxs_guard.hpp) and comes back as NXS_ERR_NOMEM with a text; with the cap lifted the same handle takes the mesh and steps."""
    import subprocess
    import sys
    code = r'''
import resource, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import cases
from nextsim_amd import dynamics
gm, p, g, lms, fields = cases.make_case("small")
lm = lms[0]
fe = dynamics.FiniteElementDynamics(p)
fe.set_mesh(lm); fe.put_state(fields[0]); fe.set_forcing(fields[0]); fe.step(); fe.synchronize()
ref = fe.get_state()["VT"].copy()
nec, nc = dynamics.mesh_connectivity(lm.indices, lm.num_nodes)
class Wide:
    shape = (lm.num_nodes, 2 ** 21)
    dtype, flags, ctypes = nec.dtype, nec.flags, nec.ctypes
soft, hard = resource.getrlimit(resource.RLIMIT_AS)
vm = int(open("/proc/self/statm").read().split()[0]) * resource.getpagesize()
resource.setrlimit(resource.RLIMIT_AS, (vm + (512 << 20), hard))
try:
    fe.set_mesh(lm, (Wide(), nc))
    print("no error")
except dynamics.NxsError as e:
    print("code", e.code, str(e))
resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
fe.set_mesh(lm); fe.put_state(fields[0]); fe.set_forcing(fields[0]); fe.step(); fe.synchronize()
print("same bits", bool(np.array_equal(fe.get_state()["VT"], ref)))
fe.close()
''' % (cases.__file__.rsplit("/", 2)[0], cases.__file__.rsplit("/", 1)[0])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "code -6" in r.stdout and "bad_alloc" in r.stdout and "same bits True" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


# ---- full-size properties (the oracle would take minutes here; use what the domain offers) ----

def test_full_size_10km_invariants_and_rigid_state():
    """BASELINE config 2 size (~60k triangles): the reference's own runtime invariants
    (FE.cpp:14541-14557) hold after 5 steps; thick*area is conserved by update() where nothing is
    capped; a state at rest with no forcing stays at rest."""
    from nextsim_amd import dynamics
    gm, p, g, lms, fields = cases.make_case("10km")
    lm, f = lms[0], fields[0]
    fe = dynamics.FiniteElementDynamics(p)
    fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
    x, y, tri = lm.coord_x, lm.coord_y, lm.indices.reshape(-1, 3) - 1
    def area(um):
        X = x + um[:lm.num_nodes]; Y = y + um[lm.num_nodes:]
        return 0.5 * np.abs((X[tri[:, 1]] - X[tri[:, 0]]) * (Y[tri[:, 2]] - Y[tri[:, 0]]) - (X[tri[:, 2]] - X[tri[:, 0]]) * (Y[tri[:, 1]] - Y[tri[:, 0]]))
    vol0 = f["thick"] * area(f["UM"])
    for _ in range(5):
        fe.step()
    fe.synchronize()
    s = fe.get_state()
    assert fe.checkFieldsFast() == 0
    assert s["damage"].min() >= 0 and s["damage"].max() <= 1
    assert s["conc"].min() >= 0 and s["conc"].max() <= 1
    Nn = lm.num_nodes
    assert np.hypot(s["VT"][:Nn], s["VT"][Nn:]).max() < 5.0
    # ice volume of elements that are not on the open boundary and never hit a cap is conserved
    on_neumann = np.isin(tri, lm.neumann_flags).any(1)
    vol1 = s["thick"] * area(s["UM"])
    ok = (~on_neumann) & (f["conc_young"] == 0) & (s["conc"] > 0) & (s["thick"] / np.maximum(s["conc"], 1e-300) < 49)
    assert ok.sum() > 0.5 * ok.size
    np.testing.assert_allclose(vol1[ok], vol0[ok], rtol=1e-12)
    # at rest + no forcing + flat ssh -> stays exactly at rest
    z = {k: v.copy() for k, v in f.items()}
    for k in ("wind", "ocean", "ssh", "VT"):
        z[k][:] = 0.0
    fe.put_state(z); fe.set_forcing(z); fe.step(); fe.synchronize()
    s = fe.get_state()
    assert np.all(s["VT"] == 0) and np.all(s["UM"] == 0) and np.all(s["sigma0"] == 0)
    fe.close()


@pytest.mark.parametrize("kind,state,options,expect_launches", [("2km", "arctic", {}, 60), ("2km", "arctic_ow", {}, 60), ("2km", "arctic", {"pair_regs": 0}, 120),
                                                                 ("h15600", "arctic", {"fused": 4}, 1), ("10km", "arctic_ow", {}, 30)])
def test_the_bench_workloads_themselves_against_the_oracle(kind, state, options, expect_launches):
    """What bench.py times, tied to the oracle DIRECTLY (not through the per-loop kernels): the 2 km mesh (BASELINE's headline configuration,
    1.46 M triangles) on a single rank with the library's automatic kernel choice -- two sub-steps per launch on 428-node patches with the stresses
    in registers (k_substep_pair), the fused prep kernel, the once-per-step ring flush, ten smoother sweeps per launch; and with one launch per
    sub-step (pair_regs = 0: streaming hints, 476-node patches in three element rounds) -- with bench.py's two states ('arctic' and 'arctic_ow': 29 % of the triangles
    ice free, 4 % in the 0 < A <= 0.1 band); the 182 k-triangle partition of aux_partition_floor in ONE resident launch (fused = 4); the 10 km
    mesh of aux_10km on the several-sub-steps kernel.  One full step (120 BBM sub-steps + 50 sweeps + update), every state array <= 1e-10 of
    the serial oracle (which takes ~9 s at 2 km and runs in a thread beside the GPU)."""
    import threading
    from nextsim_amd import dynamics
    from oracle import pyoracle as O
    gm, p, g, lms, fields = cases.make_case(kind, state)
    lm, f = lms[0], fields[0]
    ref = O.OracleRank(lm, p, f, fast=False)
    th = threading.Thread(target=ref.step)
    th.start()
    fe = dynamics.FiniteElementDynamics(p)
    for k, v in options.items():
        fe.set_option(k, v)
    fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
    fe.step(); fe.synchronize()
    got = fe.get_state()
    assert fe.timing()["substep_launches"] == expect_launches      # the kernel family bench.py reports for this workload
    assert fe.checkFieldsFast() == 0
    dg = fe.get_diag()
    fe.close()
    th.join()
    _assert_close(got, ref.arr, [k for k in STATE_KEYS if k != "ridge_ratio"], 1e-10, f"{kind} / {state}: one step of the bench workload")
    # M_ridge_ratio after ONE step is 1 - (1 - R) min(1, C) / (C_old * surface ratio) with R = 0 and a surface ratio within 1e-6 of 1 (FE.cpp:3983): a
    # difference of two numbers next to 1, i.e. values of ~1e-6 that inherit the ABSOLUTE error of the surface ratio (1e-15, the mesh displacement's) -- a few
    # 1e-10 of their own maximum; the ratio is a fraction of one and the bound that means something is absolute (measured 2e-15 at 2 km)
    assert np.abs(got["ridge_ratio"] - ref.arr["ridge_ratio"]).max() <= 1e-13, "ridge_ratio"
    Nn, Ne = lm.num_nodes, lm.num_elements
    for k, n in (("surface", Ne), ("D_tau_w", 2 * Nn), ("D_tau_a", 2 * Nn)):
        assert cases.rel_err(dg[k], ref.work_array(k, n)) <= 1e-10, k
    if state == "arctic_ow":   # the workload really has open water: nodes without mass, which the smoother moves
        assert (f["conc"] == 0).mean() > 0.2


def test_full_size_2km_bench_workload_two_kernel_families_agree_bit_for_bit():
    """The workload bench.py times (BASELINE's 2 km configuration, 1.46 M triangles) is far beyond what the oracle
    finishes in seconds.  Size-independent evidence instead: (1) the fused patch kernel and the per-loop kernels --
    two implementations with different data flow (LDS patches + ring of velocity buffers vs global corner forces),
    each pinned to the oracle at small sizes -- give the SAME BITS for every prognostic array after a full step
    (a checksum of checksums would hide which array differs; the arrays themselves are compared); (2) the
    reference's runtime invariants hold; (3) ice volume is conserved by update() where nothing is capped."""
    from nextsim_amd import dynamics
    gm, p, g, lms, fields = cases.make_case("2km")
    lm, f = lms[0], fields[0]
    out = []
    for fused in (1, 0):
        fe = dynamics.FiniteElementDynamics(p)
        fe.set_option("fused", fused)
        fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
        fe.step(); fe.synchronize()
        assert fe.checkFieldsFast() == 0
        out.append(fe.get_state())
        fe.close()
    for k in STATE_KEYS:
        assert np.array_equal(out[0][k], out[1][k]), k
    s = out[0]
    Nn = lm.num_nodes
    assert s["damage"].min() >= 0 and s["damage"].max() <= 1 and s["conc"].min() >= 0 and s["conc"].max() <= 1
    assert np.hypot(s["VT"][:Nn], s["VT"][Nn:]).max() < 5.0 and np.abs(s["UM"]).max() > 0
    x, y, tri = lm.coord_x, lm.coord_y, lm.indices.reshape(-1, 3) - 1

    def area(um):
        X = x + um[:Nn]; Y = y + um[Nn:]
        return 0.5 * np.abs((X[tri[:, 1]] - X[tri[:, 0]]) * (Y[tri[:, 2]] - Y[tri[:, 0]]) - (X[tri[:, 2]] - X[tri[:, 0]]) * (Y[tri[:, 1]] - Y[tri[:, 0]]))
    on_neumann = np.isin(tri, lm.neumann_flags).any(1)
    ok = (~on_neumann) & (f["conc_young"] == 0) & (s["conc"] > 0) & (s["thick"] / np.maximum(s["conc"], 1e-300) < 49)
    assert ok.sum() > 0.5 * ok.size
    np.testing.assert_allclose((s["thick"] * area(s["UM"]))[ok], (f["thick"] * area(f["UM"]))[ok], rtol=1e-12)


# ---- v2 fused sub-step kernel vs the v1 reference-loop kernels ----

@pytest.mark.parametrize("dyn,substeps", [("bbm", 120), ("evp", 120), ("mevp", 120), ("bbm", 3), ("bbm", 1)])
def test_fused_substep_kernel_is_bitwise_equal_to_the_per_loop_kernels(dyn, substeps):
    """The fused kernel (node patches, LDS-staged assembly, ping-pong buffers) performs the same
    operations in the same order as k_sigma_* + k_solve_move: identical bits, for even and odd numbers of
    sub-steps (the odd case ends with a copy back from the secondary buffers)."""
    outs = []
    for fused in (1, 0):
        fe, ref, lm = _pair("small", 2, options={"fused": fused}, dynamics_type=dyn, substeps=substeps,
                            dtime_step=200. * substeps / 120.)
        outs.append(fe.get_state())
        fe.close()
    for k in STATE_KEYS:
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("smooth_depth", [0, 5, 10, 25])
def test_smoother_with_open_water_several_sweeps_per_launch_equal_sweep_by_sweep(smooth_depth):
    """The toy case has an ice-free strip: the open-water smoother really changes velocities there.  fused=1 runs it 5, 10 or 25 sweeps
    per launch on node-ring patches of that many rings (k_smooth_multi; 0 = automatic: ten where they fit the LDS), fused=0 one sweep per
    launch (k_smooth): the same bits after three steps."""
    outs = []
    for options in ({"fused": 1, "smooth_depth": smooth_depth}, {"fused": 0}):
        fe, ref, lm = _pair("toy", 3, options=options)
        outs.append(fe.get_state())
        fe.close()
    for k in STATE_KEYS:
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("um_ring,substeps", [(1, 120), (5, 120), (16, 120), (120, 120), (8, 13), (7, 7)])
def test_deferred_mesh_move_ring_does_not_change_a_bit(um_ring, substeps):
    """UM/UT are advanced every um_ring sub-steps from a ring of velocity buffers: the same additions in the
    same order as moving the mesh every sub-step (v1 path), for ring sizes that do and do not divide S."""
    over = dict(substeps=substeps, dtime_step=200. * substeps / 120.)
    a, _, _ = _pair("small", 2, options={"fused": 0}, **over)
    b, _, _ = _pair("small", 2, options={"um_ring": um_ring}, **over)
    x, y = a.get_state(), b.get_state()
    for k in STATE_KEYS:
        assert np.array_equal(x[k], y[k]), k
    a.close(); b.close()


def _depth(substeps, want=4):
    d = min(want, substeps, 8)
    while d > 1 and substeps % d:
        d -= 1
    return d


@pytest.mark.parametrize("dyn,substeps,opts", [("bbm", 120, {}), ("evp", 120, {}), ("bbm", 120, {"um_ring": 16}), ("bbm", 120, {"um_ring": 2}),
                                               ("bbm", 6, {"pair_nodes": 16}), ("bbm", 120, {"pair_nodes": 300}), ("bbm", 2, {}),
                                               ("bbm", 7, {}), ("mevp", 120, {}), ("bbm", 120, {"substeps_per_launch": 2}),
                                               ("bbm", 120, {"substeps_per_launch": 3}), ("bbm", 120, {"substeps_per_launch": 8}),
                                               ("evp", 120, {"substeps_per_launch": 6, "pair_nodes": 40}), ("bbm", 10, {"substeps_per_launch": 5})])
def test_several_sub_steps_per_launch_do_not_change_a_bit(dyn, substeps, opts):
    """fused=2 (k_substep_multi): D sub-steps in one launch on patches with D rings of halo -- the rings are
    recomputed by the neighbouring patches with the same operations, so every array has the bits of the per-loop
    kernels; sub-step counts no depth divides and mEVP (no deferred mesh move) fall back to one sub-step per launch."""
    outs, launches = [], []
    for options in (dict(opts, fused=2), {"fused": 0}):
        fe, ref, lm = _pair("small", 2, options=options, dynamics_type=dyn, substeps=substeps, dtime_step=200. * substeps / 120.)
        outs.append(fe.get_state())
        launches.append(fe.timing()["substep_launches"])
        fe.close()
    for k in STATE_KEYS:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    if dyn != "mevp":
        assert launches[0] == substeps // _depth(substeps, opts.get("substeps_per_launch", 4))


@pytest.mark.parametrize("kind,over,opts,launches", [
    ("small", {}, {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1}, 60),                       # the planner's own patch size
    ("40km", {}, {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1, "pair_nodes": 150}, 60),
    ("40km", {"dynamics_type": 3}, {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1, "pair_nodes": 64}, 60),   # EVP
    ("small", {"substeps": 10, "dtime_step": 200. * 10 / 120}, {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1, "um_ring": 4}, 5),
    ("shuffled", {}, {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1}, 60),                    # patches along the Hilbert curve
    ("h15600", {}, {}, 60),                                                                          # the default above 65 k nodes
    ("h15600", {"substeps": 7, "dtime_step": 200. * 7 / 120}, {}, 7),                                 # an odd count: one sub-step per launch
    ("40km", {}, {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1, "pair_nodes": 1000}, 120),   # patches too large for it: refused, one sub-step per launch
])
def test_two_sub_steps_per_launch_with_the_stresses_in_registers_do_not_change_a_bit(kind, over, opts, launches):
    """k_substep_pair: temporal blocking at depth 2 for meshes that stream from HBM -- sub-step 0 updates every element touching the patch's
    first ring of nodes and solves that ring, sub-step 1 updates the patch's own elements and solves its own nodes; stress and damage stay in the
    registers of the thread that updates the element in both sub-steps, LDS holds staged velocities, frozen coordinates and corner forces only, two
    workgroups share a CU.  The bits of one launch per sub-step: BBM and EVP, the planner's patch size and explicit ones, a short ring, a
    numbering without locality; the default for single-rank meshes above 65 k nodes; odd sub-step counts and patches too large for it fall back."""
    from nextsim_amd import dynamics
    states = []
    for options in (opts, {"fused": 1}):
        if kind == "shuffled": p, lm, f = _shuffled_case()
        else:
            _, p, _, lms, fields = cases.make_case(kind, **over)
            lm, f = lms[0], fields[0]
        fe = dynamics.FiniteElementDynamics(p)
        for k, v in options.items(): fe.set_option(k, v)
        fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
        fe.step(); fe.step(); fe.synchronize()
        states.append((fe.get_state(), fe.timing()["substep_launches"]))
        fe.close()
    (sa, la), (sb, lb) = states
    for k in STATE_KEYS:
        assert np.array_equal(sa[k], sb[k]), k
    assert la == launches and lb == over.get("substeps", 120)


@pytest.mark.parametrize("kind,over,opts", [
    ("h15600", {}, {}),                                                                               # the planner's patches (more than one task per workgroup)
    ("small", {}, {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1}),                             # a mesh smaller than the device: fewer patches than workgroups
    ("40km", {"dynamics_type": 3}, {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1, "pair_nodes": 64}),   # EVP, small patches: many tasks per workgroup, short ones
    ("shuffled", {}, {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1}),                          # patches along the Hilbert curve
    ("h15600", {"substeps": 10, "dtime_step": 200. * 10 / 120}, {}),
])
def test_the_pairs_of_a_step_in_one_data_flow_launch_do_not_change_a_bit(kind, over, opts):
    """k_substep_flow (option pair_flow = 1): the patches of k_substep_pair, every pair of sub-steps of the step in ONE launch whose workgroups take (pair, patch)
    tasks from queues and wait for the patches around theirs only -- the device stays full from the first task to the last instead of draining 60 times a step.
    What the patches hand each other crosses memory inside the launch (write-through stores, loads past the L1, per-patch counters).  Three steps, every
    prognostic array the bits of one launch per pair and of one launch per sub-step; the kernel reports itself; one launch per step."""
    from nextsim_amd import dynamics
    states = []
    for extra in ({"pair_flow": 1}, {"pair_flow": 0}, None):
        if kind == "shuffled": p, lm, f = _shuffled_case()
        else:
            _, p, _, lms, fields = cases.make_case(kind, **over)
            lm, f = lms[0], fields[0]
        fe = dynamics.FiniteElementDynamics(p)
        for k, v in (dict(opts, **extra) if extra is not None else {"fused": 1}).items(): fe.set_option(k, v)
        fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
        fe.step(); fe.step(); fe.step(); fe.synchronize()
        states.append((fe.get_state(), fe.timing()["substep_launches"], fe.traffic_model()["substep_kernel_name"], fe.checkFieldsFast()))
        fe.close()
    S = over.get("substeps", 120)
    assert [s[1] for s in states] == [1, S // 2, S] and [s[2] for s in states] == ["k_substep_flow", "k_substep_pair", "k_substep_fused"], [s[1:] for s in states]
    assert all(s[3] == 0 for s in states)
    for k in STATE_KEYS:
        assert np.array_equal(states[0][0][k], states[1][0][k]), k
        assert np.array_equal(states[0][0][k], states[2][0][k]), k


@pytest.mark.parametrize("kind,over,opts", [
    ("h15600", {}, {}),
    ("small", {}, {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1}),
    ("40km", {"dynamics_type": 3}, {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1, "pair_nodes": 64}),   # EVP
    ("shuffled", {}, {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1}),
    ("h15600", {"substeps": 10, "dtime_step": 200. * 10 / 120}, {}),                                   # ten sub-steps: the last velocity ends in a ring slot, not in M_VT
    ("toy", {}, {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1}),                               # open boundaries: Neumann nodes keep M_UM
])
def test_the_mesh_move_inside_the_pair_launch_does_not_change_a_bit(kind, over, opts):
    """k_substep_pair<MOVE> (option pair_move = 1, single rank): the launch applies the mesh move of its two sub-steps to its own nodes itself -- M_UM and M_UT
    read and written once per launch, the two additions per component in the order k_move_ring makes them -- so the step needs no ring of 120 velocity slots
    and no flush.  Three steps: every prognostic array (M_UM and M_UT among them) the bits of the deferred move (pair_move = 0) and of one launch per sub-step."""
    from nextsim_amd import dynamics
    states = []
    for extra in ({"pair_move": 1}, {"pair_move": 0}, None):
        if kind == "shuffled": p, lm, f = _shuffled_case()
        else:
            _, p, _, lms, fields = cases.make_case(kind, **over)
            lm, f = lms[0], fields[0]
        fe = dynamics.FiniteElementDynamics(p)
        for k, v in (dict(opts, **extra) if extra is not None else {"fused": 1}).items(): fe.set_option(k, v)
        fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
        fe.step(); fe.step(); fe.step(); fe.synchronize()
        states.append((fe.get_state(), fe.timing(), fe.traffic_model(), fe.checkFieldsFast()))
        fe.close()
    S = over.get("substeps", 120)
    assert [s[1]["substep_launches"] for s in states] == [S // 2, S // 2, S], [s[1] for s in states]
    assert states[0][2]["move_ring_slots"] == 0 and states[1][2]["move_ring_slots"] == S and states[0][1]["ring_flush_ms"] == 0.
    assert all(s[3] == 0 for s in states)
    assert np.abs(states[0][0]["UM"]).max() > 0
    for k in STATE_KEYS:
        assert np.array_equal(states[0][0][k], states[1][0][k]), k
        assert np.array_equal(states[0][0][k], states[2][0][k]), k


def test_sums_without_the_literal_zero_terms_are_the_reference_sums():
    """updateSigmaDamage forms its strain rates and its stress increment with literal zeros of M_B0T and M_Dunit among the terms (FE.cpp:4167-4176, 4204-4210); the
    kernels leave those products out (adding +-0 to a sum that cannot be -0 returns it unchanged).  4 x 10^8 random operand sets, zeros of both signs among the
    velocities and the stresses, with and without the terms: the same bits."""
    from nextsim_amd import dynamics
    for seed in (3, 977):
        assert dynamics.selftest_quotients(200_000_000, seed=seed, mode=2) == 0


@pytest.mark.parametrize("mode", [0, 1])
def test_quotients_by_one_divisor_are_the_divisions(mode):
    """The six shape coefficients of a triangle are six quotients by its Jacobian; the sub-step kernels refine the Jacobian's reciprocal once and finish every
    quotient with the three operations the compiler's division ends in (quotients_by_one_divisor) -- the same bits as six divisions wherever v_div_scale leaves
    the operands alone.  2 x 10^8 sextuples per mode (triangles as meshes have them; operands over the whole range the per-step check admits, zeros among them)
    computed both ways on the device: not one quotient differs in a bit."""
    from nextsim_amd import dynamics
    for seed in (1, 20261005):
        assert dynamics.selftest_quotients(200_000_000, seed=seed, mode=mode) == 0


def test_coordinates_outside_the_checked_range_fall_back_to_six_divisions():
    """The range in which the quotients may share a reciprocal is checked once per step by the prep kernels; a single coordinate outside it (here: a node 1e-150 m
    from the y axis) raises the flag for the step, every sub-step kernel divides six times, k_update lowers the flag again.  Same bits as the per-loop kernels
    (which read shape coefficients formed by division) either way; an ordinary mesh leaves the flag down."""
    from nextsim_amd import dynamics
    import copy
    _, p, _, lms, fields = cases.make_case("small")
    lm0, f = lms[0], fields[0]
    lm = copy.deepcopy(lm0)
    k = int(np.argmin(np.abs(lm.coord_x)))
    tri0 = lm.indices.reshape(-1, 3)[0] - 1
    assert abs(lm.coord_x[k]) < 0.25 * np.hypot(lm.coord_x[tri0[1]] - lm.coord_x[tri0[0]], lm.coord_y[tri0[1]] - lm.coord_y[tri0[0]])   # (the mesh stays a mesh)
    lm.coord_x = lm.coord_x.copy(); lm.coord_x[k] = 1e-150
    flags, states = [], []
    for mesh in (lm, lm0):
        for opts in ({}, {"fused": 1}, {"fused": 0}):
            fe = dynamics.FiniteElementDynamics(p)
            for kk, v in opts.items(): fe.set_option(kk, v)
            fe.set_mesh(mesh); fe.put_state(f); fe.set_forcing(f)
            fe.explicitSolve(); fe.synchronize()
            flags.append(fe.debug_array("shape_range")[0])
            fe.update(); fe.synchronize()
            assert fe.debug_array("shape_range")[0] == 0
            fe.step(); fe.synchronize()
            states.append(fe.get_state())
            fe.close()
    assert flags == [1, 1, 1, 0, 0, 0], flags
    for base in (0, 3):
        for k in STATE_KEYS:
            assert np.array_equal(states[base][k], states[base + 2][k]), k
            assert np.array_equal(states[base + 1][k], states[base + 2][k]), k


def test_two_sub_steps_per_launch_survive_a_change_of_sub_steps_and_a_remesh():
    """What k_substep_pair's patches and ring are tied to may change under it: nxs_dyn_set_params with an odd number of sub-steps (one launch per
    sub-step from then on), back to an even one (pairs again, another ring length), nxs_dyn_set_mesh with another mesh (the planner starts from
    the patch size it kept).  Always the bits of one launch per sub-step doing the same."""
    from nextsim_amd import dynamics
    _, p1, _, lms1, f1 = cases.make_case("h15600")
    _, p2, _, lms2, f2 = cases.make_case("h19000", dtime_step=200. * 10 / 120, substeps=10)

    def run(opts):
        fe = dynamics.FiniteElementDynamics(p1)
        for k, v in opts.items(): fe.set_option(k, v)
        fe.set_mesh(lms1[0]); fe.put_state(f1[0]); fe.set_forcing(f1[0]); fe.step()
        launches = [fe.timing()["substep_launches"]]
        q = p1.copy(); q.substeps = 7; q.dtime_step = 200. * 7 / 120
        fe.set_params(q); fe.step(); fe.synchronize(); launches.append(fe.timing()["substep_launches"])
        q = p1.copy(); q.substeps = 10; q.dtime_step = 200. * 10 / 120
        fe.set_params(q); fe.step(); fe.synchronize(); launches.append(fe.timing()["substep_launches"])
        fe.set_params(p2); fe.set_mesh(lms2[0]); fe.put_state(f2[0]); fe.set_forcing(f2[0]); fe.step(); fe.synchronize()
        launches.append(fe.timing()["substep_launches"])
        return fe, launches
    (a, la), (b, lb) = run({"fused": 2, "substeps_per_launch": 2, "pair_regs": 1}), run({"fused": 1})
    sa, sb = a.get_state(), b.get_state()
    for k in STATE_KEYS:
        assert np.array_equal(sa[k], sb[k]), k
    assert la == [60, 7, 5, 5] and lb == [120, 7, 10, 10], (la, lb)
    a.close(); b.close()


def test_automatic_choice_of_the_sub_step_kernel():
    """Default (fused = 3): four sub-steps per launch on a single-rank mesh that lives in the caches, one per launch when the
    pairing is impossible (odd count) -- and the same bits either way."""
    a, _, _ = _pair("small", 1)
    assert a.timing()["substep_launches"] == 30
    b, _, _ = _pair("small", 1, options={"fused": 1})
    assert b.timing()["substep_launches"] == 120
    x, y = a.get_state(), b.get_state()
    for k in STATE_KEYS:
        assert np.array_equal(x[k], y[k]), k
    c, _, _ = _pair("small", 1, substeps=7, dtime_step=200. * 7 / 120.)
    assert c.timing()["substep_launches"] == 7
    a.close(); b.close(); c.close()


@pytest.mark.parametrize("kind,nsteps", [("toy", 100), ("10km", 60), ("h15600", 40)])
def test_long_free_run_default_kernels_equal_one_sub_step_per_launch_bit_for_bit(kind, nsteps):
    """A long free run (the mesh moves, damage localises, the toy case has an open-water strip for the smoother): the default
    kernels of a small single-rank mesh (four sub-steps and four smoother sweeps per launch, one ring flush per step) -- and, h15600, of one above
    65 k nodes (two sub-steps per launch with the stresses in registers and the mesh move inside the launch: what the 2 km headline runs on) --
    against the one-sub-step-per-launch kernels: every prognostic array bit for bit after the last step."""
    from nextsim_amd import dynamics
    gm, p, g, lms, fields = cases.make_case(kind)
    lm, f = lms[0], fields[0]
    out = []
    for fused in (3, 1):
        fe = dynamics.FiniteElementDynamics(p)
        fe.set_option("fused", fused)
        fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
        for _ in range(nsteps):
            fe.step()
        fe.synchronize()
        assert fe.timing()["substep_launches"] == ((60 if kind == "h15600" else 30) if fused == 3 else 120)
        if kind == "h15600" and fused == 3: assert fe.traffic_model()["substep_kernel_name"] == "k_substep_pair" and fe.traffic_model()["move_ring_slots"] == 0
        out.append(fe.get_state())
        fe.close()
    for k in STATE_KEYS:
        assert np.array_equal(out[0][k], out[1][k]), k
    assert np.abs(out[0]["UM"]).max() > 0 and out[0]["damage"].max() > 0


def test_full_size_2km_several_sub_steps_per_launch_agree_bit_for_bit():
    from nextsim_amd import dynamics
    gm, p, g, lms, fields = cases.make_case("2km")
    lm, f = lms[0], fields[0]
    out = []
    for fused in (2, 1):
        fe = dynamics.FiniteElementDynamics(p)
        fe.set_option("fused", fused)
        fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
        fe.step(); fe.step(); fe.synchronize()
        assert fe.checkFieldsFast() == 0
        out.append(fe.get_state())
        fe.close()
    for k in STATE_KEYS:
        assert np.array_equal(out[0][k], out[1][k]), k


@pytest.mark.parametrize("patch_nodes", [64, 200, 1024])
def test_patch_size_does_not_change_a_bit(patch_nodes):
    base, _, _ = _pair("toy", 2)
    other, _, _ = _pair("toy", 2, options={"patch_nodes": patch_nodes})
    a, b = base.get_state(), other.get_state()
    for k in STATE_KEYS:
        assert np.array_equal(a[k], b[k]), k
    base.close(); other.close()


@pytest.mark.parametrize("kind,over,opts", [("toy", {}, {"fused": 2}), ("toy", {"dynamics_type": 3}, {"fused": 2}), ("small", {}, {"fused": 2}),
                                            ("10km", {}, {})])
def test_shape_coefficients_from_records_or_rebuilt_do_not_change_a_bit(kind, over, opts):
    """Option shape_mem: the several-sub-steps kernel reads M_shape_coeff from per-step records (default) or rebuilds it from the
    staged frozen coordinates as the one-sub-step kernel does: the same quotients either way -- BBM and EVP, three meshes."""
    a, _, _ = _pair(kind, 2, options=dict(opts, shape_mem=0), **over)
    b, _, _ = _pair(kind, 2, options=dict(opts, shape_mem=1), **over)
    sa, sb = a.get_state(), b.get_state()
    for k in STATE_KEYS:
        assert np.array_equal(sa[k], sb[k]), k
    a.close(); b.close()


@pytest.mark.parametrize("kind,over", [("h15600", {}), ("small", {}), ("toy", {"dynamics_type": 3}), ("h15600", {"substeps": 7, "dtime_step": 200. * 7 / 120})])
def test_resident_sub_step_loop_does_not_change_a_bit(kind, over):
    """Option fused = 4: ONE launch for the whole sub-step loop, every patch resident on its CU and waiting for its neighbouring
    patches only (stress, damage and element constants in registers, nodal inputs and the moving mesh in LDS, velocities
    exchanged through write-through stores and cache-bypassing loads) -- bit for bit the one-kernel-per-sub-step path: 182 k
    triangles (the size it is meant for: 511 patches, two per CU), a few patches, EVP, an odd number of sub-steps."""
    a, _, _ = _pair(kind, 2, options={"fused": 1}, **over)
    b, _, _ = _pair(kind, 2, options={"fused": 4}, **over)
    sa, sb = a.get_state(), b.get_state()
    for k in STATE_KEYS:
        assert np.array_equal(sa[k], sb[k]), k
    assert b.timing()["substep_launches"] == 1 and a.timing()["substep_launches"] == over.get("substeps", 120)
    a.close(); b.close()


@pytest.mark.parametrize("kind,over,opts", [("h11000", {}, {}), ("h11000", {}, {"resident_overlap": 0}), ("40km", {}, {"patch_nodes": 600}),
                                            ("40km", {"dynamics_type": 3}, {"patch_nodes": 1000}), ("40km", {"dynamics_type": 3}, {"patch_nodes": 1000, "resident_overlap": 0}),
                                            ("small", {"substeps": 7, "dtime_step": 200. * 7 / 120}, {"patch_nodes": 700})])
def test_resident_sub_step_loop_on_one_large_patch_per_cu_does_not_change_a_bit(kind, over, opts):
    """k_substep_resident_big: where a partition is too large for one element per thread (367 k triangles, a rank of four of the 2 km mesh, cut
    automatically into 256 patches of ~720 nodes when fused = 4 is set before set_mesh; or patches of 600 - 1000 nodes asked for explicitly) the
    resident loop runs ONE 512-thread workgroup per CU with four elements and two own nodes per thread -- stress, damage, shape coefficients and the
    moving mesh in registers, element constants and nodal inputs re-read from L2 -- and gives the bits of one launch per sub-step: BBM, EVP, an
    odd number of sub-steps; with (the default here) and without the interior elements' next update computed under the wait for the neighbours."""
    a, _, _ = _pair(kind, 2, options=dict(opts, fused=1), **over)
    b, _, _ = _pair(kind, 2, options=dict({"fused": 4}, **opts), **over)
    sa, sb = a.get_state(), b.get_state()
    for k in STATE_KEYS:
        assert np.array_equal(sa[k], sb[k]), k
    assert b.timing()["substep_launches"] == 1 and a.timing()["substep_launches"] == over.get("substeps", 120)
    a.close(); b.close()


def _shuffled_case(kind="small", seed=7):
    """The mesh `kind` with node and element numbers permuted at random (no locality left): params, local mesh, fields."""
    from nextsim_amd import forcing as F, mesh as M
    gm = cases.global_mesh(kind)
    rng = np.random.default_rng(seed)
    pn = rng.permutation(gm.num_nodes); pe = rng.permutation(gm.num_elements)
    inv = np.empty_like(pn); inv[pn] = np.arange(pn.size)
    g2 = M.GlobalMesh(x=gm.x[pn].copy(), y=gm.y[pn].copy(), tri=np.ascontiguousarray(inv[gm.tri][pe].astype(np.int32)),
                      dirichlet=gm.dirichlet[pn].copy(), neumann=gm.neumann[pn].copy(), lat=gm.lat[pn].copy(), name="shuffled")
    p, C_fix, C_alea = F.scale_params_to_mesh(F.default_params(), g2, alea_factor=0.33)
    g = F.global_fields(g2, p, "arctic", C_fix, C_alea)
    lm = M.localize(g2, 1)[0]
    return p, lm, F.localize_fields(g, lm, g2.num_nodes)


@pytest.mark.parametrize("kind,over,opts", [("small", {}, {}), ("40km", {}, {}), ("40km", {"dynamics_type": 3}, {"fused": 2}), ("h15600", {}, {"fused": 4}),
                                            ("small", {"ice_cat_type": 0, "basal_stress_type": 0}, {"patch_nodes": 300}), ("toy", {}, {}), ("shuffled", {}, {})])
def test_fused_prep_kernel_does_not_change_a_bit(kind, over, opts):
    """k_prep_fused (single rank): prep elements + the nodal side of prep elements + prep nodes (FE.cpp:10235-10416) in ONE launch over the sub-step
    kernel's patches -- the elements' values reach their nodes through LDS, the six ssh-gradient products are formed again at the node from the
    Jacobian and the staged coordinates.  Every array the two separate kernels leave behind (the 48-byte element records, the 80-byte nodal
    records, the displaced coordinates, M_delta_x, M_surface, D_tau_a, VTM, node_mass) has the same bits after explicitSolve, and the state after
    two steps is the same; with and without young ice and basal stress, EVP, patches of any size, a numbering without locality, the resident
    loop's patches."""
    from nextsim_amd import dynamics
    outs = []
    for pf in (0, 1):
        if kind == "shuffled": p, lm, f = _shuffled_case()
        else:
            _, p, _, lms, fields = cases.make_case(kind, **over)
            lm, f = lms[0], fields[0]
        fe = dynamics.FiniteElementDynamics(p)
        for k, v in dict(opts, prep_fused=pf).items(): fe.set_option(k, v)
        fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
        fe.explicitSolve(); fe.synchronize()
        recs = {nm: fe.debug_array(nm).view(np.uint64).copy() for nm in ("erec", "nrec", "xy", "delta_x", "surface", "tau_a", "VTM", "node_mass")}
        fe.update(); fe.step(); fe.synchronize()
        outs.append((recs, fe.get_state()))
        fe.close()
    (ra, sa), (rb, sb) = outs
    for nm in ra:
        assert np.array_equal(ra[nm], rb[nm]), (nm, int((ra[nm] != rb[nm]).sum()))
    for k in STATE_KEYS:
        assert np.array_equal(sa[k], sb[k]), k


def test_resident_sub_step_loop_falls_back_where_it_cannot_run():
    """mEVP (its sub-steps need the velocity of the step's start) and meshes whose patches do not fit one round of resident
    workgroups run one kernel per sub-step although fused = 4 was asked for -- same results, no error."""
    from nextsim_amd import _abi
    a, ref, _ = _pair("small", 1, options={"fused": 4}, dynamics_type=_abi.NXS_DYN_MEVP)
    assert a.timing()["substep_launches"] == 120
    _assert_close(a.get_state(), ref.arr, ("VT", "sigma0", "sigma1", "sigma2"), 1e-9, "mEVP with fused = 4")
    a.close()
    b, _, _ = _pair("h9000", 1, options={"fused": 4})        # ~550 k triangles: several rounds of patches
    assert b.timing()["substep_launches"] == 120
    b.close()


def test_resident_loop_survives_a_remesh_a_change_of_sub_steps_and_a_second_handle():
    """What the resident loop's tables are tied to may change under it: (a) nxs_dyn_set_mesh on a live handle (every regrid) -- its second exchange
    buffer used to stay a pointer into the freed arrays of the OLD mesh (round 3: a device-side use-after-free found by reading, the tables now
    live in a pool of their own that goes with the mesh); (b) nxs_dyn_set_params with another number of sub-steps or another exponent (another
    kernel build, a ghost ring of another length): re-checked at the next step; (c) a SECOND handle of this process asking for the resident loop
    on the same device when the first one's workgroups fill it: refused up front by the per-device registry (one kernel per sub-step, no spin
    against each other until the time-out).  Always the bits of one kernel per sub-step."""
    from nextsim_amd import dynamics
    gm1, p1, g1, lms1, f1 = cases.make_case("h15600")
    gm2, p2, g2, lms2, f2 = cases.make_case("small")

    def run(fused, second_substeps):
        fe = dynamics.FiniteElementDynamics(p2)
        fe.set_option("fused", fused)
        fe.set_mesh(lms2[0]); fe.put_state(f2[0]); fe.set_forcing(f2[0]); fe.step(); fe.step()
        fe.set_params(p1)
        fe.set_mesh(lms1[0]); fe.put_state(f1[0]); fe.set_forcing(f1[0]); fe.step()
        q = p1.copy(); q.substeps = second_substeps; q.dtime_step = 200. * second_substeps / 120; q.exponent_relaxation_sigma = 4.
        fe.set_params(q)
        fe.step(); fe.synchronize()
        return fe
    a, b = run(1, 7), run(4, 7)
    sa, sb = a.get_state(), b.get_state()
    for k in STATE_KEYS:
        assert np.array_equal(sa[k], sb[k]), k
    assert b.timing()["substep_launches"] == 1 and a.timing()["substep_launches"] == 7
    # (c) b holds 511 of the device's 512 resident workgroup slots: c is refused the resident loop and runs one kernel per sub-step
    c = dynamics.FiniteElementDynamics(p1); c.set_option("fused", 4)
    c.set_mesh(lms1[0]); c.put_state(f1[0]); c.set_forcing(f1[0]); c.step(); c.synchronize()
    assert c.timing()["substep_launches"] == 120
    b.close()                                   # its claim goes with it
    c.set_option("fused", 4); c.step(); c.synchronize()
    assert c.timing()["substep_launches"] == 1
    a.close(); c.close()


@pytest.mark.parametrize("kind,opts,kernel,D", [("small", {"fused": 1}, "k_substep_fused", 1), ("small", {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1, "pair_nodes": 64}, "k_substep_pair", 2),
                                                  ("small", {"fused": 2, "substeps_per_launch": 2, "pair_regs": 1, "pair_nodes": 400}, "k_substep_pair", 2),
                                                  ("small", {"fused": 2, "substeps_per_launch": 4}, "k_substep_multi", 4), ("small", {"fused": 4}, "k_substep_resident", 120)])
def test_traffic_model_and_step_times(kind, opts, kernel, D):
    """nxs_dyn_get_traffic_model names the kernel the last step ran on and prices a launch from the patch tables: unique <= scheme (the halo rings are the
    difference), rereads only where a workgroup reads a record twice, the survey's model scaled by the sub-steps a launch advances; the mesh move, prep and
    update figures scale with the mesh.  nxs_dyn_get_step_times returns one device time per step since the reset, and the last ring flush is part of the
    sub-step phase."""
    from nextsim_amd import dynamics
    _, p, _, lms, fields = cases.make_case(kind)
    lm, f = lms[0], fields[0]
    fe = dynamics.FiniteElementDynamics(p)
    for k, v in opts.items():
        fe.set_option(k, v)
    fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
    fe.step(); fe.synchronize(); fe.set_option("timing_reset", 1)
    for _ in range(3):
        fe.step()
    fe.synchronize()
    t = fe.traffic_model()
    assert t["substep_kernel_name"] == kernel and t["substeps_per_launch"] == D and t["halo_in_kernel"] == 0, t
    assert 0 < t["substep_unique_bytes"] <= t["substep_scheme_bytes"], t
    # second reads: the several-sub-steps kernel always has them; k_substep_pair on one rank reads again exactly the 48-byte constants of the E_1 elements beyond the
    # first 512 of their patch (the second element round of sub-step 1; the first round's stay in registers): none in 64-node patches (E_1 ~ 170 elements), some -- whole
    # records, fewer than the mesh has elements -- in 400-node patches (E_1 ~ 900).  (Round 4 relaxed this to a vacuous ">= 0" after a failure in suite 7 -- the
    # automatic patch size of 'small' happens to have no second round; profiles/r04_experiments/r4_suite7_traffic_model_assertion.log.)
    rr = t["substep_reread_bytes"]
    if kernel == "k_substep_multi": assert rr > 0, t
    elif kernel == "k_substep_pair" and opts["pair_nodes"] == 64: assert rr == 0, t
    elif kernel == "k_substep_pair": assert rr > 0 and rr % 48 == 0 and rr / 48 < lm.num_elements, t
    else: assert rr == 0, t
    Ne, Nn = lm.num_elements, lm.num_nodes
    assert abs(t["survey_model_bytes"] - (172. * Ne + 217. * Nn) * D) < 1.
    assert t["substep_unique_bytes"] >= D * 16. * Nn and t["substep_unique_bytes"] >= 112. * Ne    # at least: every velocity out, state in + out + constants
    assert t["update_bytes"] > 100. * Ne and t["prep_unique_bytes"] > 100. * Ne and t["prep_scheme_bytes"] >= t["prep_unique_bytes"]
    if t["move_ring_slots"]:
        assert t["move_ring_bytes"] >= 16. * t["move_ring_slots"] * Nn
    st = fe.step_times()
    tm = fe.timing()
    assert st.shape == (3,) and (st > 0).all() and abs(st.mean() - tm["total_ms"]) < 1e-6 * max(1., tm["total_ms"])
    assert 0. <= tm["ring_flush_ms"] <= tm["substeps_ms"] and (tm["ring_flush_ms"] > 0) == (t["move_ring_slots"] > 0)
    fe.close()


def test_a_resident_grid_of_another_process_is_seen():
    """The registry of resident grids is a POSIX shared-memory table named after the device (csrc/nxs_resident_registry.hpp): while THIS process holds 511 of
    the device's 512 resident workgroup slots, a handle of ANOTHER process that asks for the resident loop is refused up front and steps with one kernel per
    sub-step (same results, no spinning against each other until a time-out); when this process lets go, a new process gets the loop."""
    import subprocess
    import sys
    from nextsim_amd import dynamics
    gm1, p1, g1, lms1, f1 = cases.make_case("h15600")
    b = dynamics.FiniteElementDynamics(p1); b.set_option("fused", 4)
    b.set_mesh(lms1[0]); b.put_state(f1[0]); b.set_forcing(f1[0]); b.step(); b.synchronize()
    assert b.timing()["substep_launches"] == 1
    child = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import cases
from nextsim_amd import dynamics
gm, p, g, lms, f = cases.make_case("small")
c = dynamics.FiniteElementDynamics(p); c.set_option("fused", 4)
c.set_mesh(lms[0]); c.put_state(f[0]); c.set_forcing(f[0]); c.step(); c.synchronize()
print("launches", c.timing()["substep_launches"], "crash", c.checkFieldsFast())
c.close()
''' % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "launches 120 crash 0" in r.stdout, (r.stdout, r.stderr[-2000:])
    b.close()
    r = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "launches 1 crash 0" in r.stdout, (r.stdout, r.stderr[-2000:])


def test_shuffled_numbering_still_matches_the_oracle():
    """A mesh whose node/element numbering has no locality (random permutation): patches are then cut
    along a Morton curve through the coordinates; results must still match the oracle on that mesh."""
    from nextsim_amd import dynamics
    from oracle import pyoracle as O
    p, lm, f = _shuffled_case()
    fe = dynamics.FiniteElementDynamics(p); fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
    ref = O.OracleRank(lm, p, f)
    fe.step(); ref.step(); fe.synchronize()
    _assert_close(fe.get_state(), ref.arr, STATE_KEYS, 1e-10, "shuffled numbering")
    fe.close()


def test_remesh_on_a_live_handle_equals_a_fresh_handle():
    """nxs_dyn_set_mesh is called again after every regrid (FE.cpp:3071-3154 -> distributedMeshProcessing):
    a handle that already ran on another mesh must give the same bits as a fresh one."""
    from nextsim_amd import dynamics
    gm1, p1, g1, lms1, f1 = cases.make_case("toy")
    gm2, p2, g2, lms2, f2 = cases.make_case("small")
    fe = dynamics.FiniteElementDynamics(p1)
    fe.set_mesh(lms1[0]); fe.put_state(f1[0]); fe.set_forcing(f1[0]); fe.step(); fe.step()
    fe.set_params(p2)
    fe.set_mesh(lms2[0]); fe.put_state(f2[0]); fe.set_forcing(f2[0]); fe.step(); fe.synchronize()
    a = fe.get_state()
    fresh = dynamics.FiniteElementDynamics(p2)
    fresh.set_mesh(lms2[0]); fresh.put_state(f2[0]); fresh.set_forcing(f2[0]); fresh.step(); fresh.synchronize()
    b = fresh.get_state()
    for k in STATE_KEYS:
        assert np.array_equal(a[k], b[k]), k
    fe.close(); fresh.close()


def test_checkpoint_resume_through_restart_files_is_bitwise(tmp_path):
    """Checkpoint / resume (writeRestart / readRestart, FE.cpp:9518-9925): state -> restart files in
    the reference's layout -> a NEW handle -> the run continues with the bits of the uninterrupted run."""
    from nextsim_amd import dynamics, io as nio
    gm, p, g, lms, fields = cases.make_case("small")
    lm, f = lms[0], fields[0]
    fe = dynamics.FiniteElementDynamics(p)
    fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
    fe.step(); fe.step(); fe.synchronize()
    st = fe.get_state()
    names = {"M_conc": "conc", "M_thick": "thick", "M_snow_thick": "snow_thick", "M_sigma_0": "sigma0", "M_sigma_1": "sigma1", "M_sigma_2": "sigma2",
             "M_damage": "damage", "M_ridge_ratio": "ridge_ratio", "M_conc_young": "conc_young", "M_h_young": "h_young", "M_hs_young": "hs_young",
             "M_conc_myi": "conc_myi", "M_thick_myi": "thick_myi"}
    nn = lm.num_nodes
    nio.write_restart(tmp_path, "step2", lm.coord_x, lm.coord_y, np.arange(1, nn + 1, dtype=np.int32), lm.indices, [2, 10000, 0, 0],
                      np.flatnonzero(lm.mask_dirichlet[:nn]).astype(np.int32) + 1, 42000.0 + 2 * 200. / 86400., {k: st[v] for k, v in names.items()},
                      st["VT"], st["UM"], st["UT"], np.arange(1, nn + 1, dtype=np.float64))
    fe.step(); fe.step(); fe.synchronize()
    cont = fe.get_state()
    fe.close()
    mesh, field = nio.read_restart(tmp_path, "step2")
    assert np.array_equal(mesh["Elements"], lm.indices.ravel()) and field["Misc_int"][0] == 2
    f2 = dict(f)
    for k, v in names.items():
        f2[v] = field[k]
    f2["VT"], f2["UM"], f2["UT"] = field["M_VT"], field["M_UM"], field["M_UT"]
    fe2 = dynamics.FiniteElementDynamics(p)
    fe2.set_mesh(lm); fe2.put_state(f2); fe2.set_forcing(f2)
    fe2.step(); fe2.step(); fe2.synchronize()
    res = fe2.get_state()
    fe2.close()
    for k in STATE_KEYS:
        assert np.array_equal(res[k], cont[k]), k


def test_time_interpolated_forcing_on_the_device_equals_host_evaluation():
    """nxs_dyn_set_forcing_pair + nxs_dyn_set_forcing_time (two resident snapshots, blended per step on the device with
    ExternalData::get's expression, externaldata.cpp:360-401) against uploading the host-evaluated arrays every step."""
    from nextsim_amd import dynamics
    gm, p, g, lms, fields = cases.make_case("small")
    lm, f0 = lms[0], fields[0]
    rng = np.random.default_rng(3)
    f1 = dict(f0)
    f1["wind"] = f0["wind"] * 1.3 + rng.normal(0, 1.0, f0["wind"].size); f1["ocean"] = f0["ocean"] * 0.7; f1["ssh"] = f0["ssh"] + rng.normal(0, 0.01, f0["ssh"].size)
    factor, bias = (0.625, 1.0, 1.0), (0.0, 0.0, 0.003)         # a spin-up ramp on the wind, a bias on ssh
    t0, t1, times = 100.0, 100.25, (100.03125, 100.0625, 100.2)
    a = dynamics.FiniteElementDynamics(p); b = dynamics.FiniteElementDynamics(p)
    for fe in (a, b):
        fe.set_mesh(lm); fe.put_state(f0)
    b.set_forcing_pair(f0, f1)
    for t in times:
        fdt = abs(t1 - t0)
        c0, c1 = abs(t - t1) / fdt, abs(t - t0) / fdt
        host = dict(f0)
        for k, (fa, bi) in zip(("wind", "ocean", "ssh"), zip(factor, bias)):
            host[k] = fa * (c0 * f0[k] + c1 * f1[k]) + bi
        a.set_forcing(host); a.step()
        b.set_forcing_time(c0, c1, factor, bias); b.step()
    a.synchronize(); b.synchronize()
    sa, sb = a.get_state(), b.get_state()
    for k in STATE_KEYS:
        assert np.array_equal(sa[k], sb[k]), k
    with pytest.raises(Exception, match="set_forcing_pair"):
        c = dynamics.FiniteElementDynamics(p); c.set_mesh(lm); c.set_forcing_time(0.5, 0.5)
    a.close(); b.close()


def test_handles_give_their_device_memory_back():
    """create -> set_mesh (twice: a regrid) -> steps -> forcing pair -> regrid interpolation -> destroy, over and over on the
    10 km mesh (a handle holds ~100 MB): after a few cycles the free device memory must stop moving -- every pool of the handle
    is released and the one-shot entry points free what they take.  (The HIP runtime's own pools settle during the first cycles;
    a leaked array would cost 0.5 MB per cycle, a leaked handle 100 MB.)"""
    import ctypes as C
    from nextsim_amd import dynamics
    from nextsim_amd.interp import InterpFromMeshToMesh2dx
    L = dynamics.load_library()
    L.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes():
        a, b = C.c_size_t(), C.c_size_t()
        assert L.hipMemGetInfo(C.byref(a), C.byref(b)) == 0
        return a.value
    gm, p, g, lms, fields = cases.make_case("10km")
    gm2, p2, g2, lms2, f2 = cases.make_case("small")

    def cycle():
        fe = dynamics.FiniteElementDynamics(p)
        fe.set_mesh(lms[0]); fe.put_state(fields[0]); fe.set_forcing(fields[0]); fe.step()
        fe.set_forcing_pair(fields[0], fields[0]); fe.set_forcing_time(0.5, 0.5); fe.step()
        fe.set_params(p2); fe.set_mesh(lms2[0]); fe.put_state(f2[0]); fe.set_forcing(f2[0]); fe.step(); fe.synchronize()
        fe.close()
        InterpFromMeshToMesh2dx(gm.tri + 1, gm.x, gm.y, np.ones((gm.num_nodes, 2)), gm.x[:100], gm.y[:100], False)
    for _ in range(4):
        cycle()
    mid = free_bytes()
    for _ in range(10):
        cycle()
    end = free_bytes()
    assert mid - end < 3 << 20, (mid, end, (mid - end) / 10)


def test_arctic_free_running_sixty_steps_is_statistically_the_oracle():
    """The long horizon, where point-wise comparison is meaningless (1-ulp differences grow to O(1), see
    tests/test_oracle_sensitivity.py): 60 free-running steps (3.3 hours of model time, 7200 sub-steps) on the 21 k-triangle
    Arctic-like mesh.  Both runs must stay physical and agree in everything that is not chaotic: ice volume and area
    (conserved quantities), mean drift speed, the damage distribution and the stress statistics."""
    fe, ref, lm = _pair("40km", 60)
    got = fe.get_state()
    assert fe.checkFieldsFast() == 0 == ref.check_fields_fast()
    Nn = lm.num_nodes
    surf_g = fe.get_diag()["surface"]; surf_r = ref.work_array("surface", lm.num_elements)
    for k in ("thick", "conc"):
        vg, vr = (got[k] * surf_g).sum(), (ref.arr[k] * surf_r).sum()
        assert abs(vg - vr) <= 1e-5 * vr, k                                  # volume / area
    sp_g = np.hypot(got["VT"][:Nn], got["VT"][Nn:]); sp_r = np.hypot(ref.arr["VT"][:Nn], ref.arr["VT"][Nn:])
    assert abs(sp_g.mean() - sp_r.mean()) <= 0.02 * sp_r.mean()
    assert abs(np.hypot(got["UM"][:Nn], got["UM"][Nn:]).mean() - np.hypot(ref.arr["UM"][:Nn], ref.arr["UM"][Nn:]).mean()) <= 0.02 * np.hypot(ref.arr["UM"][:Nn], ref.arr["UM"][Nn:]).mean()
    q = [0.5, 0.9, 0.99]
    assert np.abs(np.quantile(got["damage"], q) - np.quantile(ref.arr["damage"], q)).max() <= 0.03
    assert abs(got["damage"].mean() - ref.arr["damage"].mean()) <= 0.01
    sn_g, sn_r = 0.5 * (got["sigma0"] + got["sigma1"]), 0.5 * (ref.arr["sigma0"] + ref.arr["sigma1"])
    assert abs(sn_g.mean() - sn_r.mean()) <= 0.05 * abs(sn_r.mean()) + 1.0
    assert abs(np.abs(got["sigma2"]).mean() - np.abs(ref.arr["sigma2"]).mean()) <= 0.05 * np.abs(ref.arr["sigma2"]).mean() + 1.0
    fe.close()


def test_partial_state_transfers():
    """put_state / get_state with NULL members: only the arrays a host-side thermodynamics touched cross PCIe."""
    import ctypes as C
    from nextsim_amd import _abi, dynamics
    gm, p, g, lms, fields = cases.make_case("small")
    lm, f = lms[0], fields[0]
    a = dynamics.FiniteElementDynamics(p); b = dynamics.FiniteElementDynamics(p)
    for fe in (a, b):
        fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f); fe.step(); fe.synchronize()
    # "thermodynamics": thicker ice everywhere.  a: full round trip; b: only thick goes up, nothing else moves
    sa = a.get_state(); sa2 = dict(f); sa2.update(sa); sa2["thick"] = sa["thick"] * 1.01
    a.put_state(sa2)
    thick = np.empty(lm.num_elements)
    s = _abi.State(); s.thick = _abi.dptr(thick)
    assert b.L.nxs_dyn_get_state(b.h, C.byref(s)) == 0
    thick *= 1.01
    s = _abi.State(); s.thick = _abi.dptr(thick)
    assert b.L.nxs_dyn_put_state(b.h, C.byref(s)) == 0
    a.step(); b.step(); a.synchronize(); b.synchronize()
    ga, gb = a.get_state(), b.get_state()
    for k in STATE_KEYS:
        assert np.array_equal(ga[k], gb[k]), k
    # the first put of a mesh must be complete
    c = dynamics.FiniteElementDynamics(p); c.set_mesh(lm)
    assert c.L.nxs_dyn_put_state(c.h, C.byref(s)) != 0
    a.close(); b.close(); c.close()


@pytest.mark.parametrize("over", [{}, {"ice_cat_type": 1}, {"dynamics_type": 3}])
def test_pin_host_option_changes_nothing_but_the_copies(over):
    """nxs_dyn_step_host with page-locked vectors moves what the kernels do not need yet / any more on a second stream beside them (round 5: the arrays only update()
    reads go up while the sub-steps run, M_VT / M_UM / M_UT come down while update() runs, an unused young-ice trio does not come down at all): the same bits as the
    plain put + set_forcing + step + get sequence, with and without the young-ice category (whose arrays are then needed FIRST, by prep), BBM and EVP."""
    import ctypes as C
    from nextsim_amd import _abi, dynamics
    gm, p, g, lms, fields = cases.make_case("small", **over)
    lm = lms[0]
    if over.get("ice_cat_type"):   # a young-ice category that holds something
        rng = np.random.default_rng(3)
        fields[0]["conc_young"] = 0.05 * rng.random(lm.num_elements); fields[0]["h_young"] = 0.1 * rng.random(lm.num_elements); fields[0]["hs_young"] = 0.01 * rng.random(lm.num_elements)
    out = []
    for pin in (0, 1):
        f = {k: v.copy() for k, v in fields[0].items()}
        fe = dynamics.FiniteElementDynamics(p)
        fe.set_option("pin_host", pin)
        fe.set_mesh(lm)
        s = _abi.state_struct(f); fo = _abi.forcing_struct(f)
        for _ in range(3):
            assert fe.L.nxs_dyn_step_host(fe.h, C.byref(s), C.byref(fo)) == 0
        fe.set_mesh(lm)                         # a regrid drops the registrations; the vectors register anew
        assert fe.L.nxs_dyn_step_host(fe.h, C.byref(s), C.byref(fo)) == 0
        fe.close()
        out.append(f)
    for k in STATE_KEYS:
        assert np.array_equal(out[0][k], out[1][k]), k
