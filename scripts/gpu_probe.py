"""Exploratory: GPU vs oracle differences at several horizons + timings. Run on the GPU box."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from nextsim_amd import dynamics, forcing as F, mesh as M, _abi
from oracle import pyoracle as O
import cases

def ulps(a, b):
    return np.abs(a - b) / np.maximum(np.spacing(np.abs(b)), 1e-300)

def compare(tag, got, ref, keys):
    for k in keys:
        d = np.abs(got[k] - ref[k]); sc = max(np.abs(ref[k]).max(), 1e-300)
        print(f"  {tag:10s} {k:12s} maxabs {d.max():.3e} rel-to-max {d.max()/sc:.3e} nonzero-diff {np.count_nonzero(d)}/{d.size}")

for kind, nsteps, over in [("small", 1, dict(substeps=1, dtime_step=200/120.)), ("small", 1, {}), ("toy", 3, {}), ("small", 1, dict(dynamics_type="evp")), ("small", 1, dict(dynamics_type="mevp"))]:
    gm, p, g, lms, fields = cases.make_case(kind, **over)
    lm, f = lms[0], fields[0]
    print(f"== {kind} Ne={lm.num_elements} Nn={lm.num_nodes} steps={nsteps} over={over}")
    fe = dynamics.FiniteElementDynamics(p); fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
    ref = O.OracleRank(lm, p, f)
    for it in range(nsteps):
        fe.step(); fe.synchronize(); ref.step()
        if it == 0:
            for nm, n in [("rlmass", lm.num_nodes), ("node_mass", lm.num_nodes), ("C_bu", lm.num_nodes), ("grad_ssh", 2*lm.num_nodes), ("fcor", lm.num_nodes)]:
                a = fe.debug_array(nm); b = ref.work_array({"rlmass":"rlmass_matrix"}.get(nm, nm), n)
                d = np.abs(a-b); print(f"  prep {nm:10s} maxabs {d.max():.3e} rel {d.max()/max(np.abs(b).max(),1e-300):.3e} ndiff {np.count_nonzero(d)}")
            sh = fe.debug_array("shape").reshape(6, -1).T.ravel(); b = ref.work_array("shape_coeff", 6*lm.num_elements)
            print("  prep shape ndiff", np.count_nonzero(sh-b))
        got = fe.get_state()
        compare(f"step{it}", got, ref.arr, ["VT","UM","UT","sigma0","sigma1","sigma2","damage","conc","thick","snow_thick","ridge_ratio","conc_young","h_young","conc_myi","thick_myi"])
        dg = fe.get_diag()
        for k, rk, n in [("surface","surface",lm.num_elements),("delta_x","delta_x",lm.num_elements),("D_tau_a","D_tau_a",2*lm.num_nodes),("D_tau_w","D_tau_w",2*lm.num_nodes),("D_del_ci_ridge_myi","D_del_ci_ridge_myi",lm.num_elements)]:
            b = ref.work_array(rk, n); d = np.abs(dg[k]-b); print(f"  diag {k:18s} maxabs {d.max():.3e} ndiff {np.count_nonzero(d)}")
        print("  regrid gpu", fe.checkRegridding(), "ref", ref.check_regridding(), "crash", fe.checkFieldsFast(), ref.check_fields_fast())
        print("  timing", fe.timing())
    fe.close()

# timing on bigger meshes
for kind in ["10km", "2km"]:
    t = time.time(); gm, p, g, lms, fields = cases.make_case(kind); lm, f = lms[0], fields[0]
    print(f"== {kind} Ne={lm.num_elements} Nn={lm.num_nodes} gen {time.time()-t:.1f}s")
    fe = dynamics.FiniteElementDynamics(p); t=time.time(); fe.set_mesh(lm); print("  set_mesh", time.time()-t); fe.put_state(f); fe.set_forcing(f)
    for it in range(4):
        t = time.time(); fe.step(); fe.synchronize(); dt = time.time()-t
        tm = fe.timing()
        print(f"  step {it} wall {dt*1e3:.2f} ms  eu/s {lm.num_elements*120/dt:.3e}  timing {tm}  substep GB/s(280B) {280*lm.num_elements*120/(tm['substeps_ms']*1e-3)/1e9:.0f}")
    print("  crash", fe.checkFieldsFast(), "regrid", fe.checkRegridding())
    fe.close()
