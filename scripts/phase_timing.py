"""Where a workgroup of the fused sub-step kernel spends its time: builds a variant of the library with
-DNXS_PHASE_TIMING (timestamps at the phase boundaries), runs one step and prints the statistics.
    python3 scripts/phase_timing.py --build        (in the build container: writes nextsim_amd/csrc/libnxsdyn_phase.so)
    NXS_DYN_LIBRARY=$PWD/nextsim_amd/csrc/libnxsdyn_phase.so python3 scripts/phase_timing.py --mesh 2km   (on the GPU box)"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--build", action="store_true"); ap.add_argument("--mesh", default="2km"); ap.add_argument("--h", type=float, default=0.); ap.add_argument("--multi", action="store_true"); ap.add_argument("--resident", action="store_true"); ap.add_argument("--prep", action="store_true"); ap.add_argument("--pair", action="store_true"); ap.add_argument("--opt", action="append", default=[], help="key=value for set_option (before set_mesh)")
a = ap.parse_args()
csrc = os.path.join(ROOT, "nextsim_amd", "csrc")
if a.build:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
                           "-I" + os.path.join(ROOT, "include"), "-DNXS_PHASE_TIMING", "-shared", "-o", os.path.join(csrc, "libnxsdyn_phase.so")] +
                          [os.path.join(csrc, f) for f in ("nxs_dyn.hip", "nxs_interp.hip", "nxs_krylov.hip", "nxs_mesh.cpp", "nxs_io.cpp")] + ["-ldl"])
    sys.exit(0)
import numpy as np
import torch  # noqa: F401
from nextsim_amd import dynamics, forcing as F, mesh as M
gm = M.make_disc_mesh(a.h, seed=M.SEED, name="custom") if a.h > 0 else M.make_mesh(a.mesh)
p, C_fix, C_alea = F.scale_params_to_mesh(F.default_params(), gm, alea_factor=0.33)
g = F.global_fields(gm, p, "arctic", C_fix, C_alea)
lm = M.localize(gm, 1)[0]; f = F.localize_fields(g, lm, gm.num_nodes)
fe = dynamics.FiniteElementDynamics(p); fe.set_option("graph", 0)
if a.resident: fe.set_option("fused", 4)
for kv in a.opt: fe.set_option(kv.split("=")[0], float(kv.split("=")[1]))
fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
fe.step(); fe.step(); fe.synchronize()
if a.pair:  # k_substep_pair (the last launch of the step): start, end of sub-step 0, end -- and how many workgroups run at a time
    t = fe.debug_array("phase_times").reshape(8192, 8)[:, [0, 2, 4]]
    t = t[t[:, 0] > 0]
    k0, k1 = t[:, 0].min(), t[:, 2].max()
    print(f"{gm.num_elements} triangles, {t.shape[0]} workgroups; kernel span {(k1 - k0) * 10e-3:.2f} us; a workgroup takes {((t[:, 2] - t[:, 0]).mean()) * 10e-3:.2f} us "
          f"(sub-step 0: {((t[:, 1] - t[:, 0]).mean()) * 10e-3:.2f}); started in the last quarter of the span: {(t[:, 0] > k0 + 0.75 * (k1 - k0)).sum()}")
    for f0 in np.linspace(0.05, 0.95, 10):
        x = k0 + f0 * (k1 - k0)
        print(f"  at {100 * f0:3.0f} % of the span: {((t[:, 0] <= x) & (t[:, 2] > x)).sum():4d} workgroups running")
    # the phases of a workgroup (thread 0's clock): stamps 0 start, 1 staged (barrier), 5 elements of sub-step 0 done, 6 their forces visible (barrier; the loads of the
    # first round of nodes were issued in front of it), 2 nodes of sub-step 0 done + barrier, 7 elements of sub-step 1 done, 3 barrier, 4 end
    full = fe.debug_array("phase_times").reshape(8192, 8)
    full = full[full[:, 0] > 0][:, [0, 1, 5, 6, 2, 7, 3, 4]]
    d = np.diff(full, axis=1) * 10e-3
    for nm, col in zip(("index hop + staging (to barrier 1)", "elements E_2, three rounds", "node loads + barrier 2", "solve N_1, two rounds (+ barrier 3)", "elements E_1, two rounds", "node loads + barrier 4", "solve the own nodes"), d.T):
        print(f"  {nm:42s} mean {col.mean():6.2f} us   p10 {np.percentile(col, 10):6.2f}   p90 {np.percentile(col, 90):6.2f}")
    fe.close(); sys.exit(0)
if a.prep:  # k_prep_fused: start, nodes staged + barrier, elements done, barrier, nodes done
    t = fe.debug_array("phase_times_prep").reshape(8192, 8)[:, :5]
    t = t[t[:, 0] > 0]
    k0 = t[:, 0].min()
    d = np.diff(t, axis=1) * 10e-3
    print(f"{gm.num_elements} triangles, {t.shape[0]} workgroups; kernel span {(t[:, 4].max() - k0) * 10e-3:.2f} us")
    for nm, col in zip(("staging (to barrier 1)", "element phase", "wait at barrier 2", "node phase"), d.T):
        print(f"  {nm:38s} mean {col.mean():6.2f} us   p10 {np.percentile(col, 10):6.2f}   p90 {np.percentile(col, 90):6.2f}")
    print(f"  workgroup total                        mean {(t[:, 4] - t[:, 0]).mean() * 10e-3:6.2f} us; start of the last workgroup {(t[:, 0].max() - k0) * 10e-3:.2f} us after the first")
    fe.close(); sys.exit(0)
t = fe.debug_array("phase_times")
if a.resident:  # k_substep_resident, sub-step 60 of the launch: start of the element phase, barrier 1, end of the node phase, stores drained + barrier, neighbours' counters seen, halo loaded
    t = t.reshape(8192, 8)[:, :6]
    t = t[t[:, 0] > 0]
    d = np.diff(t, axis=1) * 10e-3
    print(f"{gm.num_elements} triangles, {t.shape[0]} workgroups, sub-step 60 of the resident launch; spread of the workgroups' start of that sub-step {(t[:, 0].max() - t[:, 0].min()) * 10e-3:.2f} us")
    for nm, col in zip(("element phase (to barrier 1)", "node phase + publishing stores issued", "stores drained + barrier", "wait for the neighbours' counters + barrier", "halo loads + barrier"), d.T):
        print(f"  {nm:45s} mean {col.mean():6.2f} us   p10 {np.percentile(col, 10):6.2f}   p90 {np.percentile(col, 90):6.2f}")
    print(f"  one sub-step                                  mean {(t[:, 5] - t[:, 0]).mean() * 10e-3:6.2f} us")
    fe.close(); sys.exit(0)
if a.multi:  # k_substep_multi (small single-rank meshes), D = 4: start, barrier 1, forces of sub-step 0, end of sub-steps 0..3
    t = t.reshape(8192, 8)[:, :7]
    t = t[t[:, 0] > 0]
    d = np.diff(t, axis=1) * 10e-3
    print(f"{gm.num_elements} triangles, {t.shape[0]} workgroups; kernel span {(t[:, 6].max() - t[:, 0].min()) * 10e-3:.2f} us")
    for nm, col in zip(("staging (to barrier 1)", "elements of sub-step 0 (to barrier 2)", "nodes of sub-step 0", "sub-step 1", "sub-step 2", "sub-step 3"), d.T):
        print(f"  {nm:38s} mean {col.mean():6.2f} us   p10 {np.percentile(col, 10):6.2f}   p90 {np.percentile(col, 90):6.2f}")
    fe.close(); sys.exit(0)
t = t.reshape(8192, 8)[:, :5]
t = t[t[:, 0] > 0]
k0 = t[:, 0].min()
d = np.diff(t, axis=1) * 10e-3  # 100 MHz ticks -> us
print(f"{gm.num_elements} triangles, {t.shape[0]} workgroups; kernel span {(t[:, 4].max() - k0) * 10e-3:.2f} us")
for nm, col in zip(("index hop + staging (to barrier 1)", "element rounds", "wait at barrier 2 (+ node loads)", "fan gather + solve + stores"), d.T):
    print(f"  {nm:38s} mean {col.mean():6.2f} us   p10 {np.percentile(col, 10):6.2f}   p90 {np.percentile(col, 90):6.2f}")
print(f"  workgroup total                        mean {(t[:, 4] - t[:, 0]).mean() * 10e-3:6.2f} us; start of the last workgroup {(t[:, 0].max() - k0) * 10e-3:.2f} us after the first")
fe.close()
