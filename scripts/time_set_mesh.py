"""Host cost of a (re)mesh on the dynamics handle: nxs_dyn_set_mesh + the first step (lazily built tables: patches of the sub-step kernels, smoother
patches, graphs), with and without the two-sub-steps-per-launch patches.    python3 scripts/time_set_mesh.py [mesh]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nextsim_amd import dynamics, forcing as F, mesh as M
kind = sys.argv[1] if len(sys.argv) > 1 else "2km"
gm = M.make_mesh(kind)
p, C_fix, C_alea = F.scale_params_to_mesh(F.default_params(), gm, alea_factor=0.33)
g = F.global_fields(gm, p, "arctic", C_fix, C_alea)
lm = M.localize(gm, 1)[0]; f = F.localize_fields(g, lm, gm.num_nodes)
for pair in (1, 0, -1):
    fe = dynamics.FiniteElementDynamics(p)
    if pair >= 0: fe.set_option("pair_regs", pair)
    for rep in range(2):
        t0 = time.perf_counter(); fe.set_mesh(lm); t1 = time.perf_counter()
        fe.put_state(f); fe.set_forcing(f); t2 = time.perf_counter()
        fe.step(); fe.synchronize(); t3 = time.perf_counter()
        fe.step(); fe.synchronize(); t4 = time.perf_counter()
        print(f"pair_regs {pair:2d} round {rep}: set_mesh {1e3 * (t1 - t0):8.1f} ms, put_state + set_forcing {1e3 * (t2 - t1):7.1f} ms, first step {1e3 * (t3 - t2):8.1f} ms, second step {1e3 * (t4 - t3):6.2f} ms, launches {fe.timing()['substep_launches']}", flush=True)
    fe.close()
