"""Patch-size sweep of the fused sub-step kernel: python3 scripts/sweep_patch.py --mesh 2km 400 448 476 512"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--mesh", default="2km"); ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--h", type=float, default=0.0, help="disc mesh of this resolution (m) instead of --mesh")
ap.add_argument("P", nargs="+", type=int)
a = ap.parse_args()
import torch  # noqa: F401  (its HIP runtime must be the one in the process)
from nextsim_amd import dynamics, forcing as F, mesh as M
gm = M.make_disc_mesh(a.h, seed=M.SEED, name="custom") if a.h > 0 else M.make_mesh(a.mesh)
print(f"mesh: {gm.num_elements} triangles, {gm.num_nodes} nodes", flush=True)
p, C_fix, C_alea = F.scale_params_to_mesh(F.default_params(), gm, alea_factor=0.33)
g = F.global_fields(gm, p, "arctic", C_fix, C_alea)
lm = M.localize(gm, 1)[0]; f = F.localize_fields(g, lm, gm.num_nodes)
for P in a.P:
    fe = dynamics.FiniteElementDynamics(p)
    fe.set_option("patch_nodes", P)
    fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
    fe.step(); fe.step(); fe.synchronize(); fe.set_option("timing_reset", 1)
    t = time.perf_counter()
    for _ in range(a.steps): fe.step()
    fe.synchronize(); dt = time.perf_counter() - t
    tm = fe.timing()
    print(f"P={P}: {dt/a.steps*1e3:.3f} ms/step substeps {tm.get('substeps_ms', 0):.3f} ms", flush=True)
    fe.close()
