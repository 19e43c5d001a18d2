"""Soak of the resident sub-step loop (option fused = 4): N steps on several mesh sizes, final state bitwise against one launch per sub-step.
    python3 scripts/soak_resident.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nextsim_amd import dynamics, forcing as F, mesh as M
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
# (11 000 m, 12 500 m, 14 000 m: 367 k - 227 k triangles, the large-patch kernel with the interior elements one exchange ahead)
for h_edge in (11000., 12500., 14000., 15600., 17000., 19000., 22000., 27600., 46000.):
    gm = M.make_disc_mesh(h_edge, seed=M.SEED, name="custom")
    p, C_fix, C_alea = F.scale_params_to_mesh(F.default_params(), gm, alea_factor=0.33)
    g = F.global_fields(gm, p, "arctic", C_fix, C_alea)
    lm = M.localize(gm, 1)[0]; f = F.localize_fields(g, lm, gm.num_nodes)
    out = {}
    for fused in (4, 1):
        fe = dynamics.FiniteElementDynamics(p); fe.set_option("fused", fused)
        fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
        t0 = time.perf_counter()
        for _ in range(steps): fe.step()
        fe.synchronize(); dt = time.perf_counter() - t0
        out[fused] = (fe.get_state(), fe.timing()["substep_launches"], dt / steps * 1e3, fe.checkFieldsFast())
        fe.close()
    same = all(np.array_equal(out[4][0][k], out[1][0][k]) for k in out[4][0])
    print(f"{gm.num_elements} triangles, {steps} steps: resident {out[4][2]:.3f} ms/step ({out[4][1]} launch), per sub-step {out[1][2]:.3f} ms/step; "
          f"bitwise {'IDENTICAL' if same else 'DIFFERENT'}; crash flags {out[4][3]} {out[1][3]}", flush=True)
    assert same
