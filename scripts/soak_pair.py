"""Soak of the default single-rank path above 65 k nodes (k_substep_pair + k_prep_fused): N steps, final state bitwise against one launch per sub-step with the
two separate prep kernels.    python3 scripts/soak_pair.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nextsim_amd import dynamics, forcing as F, mesh as M
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for kind in ("h15600", "h9500", "2km"):
    gm = M.make_mesh(kind)
    p, C_fix, C_alea = F.scale_params_to_mesh(F.default_params(), gm, alea_factor=0.33)
    g = F.global_fields(gm, p, "arctic_ow" if kind == "2km" else "arctic", C_fix, C_alea)
    lm = M.localize(gm, 1)[0]; f = F.localize_fields(g, lm, gm.num_nodes)
    out = {}
    for name, opts in (("default", {}), ("plain", {"fused": 1, "prep_fused": 0})):
        fe = dynamics.FiniteElementDynamics(p)
        for k, v in opts.items(): fe.set_option(k, v)
        fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
        fe.step(); fe.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps - 1): fe.step()
        fe.synchronize(); dt = time.perf_counter() - t0
        out[name] = (fe.get_state(), fe.timing()["substep_launches"], dt / max(steps - 1, 1) * 1e3, fe.checkFieldsFast())
        fe.close()
    same = all(np.array_equal(out["default"][0][k], out["plain"][0][k]) for k in out["default"][0])
    print(f"{kind}: {gm.num_elements} triangles, {steps} steps: default {out['default'][2]:.3f} ms/step ({out['default'][1]} launches), one launch per sub-step "
          f"{out['plain'][2]:.3f} ms/step ({out['plain'][1]}); bitwise {'IDENTICAL' if same else 'DIFFERENT'}; crash flags {out['default'][3]} {out['plain'][3]}", flush=True)
    assert same
