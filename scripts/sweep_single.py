"""The bench's 2 km mesh on one rank: step time against ONE swept option (set before set_mesh), other options fixed.   python3 scripts/sweep_single.py pair_threads:512,256 [key=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from nextsim_amd import dynamics, forcing as F, mesh as M
fixed = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[1:] if "=" in kv)}
sweeps = [kv.split(":") for kv in sys.argv[1:] if ":" in kv]
key, values = (sweeps[0][0], [int(v) for v in sweeps[0][1].split(",")]) if sweeps else (None, [None])
gm = M.make_mesh(os.environ.get("NXS_MESH", "2km"))
p, C_fix, C_alea = F.scale_params_to_mesh(F.default_params(), gm, alea_factor=0.33)
g = F.global_fields(gm, p, "arctic", C_fix, C_alea)
lm = M.localize(gm, 1)[0]; f = F.localize_fields(g, lm, gm.num_nodes)
for val in values:
    opts = dict(fixed)
    if key: opts[key] = val
    fe = dynamics.FiniteElementDynamics(p)
    try:
        for k, v in opts.items(): fe.set_option(k, v)
        fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
        for _ in range(20): fe.step()
        fe.synchronize(); fe.put_state(f); fe.set_option("timing_reset", 1)
        for _ in range(30): fe.step()
        fe.synchronize()
        tm, tr = fe.timing(), fe.traffic_model()
        print(f"{opts}: {tr['substep_kernel_name']} sub-steps {tm['substeps_ms']:.3f} ms ({tm['substep_launches']} launches), prep {tm['prep_ms']:.3f}, smoother {tm['smoother_ms']:.3f}, update {tm['update_ms']:.3f}, total {tm['total_ms']:.3f} ms", flush=True)
    except dynamics.NxsError as e:
        print(f"{opts}: {str(e)[:200]}", flush=True)
    finally:
        fe.close()
