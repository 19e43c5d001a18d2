"""Rank 0 of N of the 2 km mesh, looped back, resident loop: sub-step time against option band_patch_nodes (and any other key=value given).   python3 scripts/sweep_band.py 8 16 24 32 48 96 [key=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from nextsim_amd import dynamics, forcing as F, mesh as M
nparts = int(sys.argv[1])
bands = [int(v) for v in sys.argv[2:] if "=" not in v]
opts = dict(kv.split("=") for kv in sys.argv[2:] if "=" in kv)
gm = M.make_mesh("2km")
p, C_fix, C_alea = F.scale_params_to_mesh(F.default_params(), gm, alea_factor=0.33)
g = F.global_fields(gm, p, "arctic", C_fix, C_alea)
lm = M.localize(gm, nparts)[0]; f = F.localize_fields(g, lm, gm.num_nodes)
for band in bands:
    fe = dynamics.FiniteElementDynamics(p)
    fe.set_option("band_patch_nodes", band)
    for k, v in opts.items(): fe.set_option(k, int(v))
    fe.set_mesh(lm)
    assert fe.ipc_loopback()
    for k, v in {"fused": 4, "halo_fused": 1, "resident_wide": 1, **{k2: int(v2) for k2, v2 in opts.items()}}.items(): fe.set_option(k, v)   # (the caller's options win)
    fe.set_option("prepare", 1); fe.put_state(f); fe.set_forcing(f)
    for _ in range(3): fe.step()
    fe.synchronize(); fe.put_state(f); fe.set_option("timing_reset", 1)
    for _ in range(10): fe.step()
    fe.synchronize()
    tm, tr = fe.timing(), fe.traffic_model()
    print(f"band_patch_nodes {band} {opts}: {tr['substep_kernel_name']} sub-steps {tm['substeps_ms']:.3f} ms ({tm['substep_launches']} launches), smoother {tm['smoother_ms']:.3f}, prep {tm['prep_ms']:.3f}, update {tm['update_ms']:.3f}, total {tm['total_ms']:.3f} ms", flush=True)
    fe.close()
