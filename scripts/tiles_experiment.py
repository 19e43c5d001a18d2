"""How much a node numbering in COMPACT TILES would give the two-sub-steps-per-launch kernel at 2 km (is the internal renumbering behind the ABI worth
building?).  The synthetic mesh is numbered in tiles of T nodes (nextsim_amd/mesh.py::tile_numbering, NXS_MESH_CURVE=tiles) and cut into patches of T own nodes, so that
a patch IS a tile; the bench's mesh (numbered along a Hilbert curve) beside it.  Prints per configuration: ms of sub-steps, the scheme / unique bytes per launch and the
patch count -- the counters (rocprofv3 --pmc) are taken by scripts/profile_round.sh on the configuration that wins.
    python3 scripts/tiles_experiment.py [--steps 20] hilbert:0 tiles:428 tiles:416 tiles:400 ..."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--mesh", default="2km")
ap.add_argument("cfg", nargs="+", help="curve:tile[:pair_nodes]  (hilbert:0 = the bench's mesh, automatic patch size)")
a = ap.parse_args()
import torch  # noqa: F401
from nextsim_amd import dynamics, forcing as F, mesh as M
for cfg in a.cfg:
    parts = cfg.split(":")
    curve, tile = parts[0], int(parts[1])
    pair_nodes = int(parts[2]) if len(parts) > 2 else tile
    os.environ["NXS_MESH_CURVE"] = curve
    if tile: os.environ["NXS_MESH_TILE"] = str(tile)
    t0 = time.perf_counter()
    gm = M.make_mesh(a.mesh)
    p, C_fix, C_alea = F.scale_params_to_mesh(F.default_params(), gm, alea_factor=0.33)
    g = F.global_fields(gm, p, "arctic", C_fix, C_alea)
    lm = M.localize(gm, 1)[0]; f = F.localize_fields(g, lm, gm.num_nodes)
    fe = dynamics.FiniteElementDynamics(p)
    if pair_nodes: fe.set_option("pair_nodes", pair_nodes)
    fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
    for _ in range(3): fe.step()
    fe.synchronize(); fe.put_state(f); fe.set_option("timing_reset", 1)
    t = time.perf_counter()
    for _ in range(a.steps): fe.step()
    fe.synchronize(); dt = time.perf_counter() - t
    tm, tr = fe.timing(), fe.traffic_model()
    print(f"{cfg}: {gm.num_elements} triangles; {dt / a.steps * 1e3:.3f} ms/step, sub-steps {tm['substeps_ms']:.3f} ms in {tm['substep_launches']} launches of {tr['substep_kernel_name']}; "
          f"per launch scheme {tr['substep_scheme_bytes'] / 1e6:.1f} MB, unique {tr['substep_unique_bytes'] / 1e6:.1f} MB, reread {tr['substep_reread_bytes'] / 1e6:.1f} MB; prep {tm['prep_ms']:.3f} ms "
          f"(set-up {time.perf_counter() - t0:.0f} s)", flush=True)
    fe.close()
