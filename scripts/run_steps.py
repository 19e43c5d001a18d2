"""Torch-free runner of the hot path (for rocprofv3): python3 scripts/run_steps.py --mesh 2km --steps 5"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--mesh", default="2km"); ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--h", type=float, default=0., help="disc mesh of this edge length (m) instead of a named mesh: 15600 = one rank of eight of the 2 km mesh")
ap.add_argument("--opt", action="append", default=[], help="key=value for set_option (before set_mesh)")
ap.add_argument("--torch-first", action="store_true"); ap.add_argument("--graph", type=int, default=1)
ap.add_argument("--fused", type=int, default=3); ap.add_argument("--patch-nodes", type=int, default=0); ap.add_argument("--nt", type=int, default=-1); ap.add_argument("--ring", type=int, default=0); ap.add_argument("--pair-nodes", type=int, default=0); ap.add_argument("--depth", type=int, default=0); ap.add_argument("--shape-mem", type=int, default=-1); ap.add_argument("--compare-fused", type=int, default=-1, help="also run with this value of option fused and compare the states bit for bit")
ap.add_argument("--nparts", type=int, default=1, help="partitions of the mesh (with --rank and --loopback: one rank's partition stepped alone, for counter collection)")
ap.add_argument("--rank", type=int, default=0); ap.add_argument("--loopback", action="store_true", help="several partitions: this rank's mailboxes connected to themselves (dynamics.ipc_loopback)")
a = ap.parse_args()
if a.torch_first:
    import torch
    print("torch", torch.__version__, "cuda", torch.cuda.is_available(), flush=True)
from nextsim_amd import dynamics, forcing as F, mesh as M
gm = M.make_disc_mesh(a.h, seed=M.SEED, name="custom") if a.h > 0 else M.make_mesh(a.mesh)
p, C_fix, C_alea = F.scale_params_to_mesh(F.default_params(), gm, alea_factor=0.33)
g = F.global_fields(gm, p, "arctic", C_fix, C_alea)
lm = M.localize(gm, a.nparts)[a.rank]; f = F.localize_fields(g, lm, gm.num_nodes)
fe = dynamics.FiniteElementDynamics(p); fe.set_option("graph", a.graph); fe.set_option("fused", a.fused); fe.set_option("nt_mask", a.nt); fe.set_option("um_ring", a.ring)
if a.patch_nodes: fe.set_option("patch_nodes", a.patch_nodes)
if a.pair_nodes: fe.set_option("pair_nodes", a.pair_nodes)
if a.depth: fe.set_option("substeps_per_launch", a.depth)
for kv in a.opt: fe.set_option(kv.split("=")[0], int(kv.split("=")[1]))
fe.set_mesh(lm)
if a.shape_mem >= 0: fe.set_option("shape_mem", a.shape_mem)
if a.nparts > 1:
    if not a.loopback: sys.exit("--nparts > 1 needs --loopback (one rank alone on the device)")
    if not fe.ipc_loopback(): sys.exit("this partition's halo lists cannot be looped back")
    for kv in a.opt: fe.set_option(kv.split("=")[0], int(kv.split("=")[1]))   # (fused / halo_fused again: set_mesh and the transport are in place now)
    fe.set_option("prepare", 1)
fe.put_state(f); fe.set_forcing(f)
fe.step(); fe.synchronize(); fe.set_option("timing_reset", 1)
t = time.perf_counter()
for _ in range(a.steps): fe.step()
fe.synchronize(); dt = time.perf_counter() - t
print(f"shape_mem={a.shape_mem} ring={a.ring} nt={a.nt} fused={a.fused} patch_nodes={a.patch_nodes} {a.mesh}: {lm.num_elements} triangles, {a.steps} steps, {dt/a.steps*1e3:.3f} ms/step, {lm.num_elements*120*a.steps/dt:.4e} element-updates/s, timing {fe.timing()}, crash {fe.checkFieldsFast()}", flush=True)
if a.compare_fused >= 0:
    import numpy as np
    fe2 = dynamics.FiniteElementDynamics(p); fe2.set_option("fused", a.compare_fused); fe2.set_mesh(lm); fe2.put_state(f); fe2.set_forcing(f)
    fe3 = dynamics.FiniteElementDynamics(p); fe3.set_option("fused", a.fused); [fe3.set_option(kv.split("=")[0], int(kv.split("=")[1])) for kv in a.opt]
    if a.patch_nodes: fe3.set_option("patch_nodes", a.patch_nodes)
    if a.pair_nodes: fe3.set_option("pair_nodes", a.pair_nodes)
    if a.depth: fe3.set_option("substeps_per_launch", a.depth)
    fe3.set_mesh(lm); fe3.put_state(f); fe3.set_forcing(f)
    for _ in range(2): fe3.step()
    fe3.synchronize()          # (one after the other: the resident kernel needs the device to itself)
    for _ in range(2): fe2.step()
    fe2.synchronize()
    s2, s3 = fe2.get_state(), fe3.get_state()
    bad = [k for k in s2 if not np.array_equal(s2[k], s3[k])]
    print("bitwise fused=%d vs fused=%d after 2 steps:" % (a.fused, a.compare_fused), "IDENTICAL" if not bad else "DIFFERENT in %s" % bad, "launches", fe3.timing()["substep_launches"], fe2.timing()["substep_launches"], flush=True)
    fe2.close(); fe3.close()
fe.close()
