"""Can every rank's partition run the resident sub-step loop?  (One GPU: each partition is cut and its tables are built, nothing is stepped.)
    python3 scripts/check_partitions.py [mesh] [ranks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nextsim_amd import dynamics
mesh = sys.argv[1] if len(sys.argv) > 1 else "2km"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for r in range(world):
    gm, p, lm, f = bench.build_case(mesh, world, r)
    fe = dynamics.FiniteElementDynamics(p)
    fe.set_mesh(lm)
    for ovl in (0, 1):
        fe.set_option("resident_overlap", ovl)
        try:
            fe.set_option("resident_dryrun", 1); ok = "possible"
        except dynamics.NxsError as e:
            ok = "NOT possible: " + str(e)
        print(f"{mesh}/{world} rank {r}: {lm.num_elements} triangles, {lm.num_nodes} nodes; resident loop (overlap {ovl}) {ok}", flush=True)
    fe.close()
