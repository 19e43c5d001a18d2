"""N4 extension: achieved HBM bandwidth of the SpMV and of the coloured assembly on the pan-Arctic meshes.
  python scripts/bench_krylov.py [--mesh 2km] [--reps 200]
Algorithmic bytes: SpMV = 12 B per non-zero + 16 B per row (nxs_krylov_info); assembly = per element 3 node ids (12) +
9 target positions (36) + source (8) + 9 matrix entries read+written (144) + 3 rhs entries read+written (48) = 248 B."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
from nextsim_amd import krylov, mesh as M

ap = argparse.ArgumentParser()
ap.add_argument("--mesh", default="2km")
ap.add_argument("--reps", type=int, default=200)
a = ap.parse_args()
gm = M.make_mesh(a.mesh)
tri = (gm.tri + 1).astype(np.int32)
L = 2.5e6
x, y = gm.x / L, gm.y / L
bnd = (gm.dirichlet | gm.neumann).astype(np.uint8)
out = {"mesh": a.mesh, "nodes": int(gm.num_nodes), "elements": int(gm.num_elements)}

# coloured assembly + CG on the P1 Laplacian
u, info = krylov.poisson_solve(tri, x, y, bnd, np.ones(gm.num_elements), rtol=1e-8, max_iter=400)
asm_bytes = 248 * gm.num_elements
out["assembly"] = {"ms": info["ms_assembly"], "GBps": asm_bytes / info["ms_assembly"] / 1e6, "bytes": asm_bytes}
out["cg"] = {"iterations": info["iterations"], "ms_per_iteration": info["ms_solve"] / max(info["iterations"], 1), "rel_residual": info["rel_residual"]}

# SpMV: one dof per node, and the 2-dof (u, v) block pattern of a momentum matrix
rp, ci = krylov.csr_pattern(tri, gm.num_nodes)
rng = np.random.default_rng(0)
va = rng.normal(size=ci.size)
va[ci == np.repeat(np.arange(gm.num_nodes), np.diff(rp))] = 10.0
A1 = sp.csr_matrix((va, ci, rp), shape=(gm.num_nodes,) * 2)
A2 = sp.kron(A1, sp.csr_matrix(np.array([[1.0, 0.3], [-0.3, 1.0]])), format="csr")
A2.sort_indices()
for name, A in (("scalar", A1), ("block2", A2)):
    s = krylov.Solver()
    s.set_matrix(A.indptr, A.indices, A.data)
    xin = rng.normal(size=A.shape[0])
    got, ms = s.spmv(xin, reps=a.reps)
    err = float(np.abs(got - A @ xin).max() / np.abs(got).max())
    inf = s.info()
    out["spmv_" + name] = {"rows": A.shape[0], "nnz": inf["nnz"], "padding": inf["stored_entries"] / inf["nnz"] - 1.0, "us": ms * 1e3,
                           "GBps": inf["spmv_bytes"] / ms / 1e6, "frac_of_8TBps": inf["spmv_bytes"] / ms / 1e6 / 8000.0, "bytes": inf["spmv_bytes"], "err": err}
    xs = rng.normal(size=A.shape[0])
    xsol, i2 = s.solve(A @ xs, method=krylov.BICGSTAB, rtol=1e-10, max_iter=2000)
    out["bicgstab_" + name] = dict(i2, ms_per_iteration=i2["ms_solve"] / max(i2["iterations"], 1), err=float(np.abs(xsol - xs).max()))
    s.close()
print(json.dumps(out))
