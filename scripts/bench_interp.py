"""BASELINE config 5: mesh-to-mesh interpolation at regrid (6 nodal variables, FE.cpp:3131) on the 2 km
mesh: HIP gather kernel vs the real contrib/bamg InterpFromMeshToMesh2dx on the host cores."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nextsim_amd import mesh as M
from nextsim_amd.interp import InterpFromMeshToMesh2dx
kind = sys.argv[1] if len(sys.argv) > 1 else "2km"
old = M.make_mesh(kind)
h_new = {"2km": 5.8e3, "10km": 29e3, "40km": 48e3}[kind]
new = M.make_disc_mesh(h_new, seed=5, name="new")
rng = np.random.default_rng(2)
um = 200.0 * rng.standard_normal((2, old.num_nodes))
xo, yo = old.x + um[0], old.y + um[1]
data = rng.standard_normal((old.num_nodes, 6))
idx = (old.tri + 1).astype(np.int32).ravel()
for _ in range(2):
    t = time.perf_counter(); got, info = InterpFromMeshToMesh2dx(idx, xo, yo, data, new.x, new.y, True, 0.0, return_info=True); wall = time.perf_counter() - t
print(f"{kind}: {old.num_elements} data triangles -> {new.num_nodes} target nodes x 6 vars: kernel {info['kernel_ms']:.3f} ms, call incl. host grid build + PCIe {wall*1e3:.1f} ms, "
      f"{new.num_nodes/ (info['kernel_ms']*1e-3):.3e} points/s")
try:
    from oracle import pyoracle as O
    if O.bamg_shim() is not None:
        t = time.perf_counter(); ref = O.bamg_interp_mesh_to_mesh(idx, xo, yo, data, new.x, new.y, True, 0.0); cpu = time.perf_counter() - t
        same = np.all(got == ref, axis=1)
        print(f"real bamg InterpFromMeshToMesh2dx on the host (1 thread, incl. its mesh build): {cpu*1e3:.0f} ms; identical rows {same.mean()*100:.2f} %, "
              f"all others are default-vs-hull-triangle ties: {bool(np.all(got[~same] == 0.0))}")
except Exception as e:
    print("no reference on this box:", e)
