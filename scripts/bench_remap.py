"""Conservative remapping of the element variables at regrid (FE.cpp:3108, 30 variables) -- HIP kernel vs the
real contrib/bamg ConservativeRemappingMeshToMesh on one host core.  Two regimes: 'adapted' (what a regrid
does: most triangles survive, found through PreviousNumbering) and 'remeshed' (every triangle walks)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
from nextsim_amd.interp import ConservativeRemappingMeshToMesh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nv = 30
x, y, tri, ng = cases.rect_mesh(n, 7)
rng = np.random.default_rng(5)
data = rng.random((tri.shape[0], nv))
t = time.perf_counter(); xn, yn, trin, prev = cases.adapted_mesh(x, y, tri, ng, 9, frac_touched=0.1); print(f"built adapted mesh in {time.perf_counter()-t:.1f}s", flush=True)
x2, y2, tri2, _ = cases.rect_mesh(int(n * 0.83), 8)
try:
    from oracle import pyoracle as O
    have_ref = O.bamg_shim() is not None
except Exception:
    have_ref = False
for label, (a, b, c, d) in {"adapted": (xn, yn, trin, prev), "remeshed": (x2, y2, tri2, np.zeros(x2.size))}.items():
    for _ in range(2):
        t = time.perf_counter(); out, info = ConservativeRemappingMeshToMesh(data, tri + 1, x, y, c + 1, a, b, d, ng, return_info=True); wall = time.perf_counter() - t
    v = info["visits"]
    print(f"{label}: {tri.shape[0]} old -> {c.shape[0]} new triangles x {nv} vars: kernel {info['kernel_ms']:.3f} ms, call incl. host tables + PCIe {wall*1e3:.0f} ms; "
          f"failed {info['num_failed']}, unchanged {100*(v==1).mean():.1f} %, mean/max overlap {v.mean():.2f}/{v.max()}", flush=True)
    if have_ref:
        t = time.perf_counter(); ref = O.bamg_conservative_remap(tri + 1, x, y, c + 1, a, b, d, ng, data); cpu = time.perf_counter() - t
        same = np.all((out == ref) | (np.isnan(out) & np.isnan(ref)), axis=1)
        print(f"   real bamg ConservativeRemappingMeshToMesh, 1 host thread (incl. its two BamgConvertMeshx): {cpu*1e3:.0f} ms; identical rows {100*same.mean():.4f} %, "
              f"max |diff| {np.nanmax(np.abs(out-ref)):.2e}", flush=True)
