"""PCIe-inclusive rates of the drop-in boundary at 2 km (never bench.py's `value`): (a) nxs_dyn_step_host = put_state + set_forcing
+ step + get_state every step, the literal three lines of step(); (b) state resident, forcing blended on the device, only the arrays a
host-side thermodynamics touches (conc, thick, snow_thick) moved down and up."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from nextsim_amd import _abi, dynamics, forcing as F, mesh as M
gm = M.make_mesh(sys.argv[1] if len(sys.argv) > 1 else "2km")
p, C_fix, C_alea = F.scale_params_to_mesh(F.default_params(), gm, alea_factor=0.33)
g = F.global_fields(gm, p, "arctic", C_fix, C_alea)
lm = M.localize(gm, 1)[0]; f = F.localize_fields(g, lm, gm.num_nodes)
fe = dynamics.FiniteElementDynamics(p); fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
fe.step(); fe.synchronize()
n = 10
t = time.perf_counter()
for _ in range(n): fe.step()
fe.synchronize(); res = (time.perf_counter() - t) / n
st = {k: v.copy() for k, v in f.items()}
ss, fo = _abi.state_struct(st), _abi.forcing_struct(f)
t = time.perf_counter()
for _ in range(n): assert fe.L.nxs_dyn_step_host(fe.h, C.byref(ss), C.byref(fo)) == 0
host = (time.perf_counter() - t) / n
fe.set_option("pin_host", 1)
assert fe.L.nxs_dyn_step_host(fe.h, C.byref(ss), C.byref(fo)) == 0      # registers the vectors
t = time.perf_counter()
for _ in range(n): assert fe.L.nxs_dyn_step_host(fe.h, C.byref(ss), C.byref(fo)) == 0
host_pinned = (time.perf_counter() - t) / n
fe.set_forcing_pair(f, f)
arr = {k: np.empty(lm.num_elements) for k in ("conc", "thick", "snow_thick")}
s = _abi.State()
for k in arr: setattr(s, k, _abi.dptr(arr[k]))
t = time.perf_counter()
for _ in range(n):
    fe.set_forcing_time(0.5, 0.5)
    fe.L.nxs_dyn_put_state(fe.h, C.byref(s))
    fe.step()
    fe.L.nxs_dyn_get_state(fe.h, C.byref(s))
part = (time.perf_counter() - t) / n
print(f"{lm.num_elements} triangles: resident {res*1e3:.2f} ms/step; step_host (whole state + forcing over PCIe every step) {host*1e3:.2f} ms/step, {host_pinned*1e3:.2f} with option pin_host; "
      f"resident state + device-blended forcing + 3 thermodynamic arrays down and up {part*1e3:.2f} ms/step")
