"""PCIe rates of the boundary's copies with page-locked host vectors (option pin_host): put_state, set_forcing, get_state timed apart at 2 km."""
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
fe = dynamics.FiniteElementDynamics(p); fe.set_option("pin_host", 1); fe.set_mesh(lm)
st = {k: v.copy() for k, v in f.items()}
ss, fo = _abi.state_struct(st), _abi.forcing_struct(f)
Nn, Ne = lm.num_nodes, lm.num_elements
up_state = (3 * 2 * Nn + 17 * Ne) * 8; up_forc = (2 * 2 * Nn + Nn + Ne) * 8; down = (3 * 2 * Nn + 13 * Ne) * 8
for name, fn, nbytes in (("put_state", lambda: fe.L.nxs_dyn_put_state(fe.h, C.byref(ss)), up_state), ("set_forcing", lambda: fe.L.nxs_dyn_set_forcing(fe.h, C.byref(fo)), up_forc),
                         ("get_state", lambda: fe.L.nxs_dyn_get_state(fe.h, C.byref(ss)), down)):
    assert fn() == 0; assert fn() == 0
    t = time.perf_counter()
    for _ in range(10): assert fn() == 0
    dt = (time.perf_counter() - t) / 10
    print(f"{name}: {nbytes / 1e6:.1f} MB in {dt * 1e3:.2f} ms = {nbytes / dt / 1e9:.1f} GB/s", flush=True)
fe.step(); fe.synchronize()
t = time.perf_counter()
for _ in range(10): fe.step()
fe.synchronize(); print(f"step: {(time.perf_counter() - t) / 10 * 1e3:.2f} ms")
print("phases of the resident step:", {k: round(v, 3) for k, v in fe.timing().items() if k.endswith("_ms")})
fe.set_option("timing_reset", 1)
t = time.perf_counter()
for _ in range(10): assert fe.L.nxs_dyn_step_host(fe.h, C.byref(ss), C.byref(fo)) == 0
print(f"step_host: {(time.perf_counter() - t) / 10 * 1e3:.2f} ms")
print("phases of the step inside step_host:", {k: round(v, 3) for k, v in fe.timing().items() if k.endswith("_ms")})
