"""Where a sub-step of the several-rank RESIDENT launch goes on a rank of N of the 2 km mesh, alone on the device with its mailboxes looped back (no neighbour, no xGMI:
every wait of the exchange is satisfied by the rank's own stores) -- the chain VERDICT r4 item 4(a) asks about.  Needs the library built with -DNXS_PHASE_TIMING:
    python3 scripts/phase_timing.py --build ; NXS_DYN_LIBRARY=$PWD/nextsim_amd/csrc/libnxsdyn_phase.so python3 scripts/phase_timing_loopback.py 8 [key=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from nextsim_amd import dynamics, forcing as F, mesh as M
nparts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
opts = dict(kv.split("=") for kv in sys.argv[2:])
gm = M.make_mesh("2km")
p, C_fix, C_alea = F.scale_params_to_mesh(F.default_params(), gm, alea_factor=0.33)
g = F.global_fields(gm, p, "arctic", C_fix, C_alea)
lm = M.localize(gm, nparts)[0]; f = F.localize_fields(g, lm, gm.num_nodes)
for loop in (1, 0):
    fe = dynamics.FiniteElementDynamics(p)
    fe.set_mesh(lm)
    if loop:
        assert fe.ipc_loopback()
    else:   # the same partition as a single-rank mesh would be cut: no exchange at all (the ghosts keep their values; timing only)
        fe.L.nxs_dyn_set_halo  # (left set: the kernels below are chosen by the options)
        assert fe.ipc_loopback()
    for k, v in {"fused": 4, "halo_fused": 1, "resident_wide": 1, **opts}.items():
        fe.set_option(k, int(v))
    if not loop:
        fe.set_option("resident_overlap", 1)
    fe.set_option("prepare", 1); fe.put_state(f); fe.set_forcing(f)
    for _ in range(3): fe.step()
    fe.synchronize(); fe.put_state(f); fe.set_option("timing_reset", 1)
    for _ in range(5): fe.step()
    fe.synchronize()
    tm, tr = fe.timing(), fe.traffic_model()
    print(f"rank 0 of {nparts} ({lm.num_elements} triangles), {'resident_overlap 0' if loop else 'resident_overlap 1'}: {tr['substep_kernel_name']}, sub-steps {tm['substeps_ms']:.3f} ms in {tm['substep_launches']} launch(es), "
          f"smoother {tm['smoother_ms']:.3f}, prep {tm['prep_ms']:.3f}, update {tm['update_ms']:.3f}, total {tm['total_ms']:.3f} ms")
    raw = fe.debug_array("phase_times").reshape(8192, 8)
    t = raw[:, :6]
    t = t[t[:, 0] > 0]
    nb_ = int(os.environ.get("NXS_N_BOUNDARY", "54"))
    b = raw[:nb_]
    if raw[8191, 1] > 0:   # the chain of the exchange inside sub-step 60 (a library with the finer stamps): every time relative to the LAST boundary patch's drained barrier
        last_drained = b[:, 3].max()
        print(f"  exchange chain of sub-step 60: last boundary patch drained at 0; its ticket back +{(raw[8191, 1] - last_drained) * 10e-3:.2f} us; flags stored +{(raw[8191, 2] - last_drained) * 10e-3:.2f}; "
              f"flag seen by the boundary patches +{(b[:, 6].min() - last_drained) * 10e-3:.2f} .. +{(b[:, 6].max() - last_drained) * 10e-3:.2f} (mean +{(b[:, 6].mean() - last_drained) * 10e-3:.2f}); "
              f"their barrier after the waits +{(b[:, 4].mean() - last_drained) * 10e-3:.2f}; first boundary patch drained {(b[:, 3].min() - last_drained) * 10e-3:.2f}")
    d = np.diff(t, axis=1) * 10e-3
    names = ("element phase (to barrier 1)", "node phase + stores issued", "stores drained + barrier", "publish + wait (ranks and patches) + barrier", "halo loads + barrier")
    print(f"  sub-step 60, {t.shape[0]} workgroups; spread of their starts {(t[:, 0].max() - t[:, 0].min()) * 10e-3:.2f} us")
    nb = int(os.environ.get("NXS_N_BOUNDARY", "54"))
    for lab, sel in ((f"boundary patches (the first {nb})", slice(0, nb)), ("interior patches", slice(nb, None))):
        print(f"   {lab}: start of the sub-step {(t[sel, 0].mean() - t[:, 0].min()) * 10e-3:.2f} us after the first workgroup's (spread {(t[sel, 0].max() - t[sel, 0].min()) * 10e-3:.2f})")
        for nm, col in zip(names, d[sel].T):
            print(f"    {nm:48s} mean {col.mean():6.2f} us   p10 {np.percentile(col, 10):6.2f}   p90 {np.percentile(col, 90):6.2f}   max {col.max():6.2f}")
        print(f"    one sub-step {(t[sel, 5] - t[sel, 0]).mean() * 10e-3:6.2f} us")
    fe.close()
