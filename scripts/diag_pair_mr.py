"""Diagnostic: k_substep_pair<HALO> against the separate kernels, several ranks as threads of one process on one device.
python3 scripts/diag_pair_mr.py <kind> <nparts> <nsteps>  -- prints, per rank and array, how many own / ghost entries differ."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import cases
from nextsim_amd import dynamics
kind, world, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
over = {"ragged_seed": int(os.environ["RAGGED"])} if os.environ.get("RAGGED") else {}
gm, p, g, lms, fields = cases.make_case(kind, nparts=world, **over)
bar = threading.Barrier(world); slots = [None] * world; res = [None]
def gather(r, obj):
    slots[r] = obj
    if bar.wait() == 0: res[0] = list(slots)
    bar.wait(); out = res[0]; bar.wait(); return out
out = [None] * world
def run(r):
    lm = lms[r]
    fe = dynamics.FiniteElementDynamics(p); fe.set_mesh(lm)
    fe.set_option("pair_regs", int(os.environ.get("PAIR", "1")))
    assert fe.ipc_setup(lambda o: gather(r, o))
    states = []
    for hf in (1, 0):
        fe.set_option("halo_fused", hf)
        fe.put_state(fields[r]); fe.set_forcing(fields[r]); fe.set_option("prepare", 1); gather(r, 0)
        per_step = []
        for _ in range(nsteps):
            fe.step(); fe.synchronize(); per_step.append(fe.get_state())
        states.append(per_step); gather(r, 0)
    tm = fe.timing()
    lines = []
    for st in range(nsteps):
        a, b = states[0][st], states[1][st]
        for k in ("VT", "UM", "UT", "sigma0", "damage", "conc"):
            d = a[k] != b[k]
            if not d.any(): continue
            if k in ("VT", "UM", "UT"):
                Nn, No = lm.num_nodes, lm.local_ndof
                idx = np.nonzero(d)[0] % Nn
                lines.append(f"  step {st} {k}: {d.sum()} differ, own {int((idx < No).sum())} ghost {int((idx >= No).sum())}, max abs {np.abs(a[k] - b[k]).max():.3e}, first {idx[:6]}")
            else:
                idx = np.nonzero(d)[0]
                lines.append(f"  step {st} {k}: {d.sum()} differ, own elements {int((idx < lm.local_nelements).sum())} ghost elements {int((idx >= lm.local_nelements).sum())}")
    out[r] = f"rank {r}: Nn {lm.num_nodes} No {lm.local_ndof} launches {tm['substep_launches']}\n" + "\n".join(lines)
    gather(r, 0); fe.close()
th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join() for t in th]
print("\n".join(str(o) for o in out))
