"""World-size-2 run of the multi-rank path on CPU: two processes, torch.distributed (gloo, 127.0.0.1),
each rank owns one partition, halo exchange by send/recv with the pack/unpack of updateGhosts
(FE.cpp:13963-13996).  Compute per rank is the CPU oracle (allowed in tests); what is exercised is the
partition + halo lists + exchange protocol that the GPU path uses with RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import cases
from oracle import pyoracle as O
from nextsim_amd import _abi
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
gm, p, g, lms, fields = cases.make_case("small", nparts=world)
r = O.OracleRank(lms[rank], p, fields[rank]); lm = lms[rank]

def exchange():
    reqs, rbufs = [], []
    for k, q in enumerate(lm.send_procs):
        reqs.append(dist.isend(torch.from_numpy(r.pack(k)), int(q)))
    for k, q in enumerate(lm.recv_procs):
        n = int(lm.recv_offsets[k + 1] - lm.recv_offsets[k]); t = torch.empty(2 * n, dtype=torch.float64)
        reqs.append(dist.irecv(t, int(q))); rbufs.append((k, t))
    for q_ in reqs: q_.wait()
    for k, t in rbufs: r.unpack(k, t.numpy())

steps = p.substeps; dte = p.dtime_step / steps
for it in range(2):
    r.prep()
    for s in range(steps):
        r.substep_solve(); exchange(); r.move_mesh(dte)
    for nit in range(50):
        r.smoother_sweep(); exchange()
    r.ow_tail(); r.update()
# the two scalar all-reduces of step(): checkRegridding (FE.cpp:8306) and checkFieldsFast (FE.cpp:14647)
ang, flip, rg = r.check_regridding()
t = torch.tensor([ang], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MIN)
c = torch.tensor([float(r.check_fields_fast() or rg)]); dist.all_reduce(c, op=dist.ReduceOp.SUM)
np.savez(os.path.join({out!r}, f"rank{{rank}}.npz"), minang=t.numpy(), crash=c.numpy(), **{{k: r.arr[k] for k in ("VT", "UM", "sigma0", "damage", "conc")}})
dist.barrier(); dist.destroy_process_group()
'''


def test_two_process_gloo_run_matches_in_process_multirank(tmp_path):
    from oracle import pyoracle as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for pr in procs:
        assert pr.wait(timeout=600) == 0
    gm, p, g, lms, fields = cases.make_case("small", nparts=2)
    ranks = [O.OracleRank(lm, p, f) for lm, f in zip(lms, fields)]
    for _ in range(2):
        O.multirank_step(ranks)
    angs = []
    for rank in range(2):
        z = np.load(tmp_path / f"rank{rank}.npz")
        for k in ("VT", "UM", "sigma0", "damage", "conc"):
            assert np.array_equal(z[k], ranks[rank].arr[k]), (rank, k)
        angs.append(float(z["minang"][0]))
        assert z["crash"][0] == 0
    assert angs[0] == angs[1] == min(r.check_regridding()[0] for r in ranks)
