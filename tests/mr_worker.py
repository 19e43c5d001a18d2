"""Worker of tests/test_gpu_multirank.py: one rank of a multi-rank GPU run (all ranks may share GPU 0
on a one-GPU box).  Compares its partition with the in-process multi-rank oracle and writes a report."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch, torch.distributed as dist   # torch first: its bundled HIP runtime is the process's runtime
import cases
from nextsim_amd import dynamics
from oracle import pyoracle as O

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"]); out = sys.argv[1]; kind = sys.argv[2]; nsteps = int(sys.argv[3])
transport = sys.argv[4] if len(sys.argv) > 4 else "rccl"
over = json.loads(sys.argv[5]) if len(sys.argv) > 5 else {}
dev = int(os.environ.get("NXS_TEST_DEVICE", "0"))
dist.init_process_group("gloo", rank=rank, world_size=world)
report = {"rank": rank, "ok": False}
try:
    gm, p, g, lms, fields = cases.make_case(kind, nparts=world, **over)
    fe = dynamics.FiniteElementDynamics(p, device=dev)
    fe.set_mesh(lms[rank])
    lm = lms[rank]
    if transport == "rccl":
        ids = [dynamics.FiniteElementDynamics.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        try:
            fe.comm_init(ids[0], rank, world)
        except dynamics.NxsError as e:
            report["comm_error"] = str(e)
            raise
    elif transport in ("ipc", "ipc_sep"):
        def all_gather(obj):
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
        good = fe.ipc_setup(all_gather)
        report["ipc_selftest"] = bool(good)
        if not good:
            report["comm_error"] = "ipc self-test failed: " + getattr(fe, "_ipc_error", "")
            raise RuntimeError(report["comm_error"])
    else:
        # host-staged updateGhosts through "the caller's communicator" (here torch.distributed/gloo)
        def exchange(send, recv):
            reqs = []
            for k, q in enumerate(lm.send_procs):
                a, b = 2 * int(lm.send_offsets[k]), 2 * int(lm.send_offsets[k + 1])
                reqs.append(dist.isend(torch.from_numpy(send[a:b].copy()), int(q)))
            bufs = []
            for k, q in enumerate(lm.recv_procs):
                a, b = 2 * int(lm.recv_offsets[k]), 2 * int(lm.recv_offsets[k + 1])
                t = torch.empty(b - a, dtype=torch.float64)
                reqs.append(dist.irecv(t, int(q))); bufs.append((a, b, t))
            for r_ in reqs:
                r_.wait()
            for a, b, t in bufs:
                recv[a:b] = t.numpy()
        fe.set_halo_exchange(exchange)
    if transport == "ipc_sep":
        fe.set_option("halo_fused", 0)      # k_halo_push / k_halo_pull as separate kernels
    fe.put_state(fields[rank]); fe.set_forcing(fields[rank])
    for _ in range(nsteps):
        fe.step()
    fe.synchronize()
    got = fe.get_state()
    if transport == "ipc":
        # the exchange inside the sub-step kernel (default) must give the bits of the separate kernels
        report["launches_fused"] = fe.timing()["substep_launches"]
        fe.set_option("halo_fused", 0)
        fe.put_state(fields[rank]); fe.set_forcing(fields[rank])
        for _ in range(nsteps):
            fe.step()
        fe.synchronize()
        sep = fe.get_state()
        report["launches_separate"] = fe.timing()["substep_launches"]
        report["fused_equals_separate"] = bool(all(np.array_equal(got[k], sep[k]) for k in got))
    ranks = [O.OracleRank(lm, p, f) for lm, f in zip(lms, fields)]
    for _ in range(nsteps):
        O.multirank_step(ranks)
    ref = ranks[rank].arr
    errs = {k: cases.rel_err(got[k], ref[k]) for k in ("VT", "UM", "UT", "sigma0", "sigma1", "sigma2", "damage", "conc", "thick")}
    report.update(ok=True, errs=errs, crash=fe.checkFieldsFast(), timing=fe.timing())
    fe.close()
except Exception as e:  # noqa: BLE001
    report["error"] = repr(e)
json.dump(report, open(os.path.join(out, f"report{rank}.json"), "w"))
try:
    dist.barrier(); dist.destroy_process_group()
except Exception:
    pass
