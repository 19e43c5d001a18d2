"""Worker of tests/test_gpu_multirank.py: one PROCESS of a multi-rank GPU run (all ranks may share GPU 0 on a one-GPU box).

A process hosts NXS_RANKS_PER_PROC ranks (default 1), one thread and one library handle each -- a GPU box admits only a few
processes on its card, so an 8-rank run is 4 processes x 2 ranks: neighbours in another process are reached through hipIpc,
neighbours in the same process through the pointer itself (nxs_dyn_ipc_connect tells them apart).  Every rank compares its
partition with the in-process multi-rank oracle and writes a report; with over["dump"] it writes its state instead and the
parent test does the comparison (the oracle of a big mesh is then computed once, beside the workers)."""
import json, os, sys, threading, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch, torch.distributed as dist   # torch first: its bundled HIP runtime is the process's runtime
import cases
from nextsim_amd import dynamics

proc = int(os.environ["RANK"]); nproc = int(os.environ["WORLD_SIZE"]); rpp = int(os.environ.get("NXS_RANKS_PER_PROC", "1"))
world = nproc * rpp
out = sys.argv[1]; kind = sys.argv[2]; nsteps = int(sys.argv[3])
transport = sys.argv[4] if len(sys.argv) > 4 else "rccl"
over = json.loads(sys.argv[5]) if len(sys.argv) > 5 else {}
dump = bool(over.pop("dump", False))
options = over.pop("options", {})
dev = int(os.environ.get("NXS_TEST_DEVICE", "0"))
dist.init_process_group("gloo", rank=proc, world_size=nproc)
KEYS = ("VT", "UM", "UT", "sigma0", "sigma1", "sigma2", "damage", "conc", "thick")


class Gather:
    """all_gather over every rank of every process: the threads of a process meet at a barrier, one of them runs the
    process-level gloo gather, all read the flattened result."""

    def __init__(self):
        self.slots = [None] * rpp
        self.bar = threading.Barrier(rpp)
        self.result = None

    def __call__(self, li, obj):
        self.slots[li] = obj
        if self.bar.wait() == 0:
            per_proc = [None] * nproc
            dist.all_gather_object(per_proc, list(self.slots))
            self.result = [o for lst in per_proc for o in lst]
        self.bar.wait()
        res = self.result
        self.bar.wait()
        return res


gather = Gather()
gm, p, g, lms, fields = cases.make_case(kind, nparts=world, **over)
if not dump:
    from oracle import pyoracle as O


def run_rank(li, report):
    rank = proc * rpp + li
    report.update(rank=rank, ok=False)
    lm = lms[rank]
    fe = dynamics.FiniteElementDynamics(p, device=dev)
    fe.set_mesh(lm)
    for k, v in options.items():
        fe.set_option(k, v)
    all_gather = lambda obj: gather(li, obj)  # noqa: E731
    if transport == "rccl":
        ids = all_gather(dynamics.FiniteElementDynamics.comm_unique_id() if rank == 0 else None)
        try:
            fe.comm_init(ids[0], rank, world)
        except dynamics.NxsError as e:
            report["comm_error"] = str(e)
            raise
        report["comm_selftest_errors"] = fe.comm_selftest()
    elif transport in ("ipc", "ipc_sep"):
        good = fe.ipc_setup(all_gather)
        report["ipc_selftest"] = bool(good)
        if not good:
            report["comm_error"] = "ipc self-test failed: " + getattr(fe, "_ipc_error", "")
            raise RuntimeError(report["comm_error"])
    else:
        # host-staged updateGhosts through "the caller's communicator": every rank publishes its packed segments, takes its own
        def exchange(send, recv):
            segs = {int(q): send[2 * int(lm.send_offsets[k]):2 * int(lm.send_offsets[k + 1])].copy() for k, q in enumerate(lm.send_procs)}
            everyone = all_gather(segs)
            for k, q in enumerate(lm.recv_procs):
                recv[2 * int(lm.recv_offsets[k]):2 * int(lm.recv_offsets[k + 1])] = everyone[int(q)][rank]
        fe.set_halo_exchange(exchange)
    if transport == "ipc_sep":
        fe.set_option("halo_fused", 0)      # k_halo_push / k_halo_pull as separate kernels
    fe.put_state(fields[rank]); fe.set_forcing(fields[rank])
    fe.set_option("prepare", 1)             # the lazily built tables now: building them frees device memory, which synchronises the SHARED device
    all_gather(0)                           # nobody steps before every rank's state is resident
    for i in range(nsteps):
        if i == 2 and nsteps > 4:            # rehearsals time the steps after the warm-up ones (graph capture, lazily built tables)
            fe.synchronize(); all_gather(0); fe.set_option("timing_reset", 1)
        fe.step()
    fe.synchronize()
    got = fe.get_state()
    report["timing"] = fe.timing()
    if os.environ.get("NXS_PHASE_DUMP") and rank == 0:   # (a library built with -DNXS_PHASE_TIMING: scripts/phase_timing.py --build)
        np.save(os.path.join(os.environ["NXS_PHASE_DUMP"], "phase0.npy"), fe.debug_array("phase_times"))
    if transport == "ipc":
        # the exchange inside the sub-step kernel (default) must give the bits of the separate kernels
        report["launches_fused"] = report["timing"]["substep_launches"]
        all_gather(0)
        fe.set_option("halo_fused", 0)
        fe.put_state(fields[rank]); fe.set_forcing(fields[rank])
        fe.set_option("prepare", 1)
        all_gather(0)
        for _ in range(nsteps):
            fe.step()
        fe.synchronize()
        sep = fe.get_state()
        report["launches_separate"] = fe.timing()["substep_launches"]
        report["fused_equals_separate"] = bool(all(np.array_equal(got[k], sep[k]) for k in got))
    report["crash"] = fe.checkFieldsFast()
    if dump:
        np.savez(os.path.join(out, f"state{rank}.npz"), **{k: got[k] for k in KEYS})
    else:
        ranks = [O.OracleRank(lm_, p, f) for lm_, f in zip(lms, fields)]
        for _ in range(nsteps):
            O.multirank_step(ranks)
        ref = ranks[rank].arr
        report["errs"] = {k: cases.rel_err(got[k], ref[k]) for k in KEYS}
    report["ok"] = True
    all_gather(0)                           # keep every mailbox alive until all ranks are done
    fe.close()


def guarded(li, report):
    try:
        run_rank(li, report)
    except Exception as e:  # noqa: BLE001
        report["error"] = repr(e) + "\n" + traceback.format_exc()
        gather.bar.abort()                  # the sibling threads must not wait for this one for ever


reports = [dict() for _ in range(rpp)]
threads = [threading.Thread(target=guarded, args=(li, reports[li])) for li in range(rpp)]
for t in threads:
    t.start()
for t in threads:
    t.join()
for li, r in enumerate(reports):
    r.setdefault("rank", proc * rpp + li); r.setdefault("ok", False)
    json.dump(r, open(os.path.join(out, f"report{r['rank']}.json"), "w"))
try:
    dist.barrier(); dist.destroy_process_group()
except Exception:
    pass
