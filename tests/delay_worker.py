"""Worker of tests/test_gpu_protocol_delays.py: one PROCESS of a multi-rank GPU run that walks a list of cases over ONE set-up.

A case = (kernel variant, options per rank, ONE rank delayed at ONE named point of the exchange protocols -- option "ipc_delay", include/nxs_dyn.h).  Every
case starts from the same state, runs one step (or case["steps"] of them) and is compared BIT FOR BIT with the first case of the first phase (the separate push / pull kernels, nobody
delayed).  A phase may re-create round 4's defect: the halo lists taken as given (test door "halo_one_directional"), connected with the caller-side bookkeeping.

Same process layout as mr_worker.py: NXS_RANKS_PER_PROC ranks per process as threads, gloo between the processes."""
import json, os, sys, threading, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch, torch.distributed as dist   # torch first: its bundled HIP runtime is the process's runtime
import cases
from nextsim_amd import dynamics

proc = int(os.environ["RANK"]); nproc = int(os.environ["WORLD_SIZE"]); rpp = int(os.environ.get("NXS_RANKS_PER_PROC", "1"))
world = nproc * rpp
out = sys.argv[1]
spec = json.load(open(sys.argv[2]))
dev = int(os.environ.get("NXS_TEST_DEVICE", "0"))
dist.init_process_group("gloo", rank=proc, world_size=nproc)
KEYS = ("VT", "UM", "UT", "sigma0", "sigma1", "sigma2", "damage", "conc", "thick")


class Gather:
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
gm, p, g, lms, fields = cases.make_case(spec["kind"], nparts=world, **spec.get("over", {}))


def run_rank(li, report):
    rank = proc * rpp + li
    report.update(rank=rank, ok=False, phases=[])
    lm = lms[rank]
    all_gather = lambda obj: gather(li, obj)  # noqa: E731
    fe = dynamics.FiniteElementDynamics(p, device=dev)
    ref = None
    for ph in spec["phases"]:
        old = bool(ph.get("one_directional", 0))
        fe.set_option("ipc_delay", 0)
        fe.set_option("halo_one_directional", 1 if old else 0)
        fe.set_mesh(lm)                       # (sets the halo lists again: the door is read there)
        # the old protocol's self-test is itself exposed to the race it was found by: no rounds there (a delayed self-test is a case of its own)
        good = fe.ipc_setup(all_gather, selftest_rounds=0 if old else 32, low_level=old)
        prep = {"one_directional": old, "ipc": bool(good), "ipc_error": getattr(fe, "_ipc_error", ""), "cases": []}
        report["phases"].append(prep)
        if not good:
            raise RuntimeError("ipc set-up failed: " + prep["ipc_error"])
        for case in ph["cases"]:
            res = {"name": case["name"]}
            prep["cases"].append(res)
            all_gather(0)
            opts = dict(case.get("options", {}))
            for k, v in case.get("rank_options", {}).items():
                opts[k] = v[rank % len(v)]
            for k, v in opts.items():
                fe.set_option(k, v)
            d = case.get("delay")
            fe.set_option("ipc_delay", (d[0] << 16) | (d[1] << 8) | d[2] if d else 0)
            if case.get("selftest"):
                import ctypes as C
                err = C.c_int32(0)
                all_gather(0)
                fe._chk(fe.L.nxs_dyn_ipc_selftest(fe.h, int(case["selftest"]), C.byref(err)))
                res["selftest_errors"] = int(err.value)
                fe.set_option("ipc_delay", 0)
                continue
            fe.put_state(fields[rank]); fe.set_forcing(fields[rank])
            fe.set_option("prepare", 1)
            all_gather(0)                       # nobody steps before every rank's state is resident and its tables are built
            try:
                for _ in range(int(case.get("steps", 1))):
                    fe.step()
                fe.synchronize()
            except dynamics.NxsError as e:
                res["error"] = str(e)
            got = fe.get_state()
            for name in spec.get("debug_arrays", []):        # work arrays of the step too (the prep kernels' records, ghosts included)
                got["debug:" + name] = fe.debug_array(name)
            res["launches"] = fe.timing()["substep_launches"]
            res["prep"] = fe.traffic_model()["prep_kernel_name"]
            res["kernel"] = fe.traffic_model()["substep_kernel_name"]
            res["crash"] = fe.checkFieldsFast()
            if ref is None:
                ref = got
                np.savez(os.path.join(out, f"ref{rank}.npz"), **{k: got[k] for k in KEYS})
                res["equal"] = True
            else:
                res["equal"] = bool(all(np.array_equal(got[k], ref[k]) for k in got))
                if not res["equal"]:
                    res["worst"] = {k: float(cases.rel_err(got[k], ref[k])) for k in got if not np.array_equal(got[k], ref[k])}
            fe.set_option("ipc_delay", 0)
    report["ok"] = True
    all_gather(0)                               # keep every mailbox alive until all ranks are done
    fe.close()


def guarded(li, report):
    try:
        run_rank(li, report)
    except Exception as e:  # noqa: BLE001
        report["error"] = repr(e) + "\n" + traceback.format_exc()
        gather.bar.abort()


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
