#!/usr/bin/env python3
"""bench.py -- element-updates/s of one dynamics step (explicitSolve + update) on N MI355X.

    python bench.py --gpus N --steps K --warmup W [--mesh 2km|10km|40km] [--no-cpu-baseline]

N > 1 is launched by the driver as one rank per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
or simply as `python bench.py --gpus N ...` (no WORLD_SIZE in the environment): bench.py then starts its N ranks itself, as child
processes created before anything of torch or HIP is loaded (launch_ranks), and relays rank 0's line.
A "step" is one pass of the hot path: FiniteElement::step()'s dynamics block (FE.cpp:8197-8214) =
explicitSolve() with dynamics.substeps = 120 BBM sub-steps + 50 smoother sweeps + update(), on the
synthetic pan-Arctic mesh with inputs resident in HBM.  The mesh is domain-decomposed over the N
ranks (total work fixed => "strong" scaling), halo exchange of M_VT over RCCL.

Rank 0 prints ONE JSON line.  metric = element-updates/s = Ne_global * substeps * K / t.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of this pool supports dmabuf IPC only: without this hipIpcGetMemHandle (the mailboxes of the device-direct halo transport, RCCL)
# fails with "invalid argument"; it is exported on the GPU boxes already -- kept here for a launcher that builds its own environment
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# Algorithmic HBM bytes per element-update of the sub-step loop (DESIGN.md "Roofline model"):
# 172 B per element + 217 B per node per sub-step.
BYTES_PER_ELEMENT = 172.0
BYTES_PER_NODE = 217.0
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mesh", default="2km", help="2km (~1.5M triangles, default: the mesh north_star's target and "
                    "HBM roofline are quoted on), 10km (~60k), 40km, small")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the secondary 10 km measurement")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not measure roofline.traffic with two rocprofv3 --pmc child runs (replay the committed profile)")
    return ap.parse_args()


PMC_PROFILES = ("profiles/r04_pmc_traffic.json", "profiles/r03_pmc_traffic.json", "profiles/r02_pmc_traffic.json", "profiles/r01_v2_pmc_traffic.json")


def pmc_traffic(mesh, world, kernel):
    """(HBM bytes per launch of `kernel`, the file they come from) from the COMMITTED rocprofv3 --pmc passes of the single-GPU 2 km run (FETCH_SIZE x2 +
    WRITE_SIZE, KiB, separate passes -- corrected as MI355X_MICROARCH.md prescribes for gfx950).  Only a stand-in when the live measurement below cannot
    run, and labelled as a replay in the JSON line; (None, None) when no profile applies to this run."""
    if mesh != "2km" or world != 1:
        return None, None
    for rel in PMC_PROFILES:
        try:
            prof = json.load(open(os.path.join(ROOT, rel)))
            for k, v in prof["kernels"].items():
                if k.startswith(kernel):
                    return v["hbm_bytes_per_launch"], rel
        except Exception:  # noqa: BLE001
            continue
    return None, None


def pmc_traffic_live(mesh, world, rank, kernels, options=(), timeout_s=300):
    """({kernel prefix: HBM bytes per launch MEASURED NOW}, how): two child runs of one step of THIS RANK'S partition (scripts/run_steps.py) under
    `rocprofv3 --pmc FETCH_SIZE` and `rocprofv3 --pmc WRITE_SIZE` -- separate passes, counters only, no tracing beside them, the program directly behind
    `--` -- reduced as scripts/make_pmc_profile.py does: (2 x FETCH_SIZE + WRITE_SIZE) KiB averaged over the launches of each kernel (gfx950 counts half of
    the coalesced streaming reads: MI355X_MICROARCH.md; calibrated in profiles/r01_v1_pmc_traffic.json).  world > 1: the child runs the rank's partition ALONE
    on its device, its mailboxes connected to themselves (`--loopback`: every wait of the exchange inside the kernels is then satisfied by the rank's own
    stores -- counter collection serialises the kernels of a device, so two ranks waiting for each other could not be profiled -- the ghosts receive
    meaningless velocities, the launches walk the same tables and move the same bytes).  ({}, reason) when it cannot be done here."""
    import csv, glob, shutil, subprocess, tempfile
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {}, "this run is itself under a profiler"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="nxs_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "NXS_DYN_LIBRARY", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                              "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    env["TMPDIR"] = "/tmp"
    child = [sys.executable, os.path.join(ROOT, "scripts", "run_steps.py"), "--mesh", mesh, "--steps", "1", "--graph", "0"]
    if world > 1:
        child += ["--nparts", str(world), "--rank", str(rank), "--loopback"]
    for kv in options:
        child += ["--opt", kv]
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--"] + child
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            acc, n = {k: 0.0 for k in kernels}, {k: 0 for k in kernels}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] != counter:
                        continue
                    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    for k in kernels:
                        if name.startswith(k):
                            acc[k] += float(r["Counter_Value"]); n[k] += 1
            vals[counter] = {k: (acc[k] / n[k], n[k]) for k in kernels if n[k] > 0}
        got = {k: int((2.0 * vals["FETCH_SIZE"][k][0] + vals["WRITE_SIZE"][k][0]) * 1024) for k in kernels if k in vals["FETCH_SIZE"] and k in vals["WRITE_SIZE"]}
        if not got:
            return {}, "no counter rows for " + ", ".join(kernels)
        detail = "; ".join(f"{k}: FETCH_SIZE {vals['FETCH_SIZE'][k][0]:.1f} KiB, WRITE_SIZE {vals['WRITE_SIZE'][k][0]:.1f} KiB over {vals['FETCH_SIZE'][k][1]} launches" for k in got)
        return got, ("MEASURED IN THIS RUN: two child passes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate, counters only) over one step of "
                     + ("the same mesh" if world == 1 else f"rank {rank}'s partition of {world}, alone on the device with its mailboxes looped back to itself")
                     + f" (`{' '.join(child[1:])}`), (2 x FETCH_SIZE + WRITE_SIZE) KiB averaged per kernel (gfx950 reports half of the coalesced streaming reads, "
                     f"MI355X_MICROARCH.md); {detail}")
    except Exception as e:  # noqa: BLE001 -- the bench line survives without it
        return {}, "failed: " + repr(e)[:200]
    finally:
        shutil.rmtree(out, ignore_errors=True)


_MESHES = {}
_COVER = {}


def build_case(kind, nparts, rank, state="arctic"):
    from nextsim_amd import forcing as F, mesh as M
    if kind not in _MESHES:
        _MESHES[kind] = M.make_mesh(kind)
    gm = _MESHES[kind]
    p, C_fix, C_alea = F.scale_params_to_mesh(F.default_params(), gm, alea_factor=0.33)
    g = F.global_fields(gm, p, state, C_fix, C_alea)
    lm = M.localize(gm, nparts)[rank]
    f = F.localize_fields(g, lm, gm.num_nodes)
    _COVER[id(f)] = ice_cover(g)    # of the whole mesh, not of this rank's partition
    return gm, p, lm, f


def ice_cover(f):
    """Fractions of the triangles that are ice free / in the 0 < A <= 0.1 band updateSigmaDamage skips (FE.cpp:4146)."""
    if id(f) in _COVER:
        return _COVER[id(f)]
    c = f["conc"]
    return {"ice_free_fraction": float((c == 0).mean()), "low_concentration_fraction": float(((c > 0) & (c <= 0.1)).mean())}


def run_gpu(kind, args, rank, world, local_rank, dist, torch, unique_id_fn, state="arctic", own_device=False):
    from nextsim_amd import dynamics
    gm, p, lm, f = build_case(kind, world, rank, state)
    fe = dynamics.FiniteElementDynamics(p, device=local_rank)
    fe.set_mesh(lm)
    if os.environ.get("NXS_BENCH_PATCH_NODES"):   # rehearsals: e.g. two ranks of 91 k triangles with the 180-node patches a rank of eight has
        fe.set_option("patch_nodes", int(os.environ["NXS_BENCH_PATCH_NODES"]))
    if os.environ.get("NXS_BENCH_OPTIONS"):       # experiments (e.g. "pair_nodes=416" on a mesh numbered in tiles of 416 nodes): key=value,... ; they go to the counter passes too
        fe._bench_options = dict(getattr(fe, "_bench_options", {}))
        for kv in os.environ["NXS_BENCH_OPTIONS"].split(","):
            k, v = kv.split("=")
            fe.set_option(k, int(v)); fe._bench_options[k] = int(v)
    transport = "none"
    halo = {"transport": "none"}
    if world > 1:
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)       # (librccl's banner, should the RCCL transport be initialised: stdout carries the one JSON line only)
        try:
            transport, halo = setup_halo_transport(fe, lm, rank, world, dist, torch, unique_id_fn)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    fe.put_state(f)
    fe.set_forcing(f)
    if transport.startswith("device-direct"):
        text, variants = choose_halo_kernels(fe, f, rank, world, dist, torch, own_device)
        transport += text
        halo.update(variants)
        fe.put_state(f)

    def barrier():
        if world > 1:
            dist.barrier()

    fe.set_option("prepare", 1)   # (lazily built tables now: ranks that share a device must not free device memory under each other's first step)
    barrier()
    for _ in range(args.warmup):
        fe.step()
    fe.synchronize()
    fe.set_option("timing_reset", 1)
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fe.step()
    fe.synchronize(); torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    tm = fe.timing()
    traffic = fe.traffic_model()
    step_ms = fe.step_times()
    options_used = dict(getattr(fe, "_bench_options", {}))
    crash = fe.checkFieldsFast()
    fe.close()
    phases = {k: tm[k] for k in ("prep_ms", "substeps_ms", "smoother_ms", "update_ms", "total_ms")}
    if world > 1:   # the slowest rank's figure for every phase (the line is rank 0's otherwise)
        t = torch.tensor([phases[k] for k in sorted(phases)] + [float(crash)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        phases_max = {k: float(v) for k, v in zip(sorted(phases), t[:-1])}
        crash = int(t[-1])
    else:
        phases_max = dict(phases)
    return dict(gm=gm, p=p, lm=lm, f=f, dt=dt, timing=tm, crash=crash, transport=transport, halo=halo, phases_max=phases_max,  # (f: rank-local fields)
                traffic=traffic, step_ms=step_ms, options=options_used)


def choose_halo_kernels(fe, f, rank, world, dist, torch, own_device):
    """How updateGhosts runs over the device-direct mailboxes: (a) inside ONE resident launch per step (option fused = 4: needs the
    device to itself, so only tried when every rank has its own GPU), (b) inside one launch per sub-step, (c) as separate push / pull
    kernels.  Each candidate runs one full step on this machine and this partition; a candidate is kept only if it gives, on every
    rank, the bits of (c), and the fastest of those is used (NXS_HALO_VARIANT=resident|inkernel|separate forces one)."""
    import numpy as np
    from nextsim_amd import dynamics
    # name: (fused, halo_fused, pair_regs, text)
    variants = {"resident": (4, 1, -1, " + exchange inside ONE resident launch per step"),
                "resident_overlap": (4, 1, -1, " + exchange inside ONE resident launch per step, interior elements computed under the exchange"),
                "resident_norelease": (4, 1, -1, " + exchange inside ONE resident launch per step, no release fence in front of a sub-step's flags (option resident_release 0)"),
                "inkernel_pair": (3, 1, 1, " + both exchanges of TWO sub-steps inside one launch (k_substep_pair<HALO>)"),
                "inkernel": (3, 1, 0, " + exchange inside the sub-step kernel (one launch per sub-step)"),
                "separate": (3, 0, 0, ", separate push/pull kernels")}

    def select(name):
        opts = {"resident_wide": 1 if own_device else 0,   # (only matters where one workgroup per CU covers a rank's partition)
                "resident_overlap": 1 if name == "resident_overlap" else 0, "resident_release": 0 if name == "resident_norelease" else 1,
                "fused": variants[name][0], "halo_fused": variants[name][1], "pair_regs": variants[name][2]}
        for k in ("resident_wide", "resident_overlap", "resident_release", "pair_regs", "fused", "halo_fused"):
            fe.set_option(k, opts[k])
        fe._bench_options = opts   # (what the PMC child run of this partition is given)

    force = os.environ.get("NXS_HALO_VARIANT")
    if force in variants:
        select(force)
        return variants[force][3] + " (forced)", {"kept_variant": force, "forced": True, "variants": {}}

    def agree(flag):   # every rank calls this the same number of times, whatever happened to it
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t[0]) == 1.0

    def all_gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    # (NXS_BENCH_TRY_RESIDENT=1: rehearsal on a shared device with a mesh small enough for every rank's workgroups to be resident)
    order = (["resident", "resident_overlap", "resident_norelease"] if (own_device or os.environ.get("NXS_BENCH_TRY_RESIDENT") == "1") else []) + ["inkernel_pair", "inkernel", "separate"]
    results = {}
    report = {n: {"status": "not tried"} for n in variants}   # -> the JSON line (config.halo): per variant ms/step, bits, errors
    for name in order:
        if name in ("resident_overlap", "resident_norelease") and "resident" not in results:
            report[name] = {"status": "skipped: the plain resident launch did not run"}
            continue                               # (the same on every rank: `results` only holds what all ranks agreed on)
        mine, state, secs, err_text = True, None, 0.0, None
        try:
            select(name)
            fe.put_state(f)
            fe.step(); fe.synchronize()           # (graph capture happens here)
            state = fe.get_state()
            t0 = time.perf_counter()
            fe.step(); fe.step(); fe.synchronize()
            secs = time.perf_counter() - t0
            if name.startswith("resident") and fe.timing()["substep_launches"] != 1:
                mine = False                       # the library fell back (partition too large for one round of workgroups)
                err_text = "the library fell back to one kernel per sub-step (partition does not fit one round of resident workgroups, or the device's slots are taken)"
            if name == "inkernel_pair" and fe.traffic_model()["substep_kernel_name"] != "k_substep_pair":
                mine = False                       # (an odd number of sub-steps, patches that do not fit, the device's slots taken: it ran as "inkernel")
                err_text = "the library fell back to one kernel per sub-step (two sub-steps per launch not possible on this partition, or the device's slots are taken)"
        except dynamics.NxsError as e:
            print(f"[bench rank {rank}] halo variant {name}: {e}", file=sys.stderr, flush=True)
            mine = False
            err_text = str(e)[:300]
        errs = [e for e in all_gather(err_text) if e]
        if agree(mine):
            tt = torch.tensor([secs], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            results[name] = (state, float(tt[0]))
            report[name] = {"status": "ran", "ms_per_step": float(tt[0]) * 500.0}
        else:
            report[name] = {"status": "failed", "error": errs[0] if errs else "failed on another rank", "ranks_failed": len(errs)}
            # a rank that timed out is out of step with its neighbours: fresh mailboxes before the next candidate
            try:
                fe.set_option("fused", 3)
                fe.ipc_setup(all_gather)
            except Exception as e:  # noqa: BLE001 -- the line must survive: the separate kernels / the next transport follow
                print(f"[bench rank {rank}] re-connecting the mailboxes after {name}: {e}", file=sys.stderr, flush=True)
    fe.set_option("fused", 3)
    if "separate" not in results:
        fe.set_option("halo_fused", 0); fe.set_option("pair_regs", 0)
        fe._bench_options = {"fused": 3, "halo_fused": 0, "pair_regs": 0}
        return (", separate push/pull kernels (its own check step failed on some rank: see stderr)",
                {"kept_variant": "separate", "forced": False, "variants": report, "note": "the check step of the separate kernels failed on some rank"})
    ref = results["separate"][0]
    ok = {n: agree(all(np.array_equal(st[k], ref[k]) for k in ref)) for n, (st, _) in results.items()}
    for n in results:
        report[n]["bit_identical_to_separate_kernels"] = bool(ok[n])
        if not ok[n]:
            report[n]["status"] = "rejected: bits differ from the separate kernels"
    good = [n for n in order if n in results and ok[n]]
    best = min(good, key=lambda n: results[n][1])
    select(best)
    note = "; ".join(f"{n} {results[n][1] * 500:.2f} ms/step" + ("" if ok[n] else " (bits differ: rejected)") for n in order if n in results)
    dropped = [n for n in order if n not in results]
    return (variants[best][3] + f" (kept variants bit-identical to the separate kernels; {note}" + (f"; failed: {', '.join(dropped)}" if dropped else "") + ")",
            {"kept_variant": best, "forced": False, "variants": report})


def setup_halo_transport(fe, lm, rank, world, dist, torch, unique_id_fn):
    """RCCL over xGMI (device-direct send/recv of the packed M_VT halo) unless NXS_HALO_TRANSPORT=host or
    the communicator cannot be created on every rank; then the host-staged transport
    (nxs_dyn_set_halo_exchange_fn) through torch.distributed/gloo -- slower, but the job still runs."""
    from nextsim_amd import dynamics
    want = os.environ.get("NXS_HALO_TRANSPORT", "auto")

    def all_gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    info = {"transport": None, "requested": want, "mailbox_selftest": "not tried", "rccl_selftest": "not tried"}
    if want in ("auto", "ipc"):
        # device-direct peer stores over xGMI; kept only if its self-test passes on every rank
        try:
            if fe.ipc_setup(all_gather):
                info.update(transport="device-direct mailboxes", mailbox_selftest="passed on every rank (64 rounds of coded payloads through every link, both publishing protocols)")
                return "device-direct peer mailboxes (hipIpc, xGMI P2P stores)", info
            info["mailbox_selftest"] = "failed or refused on some rank: " + str(getattr(fe, "_ipc_error", ""))[:200]
        except Exception as e:  # noqa: BLE001
            print(f"[bench rank {rank}] ipc transport unavailable: {e}", file=sys.stderr, flush=True)
            info["mailbox_selftest"] = "error: " + str(e)[:200]
    ok = 0.0
    if want in ("auto", "rccl"):
        try:
            fe.comm_init(unique_id_fn(), rank, world)
            ok = 1.0
        except dynamics.NxsError as e:
            print(f"[bench rank {rank}] RCCL communicator failed: {e}", file=sys.stderr, flush=True)
    t = torch.tensor([ok], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if float(t[0]) == 1.0:
        bad = 1.0
        try:   # checked payloads through every halo segment before the transport is trusted (collective)
            bad = float(fe.comm_selftest())
        except dynamics.NxsError as e:
            print(f"[bench rank {rank}] RCCL self-test failed: {e}", file=sys.stderr, flush=True)
        t = torch.tensor([bad], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if float(t[0]) == 0.0:
            info.update(transport="rccl", rccl_selftest="passed on every rank")
            return "rccl (grouped ncclSend/ncclRecv, self-test passed)", info
        info["rccl_selftest"] = f"{int(t[0])} wrong values somewhere"
        print(f"[bench rank {rank}] RCCL self-test: {int(t[0])} wrong values somewhere; host-staged instead", file=sys.stderr, flush=True)
    else:
        info["rccl_selftest"] = "the communicator could not be created on every rank"

    def exchange(send, recv):
        reqs, bufs = [], []
        for k, q in enumerate(lm.send_procs):
            a, b = 2 * int(lm.send_offsets[k]), 2 * int(lm.send_offsets[k + 1])
            reqs.append(dist.isend(torch.from_numpy(send[a:b].copy()), int(q)))
        for k, q in enumerate(lm.recv_procs):
            a, b = 2 * int(lm.recv_offsets[k]), 2 * int(lm.recv_offsets[k + 1])
            tt = torch.empty(b - a, dtype=torch.float64)
            reqs.append(dist.irecv(tt, int(q))); bufs.append((a, b, tt))
        for r_ in reqs:
            r_.wait()
        for a, b, tt in bufs:
            recv[a:b] = tt.numpy()
    fe.set_halo_exchange(exchange)
    info["transport"] = "host-staged"
    return "host-staged (gloo)", info


def aux_rccl(kind, args, rank, world, local_rank, dist, torch, unique_id_fn, kept_ms):
    """north_star's named transport, timed beside whatever transport the main line kept: a second handle on the same partition whose
    updateGhosts (FE.cpp:13963-13996) is a grouped ncclSend / ncclRecv of the packed M_VT halo on the compute stream (RCCL over xGMI;
    one kernel per sub-step + pack / unpack kernels, eager launches).  Collective; every failure ends in a structured entry, never in
    a lost line."""
    from nextsim_amd import dynamics
    out = {"transport": "RCCL: grouped ncclSend/ncclRecv of the packed M_VT halo per updateGhosts, on the compute stream"}

    def all_ok(flag):
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t[0]) == 1.0

    def first_error(text):
        got = [None] * world
        dist.all_gather_object(got, text)
        errs = [e for e in got if e]
        return errs[0] if errs else None

    fe, err = None, None
    try:
        gm, p, lm, f = build_case(kind, world, rank)
        fe = dynamics.FiniteElementDynamics(p, device=local_rank)
        fe.set_mesh(lm)
    except Exception as e:  # noqa: BLE001
        err = "set-up: " + str(e)[:300]
    e0 = first_error(err)
    if e0:
        if fe: fe.close()
        return dict(out, status="failed", error=e0)
    uid = unique_id_fn()           # (collective: rank 0's ncclGetUniqueId, or None when that failed)
    if uid is None:
        fe.close()
        return dict(out, status="failed", error="ncclGetUniqueId failed on rank 0 (librccl not loadable?)")
    t0 = time.perf_counter()
    try:
        fe.comm_init(uid, rank, world)
    except dynamics.NxsError as e:
        err = "ncclCommInitRank: " + str(e)[:300]
    e0 = first_error(err)
    out["comm_init_s"] = time.perf_counter() - t0
    if e0:
        fe.close()
        return dict(out, status="failed", error=e0)
    bad = -1
    try:
        bad = fe.comm_selftest()   # coded payloads through every halo segment
    except dynamics.NxsError as e:
        err = "self-test: " + str(e)[:300]
    t = torch.tensor([float(bad)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    e0 = first_error(err)
    out["selftest_wrong_values"] = int(t[0])
    if e0 or int(t[0]) != 0:
        fe.close()
        return dict(out, status="failed", error=e0 or "self-test: wrong values arrived")
    try:
        fe.put_state(f); fe.set_forcing(f)
        fe.step(); fe.synchronize()
        n = max(2, min(args.steps, 5))
        fe.set_option("timing_reset", 1)
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fe.step()
        fe.synchronize(); torch.cuda.synchronize(); dist.barrier()
        dt = time.perf_counter() - t0
        tm = fe.timing()
        crash = fe.checkFieldsFast()
    except dynamics.NxsError as e:
        err = "steps: " + str(e)[:300]
    e0 = first_error(err)
    fe.close()
    if e0:
        return dict(out, status="failed", error=e0)
    t = torch.tensor([dt, float(crash)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = float(t[0]) / n * 1e3
    return dict(out, status="ran", steps=n, ms_per_step=ms, value=gm.num_elements * p.substeps / (ms * 1e-3), unit="element-updates/s",
                launches_per_step_in_the_substep_loop=int(tm["substep_launches"]), fields_ok=int(t[1]) == 0,
                kept_transport_ms_per_step=kept_ms, slowdown_vs_kept=ms / kept_ms if kept_ms else None)


def aux_spmv(gm, reps=100):
    """include/nxs_krylov.h on the bench mesh: sliced-ELLPACK SpMV, algorithmic bytes 12 B per non-zero + 16 B per row."""
    import numpy as np
    import scipy.sparse as sp
    from nextsim_amd import krylov
    rp, ci = krylov.csr_pattern((gm.tri + 1).astype(np.int32), gm.num_nodes)
    rng = np.random.default_rng(0)
    va = rng.normal(size=ci.size)
    va[ci == np.repeat(np.arange(gm.num_nodes), np.diff(rp))] = 10.0
    A = sp.kron(sp.csr_matrix((va, ci, rp), shape=(gm.num_nodes,) * 2), sp.csr_matrix(np.array([[1.0, 0.3], [-0.3, 1.0]])), format="csr")
    A.sort_indices()
    s = krylov.Solver()
    s.set_matrix(A.indptr, A.indices, A.data)
    x = rng.normal(size=A.shape[0])
    y, ms = s.spmv(x, reps=reps)
    inf = s.info()
    s.close()
    ok = bool(np.abs(y - A @ x).max() <= 1e-11 * np.abs(y).max())
    return {"workload": f"SpMV, 2-dof (u, v) block pattern of the mesh: {A.shape[0]} rows, {inf['nnz']} non-zeros (extension, no live reference)",
            "us_per_spmv": ms * 1e3, "achieved": inf["spmv_bytes"] / ms / 1e6, "unit": "GB/s", "peak": 8000.0,
            "frac": inf["spmv_bytes"] / ms / 1e6 / 8000.0, "matches_scipy": ok}


def split_adapted_mesh(gm, frac, seed):
    """What a regrid leaves behind, at bench size and vectorised: `frac` of the triangles are split by a new interior vertex
    (3 new triangles each), every other triangle survives with its three vertices, the triangles are renumbered.  Returns
    x, y, tri (0-based), PreviousNumbering (1-based old vertex number, 0 = created by the remesher)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    tri = gm.tri
    ne, nn = tri.shape[0], gm.num_nodes
    pick = np.sort(rng.choice(ne, size=max(1, int(frac * ne)), replace=False))
    w = rng.dirichlet([2.0, 2.0, 2.0], pick.size)
    a, b, c = tri[pick, 0], tri[pick, 1], tri[pick, 2]
    xm = w[:, 0] * gm.x[a] + w[:, 1] * gm.x[b] + w[:, 2] * gm.x[c]
    ym = w[:, 0] * gm.y[a] + w[:, 1] * gm.y[b] + w[:, 2] * gm.y[c]
    m = nn + np.arange(pick.size)
    new = tri.copy()
    new[pick] = np.column_stack([a, b, m])
    new = np.vstack([new, np.column_stack([b, c, m]), np.column_stack([c, a, m])]).astype(np.int32)
    new = new[rng.permutation(new.shape[0])]
    prev = np.concatenate([np.arange(1, nn + 1), np.zeros(pick.size)]).astype(np.float64)
    return np.concatenate([gm.x, xm]), np.concatenate([gm.y, ym]), np.ascontiguousarray(new), prev


def aux_regrid(gm, with_cpu=True):
    """BASELINE config 5 at 2 km size: the two interpolation calls of a regrid (FE.cpp:3071-3154) -- the conservative remapping of 30 element
    variables onto the new mesh's triangles and the P1 interpolation of the 6 nodal variables onto its nodes -- through ONE regrid context
    (nxs_regrid_*: the old mesh's bucket grid and connectivity tables built once, on the device; bamg's convex completion once, on the host),
    (a) with the variables in host arrays, as the reference's root rank holds them, and (b) with the variables resident on the device, as
    nxs_dyn holds them; the real contrib/bamg routines (oracle/_ref, one host core, what the reference's root rank runs) timed beside
    them and the results compared bit for bit."""
    import ctypes as C
    import numpy as np
    from nextsim_amd import dynamics
    from nextsim_amd.interp import Regrid
    rng = np.random.default_rng(2)
    xn, yn, trin, prev = split_adapted_mesh(gm, 0.03, 9)
    idx_old = (gm.tri + 1).astype(np.int32).ravel()
    idx_new = np.ascontiguousarray(trin + 1, np.int32).ravel()   # (1-based, as bamg's tables; prepared outside the timed calls)
    def pinned(shape):   # page-locked host arrays, as a model keeps the fields it hands to the device again and again (None: no torch / no GPU)
        try:
            import torch
            return torch.empty(shape, dtype=torch.float64, pin_memory=True).numpy()
        except Exception:  # noqa: BLE001
            return None
    nodal = rng.standard_normal((gm.num_nodes, 6))
    elemental = rng.random((gm.num_elements, 30))
    L = dynamics.load_library()
    L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; L.hipFree.argtypes = [C.c_void_p]
    L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    keep_r = keep_i = None
    pinned_row = None
    for round_ in range(2):   # second round: warm allocator, result arrays kept
        t0 = time.perf_counter(); rg = Regrid(idx_old, gm.x, gm.y); w_ctx = time.perf_counter() - t0
        t0 = time.perf_counter(); vr, ir = rg.remap_elements(elemental, idx_new, xn, yn, prev, 0, return_info=True, out=keep_r); wr = time.perf_counter() - t0
        t0 = time.perf_counter(); vi, ii = rg.interp_nodes(nodal, xn, yn, False, 0.0, return_info=True, out=keep_i); wi = time.perf_counter() - t0
        if keep_r is None:   # the second round fills the arrays of the first (a model that regrids again and again keeps them: no page faults in the copy back)
            fresh = {"remap_call_ms": wr * 1e3, "interp_call_ms": wi * 1e3}
            first_r, first_i = vr.copy(), vi.copy()
        keep_r, keep_i = vr, vi
        if round_ == 0:   # ... and the last round with page-locked arrays on both sides (what PCIe itself takes)
            pe, pn_, pr_, pi_ = pinned(elemental.shape), pinned(nodal.shape), pinned(vr.shape), pinned(vi.shape)
            have_pinned = all(a is not None for a in (pe, pn_, pr_, pi_))
        if round_ == 1 and have_pinned:
            pe[...] = elemental; pn_[...] = nodal
            rg2 = Regrid(idx_old, gm.x, gm.y)
            t0 = time.perf_counter(); _, irp = rg2.remap_elements(pe, idx_new, xn, yn, prev, 0, return_info=True, out=pr_); wrp = time.perf_counter() - t0
            t0 = time.perf_counter(); _, iip = rg2.interp_nodes(pn_, xn, yn, False, 0.0, return_info=True, out=pi_); wip = time.perf_counter() - t0
            rg2.close()
            pinned_row = {"remap_call_ms": wrp * 1e3, "remap_breakdown_ms": irp["timing"], "interp_call_ms": wip * 1e3, "interp_breakdown_ms": iip["timing"],
                          "same_bits": bool(np.array_equal(pr_, vr, equal_nan=True) and np.array_equal(pi_, vi))}
        # (b) device-resident variables
        bufs = [C.c_void_p() for _ in range(4)]
        sizes = (elemental.nbytes, trin.shape[0] * 30 * 8, nodal.nbytes, xn.size * 6 * 8)
        for bq, sz in zip(bufs, sizes):
            assert L.hipMalloc(C.byref(bq), sz) == 0
        L.hipMemcpy(bufs[0], elemental.ctypes.data, elemental.nbytes, 1); L.hipMemcpy(bufs[2], nodal.ctypes.data, nodal.nbytes, 1)
        t0 = time.perf_counter(); _, ird = rg.remap_elements(None, idx_new, xn, yn, prev, 0, in_device=(bufs[0].value, 30), out_device=bufs[1].value, return_info=True); wrd = time.perf_counter() - t0
        t0 = time.perf_counter(); _, iid = rg.interp_nodes(None, xn, yn, False, 0.0, data_device=(bufs[2].value, gm.num_nodes, 6), out_device=bufs[3].value, return_info=True); wid = time.perf_counter() - t0
        back_r = np.empty_like(vr); back_i = np.empty_like(vi)
        L.hipMemcpy(back_r.ctypes.data, bufs[1], back_r.nbytes, 2); L.hipMemcpy(back_i.ctypes.data, bufs[3], back_i.nbytes, 2)
        for bq in bufs:
            L.hipFree(bq)
        rg.close()
    same_dev = bool(np.array_equal(back_r, vr, equal_nan=True) and np.array_equal(back_i, vi))
    fresh["same_bits"] = bool(np.array_equal(first_r, vr, equal_nan=True) and np.array_equal(first_i, vi))
    out = {"workload": f"regrid of the 2 km mesh: {gm.num_elements} old triangles -> {trin.shape[0]} new ({100 * (ir['visits'] == 1).mean():.1f} % overlap a single old triangle), "
                       f"conservative remap of 30 element variables + 6 nodal variables at {xn.size} nodes (isdefault=false as FE.cpp:3131), one regrid context for both",
           "context_create_ms": w_ctx * 1e3,
           "remap_kernel_ms": ir["kernel_ms"], "remap_call_ms": wr * 1e3, "remap_call_breakdown_ms": ir["timing"],
           "interp_kernel_ms": ii["kernel_ms"], "interp_call_ms": wi * 1e3, "interp_call_breakdown_ms": ii["timing"],
           "first_regrid_into_fresh_arrays": fresh, "page_locked_host_arrays": pinned_row,
           "device_resident_variables": {"remap_call_ms": wrd * 1e3, "remap_breakdown_ms": ird["timing"], "interp_call_ms": wid * 1e3, "interp_breakdown_ms": iid["timing"],
                                         "same_bits_as_host_arrays": same_dev},
           "remap_failed": int(ir["num_failed"]),
           "note": "call = what is left on the host (index checks, integer plane, the convex completion for isdefault=false) + device-built tables (first call on the context) + "
                   "PCIe both ways for host arrays + kernel.  remap_call_ms / interp_call_ms: a new context (tables built in its first call), the results into arrays the caller kept from the regrid before; "
                   "first_regrid_into_fresh_arrays: the same with new result arrays, whose page faults fall into the copy back; the reference runs both serially on its root rank"}
    if with_cpu:
        try:
            from oracle import pyoracle as O
            if O.bamg_shim() is None:
                raise RuntimeError("oracle/_ref not built on this box")
            t0 = time.perf_counter(); ref_i = O.bamg_interp_mesh_to_mesh(idx_old, gm.x, gm.y, nodal, xn, yn, False, 0.0); ci = time.perf_counter() - t0
            t0 = time.perf_counter(); ref_r = O.bamg_conservative_remap(gm.tri + 1, gm.x, gm.y, trin + 1, xn, yn, prev, 0, elemental); cr = time.perf_counter() - t0
            out["cpu_reference"] = {"kind": "reference", "what": "contrib/bamg InterpFromMeshToMesh2dx + ConservativeRemappingMeshToMesh compiled from the reference's sources, 1 host core",
                                    "interp_ms": ci * 1e3, "remap_ms": cr * 1e3,
                                    "interp_rows_identical": float(np.all(vi == ref_i, axis=1).mean()),
                                    "remap_rows_identical": float(np.all((vr == ref_r) | (np.isnan(vr) & np.isnan(ref_r)), axis=1).mean())}
        except Exception as e:  # noqa: BLE001
            out["cpu_reference"] = {"error": repr(e)}
    return out


def aux_partition_floor(args, local_rank, torch, S):
    """The compute floor of strong scaling, measured on THIS one GPU with the SEVERAL-RANK builds of the kernels: rank 0's partition of the 2 km mesh as
    `bench.py --gpus N` cuts it (N = 2, 4, 8: own nodes + one layer of ghosts), alone on the device, its mailboxes connected to themselves
    (dynamics.ipc_loopback: every wait of the exchange inside the kernels is satisfied by the rank's own stores -- no neighbour, no xGMI; the ghosts receive
    meaningless velocities, so only times are taken from these runs).  Three ways to run the sub-step loop: both exchanges of two sub-steps inside one launch
    (k_substep_pair<HALO>), one launch per sub-step with its exchange inside (k_substep_fused<HALO>), and the whole loop as ONE resident launch (fused = 4)
    where the partition fits one round of workgroups -- the candidates bench.py --gpus N times on the real machine."""
    from nextsim_amd import dynamics
    out = []
    for share in (2, 4, 8):
        gm, p, lm, f = build_case("2km", share, 0)
        row = {"share_of_2km_mesh": f"1/{share}", "elements": int(lm.num_elements), "own_nodes": int(lm.local_ndof), "ghost_nodes": int(lm.num_nodes - lm.local_ndof)}
        for name, opts in (("two_substeps_per_launch", {"pair_regs": 1, "fused": 3, "halo_fused": 1}), ("one_launch_per_substep", {"pair_regs": 0, "fused": 3, "halo_fused": 1}),
                           ("resident_one_launch_per_step", {"resident_wide": 1, "fused": 4, "halo_fused": 1}),
                           ("resident_without_the_release_fence", {"resident_wide": 1, "fused": 4, "halo_fused": 1, "resident_release": 0})):
            fe = dynamics.FiniteElementDynamics(p, device=local_rank)
            try:
                fe.set_mesh(lm)
                if not fe.ipc_loopback():
                    row[name] = {"error": "this partition's halo lists cannot be looped back"}
                    continue
                for k, v in opts.items():
                    fe.set_option(k, v)
                fe.set_option("prepare", 1)
                fe.put_state(f); fe.set_forcing(f)
                for _ in range(2):
                    fe.step()
                fe.synchronize(); fe.put_state(f); fe.set_option("timing_reset", 1)
                n = 10
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(n):
                    fe.step()
                fe.synchronize(); torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                tm, tr = fe.timing(), fe.traffic_model()
                if opts["fused"] == 4 and tm["substep_launches"] != 1:
                    row[name] = None   # the partition does not fit one round of resident workgroups
                    continue
                row[name] = {"ms_per_step": dt / n * 1e3, "value": gm.num_elements / share * S * n / dt, "unit": "element-updates/s of this rank's share", "kernel": tr["substep_kernel_name"],
                             "substeps_per_launch": tr["substeps_per_launch"], "substeps_ms": tm["substeps_ms"], "smoother_ms": tm["smoother_ms"], "prep_ms": tm["prep_ms"], "update_ms": tm["update_ms"]}
            except dynamics.NxsError as e:
                row[name] = {"error": str(e)[:300]}
            finally:
                fe.close()
        out.append(row)
    return {"workload": "rank 0's partition of the 2 km mesh for 2 / 4 / 8 ranks, the several-rank kernels with the mailboxes looped back (no neighbour, no xGMI): what one rank "
                        "computes per step -- the ceiling of strong scaling is (ms_per_step of the whole mesh) / (this)",
            "meshes": out}


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited: a GPU box shows all the host's CPUs in its affinity
    mask but may only burn a share of them -- threads beyond the quota do not add throughput, and spinning ones take it away from the working ones."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:  # noqa: BLE001
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:  # noqa: BLE001
        return None


def host_description():
    """(logical CPUs of the host, CPUs this process may run on, CPU model) -- SURVEY 8d: the baseline states its host."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return os.cpu_count() or usable, usable, model


# nominal memory bandwidth of hosts this has run on (GB/s, all sockets): the CPU leg's bytes/s is printed beside it
HOST_MEMORY_GBS = {"AMD EPYC 9575F 64-Core Processor": ("2 sockets x 12 channels DDR5-6000", 1152.0)}


def cpu_pressure():
    """/proc/pressure/cpu 'some avg10' (percent of the last ten seconds in which some task waited for a CPU), or None: how busy the shared host is beside this run."""
    try:
        for line in open("/proc/pressure/cpu"):
            if line.startswith("some"):
                return float(line.split("avg10=")[1].split()[0])
    except Exception:  # noqa: BLE001
        pass
    return None


def cpu_baseline(kind, nsteps=1, nsteps_threaded=2):
    """The restated reference CPU path (oracle/dyn_ref.c, -O3 -march=native) on the same mesh and forcing.
    (i) one core, the serial loops (nsteps steps); (ii) the cores this process may run on: one mesh partition per thread, all threads in lock-step inside the oracle
    library (ref_mr_*: the CPU analogue of the reference's MPI run), threads kept across the steps, every partition's arrays allocated and first touched by its own
    thread (NUMA-local, like the heap of an MPI rank).  What the host gives depends on how the threads wait and where they sit -- under a CPU quota (cgroup cpu.max: a
    GPU box shows all the host's CPUs and grants a share of their time) spinning threads burn the quota the working ones need, sleeping ones pay a futex per meeting --,
    so EVERY combination is timed on the same partitions: thread counts {2 x quota, quota} (every usable CPU and half of it where there is no quota) x barrier {spin,
    sleep} x {pinned, unpinned}, nsteps_threaded steps each after one untimed step per thread count; `value` is the fastest, the table goes into the line.
    kind 'port': the reference binary itself cannot be built without Boost/Gmsh/NetCDF."""
    from nextsim_amd import forcing as F, mesh as M
    from oracle import pyoracle as O
    gm, p, lm, f = build_case(kind, 1, 0)
    r = O.OracleRank(lm, p, f, fast=True)
    t0 = time.perf_counter()
    for _ in range(nsteps):
        r.step()
    dt1 = time.perf_counter() - t0
    single = gm.num_elements * p.substeps * nsteps / dt1
    del r
    host_cores, usable, model = host_description()
    out = {"value": single, "cores": 1, "seconds": dt1, "single_core_value": single, "single_seconds": dt1, "single_steps": nsteps,
           "host_cores": host_cores, "usable_cores": usable, "cpu_model": model, "threaded_steps": 0, "thread_counts_tried": {}, "table": [], "placement": None,
           "cpu_quota_cores": None, "cpu_pressure_some_avg10": {"before": cpu_pressure()}}
    if usable > 1:
        p2, C_fix, C_alea = F.scale_params_to_mesh(F.default_params(), gm, alea_factor=0.33)
        g = F.global_fields(gm, p2, "arctic", C_fix, C_alea)
        quota = cpu_quota()
        out["cpu_quota_cores"] = quota
        top = min(usable, 512) if not quota else max(2, min(usable, 512, int(2 * quota + 0.5)))
        counts = [c for c in (top, top // 2) if c >= 2]
        if quota and 4 * quota <= usable:
            counts.append(int(4 * quota + 0.5))   # (sleeping, unpinned only: waiting threads cost no quota, smaller partitions balance the lock-step better)
        best = None
        budget_t0 = time.perf_counter()
        for cores in counts:
            lms = M.localize(gm, cores)
            ranks = [O.OracleRank(l, p2, F.localize_fields(g, l, gm.num_nodes), fast=True) for l in lms]
            ctx = O.MultirankContext(ranks, nthreads=cores, pin=True)   # (every thread copies -- first touches -- its own partition here)
            ctx.run(1)                                                   # (untimed: the work arrays are touched by their threads)
            for barrier, pin in ((("sleep", False),) if (quota and cores > 2 * quota + 0.5) else (("sleep", False), ("sleep", True), ("spin", True), ("spin", False))):
                if time.perf_counter() - budget_t0 > 150.:
                    break
                ctx.configure(barrier, pin)
                info = ctx.info()
                t0 = time.perf_counter()
                ctx.run(nsteps_threaded)
                dtn = time.perf_counter() - t0
                v = gm.num_elements * p.substeps * nsteps_threaded / dtn
                out["table"].append({"threads": cores, "barrier": barrier, "pinned": pin, "value": v, "seconds": dtn})
                key = str(cores)
                out["thread_counts_tried"][key] = max(out["thread_counts_tried"].get(key, 0.), v)
                if best is None or v > best[0]:
                    best = (v, cores, dtn, info, barrier, pin)
            ctx.close(copy_back=False)
            del ranks, ctx
        out["threaded_steps"] = nsteps_threaded
        if best and best[0] > single:
            out.update(value=best[0], cores=best[1], seconds=best[2])
            info = best[3]
            out["placement"] = {"threads": info["threads"], "pinned": best[5], "sockets_used": info["sockets_used"] if best[5] else None,
                                "distinct_cpus": len(set(info["cpus"])) if best[5] else None, "first_touch": "every partition's arrays allocated and copied in by its own thread",
                                "barrier": {"spin": "sense-reversing spin barrier", "sleep": "pthread_barrier_t (the waiting threads sleep)"}[best[4]] + ", 340 meetings per step"}
    out["cpu_pressure_some_avg10"]["after"] = cpu_pressure()
    # SURVEY 8d's 172 B per element + 217 B per node per sub-step: what the figure means as a memory rate, beside the host's nominal bandwidth
    bytes_per_update = (BYTES_PER_ELEMENT * gm.num_elements + BYTES_PER_NODE * gm.num_nodes) / gm.num_elements
    out["model_GBps"] = out["value"] * bytes_per_update / 1e9
    hm = HOST_MEMORY_GBS.get(model)
    out["host_memory"] = {"what": hm[0], "nominal_GBps": hm[1], "fraction_used_by_the_model_bytes": out["model_GBps"] / hm[1]} if hm else None
    return out


def launch_ranks(nranks, worker_cmd, grace_s=20.0, poll_s=0.2):
    """`python bench.py --gpus N` called WITHOUT a launcher (no WORLD_SIZE in the environment): start the N ranks here, one child process per GPU, with
    the environment torch.distributed.run would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT on 127.0.0.1, a free port).  This
    process never imports torch and never touches the GPU, so the children are ordinary fork + exec of a process without a HIP context (an exec from a
    process that HAS initialised the GPU takes the machine down on this pool).  Rank 0's stdout -- the one JSON line -- is relayed to this process's
    stdout; the other ranks' stdout goes to stderr.  Returns the exit code: 0 when every rank returned 0; otherwise the first failing rank's code, after
    the remaining ranks were given `grace_s` seconds and then ended (by their own pids -- never by pattern)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks), GROUP_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NXS_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen(list(worker_cmd), env=env, stdin=subprocess.DEVNULL, stdout=(subprocess.PIPE if r == 0 else sys.stderr.fileno())))
    import threading
    relayed = []

    def relay():   # rank 0's stdout, line by line as it comes (a reader thread: a full pipe must never block rank 0)
        for line in procs[0].stdout:
            relayed.append(line)
            sys.stdout.buffer.write(line)
            sys.stdout.buffer.flush()
    t = threading.Thread(target=relay, daemon=True)
    t.start()
    failed, deadline = None, None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad and failed is None:
            failed, deadline = bad[0], time.monotonic() + grace_s
            print(f"[bench launcher] rank {failed[0]} exited with code {failed[1]}; the other ranks get {grace_s:.0f} s", file=sys.stderr, flush=True)
        if deadline is not None and time.monotonic() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            end = time.monotonic() + 5.0
            while time.monotonic() < end and any(p.poll() is None for p in procs):
                time.sleep(poll_s)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            deadline = None
        time.sleep(poll_s)
    t.join(timeout=10.0)
    codes = [p.returncode for p in procs]
    if failed is not None:
        return failed[1] if failed[1] > 0 else 1
    bad = [c for c in codes if c != 0]
    return (bad[0] if bad[0] > 0 else 1) if bad else 0


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # called like the single-GPU bench (`python bench.py --gpus N ...`): be the launcher, before anything of torch / HIP is in this process
        sys.exit(launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world

    # torch FIRST: it bundles its own libamdhip64.so.7 / libhsa-runtime64 and they must be the ones in
    # the process (loading /opt/rocm's runtime first and torch second leaves HIP without devices);
    # libnxsdyn.so then binds to the runtime that is already loaded.
    import torch
    import torch.distributed as dist
    from nextsim_amd import dynamics
    dynamics.load_library()

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the dynamics path has no CPU fallback")
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    local_rank = local_rank % max(torch.cuda.device_count(), 1)  # several ranks may share a GPU on a test box

    unique_id_fn = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane (barrier, max-reduce, handle / id exchange at set-up) on gloo; the data path (halo exchange of M_VT)
        # is inside libnxsdyn.so: device-direct peer mailboxes over xGMI, RCCL or host-staged as fallbacks
        # gloo announces itself on stdout from C++; keep stdout for the one JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

        def unique_id_fn():   # collective; None when rank 0 could not make one (every rank then learns it, nobody waits in a broadcast)
            ids = [None]
            if rank == 0:
                try:
                    ids = [dynamics.FiniteElementDynamics.comm_unique_id()]
                except Exception as e:  # noqa: BLE001
                    print(f"[bench rank 0] ncclGetUniqueId: {e}", file=sys.stderr, flush=True)
            dist.broadcast_object_list(ids, src=0)
            return ids[0]

    # which physical devices the ranks really run on (a test box may wrap several ranks onto one GPU: its numbers then bound the
    # protocol overhead, they are not a scaling measurement)
    devices = [local_rank]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, (os.environ.get("GROUP_RANK", "0"), local_rank))
    distinct_devices = len(set(devices))
    res = run_gpu(args.mesh, args, rank, world, local_rank, dist, torch, unique_id_fn, own_device=distinct_devices == world)
    gm, p, lm, tm = res["gm"], res["p"], res["lm"], res["timing"]
    S = p.substeps
    value = gm.num_elements * S * args.steps / res["dt"]

    # ---- roofline of the dominant kernel (the sub-step loop) and of the once-per-step kernels.  Everything is PER LAUNCH: the bytes the launch must move,
    # from the patch tables it walks (nxs_dyn_get_traffic_model), over the HIP-event time of the launch on the library's stream.
    tr = res["traffic"]
    kernel_name = tr["substep_kernel_name"]
    D_launch = max(int(tr["substeps_per_launch"]), 1)
    n_launches = max(S // D_launch, 1)                                   # launches of the dominant kernel per step
    loop_ms = tm["substeps_ms"] - tm.get("ring_flush_ms", 0.0)           # the sub-step launches alone (the deferred mesh move is timed apart)
    launch_ms = loop_ms / n_launches
    def rate(nbytes, ms):
        return nbytes / (ms * 1e-3) / 1e9 if (nbytes and ms and ms > 0) else None
    def frac(r):
        return r / HBM_PEAK_GBS if r is not None else None
    # the roofline's numerator is the launch's UNIQUE bytes -- every array entry it touches, once: the floor of its HBM traffic whatever the caches do.  (Until the
    # middle of round 4 it was the scheme's bytes -- every list a workgroup reads, rings included; since the element constants stay in registers the COUNTED traffic of
    # k_substep_pair, 361 MB, is below those 372 MB: the L2 serves part of the rings neighbouring patches share, so the scheme's bytes bound what the workgroups pull
    # through the L2, not what crosses to memory.  unique <= counted always holds.)
    achieved = rate(tr["substep_unique_bytes"], launch_ms)
    want = [kernel_name.split(" ")[0]] + (["k_move_ring"] if tr["move_ring_slots"] else []) + {3: ["k_prep_fused"], 2: ["k_prep_elements", "k_prep_nodes"], 1: ["k_prep_elements", "k_prep_nodes"]}.get(tr["prep_kernel"], []) + ["k_update"]
    if args.no_live_pmc or rank != 0:
        counted, traffic_source = {}, "skipped (--no-live-pmc)" if args.no_live_pmc else "rank 0 measures"
    else:
        counted, traffic_source = pmc_traffic_live(args.mesh, world, rank, want, options=[f"{k}={v}" for k, v in res["options"].items()])
    traffic = counted.get(want[0])
    traffic_live = traffic is not None
    traffic_profile, profile_file = pmc_traffic(args.mesh, world, want[0])
    if not traffic_live and traffic_profile:   # replay of the committed profile, labelled as such
        traffic, traffic_source = traffic_profile, (profile_file + " -- rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; REPLAYED here, not "
                                                    "measured in this run (" + str(traffic_source) + ")")
    achieved_counter = rate(traffic, launch_ms)
    def kernel_row(name, scheme, unique, ms, what):
        c = counted.get(name)
        return {"what": what, "bytes_per_launch": unique, "scheme_bytes_per_launch": scheme, "avg_ms_per_launch": ms, "achieved": rate(unique, ms), "frac": frac(rate(unique, ms)),
                "frac_scheme": frac(rate(scheme, ms)), "traffic": c, "achieved_counter": rate(c, ms), "frac_counter": frac(rate(c, ms))}
    other = {}
    if tr["move_ring_slots"]:
        other["k_move_ring"] = kernel_row("k_move_ring", tr["move_ring_bytes"], tr["move_ring_bytes"], tm.get("ring_flush_ms", 0.0),
                                          f"the deferred mesh move M_UM / M_UT += dt * M_VT (FE.cpp:10543-10550) of {tr['move_ring_slots']} sub-steps from the velocity ring, once per step")
    if tr["prep_kernel"] == 3:
        other["k_prep_fused"] = kernel_row("k_prep_fused", tr["prep_scheme_bytes"], tr["prep_unique_bytes"], tm["prep_ms"], "prep elements + prep nodes (FE.cpp:10235-10416), one launch per step")
    elif tr["prep_kernel"] in (1, 2):
        c = (counted.get("k_prep_elements") or 0) + (counted.get("k_prep_nodes") or 0)
        other["k_prep_elements + k_prep_nodes"] = kernel_row("", tr["prep_scheme_bytes"], tr["prep_unique_bytes"], tm["prep_ms"], "prep elements, prep nodes (FE.cpp:10235-10416), two launches per step")
        if c:
            other["k_prep_elements + k_prep_nodes"].update(traffic=c, achieved_counter=rate(c, tm["prep_ms"]), frac_counter=frac(rate(c, tm["prep_ms"])))
    other["k_update"] = kernel_row("k_update", tr["update_bytes"], tr["update_bytes"], tm["update_ms"], "update() (FE.cpp:3946-4131), one launch per step")
    sm = sorted(float(x) for x in res["step_ms"])
    step_stats = None
    if sm:
        step_stats = {"n": len(sm), "median": sm[len(sm) // 2] if len(sm) % 2 else 0.5 * (sm[len(sm) // 2 - 1] + sm[len(sm) // 2]), "mean": sum(sm) / len(sm), "min": sm[0], "max": sm[-1],
                      "what": "device time of every timed step of rank 0 (HIP events on the library's stream around each nxs_dyn_step; the steps stay asynchronous)"}
    out = {
        "metric": "element-updates/sec per dynamics step",
        "value": value,
        "unit": "element-updates/s",
        "n_gpus": world,
        "distinct_devices": distinct_devices,
        "ranks_share_device": distinct_devices < world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": res["dt"] / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"pan-Arctic-like synthetic mesh '{args.mesh}' ({gm.num_elements} triangles, {gm.num_nodes} nodes), "
                        f"BBM rheology, dt=200 s, {S} sub-steps, 50 smoother sweeps, update(); "
                        f"domain-decomposed into {world} partition(s) on {distinct_devices} GPU(s)",
            "mesh": args.mesh, "elements": gm.num_elements, "nodes": gm.num_nodes, "substeps": S,
            "rheology": "bbm", "partitions": world, "halo_transport": res["transport"], "halo": res["halo"], **ice_cover(res["f"]),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": {"k_substep_resident": "k_substep_resident (ONE launch per step: the fused sub-step loop, patches waiting for their neighbours only)",
                       "k_substep_resident_big": "k_substep_resident_big (ONE launch per step, one large patch per CU: four elements and two own nodes per thread)",
                       "k_substep_pair": "k_substep_pair (TWO sub-steps per launch on patches with two rings of halo, the stresses between them in registers: stress/damage + assembly + nodal solve, twice" + ("" if tr["move_ring_slots"] else " + the mesh move of both") + ")",
                       "k_substep_flow": "k_substep_flow (ONE data-flow launch per step over k_substep_pair's patches: option pair_flow)",
                       "k_substep_multi": f"k_substep_multi ({D_launch} sub-steps per launch on patches with that many rings of halo)",
                       "k_substep_fused": "k_substep_fused (one launch per sub-step: stress/damage + assembly + nodal solve" + (" + updateGhosts through the peer mailboxes" if tr["halo_in_kernel"] else "") + ")",
                       }.get(kernel_name, kernel_name),
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": frac(achieved),
            "traffic": traffic,
            "traffic_source": traffic_source,
            "traffic_measured_in_this_run": traffic_live,
            "achieved_counter": achieved_counter,
            "frac_counter": frac(achieved_counter),
            "substeps_per_launch": D_launch,
            "launches_per_step": n_launches,
            "bytes_per_launch": tr["substep_unique_bytes"],
            "scheme_bytes_per_launch": tr["substep_scheme_bytes"],
            "frac_scheme": frac(rate(tr["substep_scheme_bytes"], launch_ms)),
            "reread_bytes_per_launch": tr["substep_reread_bytes"],
            "avg_ms_per_launch": launch_ms,
            "algorithmic_equivalent": {"bytes_per_launch": tr["survey_model_bytes"], "achieved": rate(tr["survey_model_bytes"], launch_ms), "frac": frac(rate(tr["survey_model_bytes"], launch_ms)),
                                       "note": "SURVEY 8d's model, 172 B per element + 217 B per node per sub-step x the sub-steps one launch advances: what a kernel that streamed every "
                                               "array of the reference loop once per sub-step would move.  NOT a bandwidth of this kernel (shape coefficients rebuilt on chip, M_UM / M_UT "
                                               "once per step, state and records once per TWO sub-steps): it may exceed the peak; kept for comparison with rounds 1-3"},
            "other_kernels": other,
            "note": "rank-0 partition, per launch.  bytes_per_launch (-> achieved, frac) = the launch's UNIQUE bytes: every array entry it touches, once, from the patch tables "
                    "(nxs_dyn_get_traffic_model, include/nxs_dyn.h) -- the floor of its HBM traffic whatever the caches do, so frac <= frac_counter <= ~0.79 (a copy reaches ~6.3 TB/s).  "
                    "traffic (-> achieved_counter, frac_counter) = counted bytes (2 x FETCH_SIZE + WRITE_SIZE) of the same launch.  scheme_bytes_per_launch (frac_scheme) = every list "
                    "a workgroup reads, once, plus what it writes, summed over the workgroups (the halo rings of the blocking counted): what the workgroups pull through the L2 -- the "
                    "counted traffic can be BELOW it where the L2 serves rings that neighbouring patches share; reread_bytes_per_launch = a workgroup's second reads on top of that.  "
                    "avg_ms_per_launch = HIP events around the sub-step graph on the library's stream / launches (where the mesh move is deferred to one k_move_ring per step -- several "
                    "ranks, one launch per sub-step -- that flush is timed apart: other_kernels.k_move_ring; k_substep_pair on a single rank moves the mesh inside its launches)",
        },
        "step_times_ms": step_stats,
        "value_at_median_step": (gm.num_elements * S / (step_stats["median"] * 1e-3)) if (step_stats and world == 1) else None,
        "phases_ms": res["phases_max"],   # (N > 1: the slowest rank's figure for every phase)
        "phases_ms_rank0": {k: tm[k] for k in ("prep_ms", "substeps_ms", "smoother_ms", "update_ms", "total_ms")},
        "fields_ok": res["crash"] == 0,
    }
    if world > 1:   # north_star's named transport beside the kept one (collective: every rank takes part, rank 0 reports)
        try:
            if res["halo"].get("transport") == "rccl":
                out["aux_rccl"] = {"status": "is the transport of the main line", "ms_per_step": out["ms_per_step"]}
            elif os.environ.get("NXS_BENCH_SKIP_RCCL") == "1":
                out["aux_rccl"] = {"status": "skipped (NXS_BENCH_SKIP_RCCL=1)"}
            else:
                # librccl announces itself ("RCCL version : ...") on stdout from C: keep stdout for the one JSON line
                sys.stdout.flush()
                saved = os.dup(1)
                os.dup2(2, 1)
                try:
                    out["aux_rccl"] = aux_rccl(args.mesh, args, rank, world, local_rank, dist, torch, unique_id_fn, out["ms_per_step"])
                finally:
                    sys.stdout.flush()
                    os.dup2(saved, 1)
                    os.close(saved)
        except Exception as e:  # noqa: BLE001 -- never lose the main line over it
            out["aux_rccl"] = {"status": "failed", "error": repr(e)[:300]}
    if rank == 0 and not args.no_cpu_baseline:   # (at every N: rank 0's host is the same host)
        try:
            cb = cpu_baseline(args.mesh, 1, 2)
            out["cpu_baseline"] = {
                "value": cb["value"], "unit": "element-updates/s", "cores": cb["cores"], "kind": "port",
                "single_core_value": cb["single_core_value"],
                "host_cores": cb["host_cores"], "usable_cores": cb["usable_cores"], "cpu_model": cb["cpu_model"],
                "sample": f"full dynamics steps ({S} sub-steps each) of the same '{args.mesh}' mesh and forcing, oracle/dyn_ref.c -O3 -march=native: "
                          f"{cb['single_steps']} step on 1 core ({cb['single_seconds']:.1f} s, single_core_value); {cb['threaded_steps']} steps (after one untimed per thread count) with "
                          f"one mesh partition per thread in lock-step (threads kept, partitions first touched by their threads, shared-memory halo exchange), timed in EVERY "
                          f"combination of thread count {list(cb['thread_counts_tried'])} x barrier (spin / sleeping) x (pinned / unpinned) -- `table` -- on a host where this process "
                          f"may run on {cb['usable_cores']} of {cb['host_cores']} logical CPUs ({cb['cpu_model']}) within a quota of {cb.get('cpu_quota_cores')} CPUs' time: "
                          f"the fastest, {cb['cores']} thread(s), is `value` ({cb['seconds']:.1f} s)",
                "table": cb["table"], "cpu_pressure_some_avg10": cb["cpu_pressure_some_avg10"],
                "thread_counts_tried": cb["thread_counts_tried"], "placement": cb["placement"], "cpu_quota_cores": cb.get("cpu_quota_cores"),
                "model_GBps": cb["model_GBps"], "host_memory": cb["host_memory"],
            }
        except Exception as e:  # noqa: BLE001 -- the GPU line must survive a host-side failure of the baseline leg
            out["cpu_baseline"] = {"value": None, "unit": "element-updates/s", "cores": 0, "kind": "port", "sample": "failed: " + repr(e)}
    if world == 1 and not args.no_aux and args.mesh != "10km":
        # (a 0.55 ms step is enqueued by the host as fast as the device runs it: 20 steps measure the host's mood and the device's clocks after half a minute of idling -- 0.55 to 0.80 ms in this round's records)
        aux_args = argparse.Namespace(**vars(args)); aux_args.steps = max(args.steps, 400); aux_args.warmup = max(args.warmup, 200)   # (the leg follows the CPU baseline: the device has idled for half a minute)
        try:
            r2 = run_gpu("10km", aux_args, 0, 1, local_rank, dist, torch, None)
            out["aux_10km"] = {
                "workload": f"mesh '10km' ({r2['gm'].num_elements} triangles): BASELINE config 2, latency-bound (working set in cache)",
                "value": r2["gm"].num_elements * S * aux_args.steps / r2["dt"], "unit": "element-updates/s",
                "ms_per_step": r2["dt"] / aux_args.steps * 1e3,
            }
        except Exception as e:  # noqa: BLE001 -- never lose the main line over a secondary measurement
            out["aux_10km"] = {"error": repr(e)}
    if world == 1 and not args.no_aux and args.mesh == "2km":
        aux_args = argparse.Namespace(**vars(args)); aux_args.steps = min(args.steps, 10); aux_args.warmup = min(args.warmup, 2)
        try:   # the same mesh with 29 % of the triangles ice free and 4 % in the 0 < A <= 0.1 band: the open-water smoother (FE.cpp:10578-10611),
            # its tail and the skipped-element path do real work here
            r3 = run_gpu("2km", aux_args, 0, 1, local_rank, dist, torch, None, state="arctic_ow")
            out["aux_openwater"] = {
                "workload": "mesh '2km', state 'arctic_ow' (half the rim ice free): the 50 smoother sweeps, the open-water mesh move and the "
                            "concentration cut-off of updateSigmaDamage are exercised", **ice_cover(r3["f"]),
                "value": r3["gm"].num_elements * S * aux_args.steps / r3["dt"], "unit": "element-updates/s",
                "ms_per_step": r3["dt"] / aux_args.steps * 1e3, "fields_ok": r3["crash"] == 0,
                "phases_ms": {k: r3["timing"][k] for k in ("prep_ms", "substeps_ms", "smoother_ms", "update_ms", "total_ms")},
            }
        except Exception as e:  # noqa: BLE001
            out["aux_openwater"] = {"error": repr(e)}
        try:   # what a rank of 4 / 8 would compute per step (strong-scaling ceiling, measured on this GPU)
            out["aux_partition_floor"] = aux_partition_floor(args, local_rank, torch, S)
        except Exception as e:  # noqa: BLE001
            out["aux_partition_floor"] = {"error": repr(e)}
        try:   # BASELINE config 5: the regrid interpolation kernels at 2 km size, the real bamg routines beside them
            out["aux_regrid"] = aux_regrid(res["gm"], with_cpu=not args.no_cpu_baseline)
        except Exception as e:  # noqa: BLE001
            out["aux_regrid"] = {"error": repr(e)}
        try:   # the N4 extension (no live reference): SpMV of the 2-dof block pattern a momentum matrix would have on this mesh
            out["aux_spmv"] = aux_spmv(res["gm"])
        except Exception as e:  # noqa: BLE001 -- never lose the main line over the extension
            out["aux_spmv"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
