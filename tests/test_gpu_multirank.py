"""The multi-rank GPU path (partitioned mesh, k_halo_pack -> RCCL grouped send/recv -> k_halo_unpack,
FE.cpp:13963-13996) against the in-process multi-rank oracle.

A one-GPU box has to place every rank on GPU 0; RCCL may refuse several ranks on one device, in which
case the test is skipped with RCCL's message (the 8-GPU scaling run at round end is then the first
execution of the RCCL path on distinct devices)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _start(world, kind, nsteps, tmp_path, transport="rccl", over=None, ranks_per_proc=1):
    """world ranks as world / ranks_per_proc processes (a GPU box admits at most 6 processes on its card)."""
    assert world % ranks_per_proc == 0 and world // ranks_per_proc <= 6
    nproc = world // ranks_per_proc
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(nproc):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(nproc), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NXS_RANKS_PER_PROC=str(ranks_per_proc))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "mr_worker.py"), str(tmp_path), kind, str(nsteps), transport,
                                       json.dumps(over or {})], env=env))
    return procs


def _finish(procs, world, tmp_path, timeout=300):
    for p in procs:
        try:
            p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("multi-rank workers hung")
    return [json.load(open(tmp_path / f"report{r}.json")) for r in range(world)]


def _run(world, kind, nsteps, tmp_path, transport="rccl", over=None, ranks_per_proc=1):
    return _finish(_start(world, kind, nsteps, tmp_path, transport, over, ranks_per_proc), world, tmp_path)


def _against_the_multirank_oracle(kind, world, nsteps, tmp_path, reps, tol):
    """The parent's half of a dump-mode run: the in-process multi-rank oracle, once, against every rank's dumped state."""
    import numpy as np
    import cases
    from oracle import pyoracle as O
    gm, p, g, lms, fields = cases.make_case(kind, nparts=world)
    ranks = [O.OracleRank(lm, p, f) for lm, f in zip(lms, fields)]
    for _ in range(nsteps):
        O.multirank_step(ranks)
    worst = {}
    for r in reps:
        assert r["ok"], r
        assert r["crash"] == 0
        z = np.load(tmp_path / f"state{r['rank']}.npz")
        for k in z.files:
            e = cases.rel_err(z[k], ranks[r["rank"]].arr[k])
            worst[k] = max(worst.get(k, 0.), e)
            assert e <= tol, (r["rank"], k, e)
    return worst


@pytest.mark.parametrize("world", [2, 4])
def test_partitioned_gpu_run_host_staged_halo_matches_multirank_oracle(world, tmp_path):
    """Ranks share GPU 0; updateGhosts goes device -> pinned host -> gloo -> device through
    nxs_dyn_set_halo_exchange_fn.  Exercises everything of the multi-rank path except the RCCL calls:
    owned/ghost split of k_solve_move, k_halo_pack/unpack, ghost-node mesh move, smoother with halos."""
    reps = _run(world, "small", 2, tmp_path, "host")
    for r in reps:
        assert r["ok"], r
        assert r["crash"] == 0
        for k, e in r["errs"].items():
            assert e <= 1e-10, (r["rank"], k, e)


@pytest.mark.parametrize("world,kind,nsteps", [(2, "small", 2), (4, "small", 1), (3, "toy", 2)])
def test_partitioned_gpu_run_device_direct_halo_matches_multirank_oracle(world, kind, nsteps, tmp_path):
    """Ranks share GPU 0; updateGhosts through peer-mapped mailboxes (hipIpc).  Default: the exchange runs INSIDE
    the fused sub-step kernel (boundary patches wait for the neighbours' flags, stage ghosts from the mailbox, store
    their sent nodes into the neighbours' mailboxes; one launch per sub-step); it must reproduce, bit for bit, the
    run with separate kernels (k_halo_push stores and raises flags, k_halo_pull waits and unpacks), and both must
    match the multi-rank oracle.  The whole sub-step loop replays from a hipGraph.  On one GPU this validates the protocol (lists, sequence numbers, double buffering,
    graph replay); xGMI coherence itself is probed at run time by the transport's self-test."""
    reps = _run(world, kind, nsteps, tmp_path, "ipc")
    for r in reps:
        assert r["ok"], r
        assert r.get("ipc_selftest") is True
        assert r["fused_equals_separate"] is True, r
        assert r["launches_fused"] < r["launches_separate"]
        assert r["crash"] == 0
        for k, e in r["errs"].items():
            assert e <= 1e-10, (r["rank"], k, e)


@pytest.mark.parametrize("world,seed", [(3, 1), (4, 2)])
def test_device_direct_halo_on_a_ragged_partition(world, seed, tmp_path):
    """A random mosaic partition: every rank neighbours every other, disconnected pieces, ghost elements without owned
    nodes (orphan patches), nodes shared by 3+ ranks.  In-kernel exchange == separate kernels == multi-rank oracle."""
    reps = _run(world, "small", 1, tmp_path, "ipc", over={"ragged_seed": seed})
    for r in reps:
        assert r["ok"], r
        assert r["fused_equals_separate"] is True, r
        assert r["crash"] == 0
        for k, e in r["errs"].items():
            assert e <= 1e-10, (r["rank"], k, e)


@pytest.mark.parametrize("world,kind,rpp,options", [(3, "small", 1, {}), (3, "small", 1, {"smooth_persist": 0}), (4, "40km", 2, {"fused": 4}), (2, "40km", 1, {"pair_regs": 1})])
def test_open_water_across_the_partition_boundaries(world, kind, rpp, options, tmp_path):
    """State 'arctic_ow' (29 % of the triangles ice free, half the rim): ice-free nodes on both sides of the partition boundaries, so the 50 smoother sweeps
    exchange values that CHANGE from sweep to sweep (dynamic directions) -- in one launch of persistent workgroups (k_smooth_persist, the default), in 50 launches
    of k_smooth_halo, under the resident loop and under two sub-steps per launch: bitwise the separate kernels, 1e-10 of the multi-rank oracle, two steps."""
    reps = _run(world, kind, 2, tmp_path, "ipc", over={"forcing_kind": "arctic_ow", "options": options}, ranks_per_proc=rpp)
    for r in reps:
        assert r["ok"], r
        assert r["fused_equals_separate"] is True and r["crash"] == 0, r
        for k, e in r["errs"].items():
            assert e <= 1e-10, (r["rank"], k, e)


@pytest.mark.parametrize("dyn", ["mevp", "evp"])
def test_in_kernel_halo_exchange_with_the_vp_rheologies(dyn, tmp_path):
    """mEVP moves the mesh once after the sub-step loop (no ring of velocity buffers: the ghosts are copied through
    to the two ping-pong buffers and pulled once at the end); EVP moves every sub-step like BBM."""
    from nextsim_amd import _abi
    code = {"mevp": _abi.NXS_DYN_MEVP, "evp": _abi.NXS_DYN_EVP}[dyn]
    reps = _run(3, "small", 1, tmp_path, "ipc", over={"dynamics_type": code})
    for r in reps:
        assert r["ok"], r
        assert r["fused_equals_separate"] is True, r
        assert r["crash"] == 0
        for k, e in r["errs"].items():
            assert e <= 1e-9, (r["rank"], k, e)


def test_rccl_on_one_rank_runs_the_dlopened_entry_points(tmp_path):
    """RCCL refuses two ranks on one device, but a communicator of ONE rank is legal: ncclGetUniqueId, ncclCommInitRank with
    the by-value 128-byte id, and a grouped ncclSend/ncclRecv of coded payloads to itself on the handle's stream
    (nxs_dyn_comm_selftest) -- the dlopen, the symbol table, the stream use and the error mapping of the RCCL transport
    execute on hardware, then a step runs with the communicator attached (FE.cpp:13963-13996 is the code it stands for)."""
    reps = _run(1, "small", 1, tmp_path, "rccl")
    r = reps[0]
    assert r["ok"], r
    assert r["comm_selftest_errors"] == 0
    for k, e in r["errs"].items():
        assert e <= 1e-10, (k, e)


@pytest.mark.parametrize("transport,ranks_per_proc", [("ipc", 2), ("host", 4)])
def test_config3_10km_over_8_ranks_matches_the_multirank_oracle(transport, ranks_per_proc, tmp_path):
    """BASELINE config 3 (10 km mesh domain-decomposed over 8 ranks), ranks sharing GPU 0, one step against the 8-rank oracle.
    "ipc": 4 processes x 2 ranks -- mailboxes reached through hipIpc across processes and directly inside a process, the
    exchange INSIDE the sub-step and smoother kernels (8 small grids co-schedule on one device), which must also give the bits
    of the separate push / pull kernels.  "host": 2 processes x 4 ranks through the caller's communicator."""
    procs = _start(8, "10km", 1, tmp_path, transport, over={"dump": True}, ranks_per_proc=ranks_per_proc)
    reps = _finish(procs, 8, tmp_path)
    if transport == "ipc":
        for r in reps:
            assert r["ok"], r
            assert r["ipc_selftest"] is True and r["fused_equals_separate"] is True, r
            assert r["launches_fused"] < r["launches_separate"]
    _against_the_multirank_oracle("10km", 8, 1, tmp_path, reps, 1e-10)


def test_config4_2km_over_8_ranks_matches_the_multirank_oracle(tmp_path):
    """BASELINE config 4 (2 km mesh, BBM, 120 sub-steps, 8 ranks of ~183 k triangles), ranks sharing GPU 0 as 4 processes x 2
    ranks, one step against the 8-rank oracle.  Separate push / pull kernels: eight grids that each fill the device cannot
    co-schedule, so the in-kernel exchange would starve on ONE device; on eight devices it is the default."""
    procs = _start(8, "2km", 1, tmp_path, "ipc_sep", over={"dump": True}, ranks_per_proc=2)
    import numpy as np, cases
    from oracle import pyoracle as O
    gm, p, g, lms, fields = cases.make_case("2km", nparts=8)       # the oracle runs beside the workers
    ranks = [O.OracleRank(lm, p, f) for lm, f in zip(lms, fields)]
    O.multirank_step(ranks)
    reps = _finish(procs, 8, tmp_path, timeout=600)
    for r in reps:
        assert r["ok"], r
        assert r["ipc_selftest"] is True and r["crash"] == 0
        z = np.load(tmp_path / f"state{r['rank']}.npz")
        for k in z.files:
            assert cases.rel_err(z[k], ranks[r["rank"]].arr[k]) <= 1e-10, (r["rank"], k)
    # the partition is the one the bench strong-scales: ~1/8 of the triangles each, 2-5 neighbours, ~1.2 k shared nodes per neighbour
    assert max(lm.local_nelements for lm in lms) < 1.05 * gm.num_elements / 8
    assert all(2 <= len(lm.send_procs) <= 7 for lm in lms)


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_gpu_run_matches_multirank_oracle(world, tmp_path):
    reps = _run(world, "small", 1, tmp_path)
    if any("comm_error" in r for r in reps):
        pytest.skip("RCCL refused %d ranks on one GPU: %s" % (world, [r.get("comm_error") for r in reps][0]))
    for r in reps:
        assert r["ok"], r
        assert r["crash"] == 0
        for k, e in r["errs"].items():
            assert e <= 1e-10, (r["rank"], k, e)


def test_ipc_connect_checks_its_tables_against_what_the_neighbour_published():
    """Two ranks of a partition as two handles of THIS process (a host driving several GPUs from one process): the mailboxes
    connect through plain pointers, and nxs_dyn_ipc_connect refuses tables that do not match the blob the neighbour exported
    -- a wrong offset would make the kernels store outside the neighbour's mailbox -- and blobs it did not make."""
    import ctypes as C
    import numpy as np
    import cases
    from nextsim_amd import _abi, dynamics
    gm, p, g, lms, fields = cases.make_case("small", nparts=2)
    fes = [dynamics.FiniteElementDynamics(p) for _ in range(2)]
    blobs = []
    for fe, lm in zip(fes, lms):
        fe.set_mesh(lm)
        b = C.create_string_buffer(dynamics.IPC_BLOB_BYTES)
        fe._chk(fe.L.nxs_dyn_ipc_export(fe.h, b))
        blobs.append(b.raw)

    def connect(r, off=None, tot=None, slot=None, blob=None):
        lm, other = lms[r], lms[1 - r]
        k = other.recv_procs.tolist().index(r)
        a = [np.asarray([v], np.int32) for v in (other.recv_offsets[k] if off is None else off,
                                                  other.recv_offsets[-1] if tot is None else tot, k if slot is None else slot)]
        buf = C.create_string_buffer(blobs[1 - r] if blob is None else blob, dynamics.IPC_BLOB_BYTES)
        return fes[r].L.nxs_dyn_ipc_connect(fes[r].h, buf, _abi.iptr(a[0]), _abi.iptr(a[1]), _abi.iptr(a[2]))

    err = lambda r: (fes[r].L.nxs_dyn_last_error(fes[r].h) or b"").decode()  # noqa: E731
    assert connect(0, tot=int(lms[1].recv_offsets[-1]) + 1) != 0 and "receives" in err(0)
    assert connect(0, off=int(lms[1].recv_offsets[-1])) != 0 and "does not fit" in err(0)
    assert connect(0, slot=7) != 0 and "flag slot" in err(0)
    assert connect(0, blob=bytes(dynamics.IPC_BLOB_BYTES)) != 0 and "not made by nxs_dyn_ipc_export" in err(0)
    assert connect(0) == 0 and connect(1) == 0
    assert connect(0) == 0                                   # a second connect replaces the first
    # the connected pair works: the self-test pushes both publishing protocols through the (in-process) links
    import threading
    errs = [C.c_int32(-1), C.c_int32(-1)]
    th = [threading.Thread(target=lambda r=r: fes[r]._chk(fes[r].L.nxs_dyn_ipc_selftest(fes[r].h, 16, C.byref(errs[r])))) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert errs[0].value == 0 and errs[1].value == 0
    for fe in fes:
        fe.close()


def test_one_directional_neighbours_are_padded_behind_the_abi_and_the_low_level_connect_says_so():
    """Two buffers per link are safe only with a hand-shake: a rank that sends to q without receiving from q could run two exchanges ahead of q and overwrite the
    half q still reads (round 4; deterministic in tests/test_gpu_protocol_delays.py).  The 4-rank partition of 'small' has such a link (rank 0 -> rank 3), and its
    halo lists go in VERBATIM: nxs_dyn_set_halo adds the missing direction as an empty segment behind the caller's neighbours, the record form connects all four
    ranks (handles of ONE process here, mailboxes through plain pointers); the low-level nxs_dyn_ipc_connect, whose tables know only the caller's own neighbours,
    refuses with the rank named and points at the record form."""
    import ctypes as C
    import numpy as np
    import cases
    from nextsim_amd import _abi, dynamics
    gm, p, g, lms, fields = cases.make_case("small", nparts=4)
    assert 3 in lms[0].send_procs.tolist() and 3 not in lms[0].recv_procs.tolist()
    fes = [dynamics.FiniteElementDynamics(p) for _ in range(4)]
    for fe, lm in zip(fes, lms):
        fe.set_mesh(lm)
    # low level: refused on the two ranks of the one-directional link, before any table is read
    for r, other in ((0, 3), (3, 0)):
        fe, lm = fes[r], lms[r]
        b = C.create_string_buffer(dynamics.IPC_BLOB_BYTES)
        fe._chk(fe.L.nxs_dyn_ipc_export(fe.h, b))
        ns = max(lm.send_procs.size, 1)
        z = np.zeros(ns, np.int32)
        rc = fe.L.nxs_dyn_ipc_connect(fe.h, C.create_string_buffer(ns * dynamics.IPC_BLOB_BYTES), _abi.iptr(z), _abi.iptr(z), _abi.iptr(z))
        msg = (fe.L.nxs_dyn_last_error(fe.h) or b"").decode()
        assert rc != 0 and "one direction only" in msg and f"rank {other}" in msg and "nxs_dyn_ipc_connect_records" in msg, (rc, msg)
    # the record form: sizes, records, connect -- what examples/nextsim_mpi.cpp does with MPI_Allreduce / MPI_Allgather
    sizes = []
    for fe in fes:
        n = C.c_int32()
        fe._chk(fe.L.nxs_dyn_ipc_record_bytes(fe.h, C.byref(n)))
        sizes.append(n.value)
    stride = max(sizes)
    recs = b""
    for fe, need in zip(fes, sizes):
        buf = C.create_string_buffer(stride)
        assert fe.L.nxs_dyn_ipc_export_record(fe.h, buf, need - 1) != 0          # a record that does not fit is refused, nothing is exported
        fe._chk(fe.L.nxs_dyn_ipc_export_record(fe.h, buf, stride))
        recs += buf.raw
    rb = C.create_string_buffer(recs, len(recs))
    assert fes[0].L.nxs_dyn_ipc_connect_records(fes[0].h, rb, stride, 3) != 0          # the wrong number of ranks
    for fe in fes:
        fe._chk(fe.L.nxs_dyn_ipc_connect_records(fe.h, rb, stride, 4))
    # (the self-test over these links, the steps and the delays run in tests/test_gpu_protocol_delays.py on this very partition, as 2 processes x 2 ranks: FOUR
    # handles of one process whose kernels wait for each other exceed the hardware queues a process gets -- two streams share a queue and one's spinning kernel
    # keeps the other's from starting)
    # records that disagree with this rank's lists are refused: rank 3's record with rank 0 cut out of its receive neighbours
    bad = bytearray(recs)
    head = 3 * stride + dynamics.IPC_BLOB_BYTES
    nr3 = int(np.frombuffer(bytes(bad[head + 8:head + 12]), np.int32)[0])
    rp = np.frombuffer(bytes(bad[head + 16:head + 16 + 4 * nr3]), np.int32).copy()
    assert 0 in rp.tolist()
    rp[rp.tolist().index(0)] = 2 if 2 not in rp.tolist() else 1
    bad[head + 16:head + 16 + 4 * nr3] = rp.tobytes()
    rc = fes[0].L.nxs_dyn_ipc_connect_records(fes[0].h, C.create_string_buffer(bytes(bad), len(bad)), stride, 4)
    assert rc != 0 and "does not list rank 0" in (fes[0].L.nxs_dyn_last_error(fes[0].h) or b"").decode()
    for fe in fes:
        fe.close()


@pytest.mark.parametrize("world,kind,rpp,over,overlap", [(2, "small", 1, {}, 0), (3, "small", 1, {"ragged_seed": 1}, 0), (3, "small", 1, {"dynamics_type": 3}, 0),
                                                         (8, "10km", 2, {}, 0), (2, "small", 1, {}, 1), (3, "small", 1, {"ragged_seed": 1, "dynamics_type": 3}, 1),
                                                         (8, "10km", 2, {}, 1), (2, "40km", 1, {}, 1),
                                                         # overlap = 600: patches of 600 nodes asked for (cut to ~450: more than one element per thread) -> k_substep_resident_big
                                                         # (there the interior elements run ahead by default; -600: without)
                                                         (2, "40km", 2, {}, 600), (3, "40km", 1, {"ragged_seed": 3, "dynamics_type": 3}, 600), (2, "40km", 2, {}, -600),
                                                         # four ragged ranks: own nodes that three neighbour ranks hold as ghosts (the first two destinations looked up once, the third from the tables)
                                                         (4, "40km", 2, {"ragged_seed": 3}, 600), (4, "small", 2, {}, 0)])
def test_resident_sub_step_loop_with_the_mailbox_exchange_inside(world, kind, rpp, over, overlap, tmp_path):
    """Option fused = 4 on several ranks: ONE launch per rank for the whole sub-step loop; boundary patches send into the
    neighbour ranks' mailboxes and read their ghosts from their own once per sub-step, the last boundary patch of a sub-step
    raises the flags (in sub-step order although patches far apart may be several sub-steps apart), ghost nodes are moved by
    the patch that stages them first.  Bitwise equal to separate push / pull kernels around one kernel per sub-step, and within
    1e-10 of the multi-rank oracle.  (The meshes are small enough for every rank's workgroups to be resident on ONE device.)
    overlap = 1: option resident_overlap -- the interior elements of every patch run one exchange ahead (their next update is computed
    while the exchange is awaited): the same bits."""
    options = {"fused": 4, "resident_overlap": overlap} if 0 <= overlap <= 1 else {"patch_nodes": abs(overlap), "fused": 4}
    if overlap < 0: options["resident_overlap"] = 0
    # Ranks that share a device get 70 % of its resident workgroup slots between them (headroom for each other's ordinary kernels: the registry of
    # csrc/nxs_resident_registry.hpp sees all four processes of the 8-rank cases).  Eight ranks of the 10 km mesh in the automatic 64-node patches
    # would be 8 x 58 = 464 of 512 slots; in 128-node patches they are 8 x ~33 = 52 %.
    if world >= 8: options["patch_nodes"] = 128
    reps = _run(world, kind, 1, tmp_path, "ipc", over=dict(over, options=options), ranks_per_proc=rpp)
    for r in reps:
        assert r["ok"], r
        assert r["fused_equals_separate"] is True, r
        assert r["launches_fused"] == 1, r
        assert r["crash"] == 0
        for k, e in r["errs"].items():
            assert e <= 1e-9 if over.get("dynamics_type") else e <= 1e-10, (r["rank"], k, e)


@pytest.mark.parametrize("world,kind,rpp,nsteps,over", [(2, "small", 1, 1, {}), (3, "small", 1, 1, {"ragged_seed": 1}), (2, "40km", 2, 2, {"dynamics_type": 3}),
                                                        (3, "small", 1, 1, {"ragged_seed": 5, "dynamics_type": 3}), (4, "10km", 2, 2, {}), (3, "toy", 1, 1, {"ragged_seed": 7})])
def test_two_sub_steps_per_launch_on_several_ranks(world, kind, rpp, nsteps, over, tmp_path):
    """k_substep_pair<HALO> (option pair_regs = 1 forces it on these small partitions; automatic above 65 k nodes per rank): a rank's sub-step loop is 60
    launches of TWO sub-steps, the stresses between them in registers, BOTH updateGhosts of a launch inside it -- the patches along the partition boundary
    store their first velocities into the neighbours' mailboxes, wait for the neighbours' and go on; every other patch runs the single-rank body.  Bitwise
    equal to one kernel per sub-step with separate push / pull kernels, and within 1e-10 of the multi-rank oracle: regular and ragged partitions (ghost
    elements without an own node, sent nodes that touch no ghost, every rank a neighbour of every other), BBM and EVP, one and two steps (the second starts
    from the velocity buffer again), two ranks per process."""
    reps = _run(world, kind, nsteps, tmp_path, "ipc", over=dict(over, options={"pair_regs": 1}), ranks_per_proc=rpp)
    for r in reps:
        assert r["ok"], r
        assert r["fused_equals_separate"] is True, r
        assert r["launches_fused"] == 60 and r["launches_separate"] > 120, r
        assert r["crash"] == 0
        for k, e in r["errs"].items():
            assert e <= 1e-9 if over.get("dynamics_type") else e <= 1e-10, (r["rank"], k, e)


@pytest.mark.parametrize("world,kind,rpp,over,band", [(2, "40km", 2, {}, 0), (2, "40km", 2, {}, 16), (3, "40km", 1, {"ragged_seed": 3}, 24), (4, "10km", 2, {"dynamics_type": 3}, 32)])
def test_small_patches_along_the_partition_boundary_do_not_change_a_bit(world, kind, rpp, over, band, tmp_path):
    """Option band_patch_nodes (default 48): in the cut for the resident loop of several ranks the own nodes that share an element with a ghost node
    lead the cutter's order and get patches of their own -- the patches that pay the exchange between ranks every sub-step then have a short compute
    phase.  A different cut, the same bits: explicit sizes and none at all (0) against the separate kernels and the multi-rank oracle, regular and
    ragged partitions (where a rank also sends nodes that touch no ghost of its own mesh: those stay in ordinary patches), EVP."""
    # (four ranks of the 10 km mesh in the automatic 64-node patches would claim 4 x ~125 of the device's 512 resident slots; ranks that share a device get
    # 70 % between them, csrc/nxs_resident_registry.hpp -- 160-node patches: 4 x ~52)
    reps = _run(world, kind, 1, tmp_path, "ipc", over=dict(over, options=dict({"fused": 4, "band_patch_nodes": band}, **({"patch_nodes": 160} if world >= 4 else {}))), ranks_per_proc=rpp)
    for r in reps:
        assert r["ok"], r
        assert r["fused_equals_separate"] is True and r["launches_fused"] == 1 and r["crash"] == 0, r
        for k, e in r["errs"].items():
            assert e <= 1e-9 if over.get("dynamics_type") else e <= 1e-10, (r["rank"], k, e)


@pytest.mark.parametrize("kind,world", [("2km", 8), ("10km", 8)])
def test_every_partition_of_the_baseline_configurations_can_run_the_resident_loop(kind, world):
    """BASELINE configs 3 and 4 on eight GPUs are meant to run the whole sub-step loop of a rank as ONE resident launch (option fused = 4).
    Eight ranks of 183 k triangles cannot be resident together on one device, but whether a partition CAN run it is decided by its own
    tables: every patch at most one element per thread (the RCB parts of a Hilbert-numbered mesh used to have a few patches of two distant
    blobs with 530-650 elements: the patch cutter now closes a patch at 480), one round of workgroups, the LDS of two workgroups per CU,
    at most 24 neighbouring patches.  Option resident_dryrun builds them (patches, boundary-first order, neighbour and ghost tables,
    occupancy) without stepping: every rank must say yes, with and without the interior-first element order."""
    import cases
    from nextsim_amd import dynamics
    gm, p, g, lms, fields = cases.make_case(kind, nparts=world)
    for rank, lm in enumerate(lms):
        fe = dynamics.FiniteElementDynamics(p)
        fe.set_mesh(lm)
        for overlap in (0, 1):
            fe.set_option("resident_overlap", overlap)
            fe.set_option("resident_dryrun", 1)      # raises NxsError when this partition could not run it
        fe.close()


@pytest.mark.parametrize("world,kind,over", [(2, "small", {}), (3, "40km", {"dynamics_type": 3})])
def test_resident_loop_in_the_build_with_all_its_registers(world, kind, over, tmp_path):
    """Option resident_wide: where one workgroup per CU covers a rank's partition and the device is the handle's alone, the several-rank resident
    kernel runs in its build for two waves per SIMD (148 registers instead of 128 with spills).  The few small grids of this test fit one GPU
    side by side even so: bitwise equal to the separate kernels, and the multi-rank oracle's results."""
    reps = _run(world, kind, 1, tmp_path, "ipc", over=dict(over, options={"fused": 4, "resident_wide": 1}))
    for r in reps:
        assert r["ok"], r
        assert r["fused_equals_separate"] is True, r
        assert r["launches_fused"] == 1, r
        assert r["crash"] == 0
        for k, e in r["errs"].items():
            assert e <= 1e-9 if over.get("dynamics_type") else e <= 1e-10, (r["rank"], k, e)


def test_bench_line_of_a_two_rank_run_is_one_json_line_with_every_structured_field(tmp_path):
    """bench.py as the driver launches it at N = 2 (torch.distributed.run, one rank per "GPU"; here both ranks share device 0, so the figures only
    bound the protocol): stdout is ONE JSON line (RCCL's and gloo's banners stay on stderr), it carries the contract's keys, the structured
    transport / variant report (config.halo), the slowest rank's phases, aux_rccl (RCCL refuses two ranks on one device: a recorded failure, not a
    lost line) and the CPU baseline of rank 0's host."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--mesh", "40km"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["ranks_share_device"] is True and d["scaling"] == "strong" and d["value"] > 0 and d["fields_ok"] is True
    halo = d["config"]["halo"]
    assert halo["transport"] == "device-direct mailboxes" and halo["mailbox_selftest"].startswith("passed")
    assert halo["kept_variant"] in ("inkernel", "inkernel_pair", "separate", "resident", "resident_overlap", "resident_norelease")
    assert halo["variants"]["inkernel_pair"]["status"] in ("ran", "failed")
    assert halo["variants"]["separate"]["status"] == "ran" and halo["variants"]["separate"]["bit_identical_to_separate_kernels"] is True
    assert halo["variants"]["inkernel"]["status"] in ("ran", "failed") and "ms_per_step" in halo["variants"]["separate"]
    assert set(d["phases_ms"]) == {"prep_ms", "substeps_ms", "smoother_ms", "update_ms", "total_ms"} and d["phases_ms"]["total_ms"] >= d["phases_ms_rank0"]["total_ms"] - 1e-9
    assert d["aux_rccl"]["status"] in ("ran", "failed") and ("error" in d["aux_rccl"] or "ms_per_step" in d["aux_rccl"])
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["host_cores"] >= cb["cores"] >= 1 and cb["cpu_model"] and cb["thread_counts_tried"]
