"""Deterministic tests of the ORDERINGS of the device-direct exchange protocols.

The reference exchanges ghosts with blocking send / recv (FE.cpp:13981-13985): nothing can overtake anything.  The mailbox transport replaces that with flags and
two buffers per link, in four kernel variants (separate k_halo_push / k_halo_pull; inside k_substep_fused; twice inside k_substep_pair; inside the resident loop)
plus the smoother's slots.  Round 4 found a real overwrite race there (a neighbour in one direction only) that no bitwise test had caught, because the window is a
few microseconds wide.  Here ONE rank sleeps ~0.5 ms at ONE named point of a protocol (option "ipc_delay", include/nxs_dyn.h: NXS_DELAY_*), which turns "whoever wins
the race" into "the delayed rank always loses": every (variant x point x delayed rank) of ragged 3- and 4-rank partitions must still give the bits of the undelayed
separate kernels -- and round 4's defect, re-created through the test door "halo_one_directional", must fail under the same delays, every time, in every variant.
Each case runs ONCE; nothing is repeated to provoke a failure."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

# NXS_DELAY_* of include/nxs_dyn.h
PULL_READ, PUSH_STORE, PUSH_FLAG, STAGE_READ, SEND_STORE, PAIR_SECOND_STORE, PUBLISH_FLAG, PAIR_MID_READ, SMOOTH_READ, SMOOTH_STORE, SMOOTH_FLAG, SMOOTH_PULL_READ, PAIR_SECOND_FLAG = range(1, 14)
POINT_NAMES = {1: "pull_read", 2: "push_store", 3: "push_flag", 4: "stage_read", 5: "send_store", 6: "pair_second_store", 7: "publish_flag", 8: "pair_mid_read",
               9: "smooth_read", 10: "smooth_store", 11: "smooth_flag", 12: "smooth_pull_read", 13: "pair_second_flag"}
_BASE = {"fused": 3, "pair_regs": 0, "halo_fused": 1, "smooth_persist": 1, "resident_release": 1}
VARIANTS = {   # (every variant names every option: the worker keeps ONE handle per rank for all cases)
    "separate": (dict(_BASE, halo_fused=0), (PULL_READ, PUSH_STORE, PUSH_FLAG)),
    "inkernel": (dict(_BASE), (STAGE_READ, SEND_STORE, PUBLISH_FLAG, PULL_READ, SMOOTH_READ, SMOOTH_STORE, SMOOTH_FLAG, SMOOTH_PULL_READ)),
    # the smoother as 50 launches of one sweep (k_smooth_halo) instead of one launch of persistent workgroups (k_smooth_persist, the default since round 5)
    "sweeps": (dict(_BASE, smooth_persist=0), (SMOOTH_READ, SMOOTH_STORE, SMOOTH_FLAG, SMOOTH_PULL_READ)),
    "pair": (dict(_BASE, pair_regs=1), (STAGE_READ, SEND_STORE, PUBLISH_FLAG, PAIR_MID_READ, PAIR_SECOND_STORE, PAIR_SECOND_FLAG, PULL_READ)),
    "resident": (dict(_BASE, fused=4), (STAGE_READ, SEND_STORE, PUBLISH_FLAG)),
    # the resident loop without the release fence in front of a sub-step's flags (option resident_release 0: what bench.py tries on machines with a device per rank)
    "resident_norelease": (dict(_BASE, fused=4, resident_release=0), (STAGE_READ, SEND_STORE, PUBLISH_FLAG)),
}
EXPECT_KERNEL = {"separate": "k_substep_fused", "inkernel": "k_substep_fused", "sweeps": "k_substep_fused", "pair": "k_substep_pair", "resident": "k_substep_resident", "resident_norelease": "k_substep_resident"}


def _walk(world, units, variants=VARIANTS, mixed=True):
    cases = [{"name": "reference", "options": dict(VARIANTS["separate"][0])}]
    for vname, (opts, points) in variants.items():
        cases.append({"name": f"{vname}/undelayed", "options": dict(opts)})
        for pt in points:
            for r in range(world):
                cases.append({"name": f"{vname}/{POINT_NAMES[pt]}/rank{r}", "options": dict(opts), "delay": [r, pt, units]})
    if mixed:   # k_substep_pair<HALO> on some ranks against k_substep_fused<HALO> on their neighbours, through the same two-half mailbox
        for pat in ([1, 0], [0, 1]):
            mopts = {k: v for k, v in _BASE.items() if k != "pair_regs"}
            cases.append({"name": f"mixed{pat[0]}/undelayed", "options": mopts, "rank_options": {"pair_regs": pat}})
            for pt in (STAGE_READ, PAIR_MID_READ, PAIR_SECOND_STORE, PUBLISH_FLAG, SEND_STORE):
                for r in range(world):
                    cases.append({"name": f"mixed{pat[0]}/{POINT_NAMES[pt]}/rank{r}", "options": mopts, "rank_options": {"pair_regs": pat}, "delay": [r, pt, units]})
    return cases


def _run(world, spec, tmp_path, ranks_per_proc=1, timeout=900):
    assert world % ranks_per_proc == 0 and world // ranks_per_proc <= 6
    nproc = world // ranks_per_proc
    sp = tmp_path / "spec.json"
    sp.write_text(json.dumps(spec))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(nproc):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(nproc), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NXS_RANKS_PER_PROC=str(ranks_per_proc))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "delay_worker.py"), str(tmp_path), str(sp)], env=env, stdin=subprocess.DEVNULL))
    for p in procs:
        try:
            p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("protocol-delay workers hung")
    return [json.load(open(tmp_path / f"report{r}.json")) for r in range(world)]


def _reference_against_the_oracle(kind, world, over, tmp_path):
    import numpy as np
    import cases
    from oracle import pyoracle as O
    gm, p, g, lms, fields = cases.make_case(kind, nparts=world, **over)
    ranks = [O.OracleRank(lm, p, f) for lm, f in zip(lms, fields)]
    O.multirank_step(ranks)
    for r in range(world):
        z = np.load(tmp_path / f"ref{r}.npz")
        for k in z.files:
            assert cases.rel_err(z[k], ranks[r].arr[k]) <= 1e-10, (r, k)
    return lms


@pytest.mark.parametrize("world,seed,rpp,state", [(3, 1, 1, "arctic_ow"), (4, 2, 2, "arctic")])
def test_every_protocol_point_delayed_on_every_rank_keeps_the_bits(world, seed, rpp, state, tmp_path):
    """Ragged mosaics of 'small' (every rank a neighbour of every other, elements without an own node, nodes shared by three ranks), the halo lists VERBATIM
    (nextsim_amd.mesh.localize no longer symmetrises them: nxs_dyn_set_halo does), connected through the record form.  4 ranks = 2 processes x 2 ranks: links
    through hipIpc and through plain pointers."""
    # ('arctic_ow': half the rim ice free -- open water on both sides of the partition boundaries, so that the smoother's sweeps really exchange values that change)
    over = {"ragged_seed": seed, "forcing_kind": state}
    spec = {"kind": "small", "over": over, "phases": [{"one_directional": 0, "cases": _walk(world, 50)}]}
    reps = _run(world, spec, tmp_path, ranks_per_proc=rpp)
    for r in reps:
        assert r["ok"], r.get("error", r)
    _reference_against_the_oracle("small", world, over, tmp_path)
    bad = []
    for r in reps:
        for c in r["phases"][0]["cases"]:
            if not c.get("equal") or c.get("error") or c.get("crash"):
                bad.append((r["rank"], c))
            v = c["name"].split("/")[0]
            if v in EXPECT_KERNEL:   # the variant named is the variant that ran
                assert c["kernel"].startswith(EXPECT_KERNEL[v]), (r["rank"], c)
                assert not v.startswith("resident") or c["launches"] == 1, (r["rank"], c)
            if v.startswith("mixed"):
                want = [1, 0] if v == "mixed1" else [0, 1]
                assert c["kernel"] == ("k_substep_pair" if want[r["rank"] % 2] else "k_substep_fused"), (r["rank"], c)
    assert not bad, bad[:6]


def test_the_one_directional_neighbour_of_round_4_fails_under_a_delay_and_the_padded_lists_do_not(tmp_path):
    """The regular 4-rank partition of 'small': rank 0 sends two nodes to rank 3 and receives nothing from it.  Phase 1 -- the lists as nxs_dyn_set_halo keeps them
    (the missing direction added as an empty segment): rank 3 delayed in front of every read of its mailbox, in every variant, the self-test included: the bits of
    the separate kernels.  Phase 2 -- round 4's protocol (test door halo_one_directional: lists as given, caller-side bookkeeping): the SAME delays now let rank 0
    run two exchanges ahead of rank 3 and overwrite the half rank 3 is about to read: wrong bits in every variant, a wrong payload in the self-test.  One run each."""
    import cases
    gm, p, g, lms, fields = cases.make_case("small", nparts=4)
    assert 3 in lms[0].send_procs.tolist() and 3 not in lms[0].recv_procs.tolist()          # the one-directional link is there
    U = 150   # 1.5 ms: rank 0 needs two sub-steps of four ranks sharing a device
    reads = [("separate", PULL_READ), ("inkernel", STAGE_READ), ("pair", STAGE_READ), ("pair", PAIR_MID_READ), ("resident", STAGE_READ)]
    delayed = [{"name": f"{v}/{POINT_NAMES[pt]}/rank3", "options": dict(VARIANTS[v][0]), "delay": [3, pt, U]} for v, pt in reads]
    selftest = {"name": "selftest/pull_read/rank3", "options": dict(VARIANTS["separate"][0]), "delay": [3, PULL_READ, U], "selftest": 16}
    spec = {"kind": "small", "over": {}, "phases": [
        {"one_directional": 0, "cases": [{"name": "reference", "options": dict(VARIANTS["separate"][0])}] + delayed + [selftest]},
        {"one_directional": 1, "cases": delayed + [selftest]}]}
    reps = _run(4, spec, tmp_path, ranks_per_proc=2)
    for r in reps:
        assert r["ok"], r.get("error", r)
    _reference_against_the_oracle("small", 4, {}, tmp_path)
    for r in reps:                                   # phase 1: green everywhere
        for c in r["phases"][0]["cases"]:
            assert c.get("selftest_errors", 0) == 0 and c.get("equal", True) and not c.get("error"), (r["rank"], c)
    r3 = [r for r in reps if r["rank"] == 3][0]      # phase 2: rank 3 reads what rank 0 has already overwritten
    red = {c["name"]: (not c["equal"]) if "equal" in c else c["selftest_errors"] != 0 for c in r3["phases"][1]["cases"]}
    assert all(red.values()), red


@pytest.mark.parametrize("world,kind,rpp,over", [(3, "small", 1, {"ragged_seed": 4}), (2, "40km", 2, {}), (4, "small", 2, {"forcing_kind": "arctic_ow", "ice_cat_type": 1})])
def test_fused_prep_kernel_on_several_ranks_does_not_change_a_bit(world, kind, rpp, over, tmp_path):
    """Round 5: k_prep_fused (prep elements + prep nodes in one launch over the node patches, FE.cpp:10235-10416) also on a rank of several -- over the patches of
    its OWN nodes, with k_prep_ghost_nodes for the ghosts' share of the nodal loops (partial fans, M_VT zeroed where the local mass is zero, D_tau_a, the frozen
    coordinates).  Forced on these small partitions by option prep_fused 1 (automatic from 250 k triangles): the state after a step AND the prep kernels' own arrays --
    the 80-byte nodal records, the frozen coordinates, D_tau_a, M_surface, M_delta_x, the 48-byte element records, ghosts and ghost elements included -- are bit for bit
    those of the two separate kernels, under every kernel family of the sub-step loop (whose patch cuts differ)."""
    base = dict(_BASE)
    cases = [{"name": "reference", "options": dict(base, prep_fused=0)}]
    for name, o in (("inkernel", {}), ("pair", {"pair_regs": 1}), ("resident", {"fused": 4}), ("separate", {"halo_fused": 0})):
        cases.append({"name": f"{name}/two_kernels", "options": dict(base, prep_fused=0, **o)})
        cases.append({"name": f"{name}/fused", "options": dict(base, prep_fused=1, **o)})
    spec = {"kind": kind, "over": over, "debug_arrays": ["nrec", "xy", "tau_a", "surface", "delta_x", "erec"], "phases": [{"one_directional": 0, "cases": cases}]}
    reps = _run(world, spec, tmp_path, ranks_per_proc=rpp)
    for r in reps:
        assert r["ok"], r.get("error", r)
        for c in r["phases"][0]["cases"]:
            assert c.get("equal") and not c.get("error") and not c.get("crash"), (r["rank"], c)
            if c["name"].endswith("/fused"):
                assert c["prep"] == "k_prep_fused", (r["rank"], c)
            elif c["name"] != "reference":
                assert c["prep"].startswith("k_prep_elements"), (r["rank"], c)


def test_two_hundred_steps_of_every_exchange_variant_keep_the_bits(tmp_path):
    """34 000 exchanges per variant and rank (200 steps of 120 sub-steps + 50 smoother sweeps each) on the ragged 4-rank partition, each variant from the same state:
    the state after the last step is, bit for bit, the separate kernels' -- the mesh has moved 24 000 times by then, the damage has grown, every sequence number
    and mailbox half has wrapped around thousands of times.  One run per variant."""
    steps = 200
    cases_ = [{"name": "reference", "options": dict(VARIANTS["separate"][0]), "steps": steps}]
    cases_ += [{"name": f"{v}/x{steps}", "options": dict(VARIANTS[v][0]), "steps": steps} for v in VARIANTS if v != "separate"]
    mopts = {k: v for k, v in _BASE.items() if k != "pair_regs"}
    cases_.append({"name": f"mixed/x{steps}", "options": mopts, "rank_options": {"pair_regs": [1, 0]}, "steps": steps})
    # ('arctic_ow': open water on both sides of the partition boundaries -- the smoother's sweeps exchange values that change)
    spec = {"kind": "small", "over": {"ragged_seed": 2, "forcing_kind": "arctic_ow"}, "phases": [{"one_directional": 0, "cases": cases_}]}
    reps = _run(4, spec, tmp_path, ranks_per_proc=2)
    for r in reps:
        assert r["ok"], r
        for c in r["phases"][0]["cases"]:
            assert c.get("error") is None and c["crash"] == 0, (r["rank"], c)
            assert c["equal"], (r["rank"], c["name"], c.get("worst"))
            v = c["name"].split("/")[0]
            if v in EXPECT_KERNEL:
                assert c["kernel"].startswith(EXPECT_KERNEL[v]), (r["rank"], c)

