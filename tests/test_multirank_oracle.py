"""Domain decomposition of the reference algorithm on the CPU oracle: P in-process ranks with real
updateGhosts exchanges (FE.cpp:13963-13996) against the single-rank run.

Ghost elements are recomputed, not reduced (FE.cpp:10446, 10456): no cross-rank sum exists.  The
owned results still differ from the 1-rank run in the last bits, because a rank numbers its ghost
elements after its owned ones (gmshmesh.cpp:1379-1417) and the serial element loop therefore adds a
boundary node's fan in a different order (true of the reference itself)."""
import numpy as np
import pytest

import cases
from oracle import pyoracle as O

KEYS_N = ("VT", "UM", "UT")
KEYS_E = ("sigma0", "sigma1", "sigma2", "damage", "conc", "thick", "ridge_ratio", "conc_young", "h_young")


def _run(kind, nparts, nsteps, **over):
    gm, p, g, lms, fields = cases.make_case(kind, nparts=nparts, **over)
    ranks = [O.OracleRank(lm, p, f) for lm, f in zip(lms, fields)]
    for _ in range(nsteps):
        if nparts == 1:
            ranks[0].step()
        else:
            O.multirank_step(ranks)
    return gm, lms, ranks


def _gather(gm, lms, ranks):
    Nn, Ne = gm.num_nodes, gm.num_elements
    out = {k: np.full(2 * Nn, np.nan) for k in KEYS_N}
    out.update({k: np.full(Ne, np.nan) for k in KEYS_E})
    for lm, r in zip(lms, ranks):
        No, Neo = lm.local_ndof, lm.local_nelements
        for k in KEYS_N:
            out[k][lm.node_gid[:No]] = r.arr[k][:No]
            out[k][Nn + lm.node_gid[:No]] = r.arr[k][lm.num_nodes:lm.num_nodes + No]
        for k in KEYS_E:
            out[k][lm.elem_gid[:Neo]] = r.arr[k][:Neo]
    return out


@pytest.mark.parametrize("nparts", [2, 4])
def test_full_cover_partitioned_equals_serial_to_roundoff(nparts):
    """Full ice cover (no node ever has zero mass): owned values equal to summation-order round-off."""
    def full_cover(kind, n):
        gm, p, g, lms, fields = cases.make_case(kind, nparts=n)
        for f in fields:
            f["conc"][:] = 1.0; f["thick"][:] = np.maximum(f["thick"], 0.5); f["conc_young"][:] = 0; f["h_young"][:] = 0
        ranks = [O.OracleRank(lm, p, f) for lm, f in zip(lms, fields)]
        for _ in range(2):
            O.multirank_step(ranks) if n > 1 else ranks[0].step()
        return _gather(gm, lms, ranks)
    a, b = full_cover("small", 1), full_cover("small", nparts)
    for k in KEYS_N + KEYS_E:
        assert not np.isnan(b[k]).any()
        assert cases.rel_err(b[k], a[k]) < 1e-13, k


def test_partial_cover_partitioned_matches_serial():
    """With open water the reference's per-rank semantics differ slightly from the serial run (a ghost
    node's node_mass is summed over a partial fan: FE.cpp:10366 zeroes M_VT on ghosts that only touch
    ice-free elements locally), so owned values agree to round-off rather than bitwise."""
    gm, lms1, r1 = _run("small", 1, 2)
    gm, lms3, r3 = _run("small", 3, 2)
    a, b = _gather(gm, lms1, r1), _gather(gm, lms3, r3)
    for k in KEYS_N + KEYS_E:
        assert cases.rel_err(b[k], a[k]) < 1e-9, k


def test_callback_form_of_explicit_solve_matches_phase_form():
    """ref_explicit_solve with a ghosts callback == the phase functions driven from Python."""
    import ctypes as C
    gm, p, g, lms, fields = cases.make_case("toy")
    a = O.OracleRank(lms[0], p, fields[0]); a.step()
    b = O.OracleRank(lms[0], p, fields[0])
    calls = []
    cb = O.GHOST_FN(lambda ctx, vec: calls.append(1))
    m, pp, s, f, w = b._a()
    b.L.ref_step(m, pp, s, f, w, C.cast(cb, C.c_void_p), None)
    assert len(calls) == 120 + 50
    for k in KEYS_N + KEYS_E:
        assert np.array_equal(a.arr[k], b.arr[k])


@pytest.mark.parametrize("kind,nparts,nthreads,over", [("toy", 3, 2, {}), ("small", 5, 5, {}), ("small", 4, 3, {"dynamics_type": 4}),
                                                      ("small", 3, 8, {"ragged_seed": 21}), ("toy", 2, 1, {"dynamics_type": 3})])
def test_threaded_lock_step_in_the_library_equals_the_phase_by_phase_driver(kind, nparts, nthreads, over):
    """ref_multirank_steps (pthreads, barriers, shared-memory updateGhosts: the cpu_baseline of bench.py) gives, bit for bit, what the
    phase-by-phase Python driver gives: more ranks than threads, fewer, ragged partitions, BBM / EVP / mEVP, two steps."""
    gm, p, g, lms, fields = cases.make_case(kind, nparts=nparts, **over)
    a = [O.OracleRank(lm, p, f) for lm, f in zip(lms, fields)]
    b = [O.OracleRank(lm, p, f) for lm, f in zip(lms, fields)]
    for _ in range(2):
        O.multirank_step(a)
    O.multirank_steps_native(b, nsteps=2, nthreads=nthreads)
    for ra, rb in zip(a, b):
        for k in KEYS_N + KEYS_E:
            assert np.array_equal(ra.arr[k], rb.arr[k]), (ra.lm.rank, k)


@pytest.mark.parametrize("kind,nparts,nthreads,pin,over", [("toy", 3, 2, True, {}), ("small", 5, 5, True, {}), ("small", 4, 3, False, {"dynamics_type": 4}),
                                                          ("small", 3, 8, True, {"ragged_seed": 21}), ("toy", 2, 1, True, {"dynamics_type": 3})])
def test_persistent_pinned_context_equals_the_phase_by_phase_driver(kind, nparts, nthreads, pin, over):
    """ref_mr_create / ref_mr_run / ref_mr_destroy -- threads kept across calls and pinned, every partition's arrays copied (first touched) by the
    thread that owns it, spin barriers: the form of the lock-step run that bench.py's cpu_baseline times -- gives, bit for bit, what the phase-by-phase
    driver gives; two runs on one context (one step + one step) equal two steps; the ranks' own arrays change only when the context is closed."""
    gm, p, g, lms, fields = cases.make_case(kind, nparts=nparts, **over)
    a = [O.OracleRank(lm, p, f) for lm, f in zip(lms, fields)]
    b = [O.OracleRank(lm, p, f) for lm, f in zip(lms, fields)]
    for _ in range(2):
        O.multirank_step(a)
    before = b[0].arr["VT"].copy()
    ctx = O.MultirankContext(b, nthreads=nthreads, pin=pin)
    info = ctx.info()
    assert info["threads"] == min(nthreads, nparts) and len(info["cpus"]) == info["threads"]
    if pin:
        assert all(c >= 0 for c in info["cpus"]) and info["sockets_used"] >= 1
        import os
        assert set(info["cpus"]) <= set(os.sched_getaffinity(0))
    ctx.run(1); ctx.run(1)
    assert np.array_equal(b[0].arr["VT"], before)
    ctx.close()
    for ra, rb in zip(a, b):
        for k in KEYS_N + KEYS_E:
            assert np.array_equal(ra.arr[k], rb.arr[k]), (ra.lm.rank, k)


def test_every_barrier_and_pinning_of_the_cpu_baseline_gives_the_same_bits():
    """bench.py's cpu_baseline times ONE context in every combination of barrier (spin / sleeping) and pinning (ref_mr_configure between the runs) and reports the
    fastest: four steps, one per combination, equal four steps of the phase-by-phase driver bit for bit; the info says what the threads are pinned to."""
    gm, p, g, lms, fields = cases.make_case("small", nparts=4, ragged_seed=5)
    a = [O.OracleRank(lm, p, f) for lm, f in zip(lms, fields)]
    b = [O.OracleRank(lm, p, f) for lm, f in zip(lms, fields)]
    for _ in range(4):
        O.multirank_step(a)
    ctx = O.MultirankContext(b, nthreads=3, pin=True)
    for barrier, pin in (("sleep", False), ("sleep", True), ("spin", True), ("spin", False)):
        ctx.configure(barrier, pin)
        assert all((c >= 0) == pin for c in ctx.info()["cpus"])
        ctx.run(1)
    ctx.close()
    for ra, rb in zip(a, b):
        for k in KEYS_N + KEYS_E:
            assert np.array_equal(ra.arr[k], rb.arr[k]), (ra.lm.rank, k)
