"""SURVEY.md section 8d's long-horizon tolerance with its threshold-flip set, instrumented.

The survey's 10-step figure: L2-relative <= 1e-8 on v and sigma, <= 1e-8 absolute on damage, EXCLUDING the elements whose
damage criterion came within 1e-9 of its threshold (|dcrit - 1| < 1e-9, FE.cpp:4229) or whose concentration within 1e-12 of the
hard-coded cut-off (|A - 0.1| < 1e-12, FE.cpp:4151) at any sub-step; that set must stay below 1e-4 of the elements.

Both implementations keep a per-element record of the branch updateSigmaDamage took at every sub-step (oracle: ref_work.trace in
oracle/dyn_ref.h; device: option "trace_branches" + nxs_dyn_get_branch_trace).  Equal hashes = the same branches at every sub-step
so far.  The test runs both freely (no re-seeding) and at 1, 2, 5 and 10 steps reports
    near   the survey's exclusion set (either side's flags),
    flip   the elements whose branch histories differ (what actually happened),
    and the L2-relative / absolute differences over everything, over the complement of `near`, and over the complement of
    `flip` grown by the elements around it (a flipped element changes its three nodes' velocities at once).
A second oracle whose wind differs by ONE ulp runs beside them (the "twin"): its distance from the first oracle is what the
algorithm itself makes of the smallest possible perturbation -- the yardstick for any second implementation.
What holds is asserted; what does not is printed and recorded in DESIGN.md section 2."""
import numpy as np
import pytest

import cases

CHECKPOINTS = (1, 2, 5, 10)


def _l2rel(a, b, mask=None):
    if mask is not None:
        a, b = a[mask], b[mask]
    den = float(np.sqrt((b * b).sum()))
    return float(np.sqrt(((a - b) ** 2).sum())) / max(den, 1e-300)


def _grow(tri, elem_mask, rings):
    """elements within `rings` node-adjacency rings of the marked ones"""
    m = elem_mask.copy()
    nn = tri.max() + 1
    for _ in range(rings):
        node = np.zeros(nn, bool)
        node[tri[m].ravel()] = True
        m = node[tri].any(1)
    return m


def test_oracle_branch_trace_is_a_faithful_record():
    """CPU: the trace changes nothing, counts every sub-step, and its damage count is the number of damage increments."""
    from oracle import pyoracle as O
    gm, p, g, lms, fields = cases.make_case("small")
    a = O.OracleRank(lms[0], p, fields[0]); b = O.OracleRank(lms[0], p, fields[0])
    b.enable_branch_trace()
    for _ in range(2):
        a.step(); b.step()
    for k in ("VT", "sigma0", "damage", "conc"):
        assert np.array_equal(a.arr[k], b.arr[k]), k
    t = b.branch_trace()
    assert np.all(t["substeps"] == 2 * p.substeps)
    skipped = (t["flags"] & 4) != 0
    assert np.all(skipped[fields[0]["conc"] <= 0.1]) and not skipped.all()       # the first step's skipped elements carry the flag
    assert 0 < t["damage_substeps"].max() <= 2 * p.substeps
    always = (fields[0]["conc"] <= 0.1) & (b.arr["conc"] <= 0.1)
    assert np.all(t["damage_substeps"][always] <= p.substeps)
    b.enable_branch_trace()                        # restart
    assert b.branch_trace()["substeps"].max() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["toy", "40km"])
def test_threshold_flip_set_and_the_surveys_ten_step_figure(kind, capsys):
    from nextsim_amd import dynamics
    from oracle import pyoracle as O
    gm, p, g, lms, fields = cases.make_case(kind)
    lm, f = lms[0], fields[0]
    tri = lm.indices.reshape(-1, 3) - 1
    Ne, Nn = lm.num_elements, lm.num_nodes
    fe = dynamics.FiniteElementDynamics(p)
    fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
    fe.set_option("trace_branches", 1)
    ref = O.OracleRank(lm, p, f)
    ref.enable_branch_trace()
    f_twin = {k: v.copy() for k, v in f.items()}
    f_twin["wind"] = np.nextafter(f_twin["wind"], np.inf)
    twin = O.OracleRank(lm, p, f_twin)
    twin.enable_branch_trace()
    rows, done = [], 0
    for cp in CHECKPOINTS:
        for _ in range(cp - done):
            fe.step(); ref.step(); twin.step()
        done = cp
        fe.synchronize()
        got, td, to = fe.get_state(), fe.branch_trace(), ref.branch_trace()
        assert np.array_equal(td["substeps"], to["substeps"]) and td["substeps"].max() == cp * p.substeps
        flip = td["hash"] != to["hash"]
        near = ((td["flags"] | to["flags"]) & 3) != 0
        around = _grow(tri, flip, 2)
        node_near = np.zeros(Nn, bool); node_near[tri[near].ravel()] = True
        node_around = np.zeros(Nn, bool); node_around[tri[around].ravel()] = True
        nn2 = lambda m: np.concatenate([m, m])  # noqa: E731
        row = {"steps": cp, "near_frac": near.mean(), "flip_frac": flip.mean(), "around_frac": around.mean(),
               "damage_elems": float((to["damage_substeps"] > 0).mean())}
        for tag, em, nm in (("all", None, None), ("not_near", ~near, ~nn2(node_near)), ("not_around_flip", ~around, ~nn2(node_around))):
            row[f"v_{tag}"] = _l2rel(got["VT"], ref.arr["VT"], nm)
            row[f"sig_{tag}"] = max(_l2rel(got[k], ref.arr[k], em) for k in ("sigma0", "sigma1", "sigma2"))
            d = np.abs(got["damage"] - ref.arr["damage"])
            row[f"d_{tag}"] = float((d if em is None else d[em]).max()) if (em is None or em.any()) else 0.0
        row["twin_flip_frac"] = float((twin.branch_trace()["hash"] != to["hash"]).mean())
        row["twin_v"] = _l2rel(twin.arr["VT"], ref.arr["VT"])
        row["twin_sig"] = max(_l2rel(twin.arr[k], ref.arr[k]) for k in ("sigma0", "sigma1", "sigma2"))
        row["twin_d"] = float(np.abs(twin.arr["damage"] - ref.arr["damage"]).max())
        rows.append(row)
    fe.close()
    with capsys.disabled():
        print(f"\n[flip set] mesh '{kind}' ({Ne} triangles), free-running device vs oracle, BBM, {p.substeps} sub-steps per step")
        print("  steps  near(survey)  flipped   damaging   | L2rel v / sigma, max|dd|: all elements | survey exclusion | flipped+2 rings excluded")
        for r in rows:
            print(f"  {r['steps']:5d}  {r['near_frac']:.2e}   {r['flip_frac']:.2e}  {r['damage_elems']:.2e}  | "
                  f"{r['v_all']:.1e} {r['sig_all']:.1e} {r['d_all']:.1e} | {r['v_not_near']:.1e} {r['sig_not_near']:.1e} {r['d_not_near']:.1e} | "
                  f"{r['v_not_around_flip']:.1e} {r['sig_not_around_flip']:.1e} {r['d_not_around_flip']:.1e}")
        print("  the oracle against its 1-ulp-of-wind twin:   steps  flipped   L2rel v / sigma, max|dd|")
        for r in rows:
            print(f"                                               {r['steps']:5d}  {r['twin_flip_frac']:.2e}  {r['twin_v']:.1e} {r['twin_sig']:.1e} {r['twin_d']:.1e}")
    # What holds (measured on MI355X):
    #  * the survey's exclusion set is EMPTY at every horizon on both meshes -- no criterion ever comes within 1e-9 of its threshold,
    #    so the survey's metric excludes nothing;
    #  * one and two free steps: every element within 1e-10 (measured 1e-16 .. 6e-14), not one branch differs;
    #  * from there the difference grows with NO branch flipped (40 km, 5 steps: 0 flips, sigma 2e-5) -- the growth is the algorithm's
    #    own sensitivity, flips follow it instead of causing it -- and passes 1e-8 between step 2 and step 5;
    #  * at every horizon the device is as far from the oracle as the oracle's own 1-ulp twin is (ratios 0.2 .. 1.6 measured):
    #    what any second implementation of FE.cpp:4137-4260 can achieve.  The survey's 1e-8 at 10 steps is not among those things.
    for r in rows:
        assert r["near_frac"] < 1e-4, r
    for r in rows[:2]:
        assert r["flip_frac"] == 0.0, "a branch differs inside the first two steps"
        assert r["v_all"] <= 1e-10 and r["sig_all"] <= 1e-10 and r["d_all"] <= 1e-10, r
    for r in rows:
        for dev, tw in (("v_all", "twin_v"), ("sig_all", "twin_sig"), ("d_all", "twin_d")):
            assert r[dev] <= 5.0 * r[tw] + 1e-12, (r["steps"], dev, r[dev], r[tw])   # measured 0.2 .. 1.6
    ten = rows[-1]
    assert ten["twin_sig"] > 1e-8 and ten["sig_not_near"] > 1e-8, "if this fails the survey's bound became attainable: tighten the tests"
