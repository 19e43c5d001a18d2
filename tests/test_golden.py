"""Committed fixtures (tests/golden/, made by tests/golden/make_golden.py).

oracle_tiny.npz holds the oracle's own state on the seeded 'tiny' toy case after 1 sub-step, 1 step
and 3 steps; the CPU test is a regression net for the oracle, the GPU test compares the HIP path
with the same fixture without touching the oracle at run time."""
import os

import numpy as np
import pytest

import cases

GOLD = os.path.join(os.path.dirname(__file__), "golden", "oracle_tiny.npz")
KEYS = ("VT", "UM", "UT", "sigma0", "sigma1", "sigma2", "damage", "conc", "thick", "snow_thick", "ridge_ratio",
        "conc_young", "h_young", "hs_young", "conc_myi", "thick_myi")
CASES = (("sub1", 1, dict(substeps=1, dtime_step=200. / 120.)), ("step1", 1, {}), ("step3", 3, {}))


@pytest.mark.parametrize("tag,nsteps,over", CASES)
def test_oracle_reproduces_fixture(tag, nsteps, over):
    from oracle import pyoracle as O
    z = np.load(GOLD)
    gm, p, g, lms, fields = cases.make_case("tiny", **over)
    r = O.OracleRank(lms[0], p, fields[0])
    for _ in range(nsteps):
        r.step()
    for k in KEYS:
        # same compiler flags => same bits; allow a libm update to move the last digits
        assert cases.rel_err(r.arr[k], z[f"{tag}_{k}"]) < 1e-9, k


@pytest.mark.gpu
@pytest.mark.parametrize("tag,nsteps,over,tol", [(c[0], c[1], c[2], t) for c, t in zip(CASES, (1e-13, 1e-10, 1e-9))])
def test_gpu_matches_fixture(tag, nsteps, over, tol):
    from nextsim_amd import dynamics
    z = np.load(GOLD)
    gm, p, g, lms, fields = cases.make_case("tiny", **over)
    fe = dynamics.FiniteElementDynamics(p)
    fe.set_mesh(lms[0]); fe.put_state(fields[0]); fe.set_forcing(fields[0])
    for _ in range(nsteps):
        fe.step()
    fe.synchronize()
    got = fe.get_state()
    for k in KEYS:
        assert cases.rel_err(got[k], z[f"{tag}_{k}"]) <= tol, k
    fe.close()
