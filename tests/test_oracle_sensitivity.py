"""Documents, on the CPU oracle alone, how the reference ALGORITHM amplifies round-off: a 1-ulp change of
the wind forcing stays at round-off for a couple of steps and then jumps by many orders of magnitude
(brittle damage branches FE.cpp:4229; integer-truncated M_delta_x FE.cpp:10239).  This bounds what any
parity test can assert over long horizons and is why tests/test_gpu_parity.py checks 10-step parity
along the oracle's trajectory (re-seeding every step) instead of free-running."""
import numpy as np

import cases
from oracle import pyoracle as O


def test_one_ulp_of_wind_is_roundoff_after_one_step_and_macroscopic_after_ten():
    gm, p, g, lms, fields = cases.make_case("toy")
    a = O.OracleRank(lms[0], p, fields[0])
    f2 = {k: v.copy() for k, v in fields[0].items()}
    f2["wind"] = np.nextafter(f2["wind"], np.inf)
    b = O.OracleRank(lms[0], p, f2)
    a.step(); b.step()
    one = max(cases.rel_err(b.arr[k], a.arr[k]) for k in ("VT", "sigma0", "damage"))
    for _ in range(9):
        a.step(); b.step()
    ten = max(cases.rel_err(b.arr[k], a.arr[k]) for k in ("VT", "sigma0", "damage"))
    assert one < 1e-11, one          # measured 4.5e-14
    assert ten > 1e-6, ten           # measured 2e-1: decorrelated
    assert a.check_fields_fast() == 0 and b.check_fields_fast() == 0


def test_strict_and_native_builds_of_the_oracle_agree_bitwise():
    """-std=c11 keeps gcc from contracting a*b+c even at -O3 -march=native, so the timed cpu_baseline
    build (liboracle_fast.so) computes the same bits as the strict one."""
    gm, p, g, lms, fields = cases.make_case("small")
    a = O.OracleRank(lms[0], p, fields[0]); b = O.OracleRank(lms[0], p, fields[0], fast=True)
    for _ in range(2):
        a.step(); b.step()
    for k in ("VT", "UM", "sigma0", "sigma1", "sigma2", "damage", "conc", "thick"):
        assert np.array_equal(a.arr[k], b.arr[k]), k
