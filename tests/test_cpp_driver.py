"""BASELINE config 1 from a C++ host: examples/nextsim_toy.cpp is built with plain g++ against the C ABI
(include/nxs_dyn.hpp wrapper), runs the toy configuration for 10 steps and exports the final state with the
Exporter; the result must be bit-identical to the Python-driven run of the same library and within the
one-step tolerance of the oracle for the first step."""
import os
import subprocess

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "nextsim_toy")
    lib = os.path.join(ROOT, "nextsim_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "nextsim_toy.cpp"),
                           "-L", lib, "-lnxsdyn", f"-Wl,-rpath,{lib}", "-o", exe])
    return exe


def test_cpp_driver_builds_with_gpp_and_reports_errors(tmp_path):
    """No GPU needed: the driver links against the C ABI and fails loudly on a bad case file."""
    exe = _build(tmp_path)
    r = subprocess.run([exe, str(tmp_path / "missing.bin"), "1", str(tmp_path) + "/"], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr


@pytest.mark.gpu
def test_cpp_driver_runs_config1_and_matches_python_path(tmp_path):
    from nextsim_amd import casefile, dynamics, io as nio
    from oracle import pyoracle as O
    gm, p, g, lms, fields = cases.make_case("toy")
    lm, f = lms[0], fields[0]
    case = str(tmp_path / "toy.bin")
    casefile.write_case(case, lm, p, f)
    exe = _build(tmp_path)
    prefix = str(tmp_path) + "/"
    r = subprocess.run([exe, case, "10", prefix], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rec = nio.read_exported(prefix + "field_final.bin", prefix + "field_final.dat")
    mesh = nio.read_exported(prefix + "mesh_final.bin", prefix + "mesh_final.dat")
    assert np.array_equal(mesh["Elements"], lm.indices) and np.array_equal(mesh["Nodes_x"], lm.coord_x)
    fe = dynamics.FiniteElementDynamics(p); fe.set_mesh(lm); fe.put_state(f); fe.set_forcing(f)
    for _ in range(10):
        assert fe.checkFieldsFast() == 0
        fe.step()
    fe.synchronize()
    got = fe.get_state()
    for k in ("VT", "UM", "UT", "conc", "thick", "damage", "sigma0", "sigma1", "sigma2"):
        assert np.array_equal(rec["M_" + k], got[k]), k
    assert rec["Time"][0] == 42291. + 10 * 200. / 86400.
    fe.close()
    # and one step of the same driver against the oracle
    r = subprocess.run([exe, case, "1", prefix + "s1_"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rec1 = nio.read_exported(prefix + "s1_field_final.bin", prefix + "s1_field_final.dat")
    ref = O.OracleRank(lm, p, f); ref.step()
    for k in ("VT", "sigma0", "damage", "conc"):
        assert cases.rel_err(rec1["M_" + k], ref.arr[k]) <= 1e-10, k
